#!/bin/bash
# BASELINE config 5 at its full size (10 Gbp resident in HBM) and the 150-histogram fuzz of the parity suite
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 500 python3 bench.py --workload c5 --kmer-gbp 10 --steps 3 --warmup 1 --cpu-budget 0 > "$out/bench_c5_10gbp.json" 2> "$out/bench_c5_10gbp.err" || echo "c5 10 Gbp failed"
COVEST_FUZZ_SEEDS=150 timeout -k 10 600 python3 -m pytest tests -m gpu -q -k fuzz > "$out/gpu_tests_fuzz150_seeds.log" 2>&1
tail -3 "$out/gpu_tests_fuzz150_seeds.log"
