#!/bin/bash
# The C3 part of tools/collect_profiles.sh alone (after a change to K-factored): tools/collect_c3.sh <tag>
set -u
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$repo"
w=c3
timeout -k 10 200 python3 bench.py --workload $w > "$out/bench_$w.json" 2> "$out/bench_$w.err" || echo "bench $w failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -o "$w" -- python3 bench.py --workload $w --steps 10 --warmup 2 --cpu-budget 0 > "$out/bench_${w}_under_rocprof.json" 2> "$out/trace_$w.err" || echo "trace $w failed"
find "$out/trace_$w" -name "*kernel_stats.csv" -exec cp {} "$out/${w}_kernel_stats.csv" \;
bash tools/pmc_profile.sh "$out/pmc_$w" --workload $w --steps 5 --warmup 1 > "$out/pmc_$w.log" 2>&1
python3 tools/pmc_summary.py "$out/pmc_$w" ll_ > "$out/${w}_pmc_summary.json"
rm -rf "$out/trace_$w"
find "$out/pmc_$w" -name "*.csv" -size +2000k -delete
timeout -k 10 200 python3 bench.py --scaling strong --steps 5 --warmup 1 --cpu-budget 0 > "$out/bench_c3_strong_1gpu.json" 2> "$out/bench_strong.err"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --share-gpu --steps 5 --warmup 1 --cpu-budget 0 2> "$out/bench_2rank.err" | tail -1 > "$out/bench_c3_2rank_gloo_rehearsal.json"
if [ -f tools/bin/libcovest_amd_diag.so ]; then
  COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so COVEST_FACTORED_DIAG=1 timeout -k 10 100 python3 tools/factored_diag.py > "$out/c3_factored_phase_stamps.txt" 2>&1
  COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so bash tools/phase_insts.sh "$out/phase_insts" > "$out/c3_factored_insts_by_phase.txt" 2>&1
  rm -rf "$out/phase_insts"
fi
timeout -k 10 100 python3 tools/time_host.py > "$out/time_to_argmin_split.txt" 2>&1
timeout -k 10 100 python3 tools/time_tail.py > "$out/tail_timing.txt" 2>&1
timeout -k 10 200 python3 bench.py --workload f2 > "$out/bench_f2.json" 2> "$out/bench_f2.err"
ls "$out"
