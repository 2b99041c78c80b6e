mkdir -p gpurun_out/r5z
timeout -k 10 300 python tools/record_own_optimum.py 2>&1 | tail -3 | tee gpurun_out/r5z/own.txt
timeout -k 10 300 python -m pytest tests/test_gpu_hist_steps.py -m gpu -q -x -k "whole_default_flow" 2>&1 | tail -30 | tee gpurun_out/r5z/flow.txt
for lib in tools/bin/lib_r5x.so covest_amd/lib/libcovest_amd.so; do for i in 1 2; do COVEST_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload og --steps 20 --warmup 3 --cpu-budget 0 --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib', d['ms_per_step'], d['value'])"; done; done | tee gpurun_out/r5z/og.txt
for lib in tools/bin/lib_r5x.so covest_amd/lib/libcovest_amd.so; do COVEST_AMD_LIB=$PWD/$lib timeout -k 10 200 python tools/latency.py 2>&1 | tail -4; done | tee gpurun_out/r5z/latency.txt
