mkdir -p gpurun_out/r5c
COVEST_RECORD_TAIL_BOUNDS=$PWD/gpurun_out/r5c/bounds_new.json timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "round5_tail" > gpurun_out/r5c/record.log 2>&1; tail -3 gpurun_out/r5c/record.log
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r5c/gpu_tests.log 2>&1; tail -12 gpurun_out/r5c/gpu_tests.log
WL="c3 c3t c2" AB_STEPS=200 timeout -k 10 300 bash tools/ab_many.sh tools/bin/libcovest_r5a.so covest_amd/lib/libcovest_amd.so 2>&1 | tee gpurun_out/r5c/ab.txt
for bc in 12 18 26 36 50; do echo -n "c3t build_cost=$bc: "; COVEST_AMD_LIB=$PWD/tools/bin/lib_tune.so COVEST_FACTORED_BUILD_COST=$bc python bench.py --workload c3t --steps 100 --warmup 20 --cpu-budget 0 --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])"; done 2>&1 | tee gpurun_out/r5c/sweep_bc_c3t.txt
