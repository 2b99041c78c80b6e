mkdir -p gpurun_out/r5h
timeout -k 10 800 python -m pytest tests -m gpu -q -x --deselect "tests/test_gpu_hist_steps.py::test_whole_default_flow" > gpurun_out/r5h/gpu_tests.log 2>&1; tail -5 gpurun_out/r5h/gpu_tests.log
WL="c3 c3t" AB_STEPS=200 timeout -k 10 400 bash tools/ab.sh tools/bin/libcovest_r5g.so tools/bin/lib_nodrop.so covest_amd/lib/libcovest_amd.so 2>&1 | tee gpurun_out/r5h/ab.txt
timeout -k 10 200 python bench.py --workload og --steps 5 > gpurun_out/r5h/bench_og.json 2> gpurun_out/r5h/bench_og.err; python -c "
import json; d=json.load(open('gpurun_out/r5h/bench_og.json'))
for c in d['cases']: print(c['histogram'], c['iterations'], c['time_to_argmin_s'], c['split_ms'])"
timeout -k 10 120 python tools/time_tail.py 2>&1 | tee gpurun_out/r5h/tail_timing.txt
for wl in c3 c3t; do for bc in 12 15 18 22 26 32; do echo -n "$wl build_cost=$bc: "; COVEST_AMD_LIB=$PWD/tools/bin/lib_tune.so COVEST_FACTORED_BUILD_COST=$bc python bench.py --workload $wl --steps 100 --warmup 20 --cpu-budget 0 --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])"; done; done 2>&1 | tee gpurun_out/r5h/sweep_bc.txt
