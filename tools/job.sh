mkdir -p gpurun_out/r5i
timeout -k 10 800 python -m pytest tests -m gpu -q -x --deselect "tests/test_gpu_hist_steps.py::test_whole_default_flow" > gpurun_out/r5i/gpu_tests.log 2>&1; tail -5 gpurun_out/r5i/gpu_tests.log
WL="c3 c3t c2 c2t" AB_STEPS=200 timeout -k 10 400 bash tools/ab.sh tools/bin/libcovest_r5g.so covest_amd/lib/libcovest_amd.so 2>&1 | tee gpurun_out/r5i/ab.txt
timeout -k 10 200 python bench.py --workload og --steps 5 > gpurun_out/r5i/bench_og.json 2> gpurun_out/r5i/bench_og.err; python -c "
import json; d=json.load(open('gpurun_out/r5i/bench_og.json'))
for c in d['cases']: print(c['histogram'], c['iterations'], c['time_to_argmin_s'], c['split_ms'])"
timeout -k 10 120 python tools/time_tail.py 2>&1 | tee gpurun_out/r5i/tail_timing.txt
COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so COVEST_FACTORED_DIAG=1 COVEST_DIAG_TAIL=12345 timeout -k 10 100 python tools/factored_diag.py c3 2>&1 | tee gpurun_out/r5i/stamps_c3_10k_tail.txt
