mkdir -p gpurun_out/r5l
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/r5l/gpu_tests.log 2>&1; tail -4 gpurun_out/r5l/gpu_tests.log
WL="c3 c3t c2 c2t" AB_STEPS=300 timeout -k 10 400 bash tools/ab.sh tools/bin/libcovest_r5k.so covest_amd/lib/libcovest_amd.so 2>&1 | tee gpurun_out/r5l/ab.txt
for lib in tools/bin/libcovest_r5k.so covest_amd/lib/libcovest_amd.so; do echo $lib; COVEST_AMD_LIB=$PWD/$lib timeout -k 10 120 python tools/time_tail.py; done 2>&1 | tee gpurun_out/r5l/tail_timing.txt
