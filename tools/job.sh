mkdir -p gpurun_out/r5ah
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r5ah/tests.txt
timeout -k 10 200 bash tools/kstat_ab.sh c2 tools/bin/lib_r5fin.so covest_amd/lib/libcovest_amd.so 2>&1 | tee gpurun_out/r5ah/kstat_c2.txt
timeout -k 10 120 python tools/time_tail.py 2>&1 | tee gpurun_out/r5ah/tail.txt
