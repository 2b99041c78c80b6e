mkdir -p gpurun_out/r5r
timeout -k 10 300 bash tools/kstat_ab.sh c2 tools/bin/lib_r5q.so covest_amd/lib/libcovest_amd.so 2>&1 | tee gpurun_out/r5r/kstat_c2.txt
timeout -k 10 200 bash tools/kstat_ab.sh c3 covest_amd/lib/libcovest_amd.so 2>&1 | tee gpurun_out/r5r/kstat_c3.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r5r/tests.txt
