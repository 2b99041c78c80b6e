mkdir -p gpurun_out/r5af
WL="c3 c3t" AB_STEPS=300 timeout -k 10 400 bash tools/ab.sh tools/bin/lib_r5ac.so covest_amd/lib/libcovest_amd.so 2>&1 | tee gpurun_out/r5af/ab_c3.txt
