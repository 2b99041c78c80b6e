mkdir -p gpurun_out/r5v
WL="c2" AB_STEPS=400 timeout -k 10 300 bash tools/ab.sh covest_amd/lib/libcovest_amd.so tools/bin/lib_basics1.so tools/bin/lib_basics2.so 2>&1 | tee gpurun_out/r5v/ab.txt
