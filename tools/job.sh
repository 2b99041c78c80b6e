mkdir -p gpurun_out/r5aj
timeout -k 10 300 python tools/time_rounds.py 2>&1 | tee gpurun_out/r5aj/rounds.txt
