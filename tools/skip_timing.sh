#!/bin/bash
# K-factored on C3 with phases switched off (DIAGNOSTIC library, wrong values): what the kernel's time is made of.
#   COVEST_FACTORED_SKIP bit 1 = phase A, 2 = the MFMA step loops after the first step, 4 = the logs, 8 = the shared steps
lib=${1:-tools/bin/libcovest_amd_diag.so}
for skip in 0 1 2 4 8 6 12 14 15 0; do
  echo -n "skip $skip: "
  COVEST_AMD_LIB=$PWD/$lib COVEST_FACTORED_SKIP=$skip python bench.py --workload c3 --steps 20 --warmup 3 --cpu-budget 0 --no-variants 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('step %.4f ms  kernel bracket %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms_avg']))"
done
