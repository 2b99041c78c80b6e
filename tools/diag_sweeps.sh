#!/bin/bash
# What K-factored's time is made of, on the DIAGNOSTIC build (tools/bin/libcovest_amd_diag.so:
# python -m covest_amd.build --out tools/bin/libcovest_amd_diag.so -DCOVEST_DIAG) -- the shipped library has no knobs:
#   tools/diag_sweeps.sh stamps [c3|c3t]   in-kernel s_memtime stamps per wave (tools/factored_diag.py)
#   tools/diag_sweeps.sh skip   [c3|c3t]   kernel time with one phase switched off at a time (wrong values):
#                                          COVEST_FACTORED_SKIP bit 1 = phase A, 2 = the MFMA step loops after the first
#                                          step, 4 = the logs, 8 = the shared steps
#   tools/diag_sweeps.sh charges [c3|c3t]  the planner's builder charge swept (COVEST_FACTORED_BUILD_COST)
what=${1:-stamps}; wl=${2:-c3}
export COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so
line() { python bench.py --workload $wl --steps 50 --warmup 10 --cpu-budget 0 --no-variants 2>/dev/null |
  python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])"; }
case $what in
  stamps) COVEST_FACTORED_DIAG=1 python tools/factored_diag.py $wl;;
  skip) for s in 0 2 4 8 12 14; do echo -n "skip=$s: "; COVEST_FACTORED_SKIP=$s line; done;;
  charges) for bc in 10 14 18 22 26 32 40; do echo -n "build_cost=$bc: "; COVEST_FACTORED_BUILD_COST=$bc line; done;;
esac
