#!/bin/bash
# Phase stamps and the LPT-charge sweep of K-factored on C3, on the DIAGNOSTIC build (tools/bin/libcovest_amd_diag.so:
# python -m covest_amd.build --out tools/bin/libcovest_amd_diag.so -DCOVEST_DIAG) -- the shipped library has no knobs.
export COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so
out=${1:-gpurun_out/sweep}
mkdir -p $out
COVEST_FACTORED_DIAG=1 timeout -k 10 100 python3 tools/factored_diag.py > $out/phase_stamps.txt 2>&1
cat $out/phase_stamps.txt
SWEEP_SD="${SWEEP_SD:-5}" SWEEP_UO="${SWEEP_UO:-1 2}" SWEEP_BC="${SWEEP_BC:-6 10 14 18 24 30 36}" bash tools/sweep_build_cost.sh 2>&1 | tee $out/sweep.txt
