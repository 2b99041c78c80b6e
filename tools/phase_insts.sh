#!/bin/bash
# VALU instructions of K-factored by phase: the same C3 launch with one phase switched off at a time
# (COVEST_FACTORED_SKIP bit 1 = phase A, 2 = the MFMA step loops after the first step, 4 = the logs, 8 = the shared steps) under one --pmc pass each.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$PWD}"
out=${1:-gpurun_out/phase_insts}
mkdir -p $out
for skip in 0 1 2 4 8 15; do
  COVEST_FACTORED_SKIP=$skip timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/s$skip -o p -- python3 bench.py --workload c3 --steps 3 --warmup 1 --cpu-budget 0 > /dev/null 2>&1
  echo "== skip $skip"
  python3 tools/pmc_summary.py $out/s$skip | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d.items():
    if 'factored_kernel<512,3,false' in k.replace(' ',''): print({a:round(b/1e6,1) for a,b in v.items() if a!='_dispatches'})
"
done
