#!/bin/bash
# PMC profile of bench.py on the GPU box: one rocprofv3 pass per counter group
# (8 SQ slots per pass; FETCH_SIZE / WRITE_SIZE need passes of their own), never
# combined with tracing options.  Usage: tools/pmc_profile.sh <outdir> <bench args...>
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$PWD}"
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  timeout -k 10 240 rocprofv3 --pmc $g --output-format csv -d "$out" -o "pass$i" -- python3 bench.py "$@" --cpu-budget 0 --no-variants > "$out/pass$i.json" 2> "$out/pass$i.err" || echo "pass $i failed"
  i=$((i+1))
done
ls "$out"
