#!/bin/bash
# One-off PMC passes for counters outside pmc_profile.sh's groups: tools/pmc_extra.sh <outdir> "<counters>" <bench args...>
set -u
out=$1; ctr=$2; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$out" -o extra -- python3 bench.py "$@" --cpu-budget 0 --no-variants > "$out/extra.json" 2> "$out/extra.err" || echo "extra pass failed"
python3 tools/pmc_summary.py "$out" ll_ 2>/dev/null
