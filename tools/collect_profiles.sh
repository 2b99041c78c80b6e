#!/bin/bash
# Collect the judged evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...   (copy what is wanted into profiles/)
# rocprofv3 kernel-trace stats and PMC passes are separate runs (never combined), program after `--`.
set -u
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$repo"
for w in c3 c2; do
  timeout -k 10 200 python3 bench.py --workload $w > "$out/bench_$w.json" 2> "$out/bench_$w.err" || echo "bench $w failed"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -o "$w" -- python3 bench.py --workload $w --steps 10 --warmup 2 --cpu-budget 0 > "$out/bench_${w}_under_rocprof.json" 2> "$out/trace_$w.err" || echo "trace $w failed"
  find "$out/trace_$w" -name "*kernel_stats.csv" -exec cp {} "$out/${w}_kernel_stats.csv" \;
  bash tools/pmc_profile.sh "$out/pmc_$w" --workload $w --steps 5 --warmup 1 > "$out/pmc_$w.log" 2>&1
  python3 tools/pmc_summary.py "$out/pmc_$w" ll_ > "$out/${w}_pmc_summary.json"
done
timeout -k 10 100 python3 bench.py --workload c1 > "$out/bench_c1.json" 2> "$out/bench_c1.err"
ls "$out"
