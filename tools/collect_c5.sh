#!/bin/bash
# The C5 part of tools/collect_profiles.sh alone: tools/collect_c5.sh <tag>
set -u
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$repo"
for g in 1 10; do
  timeout -k 10 400 python3 bench.py --workload c5 --kmer-gbp $g --steps 3 --warmup 1 > "$out/bench_c5_${g}gbp.json" 2> "$out/bench_c5_$g.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c5_$g" -o c5 -- python3 bench.py --workload c5 --kmer-gbp $g --steps 3 --warmup 1 --cpu-budget 0 > /dev/null 2> "$out/trace_c5_$g.err"
  find "$out/trace_c5_$g" -name "*kernel_stats.csv" -exec cp {} "$out/c5_${g}gbp_kernel_stats.csv" \;
  rm -rf "$out/trace_c5_$g"
done
bash tools/pmc_profile.sh "$out/pmc_c5" --workload c5 --kmer-gbp 1 --steps 2 --warmup 1 > "$out/pmc_c5.log" 2>&1
python3 tools/pmc_summary.py "$out/pmc_c5" kmer_ > "$out/c5_1gbp_pmc_summary.json"
find "$out/pmc_c5" -name "*.csv" -size +2000k -delete
ls "$out"
