#!/bin/bash
# Copy the evidence tools/collect_final.sh left under gpurun_out/<tag>/ into profiles/ under the round's names, and
# refresh profiles/pmc_traffic.json from the PMC summaries:   tools/publish_profiles.sh <tag> [round prefix, default r05]
set -e
tag=$1; r=${2:-r05}; src=gpurun_out/$tag
cp $src/bench_default.json profiles/${r}_bench_default_with_variants.json
for w in c3 c2 c3t c2t; do
  cp $src/bench_$w.json profiles/${r}_bench_$w.json
  cp $src/bench_${w}_under_rocprof.json profiles/${r}_bench_${w}_under_rocprof.json
  cp $src/${w}_kernel_stats.csv profiles/${r}_${w}_kernel_stats.csv
  cp $src/${w}_pmc_summary.json profiles/${r}_${w}_pmc_summary.json
done
[ -s $src/c3_factored_phase_stamps.txt ] && cp $src/c3_factored_phase_stamps.txt profiles/${r}_c3_factored_phase_stamps.txt
cp $src/time_to_argmin_split.txt profiles/${r}_time_to_argmin_split.txt
cp $src/bench_og.json profiles/${r}_bench_og.json
cp $src/bench_c3_strong_1gpu.json profiles/${r}_bench_c3_strong_1gpu.json
cp $src/gpu_tests_full.log profiles/${r}_gpu_tests_full.log
python3 tools/pmc_to_traffic.py $r > /dev/null
ls profiles | grep -c "^${r}_"
