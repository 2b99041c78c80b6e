#!/bin/bash
# Copy the evidence tools/collect_final.sh left under gpurun_out/<tag>/ into profiles/ under the round's names, and
# refresh profiles/pmc_traffic.json from the PMC summaries:   tools/publish_profiles.sh <tag> [round prefix, default r05]
tag=$1; r=${2:-r05}; src=gpurun_out/$tag
put() { [ -s "$src/$1" ] && cp "$src/$1" "profiles/${r}_$2"; }
put bench_default.json bench_default_with_variants.json
for w in c3 c2 c3t c2t; do
  put bench_$w.json bench_$w.json
  put bench_${w}_under_rocprof.json bench_${w}_under_rocprof.json
  put ${w}_kernel_stats.csv ${w}_kernel_stats.csv
  put ${w}_pmc_summary.json ${w}_pmc_summary.json
done
put c3_factored_phase_stamps.txt c3_factored_phase_stamps.txt
put c3t_factored_phase_stamps.txt c3t_factored_phase_stamps.txt
put tail_timing.txt tail_timing.txt
put time_to_argmin_split.txt time_to_argmin_split.txt
put latency_single_evaluation.txt latency_single_evaluation.txt
put bench_og.json bench_og.json
put bench_c3_strong_1gpu.json bench_c3_strong_1gpu.json
put bench_c1.json bench_c1.json
put bench_f2.json bench_f2.json
put bench_f3.json bench_f3.json
put gpu_tests_full.log gpu_tests_full.log
put gpu_tests_fuzz300_seeds.log gpu_tests_fuzz300_seeds.log
put own_optimum.log own_optimum.log
put bench_c5_10gbp.json bench_c5_10gbp.json
put c5_10gbp_kernel_stats.csv c5_10gbp_kernel_stats.csv
put c5_10gbp_pmc_summary.json c5_10gbp_pmc_summary.json
python3 tools/pmc_to_traffic.py $r > /dev/null
ls profiles | grep -c "^${r}_"
