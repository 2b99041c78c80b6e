#!/bin/bash
# The evidence the bench line's numbers rest on, for the round's FINAL library, in parts that each fit one gpurun call:
#   tools/collect_final.sh <tag> a   the default bench line (with its child-run variants); C3 and C2: the plain bench line,
#                                    rocprofv3 kernel stats and the PMC passes (separate runs, program after `--`, no child
#                                    processes under the profiler); the phase stamps of the diagnostic build
#   tools/collect_final.sh <tag> b   the same three for C3 / C2 on the trimmed histograms with their tails (c3t, c2t); the
#                                    10 000-key-with-tail timings; optimize_grid, one evaluation's latency, the host-side
#                                    split of a search, the strong-scaling grid on one GPU
#   tools/collect_final.sh <tag> c   the whole GPU test suite, the 300-seed fuzz, the library's own optimum
#   tools/collect_final.sh <tag> d   C5 at 10 Gbp: bench line, rocprofv3 kernel stats, PMC passes
# -> gpurun_out/<tag>/ ; tools/publish_profiles.sh <tag> copies what is judged into profiles/.
set -u
tag=${1:-final}; part=${2:-a}
out=gpurun_out/$tag
mkdir -p "$out"
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$repo"
profile() { # workload: bench line, kernel stats, PMC summary
  w=$1
  timeout -k 10 200 python3 bench.py --workload $w --no-variants > "$out/bench_$w.json" 2> "$out/bench_$w.err" || echo "bench $w failed"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -o "$w" -- python3 bench.py --workload $w --steps 10 --warmup 2 --cpu-budget 0 --no-variants > "$out/bench_${w}_under_rocprof.json" 2> "$out/trace_$w.err" || echo "trace $w failed"
  find "$out/trace_$w" -name "*kernel_stats.csv" -exec cp {} "$out/${w}_kernel_stats.csv" \;
  bash tools/pmc_profile.sh "$out/pmc_$w" --workload $w --steps 5 --warmup 1 > "$out/pmc_$w.log" 2>&1
  python3 tools/pmc_summary.py "$out/pmc_$w" ll_ > "$out/${w}_pmc_summary.json"
  rm -rf "$out/trace_$w"
  find "$out/pmc_$w" -name "*.csv" -size +2000k -delete
}
case $part in
a)
  timeout -k 10 600 python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" || echo "default bench failed"
  profile c3; profile c2
  if [ -f tools/bin/libcovest_amd_diag.so ]; then
    bash tools/diag_sweeps.sh stamps c3 > "$out/c3_factored_phase_stamps.txt" 2>&1
    bash tools/diag_sweeps.sh stamps c3t > "$out/c3t_factored_phase_stamps.txt" 2>&1
  fi;;
b)
  profile c3t; profile c2t
  timeout -k 10 120 python3 tools/time_tail.py > "$out/tail_timing.txt" 2>&1
  timeout -k 10 100 python3 tools/time_host.py > "$out/time_to_argmin_split.txt" 2>&1
  timeout -k 10 100 python3 tools/latency.py > "$out/latency_single_evaluation.txt" 2>&1
  timeout -k 10 200 python3 bench.py --workload og --steps 5 > "$out/bench_og.json" 2> "$out/bench_og.err"
  timeout -k 10 200 python3 bench.py --scaling strong --steps 5 --warmup 1 --cpu-budget 0 > "$out/bench_c3_strong_1gpu.json" 2> "$out/bench_strong.err"
  timeout -k 10 100 python3 bench.py --workload c1 --cpu-budget 2 > "$out/bench_c1.json" 2> "$out/bench_c1.err"
  timeout -k 10 150 python3 bench.py --workload f2 > "$out/bench_f2.json" 2> "$out/bench_f2.err"
  timeout -k 10 100 python3 bench.py --workload f3 > "$out/bench_f3.json" 2> "$out/bench_f3.err";;
c)
  timeout -k 10 100 python3 tools/record_own_optimum.py > "$out/own_optimum.log" 2>&1; cp gpurun_out/own_optimum.json "$out/" 2>/dev/null
  timeout -k 10 700 python3 -m pytest tests -m gpu -q > "$out/gpu_tests_full.log" 2>&1
  tail -3 "$out/gpu_tests_full.log"
  COVEST_FUZZ_SEEDS=300 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz > "$out/gpu_tests_fuzz300_seeds.log" 2>&1
  tail -3 "$out/gpu_tests_fuzz300_seeds.log";;
d)
  timeout -k 10 300 python3 bench.py --workload c5 --kmer-gbp 10 --steps 3 --warmup 1 > "$out/bench_c5_10gbp.json" 2> "$out/bench_c5.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c5" -o c5 -- python3 bench.py --workload c5 --kmer-gbp 10 --steps 2 --warmup 1 --cpu-budget 0 --compact > /dev/null 2> "$out/trace_c5.err"
  find "$out/trace_c5" -name "*kernel_stats.csv" -exec cp {} "$out/c5_10gbp_kernel_stats.csv" \;
  rm -rf "$out/trace_c5"
  bash tools/pmc_profile.sh "$out/pmc_c5" --workload c5 --kmer-gbp 10 --steps 2 --warmup 1 --compact > "$out/pmc_c5.log" 2>&1
  python3 tools/pmc_summary.py "$out/pmc_c5" kmer_ > "$out/c5_10gbp_pmc_summary.json"
  find "$out/pmc_c5" -name "*.csv" -size +2000k -delete;;
esac
ls "$out"
