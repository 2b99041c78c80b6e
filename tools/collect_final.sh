#!/bin/bash
# The evidence the bench line's numbers rest on, for the round's FINAL library:
#   tools/collect_final.sh <tag>   -> gpurun_out/<tag>/: the default bench line (with its child-run variants), rocprofv3
#   kernel stats and PMC passes of C3 and C2 (separate runs, program after `--`, no child processes under the profiler),
#   the phase stamps of the diagnostic build, the whole GPU test suite.
set -u
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p "$out"
repo=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$repo"
timeout -k 10 500 python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" || echo "default bench failed"
for w in c3 c2 c3t c2t; do
  timeout -k 10 200 python3 bench.py --workload $w --no-variants > "$out/bench_$w.json" 2> "$out/bench_$w.err" || echo "bench $w failed"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -o "$w" -- python3 bench.py --workload $w --steps 10 --warmup 2 --cpu-budget 0 --no-variants > "$out/bench_${w}_under_rocprof.json" 2> "$out/trace_$w.err" || echo "trace $w failed"
  find "$out/trace_$w" -name "*kernel_stats.csv" -exec cp {} "$out/${w}_kernel_stats.csv" \;
  bash tools/pmc_profile.sh "$out/pmc_$w" --workload $w --steps 5 --warmup 1 > "$out/pmc_$w.log" 2>&1
  python3 tools/pmc_summary.py "$out/pmc_$w" ll_ > "$out/${w}_pmc_summary.json"
  rm -rf "$out/trace_$w"
  find "$out/pmc_$w" -name "*.csv" -size +2000k -delete
done
if [ -f tools/bin/libcovest_amd_diag.so ]; then
  COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so COVEST_FACTORED_DIAG=1 timeout -k 10 100 python3 tools/factored_diag.py > "$out/c3_factored_phase_stamps.txt" 2>&1
fi
timeout -k 10 100 python3 tools/time_host.py > "$out/time_to_argmin_split.txt" 2>&1
timeout -k 10 200 python3 bench.py --workload og --steps 5 > "$out/bench_og.json" 2> "$out/bench_og.err"
timeout -k 10 200 python3 bench.py --scaling strong --steps 5 --warmup 1 --cpu-budget 0 > "$out/bench_c3_strong_1gpu.json" 2> "$out/bench_strong.err"
timeout -k 10 900 python3 -m pytest tests -m gpu -q > "$out/gpu_tests_full.log" 2>&1
tail -3 "$out/gpu_tests_full.log"
ls "$out"
