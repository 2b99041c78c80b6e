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
  timeout -k 10 200 python3 bench.py --workload $w --no-variants > "$out/bench_$w.json" 2> "$out/bench_$w.err" || echo "bench $w failed"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -o "$w" -- python3 bench.py --workload $w --steps 10 --warmup 2 --cpu-budget 0 --no-variants > "$out/bench_${w}_under_rocprof.json" 2> "$out/trace_$w.err" || echo "trace $w failed"
  find "$out/trace_$w" -name "*kernel_stats.csv" -exec cp {} "$out/${w}_kernel_stats.csv" \;
  bash tools/pmc_profile.sh "$out/pmc_$w" --workload $w --steps 5 --warmup 1 > "$out/pmc_$w.log" 2>&1
  python3 tools/pmc_summary.py "$out/pmc_$w" ll_ > "$out/${w}_pmc_summary.json"
  rm -rf "$out/trace_$w"
  find "$out/pmc_$w" -name "*.csv" -size +2000k -delete
done
timeout -k 10 100 python3 bench.py --workload c1 > "$out/bench_c1.json" 2> "$out/bench_c1.err"
timeout -k 10 200 python3 bench.py --workload og --steps 5 > "$out/bench_og.json" 2> "$out/bench_og.err"
timeout -k 10 200 python3 bench.py --scaling strong --steps 5 --warmup 1 --cpu-budget 0 > "$out/bench_c3_strong_1gpu.json" 2> "$out/bench_strong.err"
# the driver's N > 1 command line, rehearsed with four ranks on the one card over gloo (the pool's process guard allows six processes a GPU: five ranks and the launcher): the weak step AND variants.strong
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --backend gloo --share-gpu --steps 5 --warmup 1 --cpu-budget 0 2> "$out/bench_4rank.err" | tail -1 > "$out/bench_c3_4rank_gloo_rehearsal.json"
# phase stamps and instruction counts by phase: the DIAGNOSTIC build only (the shipped library has no such switches)
if [ -f tools/bin/libcovest_amd_diag.so ]; then
  COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so COVEST_FACTORED_DIAG=1 timeout -k 10 100 python3 tools/factored_diag.py > "$out/c3_factored_phase_stamps.txt" 2>&1
  COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so bash tools/phase_insts.sh "$out/phase_insts" > "$out/c3_factored_insts_by_phase.txt" 2>&1
  rm -rf "$out/phase_insts"
fi
timeout -k 10 100 python3 tools/time_host.py > "$out/time_to_argmin_split.txt" 2>&1
# the rows either side of the path, the shapes that used to fall back, the microbenchmarks behind DESIGN's rooflines
for g in 1 10; do
  timeout -k 10 400 python3 bench.py --workload c5 --kmer-gbp $g --steps 3 --warmup 1 > "$out/bench_c5_${g}gbp.json" 2> "$out/bench_c5_$g.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c5_$g" -o c5 -- python3 bench.py --workload c5 --kmer-gbp $g --steps 3 --warmup 1 --cpu-budget 0 > /dev/null 2> "$out/trace_c5_$g.err"
  find "$out/trace_c5_$g" -name "*kernel_stats.csv" -exec cp {} "$out/c5_${g}gbp_kernel_stats.csv" \;
  rm -rf "$out/trace_c5_$g"
done
# the table path of rounds 1-2 on the same input, for the comparison in DESIGN 6b
timeout -k 10 400 python3 bench.py --workload c5 --kmer-gbp 10 --kmer-path table --steps 2 --warmup 1 --cpu-budget 0 > "$out/bench_c5_10gbp_table_path.json" 2> "$out/bench_c5_table.err"
bash tools/pmc_profile.sh "$out/pmc_c5" --workload c5 --kmer-gbp 1 --steps 2 --warmup 1 > "$out/pmc_c5.log" 2>&1
python3 tools/pmc_summary.py "$out/pmc_c5" kmer_ > "$out/c5_1gbp_pmc_summary.json"
find "$out/pmc_c5" -name "*.csv" -size +2000k -delete
timeout -k 10 200 python3 bench.py --workload f2 > "$out/bench_f2.json" 2> "$out/bench_f2.err"
timeout -k 10 100 python3 bench.py --workload f3 > "$out/bench_f3.json" 2> "$out/bench_f3.err"
timeout -k 10 200 python3 tools/time_cliffs.py > "$out/cliffs.txt" 2>&1
timeout -k 10 100 python3 tools/time_tail.py > "$out/tail_timing.txt" 2>&1
timeout -k 10 100 python3 tools/latency.py > "$out/latency_single_evaluation.txt" 2>&1
[ -x tools/bin/microbench_atomics ] && timeout -k 10 100 tools/bin/microbench_atomics > "$out/microbench_atomics.txt" 2>&1
[ -x tools/bin/microbench_contract ] && timeout -k 10 60 tools/bin/microbench_contract > "$out/microbench_contract.txt" 2>&1
ls "$out"
