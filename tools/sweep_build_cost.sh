# sweep the LPT charges of K-factored on C3 (one process per value)
for uo in 2 3 4; do
for bc in 20 28 34 40; do
  echo -n "unit_overhead=$uo build_cost=$bc: "
  COVEST_FACTORED_UNIT_OVERHEAD=$uo COVEST_FACTORED_BUILD_COST=$bc python bench.py --workload c3 --kernel factored --steps 20 --warmup 3 --cpu-budget 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['roofline']['kernel_ms_avg'])"
done
done
