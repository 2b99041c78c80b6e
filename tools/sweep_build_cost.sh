# sweep the LPT charges of K-factored on C3 (one process per value)
for sd in ${SWEEP_SD:-5 8 16}; do
for uo in ${SWEEP_UO:-1 2 3}; do
for bc in ${SWEEP_BC:-36 48 64 90}; do
  echo -n "shared_div=$sd unit_overhead=$uo build_cost=$bc: "
  COVEST_FACTORED_SHARED_DIV=$sd COVEST_FACTORED_UNIT_OVERHEAD=$uo COVEST_FACTORED_BUILD_COST=$bc python bench.py --workload c3 --kernel factored --steps 20 --warmup 3 --cpu-budget 0 --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['roofline']['kernel_ms_avg'])"
done
done
done
