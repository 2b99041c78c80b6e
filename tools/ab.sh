# A/B of likelihood kernels on C3 (one process per run)
for i in 1 2; do
for k in factored scan; do
  echo -n "$k: "; python bench.py --workload c3 --kernel $k --steps 20 --warmup 3 --cpu-budget 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['value'], d['roofline']['frac'])"
done
done
