# phase ablation of K-scan on C3: COVEST_SCAN_SKIP bit 0/1/2 skips phase A/B/C (results wrong, timing only)
for sk in 0 4 6 7; do
  echo -n "skip=$sk: "; COVEST_SCAN_SKIP=$sk python bench.py --workload c3 --kernel scan --steps 10 --warmup 2 --cpu-budget 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['roofline']['kernel_ms_avg'])"
done
