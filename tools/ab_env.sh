# A/B of an environment knob on one build: tools/ab_env.sh VAR val_a val_b [workload]
v=$1; a=$2; b=$3; w=${4:-c3}
for i in 1 2 3; do
  for x in $a $b; do
    echo -n "$v=$x $w: "
    env $v=$x python bench.py --workload $w --steps 20 --warmup 3 --cpu-budget 0 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
  done
done
