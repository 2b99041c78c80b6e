set -e
for i in 1 2; do
for v in base new; do
  if [ $v = base ]; then export COVEST_AMD_LIB=$PWD/covest_amd/lib/libcovest_amd_base.so; else export COVEST_AMD_LIB=$PWD/covest_amd/lib/libcovest_amd.so; fi
  for w in c2 c3; do
    echo -n "$v $w: "; python bench.py --workload $w --steps 20 --warmup 3 --cpu-budget 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['frac'])"
  done
done
done
