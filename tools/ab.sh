# A/B of two builds of libcovest_amd.so in ONE GPU run (same box, same clocks):
#   tools/ab.sh <lib_a.so> <lib_b.so> [workloads...]     (default workloads: c2 c3; 400 timed steps behind 50 warm-up
#   steps: at 20 steps the device's clocks are still coming up, profiles/r04_clock_ramp.txt)
a=$1; b=$2; shift 2
wl=${*:-c2 c3}
for i in 1 2; do
  for lib in "$a" "$b"; do
    for w in $wl; do
      echo -n "$(basename $lib) $w: "
      COVEST_AMD_LIB=$PWD/$lib python bench.py --workload $w --steps ${AB_STEPS:-400} --warmup 50 --cpu-budget 0 --no-variants 2>/dev/null |
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
    done
  done
done
