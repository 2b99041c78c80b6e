# Kernel-only A/B of several builds of libcovest_amd.so in ONE GPU run (time_tail: hipEvents around the LL launch):
#   tools/ab_kernel.sh lib_a.so lib_b.so ...
for i in 1 2; do
  for lib in "$@"; do
    echo "== $(basename $lib)"
    COVEST_AMD_LIB=$PWD/$lib python tools/time_tail.py 2>/dev/null | grep "tail 0"
  done
done
