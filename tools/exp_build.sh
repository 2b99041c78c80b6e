#!/bin/bash
# An experimental build of ONE source with extra flags, the other objects being the shipped ones -- for A/B runs via
# COVEST_AMD_LIB (tools/ab.sh):
#   tools/exp_build.sh <name> <source under covest_amd/csrc> [-DFLAG ...]     -> tools/bin/lib_<name>.so
# K-factored and K-basic are compiled once per template variant (covest_amd/build.py): all of them are rebuilt.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m covest_amd.build > /dev/null
obj=covest_amd/lib/obj_exp_$name; mkdir -p $obj tools/bin
stem=${src%.*}
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -x hip"
objs=$(ls covest_amd/lib/obj/*.o | grep -v "/${stem}\(_v[0-9]*\)\?\.o$")
/opt/rocm/bin/hipcc $common "$@" -c covest_amd/csrc/$src -o $obj/$stem.o & 
case $stem in
  ll_factored) for v in 0 1 2 3 4 5 6 7 8 9; do /opt/rocm/bin/hipcc $common "$@" -DCOVEST_FACTORED_VARIANT=$v -c covest_amd/csrc/$src -o $obj/${stem}_v$v.o & done;;
  ll_basic) for v in 0 1 2 3 4 5 6 7; do /opt/rocm/bin/hipcc $common "$@" -DCOVEST_BASIC_VARIANT=$v -c covest_amd/csrc/$src -o $obj/${stem}_v$v.o & done;;
esac
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_$name.so $objs $obj/*.o
rm -rf $obj
echo tools/bin/lib_$name.so
