#!/bin/bash
# The TUNING library: the shipped objects with plan_factored.cpp alone rebuilt under -DCOVEST_TUNE, which lets the
# environment override the planner's charges (COVEST_FACTORED_BUILD_COST, _UNIT_OVERHEAD, _SHARED_DIV,
# _LAST_BUILDER_EXTRA, _MIN_SHARED) -- host side only, the kernels are the shipped ones.  For sweeps on a GPU box:
#   tools/build_tune.sh && COVEST_AMD_LIB=$PWD/tools/bin/lib_tune.so COVEST_FACTORED_BUILD_COST=24 python bench.py --no-variants
set -e
cd "$(dirname "$0")/.."
python -m covest_amd.build > /dev/null
mkdir -p covest_amd/lib/obj_tune
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -x hip -DCOVEST_TUNE \
  -c covest_amd/csrc/plan_factored.cpp -o covest_amd/lib/obj_tune/plan_factored.o
objs=$(ls covest_amd/lib/obj/*.o | grep -v plan_factored)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_tune.so $objs covest_amd/lib/obj_tune/plan_factored.o
echo tools/bin/lib_tune.so
