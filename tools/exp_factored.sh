# experimental builds of ll_factored.hip only (the other objects are the shipped ones): tools/exp_factored.sh <name> [-DFLAG ...]
# -> exp/lib_<name>.so, for A/B runs via COVEST_AMD_LIB
name=$1; shift
mkdir -p exp/obj_$name
pids=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -x hip "$@" -c covest_amd/csrc/ll_factored.hip -o exp/obj_$name/ll_factored.o & pids="$pids $!"
for v in 0 1 2 3 4 5 6 7 8 9; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -x hip "$@" -DCOVEST_FACTORED_VARIANT=$v -c covest_amd/csrc/ll_factored.hip -o exp/obj_$name/ll_factored_v$v.o & pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
objs=$(ls covest_amd/lib/obj/*.o | grep -v "ll_factored")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/lib_$name.so $objs exp/obj_$name/*.o && echo exp/lib_$name.so
