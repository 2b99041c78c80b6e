# experimental builds of kmer_bulk.hip only (the other objects are the shipped ones): tools/exp_kmer.sh <name> [-DFLAG ...]
# -> exp/lib_<name>.so, for A/B runs via COVEST_AMD_LIB
name=$1; shift
mkdir -p exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -x hip "$@" -c covest_amd/csrc/kmer_bulk.hip -o exp/kmer_bulk_$name.o || exit 1
objs=$(ls covest_amd/lib/obj/*.o | grep -v kmer_bulk.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/lib_$name.so $objs exp/kmer_bulk_$name.o && echo exp/lib_$name.so
