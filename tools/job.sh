bash tools/kstat_ab.sh c3 tools/bin/libcovest_amd_r04.so covest_amd/lib/libcovest_amd.so tools/bin/lib_fixlibexp.so 2>&1 | tail -16
bash tools/collect_final.sh r5final c
