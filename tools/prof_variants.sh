# rocprofv3 kernel stats of tools/grid_loop.py (30 C3 + 30 C2 grid evaluations) for several builds of the library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  echo "== $lib"
  export COVEST_AMD_LIB=$PWD/$lib
  rm -rf gpurun_out/trace_v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_v -o v -- python3 tools/grid_loop.py > /dev/null 2>&1
  python3 tools/kstats.py $(find gpurun_out/trace_v -name "*kernel_stats.csv") | grep -E "ll_|fix"
done
