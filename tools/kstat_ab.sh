#!/bin/bash
# rocprofv3 kernel stats of one bench workload under several builds of the library (one GPU run):
#   tools/kstat_ab.sh <workload> <lib.so> [<lib.so> ...]     -> gpurun_out/kstat_ab/<lib>_<workload>.txt
w=$1; shift
repo=$PWD
mkdir -p gpurun_out/kstat_ab
cd /tmp && export TMPDIR=/tmp && cd "$repo"
for lib in "$@"; do
  tag=$(basename $lib .so)_$w
  COVEST_AMD_LIB=$repo/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstat_ab/t_$tag -o k -- \
    python3 bench.py --workload $w --steps 10 --warmup 2 --cpu-budget 0 --no-variants > /dev/null 2> gpurun_out/kstat_ab/$tag.err
  find gpurun_out/kstat_ab/t_$tag -name "*kernel_stats.csv" -exec cat {} \; | python3 -c "
import csv, re, sys
for row in csv.DictReader(sys.stdin):  # kernel (template head only), calls, average ns
    m = re.search(r'(\\w+(?:<[^>(]*>)?)\\(', row['Name'].replace('covest::(anonymous namespace)::', '').replace('void ', ''))
    print('%-60s %6s %12s' % ((m.group(1) if m else row['Name'])[:60], row['Calls'], row['AverageNs']))" > gpurun_out/kstat_ab/$tag.txt
  rm -rf gpurun_out/kstat_ab/t_$tag
  echo "== $tag"; grep -E "ll_|argmin" gpurun_out/kstat_ab/$tag.txt
done
