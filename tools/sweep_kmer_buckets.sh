#!/bin/bash
# Bucket-count sweep of the partitioned k-mer path on the DIAGNOSTIC build (COVEST_KMER_LG is read there only):
#   GBP=10 LGS="23 22 21" tools/sweep_kmer_buckets.sh
export COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so
for lg in ${LGS:-23 22 21}; do
  COVEST_KMER_LG=$lg python3 bench.py --workload c5 --kmer-gbp ${GBP:-10} --steps 3 --warmup 1 --cpu-budget 0 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); p=d['config']['partition']; print($lg, '%.3g k-mers/s'%d['value'], '%.1f ms'%d['ms_per_step'], p['ms'], 'sampled 1 in', p['sampled_1_in'], 'room', p['room_records'], 'by workgroup', p['buckets_by_workgroup'], 'through table', p['buckets_through_table'])"
done
