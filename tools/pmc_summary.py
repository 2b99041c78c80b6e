#!/usr/bin/env python3
"""Summarise the per-pass counter CSVs of tools/pmc_profile.sh: per kernel, the mean
of every counter over its dispatches.  Usage: tools/pmc_summary.py <dir> [kernel substring]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "ll_"
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if want not in name:
                continue
            import re
            # kernel name with its template arguments (the tail / no-tail variants are different kernels)
            mm = re.search(r"(ll_[a-z_]+(?:<[^>]*>)?|argmin_stage\d|kmer_[a-z_]+(?:<[^>]*>)?)", name)
            short = mm.group(1).replace(" ", "") if mm else name[:40]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, counters in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in sorted(counters.items())}
    out[k]["_dispatches"] = max(len(v) for v in counters.values())
print(json.dumps(out, indent=1))
