#!/usr/bin/env python3
"""Short table of a rocprofv3 *_kernel_stats.csv: kernel (template head only), calls, average ns."""
import csv, re, sys
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(anonymous namespace\)::|covest::|void ", "", row["Name"])
            name = re.sub(r"\(.*", "", name)
            print("%-44s %6s calls  avg %10.1f us" % (name[:44], row["Calls"], float(row["AverageNs"]) / 1e3))
