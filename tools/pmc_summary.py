"""Averages rocprofv3 --pmc counter CSVs per kernel dispatch.  usage: pmc_summary.py file.csv ..."""
import collections
import csv
import sys

for f in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rt_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-32s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
