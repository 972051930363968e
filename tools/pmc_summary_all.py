"""Per-kernel means of rocprofv3 --pmc counter CSVs, for any program (pmc_summary.py only reads the
render kernel's rows).  usage: pmc_summary_all.py file.csv ..."""
import collections
import csv
import sys

for f in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        # the last dispatch of a kernel is the long, timed one in the micro-benchmarks
        print("%-60s %-28s n=%d last=%.6g mean=%.6g" % (k, c, len(v), v[-1], sum(v) / len(v)))
