#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats output directory.  usage: kstats.py DIR [N] [filter]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(csv.DictReader(open(f)))
print("total kernel ms %.2f" % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6))
k = 0
for r in rows:
    name = r['Name'].replace("(anonymous namespace)::", "").replace("void ", "")
    if flt not in name: continue
    print("%-84s %5s %9.1f us %6.2f%%" % (name[:84], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    k += 1
    if k >= n: break
