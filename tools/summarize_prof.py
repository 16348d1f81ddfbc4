#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile.sh: per-kernel time table + PMC averages per launch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    s = m.group(1) if m else name
    if s.startswith("at::native"):
        s = "torch:" + (re.search(r"(uniform|normal|random_from_to|FillFunctor|copy|CatArray|add|neg|Mul)", name) or [None, "other"])[1]
    return s[:90]


print("# rocprofv3 summary (%s)\n" % os.path.basename(out.rstrip("/")))
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("## kernel time (--kernel-trace --stats; bench.py --steps 2 --warmup 1)\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = short(r["Name"])
        agg[k][0] += int(r["Calls"]); agg[k][1] += float(r["TotalDurationNs"])
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
        print("| `%s` | %d | %.3f | %.1f | %.2f |" % (k, c, t / 1e6, t / c / 1e3, 100 * t / tot))
    print("\ntotal GPU kernel time: %.2f ms\n" % (tot / 1e6))

print("## PMC (per-launch averages; bench.py --steps 1 --warmup 1 --global-batch 16384)\n")
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    names = sorted({c for k in acc for c in acc[k]})
    print("### %s\n" % os.path.basename(d))
    print("| kernel | launches | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    keep = [k for k in acc if k.startswith(("k_flow_step", "k_gmm_logprob", "k_sample", "k_squeeze"))]
    for k in sorted(keep):
        n = max(cnt[k].values())
        print("| `%s` | %d | " % (k, n) + " | ".join("%.4g" % (acc[k][c] / max(cnt[k][c], 1)) for c in names) + " |")
    print()
