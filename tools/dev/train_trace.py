#!/usr/bin/env python3
"""Eager training steps of the cifar10 flow for `rocprofv3 --kernel-trace` (then: train_trace.py --summarize DIR).
usage: train_trace.py [B] [steps] [graph]   |   train_trace.py --summarize DIR"""
import csv, glob, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    marks = [i for i, r in enumerate(rows) if "k_preprocess" in r[2]]
    a, b = marks[-4], marks[-1]                      # three whole steps
    win = rows[a:b]
    span = rows[b][0] - rows[a][0]
    busy = sum(e - s for s, e, _ in win)
    print("3 steps: span %.2f ms, kernel time %.2f ms (%.1f %%), %d launches / step" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(win) // 3))
    agg = {}
    for s, e, n in win:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")[:90]
        c = agg.setdefault(n, [0, 0]); c[0] += 1; c[1] += e - s
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-90s %4d / step %8.1f us avg %7.3f ms / step %5.1f %%" % (n, c // 3, t / c / 1e3, t / 3e6, 100.0 * t / span))
    # the largest idle gaps
    gaps = sorted(((win[i + 1][0] - max(w[1] for w in win[max(0, i - 3):i + 1]), win[i][2][:60], win[i + 1][2][:60]) for i in range(len(win) - 1)), reverse=True)[:12]
    print("idle total %.2f ms" % ((span - busy) / 1e6))
    for g, p, n in gaps:
        print("gap %8.1f us after %-60s before %s" % (g / 1e3, p.replace("(anonymous namespace)::", ""), n.replace("(anonymous namespace)::", "")))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
r = bench.secondary_training("cifar10", torch.device("cuda:0"), B, steps, len(sys.argv) > 3 and sys.argv[3] == "graph")
print(r["value"], r["ms_per_step"])
