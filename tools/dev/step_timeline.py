#!/usr/bin/env python3
"""Timeline of ONE captured training step from a rocprofv3 --kernel-trace CSV (run: launch_census.py run <name> <B>):
start offset, duration, queue of every kernel of the last whole step, and the busy / idle time of the busiest queue.
usage: step_timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rd = list(csv.DictReader(open(f)))
qk = "Queue_Id" if "Queue_Id" in rd[0] else None
sk = "Stream_Id" if "Stream_Id" in rd[0] else None
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qk, "?") if qk else "?", r.get(sk, "?") if sk else "?") for r in rd))
marks = [i for i, r in enumerate(rows) if "FusedAdam" in r[2] or "fused_adam" in r[2].lower() or "k_adamw" in r[2]]
firsts = [m for j, m in enumerate(marks) if j == 0 or marks[j] - marks[j - 1] > 8]
a, b = firsts[-2], firsts[-1]
win = rows[a:b]
t0 = win[0][0]
print("columns:", list(rd[0].keys()))
print("step: %d kernels, span %.1f us" % (len(win), (rows[b][0] - t0) / 1e3))
last_end = {}
for s, e, n, q, st in win:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    key = (q, st)
    gap = (s - last_end[key]) / 1e3 if key in last_end else 0.0
    last_end[key] = e
    print("%9.1f +%7.1f us  q=%s s=%s  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, gap, n[:70]))
