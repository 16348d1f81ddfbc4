#!/usr/bin/env python3
"""Kernel timeline of the LAST `ms` milliseconds of a rocprofv3 --kernel-trace CSV: start offset, duration, queue, gap to the
previous kernel's end (any queue), and the idle total.  usage: trace_window.py DIR [ms] [min_gap_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ming = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))))
t_end = rows[-1][1]
win = [r for r in rows if r[0] >= t_end - ms * 1e6]
t0 = win[0][0]
busy_until, idle = win[0][0], 0
print("%d kernels in the last %.1f ms" % (len(win), ms))
for s, e, n, q in win:
    gap = (s - busy_until) / 1e3
    if gap > 0:
        idle += s - busy_until
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if gap >= ming or (e - s) / 1e3 >= 100:
        print("%9.1f +%8.1f us q=%s gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, n[:80]))
    busy_until = max(busy_until, e)
print("idle (no kernel on any queue): %.1f us of %.1f" % (idle / 1e3, (t_end - t0) / 1e3))
