#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/dev/resources.py contextflow_amd/csrc/cf_step_bwd.hip [filter]"""
import re, subprocess, sys, os
src = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-comment", "-c", src, "-o", "/tmp/_res.o",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd="/tmp")
cur = None; rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r": remark: +(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        name = t.split(":", 1)[1].strip()
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = {"name": d.replace("(anonymous namespace)::", "")}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for c in rows:
    if flt in c["name"]:
        print("%-95s VGPR %4s AGPR %4s spill %3s scratch %4s occ %s LDS %s" % (c["name"].split("(")[0][:95], c.get("VGPRs"), c.get("AGPRs"),
              c.get("VGPRs Spill"), c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"), c.get("LDS Size [bytes/block]")))
