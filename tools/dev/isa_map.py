#!/usr/bin/env python3
"""Where in a kernel's instruction stream the barriers, scratch (spill) accesses and global memory operations sit,
with the running MFMA / VALU counts.  usage: isa_map.py file.s 'substring of mangled name' [pattern ...]"""
import re, sys
s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]; pats = sys.argv[3:] or ["scratch_", "s_barrier"]
start = next(i for i, l in enumerate(s) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
n_mfma = n_valu = n_lds = 0
for i in range(start + 1, len(s)):
    l = s[i].strip()
    if l.startswith("s_endpgm"): break
    op = l.split(" ")[0]
    if op.startswith("v_mfma"): n_mfma += 1
    elif op.startswith("v_"): n_valu += 1
    elif op.startswith("ds_"): n_lds += 1
    if any(p in l for p in pats):
        print("%6d  mfma %5d valu %5d lds %5d  %s" % (i - start, n_mfma, n_valu, n_lds, l[:90]))
print("total: lines %d mfma %d valu %d lds %d" % (i - start, n_mfma, n_valu, n_lds))
