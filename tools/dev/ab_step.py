#!/usr/bin/env python3
"""A/B of library builds on the step kernels: runs tools/step_bench.py once per library (CONTEXTFLOW_HIP_LIB), each in its
own process.  usage: ab_step.py B flag[,flag...] lib_or_tag [lib_or_tag ...]   ('prod' = the in-tree library; a bare tag =
contextflow_amd/build/abl/libcf_abl_<tag>.so)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, flags, libs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
for lib in libs:
    env = dict(os.environ)
    if lib != "prod":
        env["CONTEXTFLOW_HIP_LIB"] = lib if os.path.sep in lib else os.path.join(root, "contextflow_amd/build/abl/libcf_abl_%s.so" % lib)
    print("==== %s" % lib, flush=True)
    subprocess.run([sys.executable, os.path.join(root, "tools/step_bench.py"), B] + flags, env=env)
