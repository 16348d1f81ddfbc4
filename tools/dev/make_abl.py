#!/usr/bin/env python3
"""Developer tool: builds contextflow_amd/build/abl/libcf_abl_<tag>.so = the product library compiled with extra -D flags
(timing-only ablations guarded by #ifdef CF_ABL_* in the sources).  Use with CONTEXTFLOW_HIP_LIB=<that path>.
usage: make_abl.py <tag> -DCF_ABL_X [...]"""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(root, "contextflow_amd/build/abl"); os.makedirs(out, exist_ok=True)
srcs = sorted(glob.glob(os.path.join(root, "contextflow_amd/csrc/*.hip")))
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-comment", "-I" + os.path.join(root, "include"),
       "-shared", "-o", os.path.join(out, "libcf_abl_%s.so" % tag)] + defs + srcs
sys.exit(subprocess.run(cmd).returncode)
