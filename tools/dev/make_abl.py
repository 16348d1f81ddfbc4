#!/usr/bin/env python3
"""Developer tool: builds contextflow_amd/build/abl/libcf_abl_<tag>.so = the product library compiled with extra -D flags
(timing-only ablations guarded by #ifdef CF_ABL_* in the sources).  Use with CONTEXTFLOW_HIP_LIB=<that path>.
usage: make_abl.py <tag> [--only=cf_step.hip] -DCF_ABL_X [...]"""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(root, "contextflow_amd/build/abl"); os.makedirs(out, exist_ok=True)
srcs = sorted(glob.glob(os.path.join(root, "contextflow_amd/csrc/*.hip")))
only = [d[len("--only="):] for d in defs if d.startswith("--only=")]
defs = [d for d in defs if not d.startswith("--only=")]
sys.path.insert(0, root)
from contextflow_amd import build as B                       # same flags as the product build (per-file extras included)
objdir = os.path.join(out, tag); os.makedirs(objdir, exist_ok=True)
objs = []
for src in srcs:
    name = os.path.basename(src)
    obj = os.path.join(objdir, name[:-4] + ".o")
    prod = os.path.join(B.OBJ, name[:-4] + ".o")
    if only and name not in only and os.path.exists(prod):   # --only=cf_step.hip: every other file's product object is reused
        objs.append(prod)
        continue
    cmd = ["hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(name, []) + defs + ["-c", src, "-o", obj]
    if subprocess.run(cmd).returncode:
        sys.exit(1)
    objs.append(obj)
lib = os.path.join(out, "libcf_abl_%s.so" % tag)
rc = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs).returncode
for junk in glob.glob(lib + ".*"):
    os.remove(junk)
sys.exit(rc)
