#!/usr/bin/env python3
"""Developer tool: builds contextflow_amd/build/abl/libcf_ticks.so = the product library with s_memtime probes at the
phase boundaries of k_flow_step_bwd (patched copy of the source, the product source is untouched).  Workgroup 0 writes
its per-phase cycle sums over gx[0..] (results of that launch are garbage).  Read with tools/dev/bwd_ticks.py on the GPU."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "contextflow_amd/csrc/cf_step_bwd.hip")).read()
marks = [
    ("    // ---------------------------------------------------------------- what the data-gradient chain needs of the forward\n", None),
    ("    // ---------------------------------------------------------------- backward\n", "tape"),
    ("        // g_h2 = (NN.4^T g_h) * [h2 > 0]\n", "gz_gh"),
    ("    __syncthreads();                     // g_h2 complete", "a3t"),
    ("            __syncthreads();             // fold slots in place\n", "bar+build"),
    ("        __syncthreads();                 // everyone done reading g_h2\n", "bar+taps"),
    ("    {   // g_y0 = NN.0^T g_h1 + g_z0", "bar+gh1"),
    ("        // g_x = (e^{-logs} Wm)^T g_y\n", "a1t"),
]
out = src
names = []
for k, (m, name) in enumerate(marks):
    assert m in out, m
    if name is None:
        out = out.replace(m, m + "    long long tacc[12]; for (int k = 0; k < 12; ++k) tacc[k] = 0; long long tprev = __builtin_readcyclecounter();\n"
                          "#define TICK(k) { long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tprev; tprev = tn; }\n")
    else:
        out = out.replace(m, "    TICK(%d);\n" % len(names) + m)
        names.append(name)
tail = "        rows_store_t<G, C, C>(gx, GX, b0, B, wave, lane);\n    }\n}"
assert tail in out
names.append("a0t")
out = out.replace(tail, "        rows_store_t<G, C, C>(gx, GX, b0, B, wave, lane);\n    }\n    TICK(%d);\n    __syncthreads();\n"
                  "    if (blockIdx.x == 0 && lane == 0) for (int k = 0; k < 12; ++k) gx[wave * 12 + k] = (float)tacc[k];\n}" % (len(names) - 1))
tmp = os.path.join(root, "contextflow_amd/build/abl")
os.makedirs(tmp, exist_ok=True)
open(os.path.join(tmp, "cf_step_bwd_ticks.hip"), "w").write(out)
open(os.path.join(tmp, "ticks_names.txt"), "w").write(" ".join(names))
srcs = [s for s in glob.glob(os.path.join(root, "contextflow_amd/csrc/*.hip")) if not s.endswith("cf_step_bwd.hip")]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-comment", "-I" + os.path.join(root, "include"),
       "-I" + os.path.join(root, "contextflow_amd/csrc"), "-shared", "-o", os.path.join(tmp, "libcf_ticks.so"),
       os.path.join(tmp, "cf_step_bwd_ticks.hip")] + srcs
print(" ".join(cmd[:12]), "...")
sys.exit(subprocess.run(cmd).returncode)
