#!/usr/bin/env python3
"""Developer tool: builds contextflow_amd/build/abl/libcf_fticks.so = the product library with s_memtime probes at the
phase boundaries of k_flow_step (patched copies of cf_step.hip / cf_step_common.h; the product sources are untouched).
Workgroup 0 writes its per-phase cycles over z[0..] (results of that launch are garbage).  Read with fwd_ticks.py."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = os.path.join(root, "contextflow_amd/build/abl/fwd")
os.makedirs(tmp, exist_ok=True)
TICKDEF = ("    long long tacc[12]; for (int k = 0; k < 12; ++k) tacc[k] = 0; long long tp = __builtin_readcyclecounter();\n"
           "#define TICK(k) { long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tp; tp = tn; }\n")
names = ["x_issue", "x_wait_lds", "p0_zstore", "p1", "bar1", "p2_mfma", "bar2_h2", "p3", "epilogue"]
def rep(s, old, new, count=1):
    assert s.count(old) >= 1, old
    return s.replace(old, new, count)
h = open(os.path.join(root, "contextflow_amd/csrc/cf_step_common.h")).read()
h = rep(h, "const int (&pin)[G::PTW], int lane, int tid, float* __restrict__ dbg,\n                                                int64_t dbg_cols, int tile) {",
        "const int (&pin)[G::PTW], int lane, int tid, float* __restrict__ dbg,\n                                                int64_t dbg_cols, int tile, long long (&tacc)[12], long long& tp) {\n#define TICK(k) { long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tp; tp = tn; }\n")
h = rep(h, "    __syncthreads();                 // h1 complete: the 3x3 taps read neighbouring waves' columns\n",
        "    TICK(3);\n    __syncthreads();                 // h1 complete\n    TICK(4);\n")
h = rep(h, "        __syncthreads();                 // every wave has finished reading h1 (taps cross pixel tiles)\n",
        "        TICK(5);\n        __syncthreads();                 // every wave has finished reading h1\n")
h = rep(h, "    // ================= phase 3: h = NN.4 h2 + b ; affine map ; log-det        (coupling.py:28,52-66)\n", "    TICK(6);\n")
h = rep(h, "    dense_phase<G, G::KS3, G::NG3, RT03>(acc3, reinterpret_cast<const float4*>(wsl + G::OFF_A3), H1, pix, lane);\n}\n",
        "    dense_phase<G, G::KS3, G::NG3, RT03>(acc3, reinterpret_cast<const float4*>(wsl + G::OFF_A3), H1, pix, lane);\n    TICK(7);\n#undef TICK\n}\n")
open(os.path.join(tmp, "cf_step_common.h"), "w").write(h)
s = open(os.path.join(root, "contextflow_amd/csrc/cf_step.hip")).read()
s = rep(s, "    float4 xr[XI];\n    const int tile = blockIdx.x;\n    x_load<G, SQ>(xr, x, xbs, tile, B, wave, lane);\n",
        "    float4 xr[XI];\n    const int tile = blockIdx.x;\n" + TICKDEF + "    x_load<G, SQ>(xr, x, xbs, tile, B, wave, lane);\n    TICK(0);\n")
s = rep(s, "        x_to_lds<G, SQ>(xr, H1, wave, lane);\n\n        // ================= phase 0", "        x_to_lds<G, SQ>(xr, H1, wave, lane);\n        TICK(1);\n\n        // ================= phase 0")
s = rep(s, "        f32x16 acc3[RT03][PTW];\n        conditioner_net<G>(acc3, lds, wsl, pix, pin, lane, tid, dbg, dbg_cols, tile);\n",
        "        TICK(2);\n        f32x16 acc3[RT03][PTW];\n        conditioner_net<G>(acc3, lds, wsl, pix, pin, lane, tid, dbg, dbg_cols, tile, tacc, tp);\n")
s = rep(s, "    conditioner_net<G>(acc3, lds, ws, pix, pin, lane, tid, nullptr, 0, tile);\n",
        "    long long tdum[12]; long long tpd = 0;\n    conditioner_net<G>(acc3, lds, ws, pix, pin, lane, tid, nullptr, 0, tile, tdum, tpd);\n")
tail = "                ldj_acc[b0 + tid] += wsl[0] + sum;\n            }\n            __syncthreads();\n        }\n    }\n}\n"
s = rep(s, tail, tail[:-2] + "    TICK(8);\n    __syncthreads();\n    if (blockIdx.x == 0 && lane == 0) for (int k = 0; k < 12; ++k) z[wave * 12 + k] = (float)tacc[k];\n}\n")
open(os.path.join(tmp, "cf_step.hip"), "w").write(s)
open(os.path.join(tmp, "ticks_names.txt"), "w").write(" ".join(names))
srcs = [x for x in glob.glob(os.path.join(root, "contextflow_amd/csrc/*.hip")) if not x.endswith("cf_step.hip")]
# the patched header shadows the product one only for the patched cf_step.hip (quote-include resolves next to the file)
objs = []
for src in [os.path.join(tmp, "cf_step.hip")] + srcs:
    obj = os.path.join(tmp, os.path.basename(src) + ".o")
    inc = ["-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "contextflow_amd/csrc")]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-comment"] + inc + ["-c", src, "-o", obj]
    objs.append((cmd, obj))
procs = [subprocess.Popen(c) for c, _ in objs]
if any(p.wait() for p in procs):
    sys.exit(1)
sys.exit(subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(root, "contextflow_amd/build/abl/libcf_fticks.so")] + [o for _, o in objs]).returncode)
