#!/usr/bin/env python3
"""Per-phase cycles of k_flow_step_rs (workgroup 0, every wave) from the probe build `tools/dev/make_abl.py ticks
--only=cf_step.hip -DCF_RS_TICKS`.  usage: CONTEXTFLOW_HIP_LIB=contextflow_amd/build/abl/libcf_abl_ticks.so rs_ticks.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
from contextflow_amd.layers import _hip
L = cfa.layers
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda:0"
names = ["x load+bar", "phase 0+bar", "y0 store", "phase 1+bar", "phase 2", "wait others", "combine+relu+bar", "phase 3+bar", "epilogue"]
for C, H in ((32, 8), (64, 4)):
    torch.manual_seed(0)
    conv, act, cpl = L.Conv1x1((C, H, H)).to(dev), L.ActNorm((C, H, H)).to(dev), L.Coupling(C, (3, 3), (1, 1)).to(dev)
    lib = _hip.lib()
    ws = torch.empty(lib.cf_flow_step_ws_bytes(C, H, H), device=dev, dtype=torch.uint8)
    f, pp = _hip.f32, _hip.p
    c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
    _hip.call("cf_flow_step_prepare", pp(f(conv.NN.detach())), pp(f(act.NN_t.detach())), pp(f(act.NN_logs.detach())),
              pp(f(c1.weight.detach())), pp(f(c1.bias.detach())), pp(f(c2.weight.detach())), pp(f(c2.bias.detach())),
              pp(f(c3.weight.detach())), pp(f(c3.bias.detach())), pp(ws), C, H, H, _hip.stream())
    x = torch.randn(B, C, H, H, device=dev)
    z, ldj = torch.empty_like(x), torch.zeros(B, device=dev)
    for _ in range(5):
        _hip.call("cf_flow_step_fwd", pp(x), pp(z), pp(ldj), pp(ws), B, C, H, H, C * H * H, 0, _hip.stream())
    torch.cuda.synchronize()
    t = z.flatten()[:8 * 16].reshape(8, 16)[:, :9].cpu()
    print("C=%d %dx%d B=%d: cycles per phase (rows = waves 0..7)" % (C, H, H, B))
    print("   " + " | ".join("%-16s" % n for n in names))
    for w in range(8):
        print("w%d " % w + " | ".join("%-16d" % int(v) for v in t[w]))
    print("   total of wave 0: %d cycles" % int(t[0].sum()))
