#!/usr/bin/env python3
"""Micro-benchmark of the fused step kernel alone (GPU box).  usage: step_bench.py [B] ; prints TFLOP/s per shape
and experiment flag (bit0 stagger, bit1 one workgroup per CU; flags>>8 = stagger units of 64*64 cycles)."""
import ctypes
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa
from contextflow_amd.layers import _hip

L = cfa.layers
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
lib = _hip.lib()
fn = lib.cf_flow_step_fwd_debug
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
FLOP = None
dev = "cuda:0"
variants = [("base", 0)]
for a in sys.argv[2:]:
    variants.append((a, int(a, 0)))
for C, H, W, nslots in ((8, 16, 16, 4), (16, 16, 16, 4), (32, 8, 8, 2), (64, 4, 4, 2)):
    torch.manual_seed(0)
    conv, act, cpl = L.Conv1x1((C, H, W)).to(dev), L.ActNorm((C, H, W)).to(dev), L.Coupling(C, (3, 3), (1, 1)).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    ws = torch.empty(lib.cf_flow_step_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
    f, pp = _hip.f32, _hip.p
    c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
    _hip.call("cf_flow_step_prepare", pp(f(conv.NN.detach())), pp(f(act.NN_t.detach())), pp(f(act.NN_logs.detach())),
              pp(f(c1.weight.detach())), pp(f(c1.bias.detach())), pp(f(c2.weight.detach())), pp(f(c2.bias.detach())),
              pp(f(c3.weight.detach())), pp(f(c3.bias.detach())), pp(ws), C, H, W, _hip.stream())
    z = torch.empty_like(x)
    ldj = torch.zeros(B, device=dev)
    res = []
    for name, flags in variants:
        if flags & 1:
            flags |= nslots << 4
        try:
            for _ in range(3):
                _hip.check(fn(pp(x), pp(z), pp(ldj), pp(ws), B, C, H, W, C * H * W, 0, None, flags, _hip.stream()))
        except RuntimeError:
            res.append("%s: n/a" % name)         # this shape has no such variant
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            _hip.check(fn(pp(x), pp(z), pp(ldj), pp(ws), B, C, H, W, C * H * W, 0, None, flags, _hip.stream()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res.append("%s: %.3f ms %.1f TF" % (name, ms, B * 80 * H * W * C * C / ms / 1e9))
    print("C%d B%d | " % (C, B) + " | ".join(res), flush=True)
