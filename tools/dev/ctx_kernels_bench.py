#!/usr/bin/env python3
"""Per-shape timings of the per-sample (context) Conv1x1 / ActNorm kernels at the cifar10 level shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from contextflow_amd.layers import _hip
dev = "cuda:0"; p, st = _hip.p, _hip.stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for C, HW in ((16, 256), (32, 64), (64, 16), (76, 72)):
    b = B if C != 76 else B // 8
    x = torch.randn(b, C, HW, device=dev); m = 0.1 * torch.randn(b, C * C, device=dev); Wm = torch.eye(C, device=dev)
    z = torch.empty_like(x); ldj = torch.empty(b, device=dev)
    t = timeit(lambda: _hip.call("cf_conv1x1_ctx", p(x), p(m), p(Wm), p(z), p(ldj), b, C, HW, C * HW, st()))
    byts = b * (2 * C * HW + C * C) * 4
    gz = torch.randn_like(x); gld = torch.randn(b, device=dev); gx = torch.empty_like(x); gm = torch.empty_like(m)
    t2 = timeit(lambda: _hip.call("cf_conv1x1_ctx_bwd", p(x), p(m), p(Wm), p(gz), p(gld), p(gx), p(gm), b, C, HW, C * HW, C * HW, st()))
    print("C=%3d HW=%3d B=%6d: conv1x1_ctx %7.1f us = %4.2f TB/s ; bwd %7.1f us = %4.2f TB/s" % (
        C, HW, b, t, byts / t / 1e6, t2, b * (4 * C * HW + 2 * C * C) * 4 / t2 / 1e6))
