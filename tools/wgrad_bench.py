#!/usr/bin/env python3
"""Times cf_wgrad (split-K MFMA weight-gradient GEMM) on the shapes of the cifar10 conv flow.  usage: wgrad_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from contextflow_amd.layers import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda"
L = _hip.lib()
for (H, MR, NR, taps) in [(16, 32, 32, 9), (8, 64, 64, 9), (4, 128, 128, 9), (16, 16, 32, 1), (16, 32, 8, 1), (16, 16, 16, 1),
                          (8, 32, 64, 1), (8, 64, 16, 1), (8, 32, 32, 1), (4, 64, 128, 1), (4, 128, 32, 1), (4, 64, 64, 1)]:
    A = torch.randn(B, MR, H * H, device=dev)
    Bm = torch.randn(B, NR, H * H, device=dev)
    gw = torch.empty(taps, MR, NR, device=dev)
    gb = torch.empty(MR, device=dev)
    ws = torch.empty(L.cf_wgrad_ws_bytes(B, MR, NR, H, H, taps), device=dev, dtype=torch.uint8)
    run = lambda: _hip.call("cf_wgrad", _hip.p(A), _hip.p(Bm), _hip.p(gw), _hip.p(gb), _hip.p(ws), B, MR, NR, H, H, taps, _hip.stream())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * MR * NR * taps * H * H * B
    by = 4.0 * (MR + NR) * H * H * B
    print("H=%2d MR=%3d NR=%3d taps=%d: %7.1f us  %6.1f TFLOP/s  %5.2f TB/s (operands)" % (H, MR, NR, taps, us, fl / us / 1e6, by / us / 1e6))
