#!/usr/bin/env python3
"""Kernel-only timing of cf_wgrad (3x3) per level of the cifar10 flow + error against fp64 torch.  usage: wgrad3x3_bench.py [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextflow_amd.layers import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
L = _hip.lib(); P = _hip.p
for (HID, H) in [(32, 16), (64, 8), (128, 4)]:
    g = torch.Generator().manual_seed(HID)
    A = torch.randn(B, HID, H * H, generator=g).cuda(); Bm = torch.randn(B, HID, H * H, generator=g).cuda()
    gw = torch.empty(9, HID, HID, device="cuda"); gb = torch.empty(HID, device="cuda")
    ws = torch.empty(L.cf_wgrad_ws_bytes(B, HID, HID, H, H, 9), device="cuda", dtype=torch.uint8)
    run = lambda: _hip.call("cf_wgrad", P(A), P(Bm), P(gw), P(gb), P(ws), B, HID, HID, H, H, 9, _hip.stream())
    for _ in range(10): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * 9 * HID * HID * H * H * B
    # error on a slice of the batch (fp64 reference)
    n = min(B, 512)
    _hip.call("cf_wgrad", P(A), P(Bm), P(gw), P(gb), P(ws), n, HID, HID, H, H, 9, _hip.stream())
    Ad, Bd = A[:n].double().view(n, HID, H, H), torch.nn.functional.pad(Bm[:n].double().view(n, HID, H, H), (1, 1, 1, 1), mode="reflect")
    pat = Bd.unfold(2, H, 1).unfold(3, H, 1)                      # (n, HID, 3, 3, H, H)
    ref = torch.einsum("nmyx,nkabyx->abmk", Ad, pat).reshape(9, HID, HID)
    err = (gw.double() - ref).abs().max().item() / ref.abs().max().item()
    berr = (gb.double() - Ad.sum((0, 2, 3))).abs().max().item()
    print("HID=%d %dx%d B=%d: %.1f us = %.1f TFLOP/s algorithmic (%.2f of fp32 MFMA peak); rel err %.2e bias err %.2e" % (HID, H, H, B, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3, err, berr), flush=True)
