#!/usr/bin/env python3
"""Kernel-only timing of the taped step-backward kernel (cf_flow_step_bwd_taped) per level of the cifar10 flow.
Algorithmic work of one sample-step: 80 C^2 HW flop (NN.4^T 4 C^2, transposed 3x3 72 C^2, NN.0^T 2 C^2, W'^T 2 C^2).  usage: bwd_bench.py [B] [iters]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextflow_amd.layers import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = "cuda"
L = _hip.lib()
P, I = _hip.p, ctypes.c_int
for (C, H) in [(16, 16), (32, 8), (64, 4)]:
    HID, HALF, HW = 2 * C, C // 2, H * H
    g = torch.Generator().manual_seed(C)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)
    Wm = (torch.linalg.qr(torch.randn(C, C, generator=g))[0]).contiguous().to(dev)
    t, logs = r(C), r(C)
    w1, b1, w2, b2, w3, b3 = r(HID, HALF), r(HID), r(HID, HID, 3, 3), r(HID), r(C, HID), r(C)
    ws = torch.empty(L.cf_flow_step_ws_bytes(C, H, H), device=dev, dtype=torch.uint8)
    wsb = torch.empty(L.cf_flow_step_bwd_ws_bytes(C, H, H), device=dev, dtype=torch.uint8)
    st = _hip.stream()
    _hip.call("cf_flow_step_prepare", P(Wm), P(t), P(logs), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), P(ws), C, H, H, st)
    _hip.call("cf_flow_step_bwd_prepare", P(Wm), P(logs), P(w1), P(w2), P(w3), P(wsb), C, H, H, st)
    x, gz, gld = r(B, C, H, H), r(B, C, H, H), r(B)
    # a real tape: the training forward of the same step
    z, ld = torch.empty_like(x), torch.zeros(B, device=dev)
    y0, h1, h2 = (torch.empty(B, rows, HW, device=dev) for rows in (HALF, HID, HID))
    aux = torch.empty(L.cf_flow_step_tape_aux_bytes(B, C, H, H), device=dev, dtype=torch.uint8)
    _hip.call("cf_flow_step_fwd_taped", P(x), P(z), P(ld), P(ws), P(y0), P(h1), P(h2), P(aux), B, C, H, H, C * HW, 0, st)
    new = lambda rows: torch.empty(B, rows, HW, device=dev)
    gx = torch.empty(B, C, H, H, device=dev)
    s_gh, s_gh2, s_gh1, s_gy = new(C), new(HID), new(HID), new(C)
    run = lambda: _hip.call("cf_flow_step_bwd_taped", P(gz), P(gld), P(wsb), P(aux), P(gx), P(s_gh), P(s_gh2), P(s_gh1), P(s_gy),
                            B, C, H, H, 0, st)
    for _ in range(30):                     # clocks up
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 80.0 * C * C * HW * B
    print("C=%d %dx%d B=%d: %.1f us/launch = %.1f TFLOP/s algorithmic (%.2f of the fp32 MFMA peak)" % (C, H, H, B, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3))
