#!/usr/bin/env python3
"""Dev check: the Winograd-form step kernel (cf_flow_step_fwd_debug, variant 4) against the production step kernel on the
same operands, and their kernel times.  usage: wino_check.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from contextflow_amd.layers import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda"
L = _hip.lib()
fn = L.cf_flow_step_fwd_debug
fn.restype = ctypes.c_int
P = _hip.p
for (C, H) in [(32, 8), (64, 4), (16, 16), (8, 16)]:
    HID, HALF, HW = 2 * C, C // 2, H * H
    g = torch.Generator().manual_seed(C)
    r = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    Wm = (torch.linalg.qr(torch.randn(C, C, generator=g))[0]).contiguous().to(dev)
    t, logs = 0.1 * r(C), 0.1 * r(C)
    w1, b1 = r(HID, HALF) / HALF ** 0.5, 0.1 * r(HID)
    w2, b2 = r(HID, HID, 3, 3) / (9 * HID) ** 0.5, 0.1 * r(HID)
    w3, b3 = r(C, HID) / HID ** 0.5, 0.1 * r(C)
    ws = torch.empty(L.cf_flow_step_ws_bytes(C, H, H), device=dev, dtype=torch.uint8)
    st = _hip.stream()
    _hip.call("cf_flow_step_prepare", P(Wm), P(t), P(logs), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), P(ws), C, H, H, st)
    x = r(B, C, H, H)
    outs = {}
    for name, v in (("production", None), ("variant %d" % variant, variant)):
        z, ld = torch.empty_like(x), torch.zeros(B, device=dev)
        def run():
            if v is None:
                _hip.call("cf_flow_step_fwd", P(x), P(z), P(ld), P(ws), B, C, H, H, C * HW, 0, st)
            else:
                _hip.check(fn(P(x), P(z), P(ld), P(ws), B, C, H, H, ctypes.c_int64(C * HW), 0, None, v << 16, st), "dbg")
        try:
            run()
        except RuntimeError as e:
            print("C=%d: %s n/a (%s)" % (C, name, str(e)[:60])); continue
        torch.cuda.synchronize()
        outs[name] = (z.clone(), ld.clone())
        for _ in range(20):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print("C=%d %dx%d B=%d %-12s %.1f us = %.1f TFLOP/s algorithmic (%.2f)" % (C, H, H, B, name, us, 80.0 * C * C * HW * B / us * 1e-6,
                                                                                 80.0 * C * C * HW * B / us * 1e-6 / 157.3))
    if len(outs) == 2:
        (z0, l0), (z1, l1) = outs.values()
        print("      max |dz| %.3e (scale %.2f)   max |d ldj| %.3e (scale %.1f)" % ((z0 - z1).abs().max(), z0.abs().max(), (l0 - l1 / 51 * 1).abs().max() if False else (l0 - l1).abs().max(), l0.abs().max()))
