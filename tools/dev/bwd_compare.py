#!/usr/bin/env python3
"""Dev check: taped step backward (forward tape + cf_flow_step_bwd_taped) against the recompute form (cf_flow_step_bwd) on the
same operands, plane by plane.  usage: bwd_compare.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from contextflow_amd.layers import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
SQ = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = "cuda"; L = _hip.lib(); P = _hip.p
for (C, H) in [(16, 16), (32, 8), (64, 4)]:
    HID, HALF, HW = 2 * C, C // 2, H * H
    g = torch.Generator().manual_seed(C)
    r = lambda *s: (torch.randn(*s, generator=g)).to(dev)
    Wm = (torch.linalg.qr(torch.randn(C, C, generator=g))[0]).contiguous().to(dev)
    t, logs = 0.1 * r(C), 0.1 * r(C)
    w1, b1 = r(HID, HALF) / HALF ** 0.5, 0.1 * r(HID)
    w2, b2 = r(HID, HID, 3, 3) / (9 * HID) ** 0.5, 0.1 * r(HID)
    w3, b3 = r(C, HID) / HID ** 0.5, 0.1 * r(C)
    ws = torch.empty(L.cf_flow_step_ws_bytes(C, H, H), device=dev, dtype=torch.uint8)
    wsb = torch.empty(L.cf_flow_step_bwd_ws_bytes(C, H, H), device=dev, dtype=torch.uint8)
    st = _hip.stream()
    _hip.call("cf_flow_step_prepare", P(Wm), P(t), P(logs), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), P(ws), C, H, H, st)
    _hip.call("cf_flow_step_bwd_prepare", P(Wm), P(logs), P(w1), P(w2), P(w3), P(wsb), C, H, H, st)
    x, gz, gld = (r(B, C // 4, 2 * H, 2 * H) if SQ else r(B, C, H, H)), r(B, C, H, H), r(B)
    nanp = lambda rows: torch.full((B, rows, HW), float("nan"), device=dev)
    z, ld = torch.empty(B, C, H, H, device=dev), torch.zeros(B, device=dev)
    y0, h1, h2 = nanp(HALF), nanp(HID), nanp(HID)
    aux = torch.full((L.cf_flow_step_tape_aux_bytes(B, C, H, H),), 255, device=dev, dtype=torch.uint8)
    _hip.call("cf_flow_step_fwd_taped", P(x), P(z), P(ld), P(ws), P(y0), P(h1), P(h2), P(aux), B, C, H, H, C * HW, SQ, st)
    gx = torch.full((B, C, H, H), float("nan"), device=dev)
    a = [nanp(C), nanp(HID), nanp(HID), nanp(C)]
    _hip.call("cf_flow_step_bwd_taped", P(gz), P(gld), P(wsb), P(aux), P(gx), *[P(v) for v in a], B, C, H, H, st)
    gx2 = torch.full((B, C, H, H), float("nan"), device=dev)
    r3 = [nanp(HALF), nanp(HID), nanp(HID)]
    b = [nanp(C), nanp(HID), nanp(HID), nanp(C)]
    _hip.call("cf_flow_step_bwd", P(x), P(gz), P(gld), P(ws), P(wsb), P(gx2), *[P(v) for v in r3], *[P(v) for v in b], B, C, H, H, C * HW, SQ, st)
    torch.cuda.synchronize()
    names = ["s_gh", "s_gh2", "s_gh1", "s_gy"]
    out = ["C=%d B=%d" % (C, B)]
    for n, u, v in [("gx", gx, gx2)] + list(zip(names, a, b)) + [("y0", y0, r3[0]), ("h1", h1, r3[1]), ("h2", h2, r3[2])]:
        d = (u - v).abs()
        bad = ~torch.isfinite(u)
        out.append("%s: max|d| %.2e nonfinite(taped) %d nonfinite(recompute) %d" % (n, d[torch.isfinite(d)].max().item() if torch.isfinite(d).any() else float("nan"),
                                                                                  int(bad.sum()), int((~torch.isfinite(v)).sum())))
        if bad.any():
            idx = bad.nonzero()[0].tolist(); out.append("  first bad index %s" % idx)
    print("\n   ".join(out))
