import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from contextflow_amd.layers import _hip
L = _hip.lib(); P = _hip.p; dev = "cuda"
for B in (8192, 9216, 12288):
    for (MR, NR, H, taps) in [(64, 64, 8, 9), (128, 128, 4, 9), (32, 32, 16, 9), (64, 16, 8, 1)]:
        HW = H * H
        A = torch.randn(B, MR, HW, device=dev); Bm = torch.randn(B, NR, HW, device=dev)
        outs = []
        for it in range(6):
            gw = torch.empty(taps, MR, NR, device=dev); gb = torch.empty(MR, device=dev)
            wsw = torch.full((L.cf_wgrad_ws_bytes(B, MR, NR, H, H, taps) // 4,), float("nan"), device=dev)
            _hip.call("cf_wgrad", P(A), P(Bm), P(gw), P(gb), P(wsw), B, MR, NR, H, H, taps, _hip.stream())
            torch.cuda.synchronize()
            outs.append((gw.clone(), gb.clone()))
        ref = torch.einsum("bmp,bnp->mn", A.double(), Bm.double()) if taps == 1 else None
        same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs)
        fin = all(torch.isfinite(o[0]).all().item() and torch.isfinite(o[1]).all().item() for o in outs)
        print("B=%d MR=%d NR=%d %dx%d taps=%d: deterministic %s finite %s |gw|max %.3g" % (B, MR, NR, H, H, taps, same, fin, outs[0][0].abs().max().item()))
