#!/usr/bin/env python3
"""Numerics experiment (CPU): would a Winograd F(2x2,3x3) form of the reflect-padded 3x3 convolution of the coupling nets,
evaluated in fp32, still meet the 1e-5 bits/dim bar?  Runs the oracle's flow on the committed end-to-end fixtures with the
3x3 replaced by an fp32 Winograd restatement (weights transformed in fp64, rounded once) and prints the bits/dim error
against the reference's fp32 and fp64 outputs, next to the direct fp32 convolution's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import flow_oracle as fo
from tests.helpers import load_e2e, e2e_inputs, bpd

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd3x3(h, w, b):
    """h (B,Ci,H,W) fp32 (reflect padded here), w (Co,Ci,3,3) -> (B,Co,H,W); F(2x2,3x3), fp32 arithmetic in the order a
    kernel would use: input transform (adds), per-position channel contraction (fp32 accumulate), output transform."""
    B, Ci, H, W = h.shape
    U = torch.einsum("xa,oiab,yb->xyoi", G, w.double(), G).float()               # (4,4,Co,Ci), rounded once
    hp = F.pad(h, (1, 1, 1, 1), mode="reflect")
    d = hp.unfold(2, 4, 2).unfold(3, 4, 2)                                         # (B,Ci,H/2,W/2,4,4)
    V = torch.einsum("xa,ncijab->ncijxb", BT, d)
    V = torch.einsum("ncijxb,yb->ncijxy", V, BT)                                   # (B,Ci,th,tw,4,4)
    M = torch.einsum("xyoc,ncijxy->noijxy", U, V)
    Y = torch.einsum("px,noijxy->noijpy", AT, M)
    Y = torch.einsum("noijpy,qy->noijpq", Y, AT)                                   # (B,Co,th,tw,2,2)
    out = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, H, W)
    return out + b.view(1, -1, 1, 1)


def coupling_net_wino(x0, p, prefix, pad):
    h = F.relu(F.conv2d(x0, p[prefix + "NN.0.weight"], p[prefix + "NN.0.bias"]))
    if h.dtype == torch.float32 and pad == (1, 1):
        h = F.relu(winograd3x3(h, p[prefix + "NN.2.weight"], p[prefix + "NN.2.bias"]))
    else:
        if pad[0] or pad[1]:
            h = F.pad(h, (pad[1], pad[1], pad[0], pad[0]), mode="reflect")
        h = F.relu(F.conv2d(h, p[prefix + "NN.2.weight"], p[prefix + "NN.2.bias"]))
    return F.conv2d(h, p[prefix + "NN.4.weight"], p[prefix + "NN.4.bias"])


direct = fo.coupling_net
for name in ("mnist", "cifar10"):
    for tag in (None, "stress", "extreme"):
        ops, _, M, params, fx = load_e2e(name, tag)
        x, u, eps = e2e_inputs(name, fx)
        ref = torch.from_numpy(fx["logp"])
        ref64 = torch.from_numpy(fx["logp_f64"]) if "logp_f64" in fx else None
        res = {}
        for label, fn in (("direct", direct), ("winograd", coupling_net_wino)):
            fo.coupling_net = fn
            _, logp = fo.flow_forward(ops, params, x, u, eps)
            e32 = (bpd(logp, name) - bpd(ref, name)).abs().max().item()
            e64 = (bpd(logp, name) - bpd(ref64, name)).abs().max().item() if ref64 is not None else float("nan")
            res[label] = (e32, e64)
        fo.coupling_net = direct
        floor = (bpd(ref, name) - bpd(ref64, name)).abs().max().item() if ref64 is not None else float("nan")
        print("%-8s %-8s B=%d  |bpd err| vs ref fp32 / vs ref fp64:  direct %.2e / %.2e   winograd %.2e / %.2e   (reference fp32 vs its fp64: %.2e)" % (
            name, tag or "-", x.shape[0], res["direct"][0], res["direct"][1], res["winograd"][0], res["winograd"][1], floor))
