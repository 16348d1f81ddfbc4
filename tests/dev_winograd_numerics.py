#!/usr/bin/env python3
"""Numerics experiment (CPU): would a Winograd F(2x2,3x3) form of the reflect-padded 3x3 convolution of the coupling nets,
evaluated in fp32, still meet the 1e-5 bits/dim bar?  Runs the oracle's flow on the committed end-to-end fixtures with the
3x3 replaced by an fp32 Winograd restatement (weights transformed in fp64, rounded once) and prints the bits/dim error
against the reference's fp32 and fp64 outputs, next to the direct fp32 convolution's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import flow_oracle as fo
from tests.helpers import load_e2e, e2e_inputs, bpd

from oracle.winograd import coupling_net_winograd as coupling_net_wino      # the restatement lives with the oracle
from oracle.winograd import coupling_net_winograd_f4                         # F(4x4,3x3) on the 16x16 / 8x8 levels (argument "f4")


direct = fo.coupling_net
for name in ("mnist", "cifar10"):
    for tag in (None, "stress", "extreme"):
        ops, _, M, params, fx = load_e2e(name, tag)
        x, u, eps = e2e_inputs(name, fx)
        ref = torch.from_numpy(fx["logp"])
        ref64 = torch.from_numpy(fx["logp_f64"]) if "logp_f64" in fx else None
        res = {}
        forms = (("direct", direct), ("winograd", coupling_net_wino)) + ((("f4", coupling_net_winograd_f4),) if "f4" in sys.argv else ())
        for label, fn in forms:
            fo.coupling_net = fn
            _, logp = fo.flow_forward(ops, params, x, u, eps)
            e32 = (bpd(logp, name) - bpd(ref, name)).abs().max().item()
            e64 = (bpd(logp, name) - bpd(ref64, name)).abs().max().item() if ref64 is not None else float("nan")
            res[label] = (e32, e64)
        fo.coupling_net = direct
        floor = (bpd(ref, name) - bpd(ref64, name)).abs().max().item() if ref64 is not None else float("nan")
        print("%-8s %-8s B=%d  |bpd err| vs ref fp32 / vs ref fp64:  direct %.2e / %.2e   winograd %.2e / %.2e   (reference fp32 vs its fp64: %.2e)" % (
            name, tag or "-", x.shape[0], res["direct"][0], res["direct"][1], res["winograd"][0], res["winograd"][1], floor)
              + ("   F(4x4,3x3) on 16x16 / 8x8: %.2e / %.2e" % res["f4"] if "f4" in res else ""))
