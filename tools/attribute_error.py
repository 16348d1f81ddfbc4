#!/usr/bin/env python3
"""Where does a fixture's bits/dim distance come from?  Runs the GPU layers one by one against the fp64 oracle trace:
  * propagated: every layer gets the GPU's own running activation (what the model does);
  * local:      every layer gets the fp64 trace's input rounded to fp32 (the layer's OWN rounding, nothing propagated);
next to the same two numbers for the fp32 oracle (= the reference's fp32 arithmetic, bit for bit on the fixtures).
Finally the prior on the GPU's z, on the fp32 oracle's z and on the exact z: the part of the distance that is the prior's
own rounding vs the part that is the activation error amplified by 1/sigma.
usage: attribute_error.py [name=smap] [tag=extreme]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import build_model, set_noise
from tests.helpers import load_e2e, e2e_inputs
import oracle.flow_oracle as fo

name = sys.argv[1] if len(sys.argv) > 1 else "smap"
tag = sys.argv[2] if len(sys.argv) > 2 else "extreme"
DEV = "cuda:0"
ops, _, M, params, fx = load_e2e(name, None if tag == "none" else tag)
x, u, eps = e2e_inputs(name, fx)
tr32, tr64 = [], []
z32, lp32 = fo.flow_forward(ops, params, x, u, eps, trace=tr32)
p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
z64, lp64 = fo.flow_forward(ops, p64, x.double(), None if u is None else u.double(), [e.double() for e in eps], trace=tr64)
D = {"smap": 200, "mnist": 1024, "cifar10": 3072}[name] * math.log(2.0)

model = build_model(name, params)
model.fused = False
set_noise(model, u, eps)
mods = list(model.sequence_modules)
assert len(mods) == len(tr64), (len(mods), len(tr64))
rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
xin = x.to(DEV)
prev64 = x.double()
print("%3s %-14s | z rel err vs fp64: gpu-prop  o32-prop | gpu-local | ldj abs err: gpu  o32" % ("i", "layer"))
with torch.no_grad():
    for i, (m, t32, t64) in enumerate(zip(mods, tr32, tr64)):
        zg, lg = m(xin, None)
        # local: the layer on the exact input (rounded)
        if prev64.shape == xin.shape:
            zl, _ = m(prev64.float().to(DEV), None)
            loc = rel(zl, t64[2]) if zl.shape == t64[2].shape else float("nan")
        else:
            loc = float("nan")
        l64 = t64[3]
        print("%3d %-14s | %.2e  %.2e | %.2e | %.2e  %.2e" % (
            i, t64[0], rel(zg, t64[2]), rel(t32[2], t64[2]), loc,
            (lg.double().cpu().reshape(l64.shape) - l64).abs().max().item(), (t32[3].double() - l64).abs().max().item()))
        xin, prev64 = zg, t64[2]
    pg = model.dist.log_prob(xin, None).double().cpu()
    p_exact_gpu = model.dist.log_prob(z64.float().to(DEV), None).double().cpu()
pr64 = lp64 - sum((t[3] if t[3].dim() == 2 else t[3].unsqueeze(-1)) for t in tr64)
pr32 = lp32.double() - sum((t[3].double() if t[3].dim() == 2 else t[3].double().unsqueeze(-1)) for t in tr32)
print("prior, nats: gpu on gpu z %.2e | oracle32 on its z %.2e | gpu prior on exact z %.2e   (1e-5 bits/dim = %.2e nats)" % (
    (pg - pr64).abs().max().item(), (pr32 - pr64).abs().max().item(), (p_exact_gpu - pr64).abs().max().item(), 1e-5 * D))
