#!/usr/bin/env python3
"""Training-step throughput (forward + hand-written backward + AdamW) of the cifar10 conv flow on one GPU, and the
worst relative gradient error against nothing (timing only).  usage: train_bench.py [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa
from contextflow_amd.layers import flowsequential as _fs

if os.environ.get("CF_TAPE_PLANES") is not None:     # A/B: 0 = the backward recomputes the conditioner planes
    _fs.TAPE_PLANES = os.environ["CF_TAPE_PLANES"] != "0"

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
name = sys.argv[3] if len(sys.argv) > 3 else "cifar10"
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.rand(B, *ds, device=dev) if name in ("smap", "atm", "msl", "smd") else torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
with torch.no_grad():
    model(x[:256])                                   # ActNorm init
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
dim_inv = 1.0 / (ds[0] * ds[1] * ds[2])


def step():
    opt.zero_grad(set_to_none=True)
    _, logp = model(x)
    logp = dim_inv * logp                              # experiment_cl.py:127
    if M == 1:
        loss = -logp.mean()                            # experiment_ad.py:204-210 (anomaly detection: NLL)
    else:
        loss = torch.nn.functional.cross_entropy(logp, gt) + 1e-3 * (-torch.nn.functional.logsigmoid(torch.logsumexp(logp, -1))).mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    l = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
with torch.no_grad():
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(iters):
        model(x)
    torch.cuda.synchronize(); df = (time.perf_counter() - t1) / iters
print("%s B=%d: train step %.2f ms = %.0f samples/s (loss %.4f); forward only %.2f ms = %.0f samples/s; bwd/fwd = %.2f" % (
    name, B, dt * 1e3, B / dt, float(l), df * 1e3, B / df, (dt - df) / df))
