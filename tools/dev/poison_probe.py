#!/usr/bin/env python3
"""Uninitialised-read probe: the eager training loop with the caching allocator's free blocks filled with NaN (or zeros)
between the steps.  Any kernel that reads a word nobody wrote shows up as a NaN / a different loss.
usage: poison_probe.py [name] [B] [nan|zero|big]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
name = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
mode = sys.argv[3] if len(sys.argv) > 3 else "nan"
dev = torch.device("cuda", 0)
fillv = {"nan": float("nan"), "zero": 0.0, "big": 3.0e38}[mode]

def poison():
    torch.cuda.synchronize()
    xs = []
    for sz in [128, 512, 2048, 8192, 32768, 131072, 262144, 524288, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26]:
        for _ in range(6):
            xs.append(torch.full((sz // 4,), fillv, device=dev))
    torch.cuda.synchronize()
    del xs

cfg, ds, M = cfa.preset_config(name)
g = torch.Generator().manual_seed(1)
xg = (torch.rand(B, *ds, generator=g) if name == "smap" else torch.randint(0, 256, (B, *ds), generator=g).float()).to(dev)
yg = torch.randint(0, max(M, 1), (B,), generator=g).to(dev)
inv = 1.0 / xg[0].numel()
loss_fn = lambda lp, y: torch.nn.functional.cross_entropy(lp * inv, y) if M > 1 else -(lp * inv).mean()
torch.manual_seed(0)
m = cfa.create_model(cfg, ds, M).to(dev)
with torch.no_grad():
    m(xg)
m.train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True, capturable=True)
losses = []
for it in range(4):
    poison()
    torch.manual_seed(100 + it)
    opt.zero_grad(set_to_none=True)
    l = loss_fn(m.log_prob(xg), yg)
    l.backward()
    bad = [k for k, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    if bad:
        print("step %d: non-finite gradients in %d tensors, e.g. %s" % (it, len(bad), bad[:6]))
    opt.step()
    losses.append(float(l.detach()))
print(name, B, mode, "losses", ["%.9g" % v for v in losses])
