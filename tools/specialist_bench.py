#!/usr/bin/env python3
"""Throughput of the specialist (context-conditioned) forward, layer by layer.
usage: specialist_bench.py [B] [iters] [only: index of one configuration]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
for idx, (name, contexts, emb, cf, typ) in enumerate((("cifar10", [15, 5], "onehot", True, "uniform"), ("cifar10", [15, 5], "eye", False, "uniform"),
                                     ("cifar10", [15, 5], "onehot", False, "vardeq"), ("mnist", [64], "eye", True, "uniform"),
                                     ("smap", [55], "onehot", True, "uniform"))):
    if only is not None and idx != only:
        continue
    torch.manual_seed(0)
    cfg, ds, M = cfa.preset_config(name)
    cfg.update(generalist=False, enc_emb=emb, enc_type=typ, contextflow=cf)
    model = cfa.create_model(cfg, ds, M, contexts=contexts).to(dev).eval()
    for p in model.parameters():                       # CN nets are zero-initialised: perturb so that they do something
        if p.abs().max() == 0:
            p.data.normal_(0, 0.02)
    x = torch.rand(B, *ds, device=dev) if name == "smap" else torch.randint(0, 256, (B, *ds), device=dev).float()
    ctx = torch.stack([torch.randint(0, k, (B,), device=dev) for k in contexts], 1)
    with torch.no_grad():
        for _ in range(2):
            model(x, ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            _, logp = model(x, ctx)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("%s enc=%s+%s contextflow=%s B=%d: %.2f ms = %.0f samples/s (logp finite: %s)" % (
        name, emb, typ, cf, B, dt * 1e3, B / dt, bool(torch.isfinite(logp).all())))
