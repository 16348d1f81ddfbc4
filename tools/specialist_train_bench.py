#!/usr/bin/env python3
"""Specialist training step: contextflow (frozen generalist, CN nets + prior embeddings train) and, for the conv flows,
without contextflow (README.md:56: every parameter trains).  usage: [B] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
for name, contexts, emb, cflow in (("cifar10", [15, 5], "onehot", True), ("mnist", [64], "eye", True), ("smap", [55], "onehot", True),
                                   ("atm", [68], "onehot", True), ("mnist", [64], "eye", False), ("cifar10", [15, 5], "eye", False)):
    torch.manual_seed(0)
    cfg, ds, M = cfa.preset_config(name)
    cfg.update(generalist=False, enc_emb=emb, enc_type="uniform", contextflow=cflow)
    model = cfa.create_model(cfg, ds, M, contexts=contexts).to(dev)
    x = torch.rand(B, *ds, device=dev) if name in ("smap", "atm") else torch.randint(0, 256, (B, *ds), device=dev).float()
    gt = torch.randint(0, M, (B,), device=dev)
    ctx = torch.stack([torch.randint(0, k, (B,), device=dev) for k in contexts], 1)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3)
    dim_inv = 1.0 / (ds[0] * ds[1] * ds[2])
    losses = []

    def step():
        opt.zero_grad(set_to_none=True)
        logp = dim_inv * model.log_prob(x, ctx)
        loss = torch.nn.functional.cross_entropy(logp, gt) if M > 1 else -logp.mean()
        loss.backward()
        opt.step()
        return loss.detach()
    for _ in range(2):
        losses.append(float(step()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        losses.append(float(step()))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("%s specialist contextflow=%s (%d trainable tensors) B=%d: train step %.2f ms = %.0f samples/s; loss %.4f -> %.4f" % (
        name, cflow, len(params), B, dt * 1e3, B / dt, losses[0], losses[-1]))
