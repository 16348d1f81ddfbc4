#!/usr/bin/env python3
"""Dev soak (GPU): bitwise repeatability of the evaluation forward and of a training backward pass over repeated runs at
several batch sizes - races and unguarded hardware hazards show up as run-to-run differences (the round-2 store hazard did:
tools/micro/store_hazard.hip).  usage: dev_soak.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import contextflow_amd as cfa
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = "cuda:0"
bad = 0
for name in ("cifar10", "mnist", "smap"):
    torch.manual_seed(0)
    cfg, ds, M = cfa.preset_config(name)
    model = cfa.create_model(cfg, ds, M).to(dev)
    model.auto_graph = False
    mk = (lambda B: torch.rand(B, *ds, device=dev)) if M == 1 else (lambda B: torch.randint(0, 256, (B, *ds), device=dev).float())
    with torch.no_grad():
        model(mk(256))
    for B in (300, 1100, 4100, 16384):
        x = mk(B)
        gt = torch.randint(0, max(M, 2), (B,), device=dev) % M
        ref = None
        with torch.no_grad():
            for it in range(R):
                torch.manual_seed(3)
                lp = model(x)[1]
                if ref is None: ref = lp.clone()
                elif not torch.equal(ref, lp): bad += 1; print("EVAL differs:", name, B, it, (ref - lp).abs().max().item())
        gref = None
        for it in range(max(2, R // 3)):
            model.zero_grad(set_to_none=True)
            torch.manual_seed(3)
            lp = model(x)[1]
            loss = -lp.mean() / 1000.0 if M == 1 else torch.nn.functional.cross_entropy(lp / 1000.0, gt)
            loss.backward()
            g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
            if not torch.isfinite(g).all(): bad += 1; print("non-finite gradient:", name, B, it)
            if gref is None: gref = g.clone()
            elif not torch.equal(gref, g): bad += 1; print("GRAD differs:", name, B, it, (gref - g).abs().max().item(), "scale", gref.abs().max().item())
        print("%s B=%d: eval x%d, train x%d repeat bitwise" % (name, B, R, max(2, R // 3)), "OK" if bad == 0 else "(failures so far: %d)" % bad, flush=True)
print("soak failures:", bad)
sys.exit(1 if bad else 0)
