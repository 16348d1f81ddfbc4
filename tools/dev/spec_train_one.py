#!/usr/bin/env python3
"""One specialist training configuration (cifar10, onehot + uniform encoders, --contextflow) for profiling.  usage: spec_train_one.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda:0"
name, contexts, emb, cflow = "cifar10", [15, 5], "onehot", True
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
cfg.update(generalist=False, enc_emb=emb, enc_type="uniform", contextflow=cflow)
model = cfa.create_model(cfg, ds, M, contexts=contexts).to(dev)
x = torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
ctx = torch.stack([torch.randint(0, k, (B,), device=dev) for k in contexts], 1)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-3)
dim_inv = 1.0 / (ds[0] * ds[1] * ds[2])
def step():
    opt.zero_grad(set_to_none=True)
    logp = dim_inv * model.log_prob(x, ctx)
    loss = torch.nn.functional.cross_entropy(logp, gt)
    loss.backward()
    opt.step()
    return loss.detach()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("spec train cifar10 contextflow B=%d: %.2f ms = %.0f samples/s loss %.4f" % (B, dt * 1e3, B / dt, float(l)))
