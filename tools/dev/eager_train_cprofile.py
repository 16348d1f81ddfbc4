#!/usr/bin/env python3
"""Host-side profile (cProfile) of the EAGER training step at a small batch: where the Python time of a step goes.
usage: eager_train_cprofile.py [name] [B] [steps]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
name = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
dim_inv = 1.0 / (ds[0] * ds[1] * ds[2])
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(dim_inv * model.log_prob(x), gt)
    loss.backward()
    opt.step()
with torch.no_grad():
    model(x)
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print("eager step %.3f ms" % ((time.perf_counter() - t0) / steps * 1e3))
torch.autograd.set_multithreading_enabled(False)       # the backward runs in this thread: visible to cProfile
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
