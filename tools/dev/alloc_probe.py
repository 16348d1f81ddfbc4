#!/usr/bin/env python3
"""Dev check (GPU): host time, device allocations and frees of the caching allocator per EAGER cifar10 training step.
usage: alloc_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
B = 16384
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config("cifar10")
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
with torch.no_grad():
    model(x[:256])
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
inv = 1.0 / (ds[0] * ds[1] * ds[2])
def step():
    opt.zero_grad(set_to_none=True)
    _, logp = model(x)
    loss = torch.nn.functional.cross_entropy(logp * inv, gt)
    loss.backward()
    opt.step()
for i in range(8):
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print("step %d: host %.1f ms, total %.1f ms, device_alloc +%d, device_free +%d, retries +%d, reserved %.1f GB, peak alloc %.1f GB" % (
        i, (t1 - t0) * 1e3, (t2 - t0) * 1e3, s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"],
        s1["num_alloc_retries"] - s0["num_alloc_retries"], s1["reserved_bytes.all.current"] / 2**30, s1["allocated_bytes.all.peak"] / 2**30), flush=True)
# the same step, host free to run ahead (no synchronisation between steps): wall time per step over three groups of ten
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        step()
    th = time.perf_counter()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print("10 steps, no sync between: %.2f ms per step (host done after %.1f ms), reserved %.1f GB, peak alloc %.1f GB, device_alloc total %d" % (
        (t1 - t0) * 100, (th - t0) * 1e3, s1["reserved_bytes.all.current"] / 2**30, s1["allocated_bytes.all.peak"] / 2**30, s1["num_device_alloc"]), flush=True)
