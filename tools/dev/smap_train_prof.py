#!/usr/bin/env python3
"""One captured SMAP training step at a batch of B, replayed N times (run under rocprofv3 --kernel-trace --stats to see which kernels a
step spends its time in).  usage: smap_train_prof.py [B=256] [N=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
model, cfg = bench.build("smap", dev)
x = bench.synth("smap", B, dev, seed=4000)
gt = torch.zeros(B, dtype=torch.long, device=dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)
step = model.capture_train_step(x, bench.reference_loss("smap"), opt, data_parallel=False)
for _ in range(3):
    step(x, gt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    step(x, gt)
torch.cuda.synchronize()
print("smap B=%d: %.3f ms per captured step" % (B, (time.perf_counter() - t0) / N * 1e3))
