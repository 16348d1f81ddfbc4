"""Kernel-level profile target for the small-batch regime: N eager no_grad forwards of one batch.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lat -- python3 tools/dev/lat_profile.py cifar10 256"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
name = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config(name)
model = cfa.create_model(cfg, ds, M).to(dev)
x = (torch.rand(B, *ds, device=dev) if M == 1 else torch.randint(0, 256, (B, *ds), device=dev).float())
model.auto_graph = False
with torch.no_grad():
    for _ in range(105):
        model(x)
torch.cuda.synchronize()
