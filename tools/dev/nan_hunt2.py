import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
from contextflow_amd.layers import _hip, autograd as ag
B = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
mode = sys.argv[2] if len(sys.argv) > 2 else "none"
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config("cifar10")
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
with torch.no_grad():
    model(x[:256])
orig_call = _hip.call
def call(name, *a):
    if mode == "pre_" + name or mode == "pre_all":
        torch.cuda.synchronize()
    r = orig_call(name, *a)
    if mode == "post_" + name or mode == "post_all":
        torch.cuda.synchronize()
    return r
_hip.call = call
ag._hip.call = call
nbad = 0
for it in range(6):
    model.zero_grad(set_to_none=True)
    _, lp = model(x)
    torch.nn.functional.cross_entropy(lp / 3072.0, gt).backward()
    torch.cuda.synchronize()
    bad = [k for k, p in model.named_parameters() if p.grad is not None and (not torch.isfinite(p.grad).all() or p.grad.abs().max() > 1e3)]
    nbad += len(bad) > 0
print("mode", mode, "B", B, "iterations with bad grads:", nbad, "of 6", bad[:4])
