import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
from contextflow_amd.layers import _hip, autograd as ag
B = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
dev = "cuda:0"
torch.manual_seed(0)
cfg, ds, M = cfa.preset_config("cifar10")
model = cfa.create_model(cfg, ds, M).to(dev)
x = torch.randint(0, 256, (B, *ds), device=dev).float()
gt = torch.randint(0, M, (B,), device=dev)
with torch.no_grad():
    model(x[:256])
orig = ag.step_backward
def wrapped(xin, squeeze, conv, act, cpl, shape, ws, gz, gld, winv=None, planes=None, gsum=None):
    gx, grads = orig(xin, squeeze, conv, act, cpl, shape, ws, gz, gld, winv, planes, gsum)
    torch.cuda.synchronize()
    bad = [str(tuple(k.shape)) for k, v in grads.items() if not torch.isfinite(v).all()]
    info = "step %s sq=%d gz finite %s |gz|max %.3g gx finite %s" % (shape, int(squeeze), torch.isfinite(gz).all().item(), gz.abs().max().item(), torch.isfinite(gx).all().item())
    if planes is not None:
        info += " planes finite " + str([torch.isfinite(p).all().item() for p in planes[:3]]) + " |h1|max %.3g" % planes[1].abs().max().item()
    print(info, "BAD" if bad else "", bad)
    return gx, grads
ag.step_backward = wrapped
for it in range(3):
    model.zero_grad(set_to_none=True)
    _, lp = model(x)
    torch.nn.functional.cross_entropy(lp / 3072.0, gt).backward()
    print("---- iteration", it)
