"""Per-tensor relative gradient error of the hand-written backward vs fp64 autograd through the oracle, next to the
error of fp32 autograd through the same oracle (the fp32 noise floor of the gradient itself).  Developer tool:
    python tests/dev_bwd_errors.py [mnist cifar10 smap atm]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import flow_oracle as fo                      # noqa: E402
from tests.gpu_util import build_model, set_noise         # noqa: E402
from tests.helpers import load_e2e                        # noqa: E402

DEV = "cuda:0"
for name, B in (("mnist", 6), ("cifar10", 5), ("smap", 7), ("atm", 3)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    ops, _, M, params, fx = load_e2e(name)
    C, H, W = fo.CONFIGS[name][0]
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, C, H, W, generator=g) if name in ("smap", "atm") else torch.randint(0, 256, (B, C, H, W), generator=g).float()
    u = torch.rand(B, C, H, W, generator=g)
    eps = [torch.randn(B, 1, H, W, generator=g)]
    wts = torch.randn(B, M, generator=g)

    def oracle(dt):
        p = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in params.items()}
        _, lp = fo.flow_forward(ops, p, x.to(dt), u.to(dt), [e.to(dt) for e in eps])
        (lp * wts.to(dt)).sum().backward()
        return p
    p64, p32 = oracle(torch.float64), oracle(torch.float32)
    model = build_model(name, params)
    set_noise(model, u, eps)
    model.train()
    z, logp = model(x.to(DEV))
    (logp * wts.to(DEV)).sum().backward()
    rows = []
    for k, p in model.named_parameters():
        ref = p64[k].grad
        if ref is None:
            continue
        scale = max(ref.abs().max().item(), 1e-3)
        e_gpu = (p.grad.detach().cpu().double() - ref).abs().max().item() / scale
        e_32 = (p32[k].grad.double() - ref).abs().max().item() / scale
        rows.append((e_gpu, e_32, scale, k))
    rows.sort(reverse=True)
    print("== %s: %d tensors; worst by our error (rel err ours | rel err fp32 autograd | scale)" % (name, len(rows)))
    for e_gpu, e_32, scale, k in rows[:12]:
        print("   %.2e | %.2e | %.2e  %s" % (e_gpu, e_32, scale, k))
    n4 = sum(1 for r in rows if r[0] > 1e-4)
    print("   tensors above 1e-4: %d; max ratio ours / max(fp32 autograd, 1e-6): %.1f"
          % (n4, max(r[0] / max(r[1], 1e-6) for r in rows)))
