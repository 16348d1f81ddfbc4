#!/usr/bin/env python3
"""Dev check (GPU): the four-update cifar10 fixture loop of test_captured_train_step_matches_the_eager_loop under three
drivers - eager + torch's default AdamW, eager + fused capturable AdamW, captured step + fused capturable AdamW - to tell the
rounding of the two AdamW implementations from anything the capture does.  usage: capture_vs_eager.py [name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_parity import load_e2e, e2e_inputs, DEV
from tests.gpu_util import build_model, set_noise

name = sys.argv[1] if len(sys.argv) > 1 else "cifar10"
ops, _, M, params, fx = load_e2e(name)
x, u, eps = e2e_inputs(name, fx)
xd = x.to(DEV)
gt = (torch.arange(x.shape[0]) % M).to(DEV)
inv = 1.0 / x[0].numel()
loss_fn = lambda lp, y: torch.nn.functional.cross_entropy(lp * inv, y) if M > 1 else -(lp * inv).mean()


def run(fused, capture):
    m = build_model(name, params)
    set_noise(m, u, eps)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, **(dict(fused=True, capturable=True) if fused else {}))
    out = []
    if capture:
        step = m.capture_train_step(xd, loss_fn, opt)
        return [float(step(xd, gt).detach()) for _ in range(4)]
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(m.log_prob(xd), gt)
        loss.backward()
        opt.step()
        out.append(float(loss.detach()))
    return out


for tag, f, c in (("eager, default AdamW", False, False), ("eager, fused AdamW  ", True, False), ("captured, fused     ", True, True)):
    print(tag, ["%.9f" % v for v in run(f, c)], flush=True)
