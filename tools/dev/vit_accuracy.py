#!/usr/bin/env python3
"""Which of the transformer-step forms is closest to the fp64 oracle?  4096 random SMAP samples through the fixture's
model: one-kernel step 'wave' (cf_vit_step_fwd), 'rs' (cf_vit_step_rs_fwd), layer mode (cf_vit_coupling); the fp64 and fp32
oracles on the 64 samples where the forms disagree most.  Prints max |d logp| in nats and bits/dim."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.gpu_util import build_model, set_noise
from tests.helpers import load_e2e
import oracle.flow_oracle as fo
from contextflow_amd.layers.coupling import TransCoupling

DEV = "cuda:0"
ops, _, M, params, fx = load_e2e("smap")
B = 4096
g = torch.Generator().manual_seed(5)
x = torch.rand(B, 25, 8, 1, generator=g)
eps = [torch.randn(B, 1, 8, 1, generator=g)]
model = build_model("smap", params)
set_noise(model, None, eps)
out = {}
with torch.no_grad():
    model.auto_graph = False
    for tag, thr, fused in (("wave", 0, True), ("rs", 1 << 30, True), ("layer", 0, False)):
        TransCoupling.STEP_RS_MAX_BATCH = thr
        model.fused = fused
        out[tag] = model(x.to(DEV))[1].double().cpu()
d = (out["wave"] - out["layer"]).abs().flatten() + (out["rs"] - out["layer"]).abs().flatten()
idx = torch.topk(d, 64).indices
p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
_, ref64 = fo.flow_forward(ops, p64, x[idx].double(), None, [eps[0][idx].double()])
_, ref32 = fo.flow_forward(ops, params, x[idx], None, [eps[0][idx]])
D = 200 * math.log(2)
for tag in ("wave", "rs", "layer"):
    e = (out[tag][idx] - ref64).abs().max().item()
    print("%-6s max |logp - fp64 oracle| over the 64 most-disagreeing samples: %.2e nats = %.2e bits/dim" % (tag, e, e / D))
e = (ref32.double() - ref64).abs().max().item()
print("fp32 oracle (= reference arithmetic): %.2e nats = %.2e bits/dim" % (e, e / D))
for a, b in (("wave", "layer"), ("rs", "layer"), ("wave", "rs")):
    e = (out[a] - out[b]).abs().max().item()
    print("%s vs %s over all %d samples: %.2e nats = %.2e bits/dim" % (a, b, B, e, e / D))
