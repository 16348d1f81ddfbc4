#!/usr/bin/env python3
"""Which of the transformer-step forms is closest to the fp64 oracle?  N random SMAP samples through the fixture's model
(random init, and the "stress" parameter set): one-kernel step 'wave' (cf_vit_step_fwd), 'rs' (cf_vit_step_rs_fwd), layer
mode (cf_vit_coupling), and the fp32 oracle (= the reference's arithmetic) - each against the fp64 oracle on ALL samples:
max and rms |d bits/dim|.  usage: vit_accuracy.py [N=4096]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.gpu_util import build_model, set_noise
from tests.helpers import load_e2e
import oracle.flow_oracle as fo
from contextflow_amd.layers.coupling import TransCoupling

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = 200 * math.log(2)
print("library:", os.environ.get("CONTEXTFLOW_HIP_LIB", "product"))
for tag in (None, "stress"):
    ops, _, M, params, fx = load_e2e("smap", tag)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 25, 8, 1, generator=g)
    eps = [torch.randn(B, 1, 8, 1, generator=g)]
    model = build_model("smap", params)
    set_noise(model, None, eps)
    out = {}
    with torch.no_grad():
        model.auto_graph = False
        for form, thr, fused in (("wave", 0, True), ("rs", 1 << 40, True), ("layer", 0, False)):
            TransCoupling.STEP_RS_MAX_BATCH = thr
            model.fused = fused
            out[form] = model(x.to(DEV))[1].double().cpu()
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    _, ref64 = fo.flow_forward(ops, p64, x.double(), None, [eps[0].double()])
    _, ref32 = fo.flow_forward(ops, params, x, None, [eps[0]])
    out["fp32 oracle"] = ref32.double()
    print("parameters: %s, %d samples, |logp| up to %.0f nats" % (tag or "random init", B, ref64.abs().max()))
    for form in ("wave", "rs", "layer", "fp32 oracle"):
        e = (out[form] - ref64).abs().flatten() / D
        print("  %-12s vs fp64 oracle: max %.2e  rms %.2e  99.9%% %.2e bits/dim" % (form, e.max(), e.pow(2).mean().sqrt(), e.quantile(0.999)))
    for a, b in (("wave", "fp32 oracle"), ("rs", "fp32 oracle"), ("wave", "rs"), ("wave", "layer")):
        e = (out[a] - out[b]).abs().flatten() / D
        print("  %-5s vs %-12s max %.2e  rms %.2e bits/dim" % (a, b, e.max(), e.pow(2).mean().sqrt()))
