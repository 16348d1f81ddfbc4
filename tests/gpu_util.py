"""Helpers for the -m gpu tests: build contextflow_amd models from the golden fixtures."""
import ctypes

import numpy as np
import torch

import contextflow_amd as cfa
from contextflow_amd.layers import _hip
from tests.helpers import load_e2e, pre_init_params, e2e_inputs

DEV = "cuda:0"


def build_model(name, params, dev=DEV):
    cfg, data_size, M = cfa.preset_config(name)
    model = cfa.create_model(cfg, data_size, M)
    model.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    return model.to(dev)


def set_noise(model, u, eps, dev=DEV):
    eps = list(eps)
    for m in model.sequence_modules:
        if isinstance(m, cfa.layers.Dequantization):
            m.dist.fixed_noise = None if u is None else u.to(dev)
        if isinstance(m, cfa.layers.Augment):
            m.distribution.fixed_noise = eps.pop(0).to(dev) if eps else None


def fused_step_debug(x, conv, act, cpl, squeeze=False, variant=0):
    """Run cf_flow_step_prepare + the debug variant of the step kernel; returns z, ldj, dumps.  variant 3 (16x16 images):
    the dumps of k_flow_step_small."""
    L = _hip.lib()
    fn = L.cf_flow_step_fwd_debug
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    B = x.shape[0]
    C, H, W = (x.shape[1] * 4, x.shape[2] // 2, x.shape[3] // 2) if squeeze else tuple(x.shape[1:])
    ws = torch.empty(L.cf_flow_step_ws_bytes(C, H, W), device=x.device, dtype=torch.uint8)
    f, pp = _hip.f32, _hip.p
    c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
    _hip.call("cf_flow_step_prepare", pp(f(conv.NN.detach())), pp(f(act.NN_t.detach())), pp(f(act.NN_logs.detach())),
              pp(f(c1.weight.detach())), pp(f(c1.bias.detach())), pp(f(c2.weight.detach())), pp(f(c2.bias.detach())),
              pp(f(c3.weight.detach())), pp(f(c3.bias.detach())), pp(ws), C, H, W, _hip.stream())
    spw = {256: 1, 64: 4, 16: 16}[H * W]
    nwg = (B + spw - 1) // spw
    cols = nwg * spw * H * W
    dbg = torch.full((2 * C + 4 * C, cols), float("nan"), device=x.device)
    z = torch.empty(B, C, H, W, device=x.device)
    ldj = torch.zeros(B, device=x.device)
    _hip.check(fn(pp(x), pp(z), pp(ldj), pp(ws), B, C, H, W, C * H * W, int(squeeze), pp(dbg), variant << 16, _hip.stream()),
               "cf_flow_step_fwd_debug")
    torch.cuda.synchronize()
    HID, HALF = 2 * C, C // 2

    def plane(r0, rows):     # (rows, cols) -> (B, rows, H, W)
        v = dbg[r0:r0 + rows, : B * H * W].reshape(rows, B, H, W).permute(1, 0, 2, 3)
        return v.contiguous()

    # the production entry point on the same operands (small batches / 16x16 images take other kernel variants than the
    # dump kernel above: half-size workgroups, k_flow_step_small)
    z2 = torch.full((B, C, H, W), float("nan"), device=x.device)
    ldj2 = torch.zeros(B, device=x.device)
    _hip.call("cf_flow_step_fwd", pp(x), pp(z2), pp(ldj2), pp(ws), B, C, H, W, C * H * W, int(squeeze), _hip.stream())
    torch.cuda.synchronize()
    return z, ldj, dict(y0=plane(0, HALF), h1=plane(C, HID), h2=plane(C + HID, HID), h=plane(C + 2 * HID, C), z_prod=z2,
                        ldj_prod=ldj2)
