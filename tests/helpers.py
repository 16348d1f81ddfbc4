"""Shared helpers for the tests: fixture loading and parameter reconstruction."""
import math
import os

import numpy as np
import torch

from oracle import flow_oracle as fo
from oracle import params as op

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_e2e(name, tag=None):
    """Returns (ops, prior_size, M, params(post ActNorm init), fixture dict).  tag = "stress" | "extreme": the
    trained-like parameter regimes of make_golden.end_to_end_stress (oracle.params.stress_params)."""
    fx = dict(np.load(os.path.join(GOLDEN, "e2e_%s%s.npz" % (name, "_" + tag if tag else ""))))
    ops, prior_size, M = fo.program(name)
    spec = op.param_spec(ops, prior_size, M)
    params = op.gen_params(spec, int(fx["seed"]))
    if tag:
        params = op.stress_params(params, spec, int(fx["seed"]), float(fx["raw_gain"]), float(fx["sg_lo"]))
    for k, v in fx.items():
        if k.startswith("param:"):
            params[k[6:]] = torch.from_numpy(v)
    return ops, prior_size, M, params, fx


def pre_init_params(name, fx):
    """Same parameters but with ActNorm still un-initialised (as before the first call)."""
    ops, prior_size, M = fo.program(name)
    spec = op.param_spec(ops, prior_size, M)
    params = op.gen_params(spec, int(fx["seed"]))
    if "raw_gain" in fx:
        params = op.stress_params(params, spec, int(fx["seed"]), float(fx["raw_gain"]), float(fx["sg_lo"]))
    for k, v in fx.items():
        if k.startswith("param:") and (spec[k[6:]][1] == "orthogonal" or k.endswith("mG")):
            params[k[6:]] = torch.from_numpy(v)
    return params


def stress_tolerance(fx, tag):
    """bits/dim bar of a stress fixture: 1e-5 (BASELINE.json) where the reference's own fp32 answer sits well inside it
    ("stress": measured 2.6e-6 .. 3.0e-6 from its fp64 run); for "extreme" the reference's fp32 result is itself up to
    7.5e-6 from its fp64 run, so the bar is max(1e-5, 1.5 x that floor).  (Round 2 needed 3 x: the fused transformer kernels
    summed every residual product on top of the residual stream and lost 2-3x the reference's accuracy per layer; measured
    now: smap extreme 8.8e-6 from the reference's fp32 and 3.0e-6 from its fp64 answer.)"""
    floor = float(fx["floor_bpd"])
    if tag == "stress":
        assert floor < 3.5e-6
        return 1e-5
    return max(1e-5, 1.5 * floor)


def e2e_inputs(name, fx):
    x = torch.from_numpy(fx["x"].astype(np.float32))
    u = torch.from_numpy(fx["u"]) if "u" in fx else None
    eps = [torch.from_numpy(fx["eps%d" % j]) for j in range(8) if "eps%d" % j in fx]
    return x, u, eps


def unit(tag):
    fx = np.load(os.path.join(GOLDEN, "unit_layers.npz"))
    out, sd = {}, {}
    for k in fx.files:
        if k.startswith(tag + "/"):
            key = k[len(tag) + 1:]
            if key.startswith("sd:"):
                sd[key[3:]] = torch.from_numpy(fx[k])
            else:
                out[key] = torch.from_numpy(fx[k]) if fx[k].dtype.kind == "f" else fx[k]
    return out, sd


def bpd(logp, name):
    D = int(np.prod(fo.CONFIGS[name][0]))
    return -torch.logsumexp(logp.double(), dim=-1) / (D * math.log(2.0))


SPECIALIST = {
    # fixture: (dataset, ctx description) — see tests/golden/make_golden_specialist.py
    "mnist_eye_cf": ("mnist", dict(contexts=[64], enc_emb="eye", contextflow=True)),
    "mnist_onehot": ("mnist", dict(contexts=[64], enc_emb="onehot", contextflow=False)),
    "cifar10_onehot_cf": ("cifar10", dict(contexts=[15, 5], enc_emb="onehot", contextflow=True)),
    "cifar10_eye": ("cifar10", dict(contexts=[15, 5], enc_emb="eye", contextflow=False)),
    "cifar10_onehot_vardeq": ("cifar10", dict(contexts=[15, 5], enc_emb="onehot", contextflow=False, enc_type="vardeq")),
    "cifar10_eye_vardeq_cf": ("cifar10", dict(contexts=[15, 5], enc_emb="eye", contextflow=True, enc_type="vardeq")),
    "smap_onehot_cf": ("smap", dict(contexts=[55], enc_emb="onehot", contextflow=True)),
    "smap_eye": ("smap", dict(contexts=[55], enc_emb="eye", contextflow=False)),
    "cifar10_eye_argmax_cf": ("cifar10", dict(contexts=[15, 5], enc_emb="eye", contextflow=True, enc_type="argmax")),
    "cifar10_embed_eyesample": ("cifar10", dict(contexts=[15, 5], enc_emb="embed", contextflow=False, enc_type="eyesample")),
    "mnist_embed_probsample_cf": ("mnist", dict(contexts=[64], enc_emb="embed", contextflow=True, enc_type="probsample")),
    "atm_onehot_cf": ("atm", dict(contexts=[68], enc_emb="onehot", contextflow=True)),
    "atm_embed_eyesample_cf": ("atm", dict(contexts=[68], enc_emb="embed", contextflow=True, enc_type="eyesample")),
    "atm_onehot_vardeq_cf": ("atm", dict(contexts=[68], enc_emb="onehot", contextflow=True, enc_type="vardeq")),
    "atm_eye_argmax_cf": ("atm", dict(contexts=[68], enc_emb="eye", contextflow=True, enc_type="argmax")),
    "atm_embed_probsample_cf": ("atm", dict(contexts=[68], enc_emb="embed", contextflow=True, enc_type="probsample")),
}


def load_specialist(fxname):
    """Returns (dataset, ctx, ops, M, params(post first call), inputs dict) of a specialist fixture."""
    name, ctx = SPECIALIST[fxname]
    fx = dict(np.load(os.path.join(GOLDEN, "spec_%s.npz" % fxname)))
    ops, prior_size, M = fo.program(name)
    spec = op.param_spec(ops, prior_size, M, ctx)
    params = op.gen_params(spec, int(fx["seed"]))
    for k, v in fx.items():
        if k.startswith("param:"):
            params[k[6:]] = torch.from_numpy(v)
    inp = dict(x=torch.from_numpy(fx["x"].astype(np.float32)), u=torch.from_numpy(fx["u"]),
               eps=[torch.from_numpy(fx["eps%d" % j]) for j in range(8) if "eps%d" % j in fx],
               context=torch.from_numpy(fx["context"]),
               cnoise=[torch.from_numpy(fx["cnoise%d" % j]) for j in range(200) if "cnoise%d" % j in fx],
               logp=torch.from_numpy(fx["logp"]), z=torch.from_numpy(fx["z"]))
    return name, ctx, ops, M, params, inp
