"""Parameter specification + deterministic parameter generation for the oracle and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/flow_oracle.py header).

`param_spec` lists the reference's `state_dict` entries (names, shapes, dtypes) for a program from
`flow_oracle.program` — SURVEY.md Appendix B; `tests/golden/make_golden.py` asserts it equals the
real reference `state_dict` key for key.  `gen_params` fills it from `numpy.random.RandomState`
streams (frozen by NumPy's compatibility policy, so the same bits on every box), which keeps the
committed fixtures small: big tensors are regenerated from the seed instead of being stored.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

from .flow_oracle import K_COMPONENTS, ctx_width, vit_dims


def _gmm(prefix, size, M, spec, ctx=None):
    D, H, W = size
    spec[prefix + "mG"] = ((M, K_COMPONENTS, D, H, W), "normal")
    spec[prefix + "sG"] = ((M, K_COMPONENTS, D, H, W), "scale")
    spec[prefix + "wG"] = ((M, K_COMPONENTS), "normal")
    if ctx is not None:               # model.py:157,162: CatEmbeddings(contexts, 2*M*K*D // len(contexts)), zero init
        d = 2 * M * K_COMPONENTS * D // len(ctx["contexts"])
        for i, k in enumerate(ctx["contexts"]):
            spec[prefix + "context_net.0._embeddings.%d.weight" % i] = ((k, d), "small")


def _enc_flow(e, K, n, spec):
    """FlowInvSequential(ConditionalGaussianDistribution(CatEmbeddings), 2 x [FC, ActNormFC, CouplingFC]) (model.py:52-79)."""
    for i, k in enumerate(K):
        spec[e + "dist.context_net._embeddings.%d.weight" % i] = ((k, 2 * n // len(K)), "small")
    for l in range(2):
        spec[e + "%d.NN" % (3 * l)] = ((n, n), "orthogonal")
        spec[e + "%d.NN_t" % (3 * l + 1)] = ((n,), "zeros")
        spec[e + "%d.NN_logs" % (3 * l + 1)] = ((n,), "zeros")
        spec[e + "%d.initialized" % (3 * l + 1)] = ((), "flag")
        q = e + "%d." % (3 * l + 2)
        for name, shp in (("NN.0", (2 * n, n // 2, 1, 1)), ("NN.2", (2 * n, 2 * n, 1, 1)), ("NN.4", (n, 2 * n, 1, 1))):
            spec[q + name + ".weight"] = (shp, ("uniform", shp[1]))
            spec[q + name + ".bias"] = ((shp[0],), ("uniform", shp[1]))


def _encoder(prefix, ctx, spec, data_dim):
    """Entries of ContextEncoder(contexts, enc_emb, enc_type, (data_dim,)) = Sequential(embedding, encoder)
    (model.py:30-90; rtdl/nn/_embeddings.py:76-285; dequantize.py:26-262).  Returns the encoder's output width C."""
    K = ctx["contexts"]
    et = ctx.get("enc_type", "uniform")
    C = ctx_width(ctx, data_dim)
    p0, p1 = prefix + "context_net.0.", prefix + "context_net.1."
    if ctx["enc_emb"] == "onehot":
        spec[p0 + "cardinalities"] = ((len(K),), ("ints", tuple(K)))
        cats = [1] * sum(K)
    elif ctx["enc_emb"] == "embed":
        for i, k in enumerate(K):
            spec[p0 + "_embeddings.%d.weight" % i] = ((k, data_dim), "small")
        cats = None
    else:
        cats = list(K)
    if et in ("uniform", "vardeq"):
        spec[p1 + "qbins"] = ((len(cats),), ("floats", tuple(float(v) for v in cats)))
        spec[p1 + "ldj_per_dim"] = ((len(cats),), ("floats", tuple(-float(np.log(np.float32(v))) for v in cats)))
    if et in ("vardeq", "argmax", "probsample"):
        _enc_flow(p1 + "encoder.", K, C, spec)
        spec[p1 + "sigmoid.temperature"] = ((1,), ("floats", (1.0,)))
    return C


def param_spec(ops, prior_size, mixtures, ctx=None):
    """OrderedDict name -> (shape, kind) in the reference's state_dict order
    (flowsequential.py:8-12: `dist` is registered before the numbered layers).  `ctx`: specialist description
    (flow_oracle.ctx_width)."""
    spec = OrderedDict()
    _gmm("dist.", prior_size, mixtures, spec, ctx)
    for op in ops:
        kind, idx = op[0], op[1]
        pre = "%d." % idx
        if kind == "dequant":
            spec[pre + "dist.empty"] = ((1,), "zeros")                       # uniform.py:15
        elif kind == "affine":
            spec[pre + "translation"] = ((1,), ("const", op[2]))             # normalize.py:24-25
            spec[pre + "scale"] = ((1,), ("const", op[3]))
        elif kind == "augment":
            spec[pre + "distribution.buffer"] = ((1,), "zeros")              # gaussian.py:19
        elif kind == "conv1x1":
            C = op[2][0]
            spec[pre + "NN"] = ((C, C), "orthogonal")                         # conv1x1.py:16-17
            if ctx is not None:                                               # conv1x1.py:20-26 (zero init there)
                Cc = _encoder(pre, ctx, spec, C)
                spec[pre + "CN.weight"] = ((C * C, Cc), "small")
                spec[pre + "CN.bias"] = ((C * C,), "small")
        elif kind == "actnorm":
            C = op[2][0]
            spec[pre + "NN_t"] = ((C,), "zeros")                              # actnorm.py:14-16
            spec[pre + "NN_logs"] = ((C,), "zeros")
            spec[pre + "initialized"] = ((), "flag")
            if ctx is not None:                                               # actnorm.py:19-26
                Cc = _encoder(pre, ctx, spec, 2 * C)
                spec[pre + "CN.weight"] = ((2 * C, Cc), "small")
                spec[pre + "CN.bias"] = ((2 * C,), "small")
        elif kind == "coupling":
            C = op[2][0]
            kh, kw = op[3]
            D, Hd, O = C // 2, C * 2, C                                        # coupling.py:17-19
            if ctx is not None:                                               # coupling.py:23: registered before NN
                Cc = _encoder(pre, ctx, spec, C)
            Din = D + O if (ctx is not None and not ctx["contextflow"]) else D   # coupling.py:33-34 (concat)
            for name, shp in (("NN.0", (Hd, Din, 1, 1)), ("NN.2", (Hd, Hd, kh, kw)), ("NN.4", (O, Hd, 1, 1))):
                fan_in = shp[1] * shp[2] * shp[3]
                spec[pre + name + ".weight"] = (shp, ("uniform", fan_in))
                spec[pre + name + ".bias"] = ((shp[0],), ("uniform", fan_in))
            if ctx is not None:                                               # coupling.py:37
                for name, shp in (("CN.0", (Hd, Cc)), ("CN.2", (Hd, Hd)), ("CN.4", (O, Hd))):
                    spec[pre + name + ".weight"] = (shp, ("uniform", shp[1]))
                    spec[pre + name + ".bias"] = ((shp[0],), ("uniform", shp[1]))
        elif kind == "transcoupling":
            d = vit_dims(op[2], op[3])
            dim, pd, inner = d["dim"], d["patch_dim"], d["dim_head"]
            q = pre + "NN.0."
            if ctx is not None:                                               # coupling.py:107,113-119
                Cc = _encoder(pre, ctx, spec, op[2][0])
                if not ctx["contextflow"]:                                    # ViT over [x0 ; CN(c)], direct child
                    C = op[2][0]
                    pd, q = (C // 2 + C) * op[3][0] * op[3][1], pre + "NN."
            spec[q + "to_patch_embedding.1.weight"] = ((pd,), "ln_w")
            spec[q + "to_patch_embedding.1.bias"] = ((pd,), "ln_b")
            spec[q + "to_patch_embedding.2.weight"] = ((dim, pd), ("uniform", pd))
            spec[q + "to_patch_embedding.2.bias"] = ((dim,), ("uniform", pd))
            spec[q + "to_patch_embedding.3.weight"] = ((dim,), "ln_w")
            spec[q + "to_patch_embedding.3.bias"] = ((dim,), "ln_b")
            spec[q + "transformer.norm.weight"] = ((dim,), "ln_w")
            spec[q + "transformer.norm.bias"] = ((dim,), "ln_b")
            for l in range(d["depth"]):
                a = q + "transformer.layers.%d.0." % l
                f = q + "transformer.layers.%d.1.net." % l
                spec[a + "norm.weight"] = ((dim,), "ln_w")
                spec[a + "norm.bias"] = ((dim,), "ln_b")
                spec[a + "to_qkv.weight"] = ((3 * inner, dim), ("uniform", dim))
                spec[a + "to_out.weight"] = ((dim, inner), ("uniform", inner))
                spec[f + "0.weight"] = ((dim,), "ln_w")
                spec[f + "0.bias"] = ((dim,), "ln_b")
                spec[f + "1.weight"] = ((dim, dim), ("uniform", dim))
                spec[f + "1.bias"] = ((dim,), ("uniform", dim))
                spec[f + "3.weight"] = ((dim, dim), ("uniform", dim))
                spec[f + "3.bias"] = ((dim,), ("uniform", dim))
            if ctx is not None:
                C = op[2][0]
                for name, shp in (("CN.0", (2 * C, Cc)), ("CN.2", (2 * C, 2 * C)), ("CN.4", (C, 2 * C))):
                    spec[pre + name + ".weight"] = (shp, ("uniform", shp[1]))
                    spec[pre + name + ".bias"] = ((shp[0],), ("uniform", shp[1]))
        elif kind == "split":
            _gmm(pre + "dist.", op[2], mixtures, spec, ctx)
    return spec


def gen_params(spec, seed=0, dtype=torch.float32):
    """Deterministic parameters.  Each entry has its own RandomState stream keyed by its name."""
    out = OrderedDict()
    for name, (shape, kind) in spec.items():
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 32))
        if kind == "zeros":
            v = np.zeros(shape)
        elif kind == "flag":
            out[name] = torch.tensor(0); continue
        elif kind == "normal":
            v = rs.standard_normal(shape)
        elif kind == "scale":          # pre-softplus scale, perturbed so that softplus is exercised
            v = 1.0 + 0.2 * rs.standard_normal(shape)
        elif kind == "small":          # zero-initialised in the reference; small values so that the branch is exercised
            v = 0.05 * rs.standard_normal(shape)
        elif kind[0] == "ints":
            out[name] = torch.tensor(kind[1], dtype=torch.int64); continue
        elif kind[0] == "floats":
            out[name] = torch.tensor(kind[1], dtype=torch.float32); continue
        elif kind == "ln_w":
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        elif kind == "ln_b":
            v = 0.1 * rs.standard_normal(shape)
        elif kind == "orthogonal":     # LAPACK-dependent in the last bits -> fixtures store these
            q, r = np.linalg.qr(rs.standard_normal(shape))
            v = q * np.sign(np.diag(r))[None, :]
        elif kind[0] == "const":
            v = np.full(shape, np.float32(kind[1]))
        elif kind[0] == "uniform":
            b = 1.0 / np.sqrt(kind[1])
            v = rs.uniform(-b, b, size=shape)
        else:
            raise ValueError(kind)
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float64)).to(dtype)
    return out


def stress_params(params, spec, seed, raw_gain, sg_lo=-4.0, sg_hi=6.0):
    """Trained-like parameter regime for the stress fixtures (tests/golden/e2e_*_stress.npz), applied to `gen_params`
    output BEFORE the first (ActNorm-initialising) call:
      * Conv1x1 weights W = diag(d) Q, d log-spaced over three decades in shuffled order (condition number 1e3): output
        channel c has scale d_c, so the data-dependent ActNorm init that follows lands on log-scales of both signs
        spread over +-3.45, as after training;
      * the conditioner's output layer scaled by `raw_gain` on the rows that produce the raw log-scale (conv couplings:
        second half of NN.4; transformer couplings: the final LayerNorm's affine), so that raw / 2 drives tanh deep into
        saturation on part of the elements (coupling.py:52-57);
      * mixture scales sG uniform in [sg_lo, sg_hi] (default softplus: 0.018 ... 6) instead of ~1 (gaussian.py:142-161);
        the fixture generator then moves the component means onto latent samples (stored in the fixture), as a fitted
        mixture has them.
    Deterministic given (params, seed) except the LAPACK-dependent Conv1x1 factors, which the fixtures store."""
    out = OrderedDict((k, v.clone()) for k, v in params.items())
    for name, (shape, kind) in spec.items():
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(("stress:" + name).encode())) % (2 ** 32))
        if kind == "orthogonal" and name.endswith(".NN") and len(shape) == 2:
            C = shape[0]
            q, r = np.linalg.qr(rs.standard_normal(shape))
            d = rs.permutation(np.logspace(-1.5, 1.5, C)) if C > 1 else np.ones(1)
            out[name] = torch.from_numpy(d[:, None] * (q * np.sign(np.diag(r))[None, :])).to(params[name].dtype)
        elif kind == "scale":
            out[name] = torch.from_numpy(rs.uniform(sg_lo, sg_hi, size=shape)).to(params[name].dtype)
        elif name.endswith(("NN.4.weight", "NN.4.bias")) and ".CN." not in name:
            half = shape[0] // 2
            out[name][half:] *= raw_gain
        elif name.endswith(("transformer.norm.weight", "transformer.norm.bias")):
            out[name] *= raw_gain
    return out


def stress_means(zin, sG, seed, name):
    """Component means of a fitted mixture for the stress fixtures: slot j = (m, k) sits on latent sample j mod B,
    offset by half a standard deviation of its own (softplus(sG)) per element."""
    MK = sG.shape[0] * sG.shape[1]
    rs = np.random.RandomState((seed * 1000003 + zlib.crc32(("stress-mean:" + name).encode())) % (2 ** 32))
    delta = torch.from_numpy(0.5 * rs.standard_normal((MK,) + tuple(sG.shape[2:]))).to(sG.dtype)
    idx = torch.arange(MK) % zin.shape[0]
    sig = torch.nn.functional.softplus(sG.reshape((MK,) + tuple(sG.shape[2:])))
    return (zin[idx].to(sG.dtype) + sig * delta).reshape(sG.shape)


LAPACK_DEPENDENT = ("orthogonal",)


def stored_keys(spec):
    """Entries a fixture must store explicitly (not bit-reproducible from the seed everywhere,
    or produced by the reference's data-dependent ActNorm init)."""
    keys = []
    for name, (shape, kind) in spec.items():
        if kind in LAPACK_DEPENDENT or name.endswith(("NN_t", "NN_logs", "initialized")):
            keys.append(name)
    return keys
