"""CPU: the oracle (oracle/flow_oracle.py) against the reference's own outputs (tests/golden)."""
import os

import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo
from tests.helpers import load_e2e, pre_init_params, e2e_inputs, unit, bpd, stress_tolerance, GOLDEN

BPD_TOL = 1e-5     # BASELINE.json: bits/dim within 1e-5 of the reference
Z_TOL = 1e-5


@pytest.mark.parametrize("name", ["mnist", "cifar10", "smap", "atm"])
def test_e2e_logp_and_trace(name):
    ops, _, M, params, fx = load_e2e(name)
    x, u, eps = e2e_inputs(name, fx)
    trace = []
    z, logp = fo.flow_forward(ops, params, x, u, eps, trace=trace)
    ref = torch.from_numpy(fx["logp"])
    assert logp.shape == ref.shape == (x.shape[0], M)
    assert (bpd(logp, name) - bpd(ref, name)).abs().max() < BPD_TOL
    # per-mixture logp, expressed in bits/dim units
    D = np.prod(fo.CONFIGS[name][0])
    assert ((logp - ref).abs().max() / (D * np.log(2))) < BPD_TOL
    assert (z - torch.from_numpy(fx["z"])).abs().max() < 2e-4
    for i, (kind, idx, zi, ldj) in enumerate(trace):
        r = torch.from_numpy(fx["ldj%d" % i])
        assert ldj.shape == r.shape, (i, kind)
        assert torch.allclose(ldj, r, rtol=2e-6, atol=2e-3), (i, kind, (ldj - r).abs().max())
        if "z%d" % i in fx:
            zr = torch.from_numpy(fx["z%d" % i])
            assert (zi - zr).abs().max() <= 1e-4 * max(1.0, zr.abs().max().item()), (i, kind)


@pytest.mark.parametrize("name", ["mnist", "cifar10", "smap", "atm"])
def test_e2e_actnorm_init(name):
    """First call: data-dependent ActNorm init must reproduce the reference's post-init state."""
    ops, _, M, post, fx = load_e2e(name)
    params = pre_init_params(name, fx)
    x, u, eps = e2e_inputs(name, fx)
    _, logp = fo.flow_forward(ops, params, x, u, eps, init_actnorm=True)
    for k in post:
        if k.endswith(("NN_t", "NN_logs")):
            assert torch.allclose(params[k], post[k], rtol=1e-4, atol=2e-5), k
    assert (bpd(logp, name) - bpd(torch.from_numpy(fx["logp"]), name)).abs().max() < BPD_TOL


@pytest.mark.parametrize("name", ["mnist", "cifar10", "smap", "atm"])
def test_fp64_noise_floor(name):
    """The oracle in fp64 against the reference in fp64: restatement is exact up to fp64 rounding."""
    ops, _, M, params, fx = load_e2e(name)
    x, u, eps = e2e_inputs(name, fx)
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    _, logp = fo.flow_forward(ops, p64, x.double(), None if u is None else u.double(), [e.double() for e in eps])
    ref = torch.from_numpy(fx["logp_f64"])
    assert (logp - ref).abs().max() < 1e-7 * ref.abs().max()


@pytest.mark.parametrize("tag", ["stress", "extreme"])
@pytest.mark.parametrize("name", ["mnist", "cifar10", "smap"])
def test_e2e_stress_regimes(name, tag):
    """Trained-like parameters (saturated coupling log-scales, ActNorm log-scales of +-3..5, Conv1x1 of condition number
    1e3, mixture scales far from 1): the oracle against the reference's own output, in fp32 within the bits/dim bar and
    in fp64 to rounding; the fixture really is in the regime it claims."""
    ops, _, M, params, fx = load_e2e(name, tag)
    x, u, eps = e2e_inputs(name, fx)
    tol = stress_tolerance(fx, tag)
    _, logp = fo.flow_forward(ops, params, x, u, eps)
    ref, ref64 = torch.from_numpy(fx["logp"]), torch.from_numpy(fx["logp_f64"])
    assert (bpd(logp, name) - bpd(ref, name)).abs().max() < tol
    assert (bpd(logp, name) - bpd(ref64, name)).abs().max() < tol
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    _, logp64 = fo.flow_forward(ops, p64, x.double(), None if u is None else u.double(), [e.double() for e in eps])
    assert (bpd(logp64, name) - bpd(ref64, name)).abs().max() < 1e-9
    # regime checks
    assert fx["raw_absmax"].max() > 8.0 and np.median(fx["raw_absmax"]) > 7.0
    logs = torch.cat([v for k, v in params.items() if k.endswith("NN_logs")])
    assert logs.min() < -3.0 and logs.max() > 3.0
    conds = [torch.linalg.cond(v.double()).item() for k, v in params.items() if k.endswith(".NN") and v.dim() == 2]
    assert min(conds) > 900
    sg = torch.cat([v.flatten() for k, v in params.items() if k.endswith("sG")])
    assert sg.max() > 5.9 and sg.min() < float(fx["sg_lo"]) + 0.1
    # ActNorm first call (data-dependent init on the ill-conditioned Conv1x1 outputs)
    pre = pre_init_params(name, fx)
    _, logp_first = fo.flow_forward(ops, pre, x, u, eps, init_actnorm=True)
    for k in params:
        if k.endswith(("NN_t", "NN_logs")):
            assert torch.allclose(pre[k], params[k], rtol=1e-4, atol=5e-5), k
    assert (bpd(logp_first, name) - bpd(ref, name)).abs().max() < tol


def test_wide_smap_fixture():
    """512 distinct SMAP samples (tests/golden/e2e_smap_wide.npz, make_golden.end_to_end_wide): the oracle in fp32 within the
    bits/dim bar of the reference's fp32 answer, in fp64 equal to the reference's fp64 run."""
    ops, _, M, params, _ = load_e2e("smap")
    fx = dict(np.load(os.path.join(GOLDEN, "e2e_smap_wide.npz")))
    x, eps = torch.from_numpy(fx["x"]), [torch.from_numpy(fx["eps0"])]
    _, logp = fo.flow_forward(ops, params, x, None, eps)
    ref, ref64 = torch.from_numpy(fx["logp"]), torch.from_numpy(fx["logp_f64"])
    assert (bpd(logp, "smap") - bpd(ref, "smap")).abs().max() < BPD_TOL
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    _, logp64 = fo.flow_forward(ops, p64, x.double(), None, [e.double() for e in eps])
    assert (bpd(logp64, "smap") - bpd(ref64, "smap")).abs().max() < 1e-9
    assert (bpd(ref, "smap") - bpd(ref64, "smap")).abs().max().item() < 8e-6          # the reference's own fp32-vs-fp64 distance


def test_inverse_mnist():
    import os
    from tests.helpers import GOLDEN
    from oracle import params as op
    fx = dict(np.load(os.path.join(GOLDEN, "inverse_mnist.npz")))
    ops, prior_size, M = fo.program("mnist")
    params = op.gen_params(op.param_spec(ops, prior_size, M), int(fx["seed"]))
    for k, v in fx.items():
        if k.startswith("param:"):
            params[k[6:]] = torch.from_numpy(v)
    z = torch.from_numpy(fx["z"])
    # layers after the Augment (index 4): reverse chain; Augment.reverse drops the noise channel
    h = fo.flow_inverse_layers(ops[5:], params, z)
    h = h[:, :1]
    h = fo.flow_inverse_layers(ops[:4], params, h)
    ref = torch.from_numpy(fx["x"])
    # floor() of the dequantisation makes this integer-valued: allow a flip only at exact boundaries
    assert (h - ref).abs().max() <= 1.0 and (h != ref).float().mean() < 2e-3


@pytest.mark.parametrize("tag", ["coupling_3x3", "coupling_3x1", "coupling_c16"])
def test_unit_coupling(tag):
    t, sd = unit(tag)
    pad = tuple(int(v) for v in t["pad"])
    p = {"0." + k: v for k, v in sd.items()}
    h = fo.coupling_net(t["x"][:, : t["x"].shape[1] // 2], p, "0.", pad)
    assert torch.allclose(h, t["h"], rtol=1e-5, atol=1e-5)
    z, ldj = fo.coupling_fwd(t["x"], p, "0.", pad)
    assert torch.allclose(z, t["z"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(ldj, t["ldj"], rtol=1e-5, atol=1e-4)
    assert torch.allclose(fo.coupling_inv(t["z"], p, "0.", pad), t["xrec"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["conv1x1_c26", "conv1x1_c64"])
def test_unit_conv1x1(tag):
    t, sd = unit(tag)
    z, ldj = fo.conv1x1_fwd(t["x"], sd["NN"])
    assert torch.allclose(z, t["z"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(ldj, t["ldj"].expand_as(ldj), rtol=1e-5, atol=1e-5)
    assert torch.allclose(fo.conv1x1_inv(t["z"], sd["NN"]), t["xrec"], rtol=1e-4, atol=1e-4)


def test_unit_actnorm():
    t, sd = unit("actnorm")
    mean, logs = fo.actnorm_stats(t["x"])
    assert torch.allclose(mean, sd["NN_t"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(logs, sd["NN_logs"], rtol=1e-6, atol=1e-6)
    z, ldj = fo.actnorm_fwd(t["x2"], sd["NN_t"], sd["NN_logs"])
    assert torch.allclose(z, t["z2"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(ldj, t["ldj2"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(fo.actnorm_inv(t["z2"], sd["NN_t"], sd["NN_logs"]), t["x2rec"], rtol=1e-6, atol=1e-6)


def test_unit_squeeze():
    t, _ = unit("squeeze22")
    assert torch.equal(fo.squeeze_fwd(t["x"], (2, 2)), t["z"])
    assert torch.equal(fo.squeeze_inv(t["z"], (2, 2)), t["xrec"])
    t, _ = unit("squeeze21")
    assert torch.equal(fo.squeeze_fwd(t["x"], (2, 1)), t["z"])


def test_unit_gmm_and_split():
    t, sd = unit("gmm")
    lp = fo.gmm_logprob(t["x"], sd["mG"], sd["sG"], sd["wG"])
    assert torch.allclose(lp, t["logp"], rtol=1e-5, atol=1e-4)
    t, sd = unit("split")
    c = t["x"].shape[1] // 2
    assert torch.equal(t["x"][:, :c], t["z"])
    lp = fo.gmm_logprob(t["x"][:, c:], sd["dist.mG"], sd["dist.sG"], sd["dist.wG"])
    assert torch.allclose(lp, t["ldj"], rtol=1e-5, atol=1e-4)


def test_unit_preprocessing():
    t, _ = unit("normalize")
    z, ldj = fo.affine_fwd(t["x"], 1e-4, 1 / (1 - 2e-4))
    assert torch.equal(z, t["z"]) and torch.allclose(ldj, t["ldj"], rtol=1e-6)
    assert torch.allclose(fo.affine_inv(t["z"], 1e-4, 1 / (1 - 2e-4)), t["xrec"], rtol=1e-6, atol=1e-7)
    t, _ = unit("normalize256")
    z, ldj = fo.affine_fwd(t["x"], 0.0, 256.0)
    assert torch.equal(z, t["z"]) and torch.allclose(ldj, t["ldj"], rtol=1e-6)
    t, _ = unit("logit")
    z, ldj = fo.logit_fwd(t["x"])
    assert torch.allclose(z, t["z"], rtol=1e-6, atol=1e-6) and torch.allclose(ldj, t["ldj"], rtol=1e-6)
    t, _ = unit("stdnormal")
    assert torch.allclose(-fo.std_normal_neg_logq(t["x"]), t["logp"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["trans_ts", "trans_img"])
def test_unit_transcoupling(tag):
    t, sd = unit(tag)
    sz = tuple(int(v) for v in t["in_sz"])
    patch = tuple(int(v) for v in t["p"])
    p = {"0." + k: v for k, v in sd.items()}
    h = fo.vit_net(t["x"][:, : sz[0] // 2], p, "0.", sz, patch)
    assert torch.allclose(h, t["h"], rtol=1e-4, atol=1e-5)
    z, ldj = fo.transcoupling_fwd(t["x"], p, "0.", sz, patch)
    assert torch.allclose(z, t["z"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(ldj, t["ldj"], rtol=1e-5, atol=1e-4)
    assert torch.allclose(fo.transcoupling_inv(t["z"], p, "0.", sz, patch), t["xrec"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["spline_shared", "spline_indiv"])
def test_unit_spline_activation(tag):
    t, sd = unit(tag)
    uw, uh, ud = sd["unnormalized_widths"], sd["unnormalized_heights"], sd["unnormalized_derivatives"]
    z, ldj = fo.spline_activation_fwd(t["x"], uw, uh, ud, 10.0)
    assert torch.allclose(z, t["z"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(ldj, t["ldj"], rtol=1e-5, atol=1e-4)
    assert torch.allclose(fo.spline_activation_inv(t["z"], uw, uh, ud, 10.0), t["xrec"], rtol=1e-5, atol=1e-5)
    outside = t["x"].abs() > 10.0
    assert outside.any() and torch.equal(z[outside], t["x"][outside])


@pytest.mark.parametrize("fxname", ["mnist_eye_cf", "mnist_onehot", "cifar10_onehot_cf", "cifar10_eye",
                                    "cifar10_onehot_vardeq", "cifar10_eye_vardeq_cf", "smap_onehot_cf", "smap_eye",
                                    "cifar10_eye_argmax_cf", "cifar10_embed_eyesample", "mnist_embed_probsample_cf",
                                    "atm_onehot_cf", "atm_embed_eyesample_cf", "atm_onehot_vardeq_cf", "atm_eye_argmax_cf",
                                    "atm_embed_probsample_cf"])
def test_specialist_oracle_matches_reference(fxname):
    """Context-conditioned (specialist) forward: every Conv1x1 / ActNorm / Coupling with its ContextEncoder + CN net,
    context-shifted GMM priors — oracle vs the reference's logp on the captured noise."""
    from tests.helpers import load_specialist
    name, ctx, ops, M, params, inp = load_specialist(fxname)
    z, lp = fo.flow_forward(ops, params, inp["x"], inp["u"], inp["eps"], ctx=ctx, context=inp["context"], cnoise=inp["cnoise"])
    D = int(np.prod(fo.CONFIGS[name][0]))
    ref = fo.bits_per_dim(inp["logp"], D)
    tol = max(1e-5, 1e-5 * ref.abs().max().item())      # |logp| ~ 1e6 in the un-normalised smap_eye configuration
    assert (fo.bits_per_dim(lp, D) - ref).abs().max().item() < tol
    assert (z - inp["z"]).abs().max().item() < 2e-3 * max(1.0, inp["z"].abs().max().item())


@pytest.mark.parametrize("tag", ["maf_3x3", "maf_3x1"])
def test_masked_coupling_oracle(tag):
    """MaskedCoupling (--coupling maf) restatement against the reference layer's outputs (tests/golden/unit_maf.npz)."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "unit_maf.npz"))
    p = {"0." + k[len(tag) + 4:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(tag + "/sd:")}
    x = torch.from_numpy(fx[tag + "/x"])
    z, ldj = fo.masked_coupling_fwd(x, p, "0.", tuple(int(v) for v in fx[tag + "/pad"]))
    assert (z - torch.from_numpy(fx[tag + "/z"])).abs().max() < 1e-5
    assert (ldj - torch.from_numpy(fx[tag + "/ldj"])).abs().max() < 1e-4


ACT_CASES = {"identity": "identity", "leaky": "leaky", "smooth_leaky": "smooth_leaky", "smooth_tanh": "smooth_tanh",
             "learnable_leaky": "leaky", "sigmoid": "sigmoid"}


@pytest.mark.parametrize("tag", sorted(ACT_CASES))
def test_activation_oracle(tag):
    """Elementwise activation layers (activations.py:34-118, 213-245): restatement against the reference classes' forward,
    log-det and inverse (tests/golden/unit_act.npz)."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "unit_act.npz"))
    a, b = (float(v) for v in fx[tag + "/ab"])
    x = torch.from_numpy(fx[tag + "/x"])
    z, ldj = fo.activation_fwd(ACT_CASES[tag], x, a, b)
    assert (z - torch.from_numpy(fx[tag + "/z"])).abs().max() < 1e-6
    assert (ldj - torch.from_numpy(fx[tag + "/ldj"])).abs().max() < 1e-4
    xr = fo.activation_inv(ACT_CASES[tag], torch.from_numpy(fx[tag + "/z"]), a, b, eps=1e-4)
    assert (xr - torch.from_numpy(fx[tag + "/xr"])).abs().max() < 1e-5


def test_winograd_form_of_the_3x3_against_the_direct_convolution():
    """The F(2x2,3x3) restatement (oracle/winograd.py = what winograd_phase2 computes) against the fp64 direct convolution:
    same result to fp32 rounding, error within 3x the fp32 direct convolution's own."""
    from oracle.winograd import winograd3x3_reflect
    g = torch.Generator().manual_seed(0)
    for (C, H) in [(32, 16), (64, 8), (128, 4)]:
        h = torch.relu(torch.randn(3, C, H, H, generator=g))
        w = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
        b = 0.1 * torch.randn(C, generator=g)
        ref = torch.nn.functional.conv2d(torch.nn.functional.pad(h.double(), (1, 1, 1, 1), mode="reflect"), w.double(), b.double())
        direct = torch.nn.functional.conv2d(torch.nn.functional.pad(h, (1, 1, 1, 1), mode="reflect"), w, b)
        wino = winograd3x3_reflect(h, w, b)
        e_direct = (direct.double() - ref).abs().max().item()
        e_wino = (wino.double() - ref).abs().max().item()
        assert e_wino <= 1e-5 * ref.abs().max().item()
        assert e_wino <= 3.0 * e_direct + 1e-7, (C, H, e_wino, e_direct)


def test_winograd_form_of_the_3x3_weight_gradient_against_autograd():
    """The F(3x3,2x2) restatement of the weight gradient (oracle/winograd.py = what k_wgrad<..., WINO> computes) against
    torch.autograd through the fp64 reflect-padded convolution: <= 2e-6 of the largest entry, at the three image sizes of
    the cifar10 flow (reflect borders on every tile at 4x4) and with an odd batch."""
    from oracle.winograd import winograd3x3_wgrad_reflect
    F = torch.nn.functional
    g = torch.Generator().manual_seed(11)
    for (B, Co, Ci, H) in [(3, 8, 6, 16), (5, 16, 16, 8), (7, 12, 20, 4)]:
        gy = torch.randn(B, Co, H, H, generator=g)
        h = torch.randn(B, Ci, H, H, generator=g)
        w = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
        (F.conv2d(F.pad(h.double(), (1, 1, 1, 1), mode="reflect"), w) * gy.double()).sum().backward()
        got = winograd3x3_wgrad_reflect(gy, h)
        assert (got.double() - w.grad).abs().max().item() < 2e-6 * w.grad.abs().max().item()


@pytest.mark.parametrize("name,tag", [("cifar10", None), ("mnist", "stress"), ("cifar10", "extreme")])
def test_e2e_with_the_winograd_form_keeps_the_bits_per_dim(name, tag, monkeypatch):
    """The whole flow with every coupling net's 3x3 in Winograd form (fp32) against the reference's outputs: the bar of the
    direct form (1e-5 bits/dim; "extreme": the fixture's own) holds - the evidence behind dispatching the Winograd kernels."""
    from oracle import winograd
    ops, _, M, params, fx = load_e2e(name, tag)
    x, u, eps = e2e_inputs(name, fx)
    tol = stress_tolerance(fx, tag) if tag else 1e-5
    monkeypatch.setattr(fo, "coupling_net", winograd.coupling_net_winograd)
    _, logp = fo.flow_forward(ops, params, x, u, eps)
    assert (bpd(logp, name) - bpd(torch.from_numpy(fx["logp"]), name)).abs().max() < tol
    if "logp_f64" in fx:
        assert (bpd(logp, name) - bpd(torch.from_numpy(fx["logp_f64"]), name)).abs().max() < tol
