"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden.py

The reference (/root/reference) cannot travel to the GPU box, so its outputs are committed here
as data: inputs, captured noise, the few parameters that are not bit-reproducible from a seed,
and the reference's outputs.  Large parameters are regenerated at test time from
`oracle.params.gen_params` (NumPy RandomState streams) and are loaded INTO the reference here
with `load_state_dict(strict=True)`, which also proves that `oracle.params.param_spec` names every
reference `state_dict` entry with the right shape.

Import recipe: SURVEY.md Appendix C (stub the dataset / reporting modules that need absent
third-party packages; they are not on the density path).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference/contextflow"


def _import_reference():
    sys.path.insert(0, REF)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m

    class InputDropout(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    stub("datasets")
    stub("datasets.ts", InputDropout=InputDropout, load_data_ts=None)
    stub("datasets.mnist", load_data=None)
    stub("datasets.cifar10", load_data=None)
    stub("datasets.mtad_dataloader", mtad_entities={})
    stub("torchinfo", summary=None)
    stub("ood_metrics", auroc=None, aupr=None, fpr_at_95_tpr=None)
    import model as ref_model
    import layers as ref_layers
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.")]:
        del sys.modules[k]
    return ref_model, ref_layers


REF_MODEL, REF_LAYERS = _import_reference()

from oracle import flow_oracle as fo          # noqa: E402
from oracle import params as op               # noqa: E402

REF_CFG = {
    "mnist": dict(dataset="mnist", num_blocks=2, block_size=2, split_prior=False, coupling="conv", contexts=[-1]),
    "cifar10": dict(dataset="cifar10", num_blocks=3, block_size=4, split_prior=True, coupling="conv", contexts=[-1, -1]),
    "smap": dict(dataset="smap", num_blocks=2, block_size=4, split_prior=False, coupling="trans", contexts=[55]),
    "atm": dict(dataset="atm", num_blocks=3, block_size=4, split_prior=True, coupling="trans", contexts=[68]),
}


def build_reference(name):
    data_size, mixtures = fo.CONFIGS[name][0], fo.CONFIGS[name][1]
    c = REF_CFG[name]
    REF_MODEL.c = types.SimpleNamespace(dataset=name)
    cfg = dict(dataset=name, contextflow=False, generalist=True, enc_emb="onehot", enc_type="uniform",
               num_blocks=c["num_blocks"], block_size=c["block_size"], actnorm=True, coupling=c["coupling"],
               split_prior=c["split_prior"], dist="gauss")
    flow = REF_MODEL.create_model(cfg, data_size=data_size, mixtures=mixtures, contexts=c["contexts"]).eval()
    return flow, len(c["contexts"])


def synth_input(name, B, seed):
    g = torch.Generator().manual_seed(seed)
    C, H, W = fo.CONFIGS[name][0]
    if name in ("smap", "atm"):
        return torch.rand(B, C, H, W, generator=g)
    return torch.randint(0, 256, (B, C, H, W), generator=g).float()


def end_to_end(name, B=4, seed=0):
    flow, nctx = build_reference(name)
    ops, prior_size, M = fo.program(name)
    spec = op.param_spec(ops, prior_size, M)
    sd = flow.state_dict()
    assert list(sd.keys()) == list(spec.keys()), (name, set(sd) ^ set(spec))
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(spec[k][0]), (k, v.shape, spec[k][0])
    params = op.gen_params(spec, seed)
    flow.load_state_dict(params, strict=True)

    x = synth_input(name, B, seed)
    ctx = torch.zeros(B, nctx, dtype=torch.long)
    with torch.no_grad():
        torch.manual_seed(1234 + seed)
        z_first, logp_first = flow(x, ctx)            # first call: runs the ActNorm data-dependent init
        post = {k: v.clone() for k, v in flow.state_dict().items()}
        # second pass, layer by layer with the same RNG stream, to capture noise and a trace
        torch.manual_seed(1234 + seed)
        h = x
        logdet = torch.zeros(B, M)
        trace, noise_u, noise_eps = [], None, []
        for i, m in enumerate(flow.sequence_modules):
            out, ldj = m(h, ctx)
            kind = ops[i][0]
            if kind == "dequant":
                noise_u = out - h                      # exact (Sterbenz): h + noise_u == out bitwise
                assert torch.equal(h + noise_u, out)
            if kind == "augment":
                noise_eps.append(out[:, h.shape[1]:].clone())
            logdet += ldj if ldj.dim() == 2 else ldj.unsqueeze(-1)
            trace.append((kind, out.clone(), ldj.clone()))
            h = out
        logp = flow.dist.log_prob(h, ctx) + logdet
    assert torch.equal(logp, logp_first) and torch.equal(h, z_first), name

    fx = dict(x=x.numpy().astype(np.uint8) if name not in ("smap", "atm") else x.numpy(), logp=logp.numpy(), z=h.numpy(),
              seed=np.int64(seed))
    if noise_u is not None:
        fx["u"] = noise_u.numpy()
    for j, e in enumerate(noise_eps):
        fx["eps%d" % j] = e.numpy()
    for k in op.stored_keys(spec):
        fx["param:" + k] = post[k].numpy()
    keep = set()
    first_of = {}
    for i, (kind, _, _) in enumerate(trace):
        first_of.setdefault(kind, i)
    keep.update(first_of.values())
    # first and last flow step of every resolution level + everything around split/squeeze
    for i, (kind, _, _) in enumerate(trace):
        if kind in ("squeeze", "split", "augment", "logit") and name != "atm":
            keep.update({i, min(i + 1, len(trace) - 1), min(i + 2, len(trace) - 1), min(i + 3, len(trace) - 1)})
    keep.update({len(trace) - 1, len(trace) - 2, len(trace) - 3})
    for i, (kind, out, ldj) in enumerate(trace):
        fx["ldj%d" % i] = ldj.numpy()
        if i in keep:
            fx["z%d" % i] = out.numpy()
    # fp64 evaluation of the same function (noise floor reference): run the reference itself in double
    flow64 = flow.double()
    with torch.no_grad():
        h = x.double()
        logdet = torch.zeros(B, M, dtype=torch.float64)
        eps_iter = iter(noise_eps)
        for i, m in enumerate(flow64.sequence_modules):
            kind = ops[i][0]
            if kind == "dequant":
                out, ldj = h + noise_u.double(), torch.zeros(B, dtype=torch.float64)
            elif kind == "augment":
                e = next(eps_iter).double()
                out = torch.cat([h, e], 1)
                ldj = -m.distribution.log_prob(e)
            else:
                out, ldj = m(h, ctx)
            logdet += ldj if ldj.dim() == 2 else ldj.unsqueeze(-1)
            h = out
        logp64 = flow64.dist.log_prob(h, ctx) + logdet
    fx["logp_f64"] = logp64.numpy()
    np.savez(os.path.join(HERE, "e2e_%s.npz" % name), **fx)
    D = int(np.prod(fo.CONFIGS[name][0]))
    bpd = fo.bits_per_dim(logp, D)
    bpd64 = fo.bits_per_dim(logp64, D)
    print("%-8s logp[0,:3]=%s  bpd=%s  fp32-vs-fp64 max|dbpd|=%.3e  (%d layers, %d kept z)" % (
        name, logp[0, :3].tolist(), bpd.tolist(), (bpd.double() - bpd64).abs().max().item(), len(trace), len(keep)))
    return flow.float(), ops, post


def sample_inverse_mnist(seed=0):
    """FlowSequential.sample (flowsequential.py:32-39) only works for the mnist topology
    (SURVEY §3.3).  Capture z -> x through every layer's `reverse`."""
    flow, _ = build_reference("mnist")
    ops, prior_size, M = fo.program("mnist")
    spec = op.param_spec(ops, prior_size, M)
    flow.load_state_dict(op.gen_params(spec, seed), strict=True)
    x = synth_input("mnist", 4, seed)
    with torch.no_grad():
        torch.manual_seed(7)
        flow(x, torch.zeros(4, 1, dtype=torch.long))
        post = {k: v.clone() for k, v in flow.state_dict().items()}
        torch.manual_seed(11)
        z, _ = flow.dist.sample(3)
        h = z
        for m in reversed(flow.sequence_modules):
            h = m.reverse(h, None)
        torch.manual_seed(11)
        assert torch.equal(flow.sample(3), h)
    fx = dict(z=z.numpy(), x=h.numpy(), seed=np.int64(seed))
    for k in op.stored_keys(spec):
        fx["param:" + k] = post[k].numpy()
    np.savez(os.path.join(HERE, "inverse_mnist.npz"), **fx)
    print("inverse_mnist: x range", h.min().item(), h.max().item())


def unit_layers():
    """Per-layer vectors at awkward shapes (odd sizes, (3,1) kernels, C=26 ...)."""
    L = REF_LAYERS
    fx = {}

    def put(tag, **kw):
        for k, v in kw.items():
            fx["%s/%s" % (tag, k)] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)

    def sd(tag, module):
        for k, v in module.state_dict().items():
            fx["%s/sd:%s" % (tag, k)] = v.numpy()

    torch.manual_seed(42)
    with torch.no_grad():
        # Coupling, 3x3 reflect, non-square odd image
        for tag, C, krn, pad, shape in (("coupling_3x3", 12, (3, 3), (1, 1), (3, 12, 6, 10)),
                                        ("coupling_3x1", 8, (3, 1), (1, 0), (2, 8, 9, 1)),
                                        ("coupling_c16", 16, (3, 3), (1, 1), (2, 16, 16, 16))):
            m = L.Coupling(C, kernel_size=krn, padding=pad)
            x = torch.randn(*shape)
            z, ldj = m(x)
            put(tag, x=x, z=z, ldj=ldj, h=m.NN(x[:, :C // 2]), xrec=m.reverse(z), krn=krn, pad=pad)
            sd(tag, m)
        # Conv1x1
        for tag, size, shape in (("conv1x1_c26", (26, 8, 1), (3, 26, 8, 1)), ("conv1x1_c64", (64, 4, 4), (2, 64, 4, 4))):
            m = L.Conv1x1(size)
            m.NN.add_(0.05 * torch.randn_like(m.NN))       # leave exact orthogonality: |det| != 1
            x = torch.randn(*shape)
            z, ldj = m(x)
            put(tag, x=x, z=z, ldj=ldj, xrec=m.reverse(z))
            sd(tag, m)
        # ActNorm: data-dependent init on first call, then a second batch
        m = L.ActNorm((7, 3, 4))
        x = 3.0 * torch.randn(5, 7, 3, 4) + torch.arange(7.0).view(1, 7, 1, 1)
        z, ldj = m(x)
        x2 = torch.randn(2, 7, 3, 4)
        z2, ldj2 = m(x2)
        put("actnorm", x=x, z=z, ldj=ldj, x2=x2, z2=z2, ldj2=ldj2, x2rec=m.reverse(z2))
        sd("actnorm", m)
        # Squeeze
        x = torch.randn(2, 3, 4, 6)
        put("squeeze22", x=x, z=L.Squeeze((2, 2))(x)[0], xrec=L.Squeeze((2, 2)).reverse(L.Squeeze((2, 2))(x)[0]))
        x = torch.randn(2, 5, 6, 1)
        put("squeeze21", x=x, z=L.Squeeze((2, 1))(x)[0])
        # GMM prior and SplitPrior
        m = L.GaussianMixtureDistribution(size=(5, 3, 2), mixtures=3, components=8)
        m.sG.add_(0.3 * torch.randn_like(m.sG))
        x = torch.randn(6, 5, 3, 2)
        put("gmm", x=x, logp=m.log_prob(x))
        sd("gmm", m)
        sp = L.SplitPrior(L.GaussianMixtureDistribution(size=(5, 3, 2), mixtures=2, components=8))
        x = torch.randn(4, 10, 3, 2)
        z, ldj = sp(x)
        put("split", x=x, z=z, ldj=ldj)
        sd("split", sp)
        # pre-processing
        n = L.Normalization(translation=1e-4, scale=1 / (1 - 2e-4))
        x = torch.rand(3, 2, 4, 4)
        z, ldj = n(x)
        put("normalize", x=x, z=z, ldj=ldj, xrec=n.reverse(z))
        n = L.Normalization(translation=0.0, scale=256.0)
        x = torch.randint(0, 256, (3, 2, 4, 4)).float() + torch.rand(3, 2, 4, 4)
        z, ldj = n(x)
        put("normalize256", x=x, z=z, ldj=ldj)
        lt = L.LogitTransform()
        x = torch.rand(3, 2, 4, 4) * 0.98 + 0.01
        z, ldj = lt(x)
        put("logit", x=x, z=z, ldj=ldj, xrec=lt.reverse(z))
        sn = L.StandardNormal((1, 4, 4))
        e = torch.randn(3, 1, 4, 4)
        put("stdnormal", x=e, logp=sn.log_prob(e))
        # TransCoupling (time-series geometry of smap and a small image geometry)
        for tag, in_sz, p, B in (("trans_ts", (26, 8, 1), (2, 1), 3), ("trans_img", (8, 4, 4), (2, 2), 2)):
            m = L.TransCoupling(in_sz, p)
            for prm in m.parameters():                       # LayerNorm affine away from (1, 0)
                if prm.dim() == 1:
                    prm.add_(0.1 * torch.randn_like(prm))
            x = torch.randn(B, *in_sz)
            z, ldj = m(x)
            put(tag, x=x, z=z, ldj=ldj, h=m.NN(x[:, :in_sz[0] // 2]), xrec=m.reverse(z), in_sz=in_sz, p=p)
            sd(tag, m)
        # SplineActivation (RQ spline, linear tails at +-10): shared and per-position knots, inputs inside and outside
        for tag, indiv in (("spline_shared", False), ("spline_indiv", True)):
            m = L.SplineActivation((3, 4, 5), n_bins=5, tail_bound=10., individual_weights=indiv)
            for prm in m.parameters():
                prm.mul_(60.0)                                 # default init (0.01 * randn) is almost the identity
            x = 6.0 * torch.randn(4, 3, 4, 5)
            x[0, 0, 0, :3] = torch.tensor([-10.0, 10.0, 0.0])  # exact knots / bounds
            z, ldj = m(x)
            put(tag, x=x, z=z, ldj=ldj, xrec=m.reverse(z))
            sd(tag, m)
    np.savez(os.path.join(HERE, "unit_layers.npz"), **fx)
    print("unit_layers: %d arrays" % len(fx))


# fixture tag -> dataset -> (B, lowest pre-softplus mixture scale, median max|raw| aimed at).
#  "stress":  the regime in which the reference's own fp32 answer still sits within 3e-6 bits/dim of its fp64 evaluation,
#             so the 1e-5 bits/dim parity bar is meaningful: sigma >= 0.13 (smap: >= 0.69, its M*K = 8 component slots leave
#             most of the 64 samples without a fitted component), max|raw| 6..25 (tanh saturated on most couplings).
#  "extreme": sG in [-4, 6] (sigma down to 0.018) and max|raw| 15..35: here the reference's fp32 result is itself 1e-5 ..
#             5e-5 bits/dim away from its fp64 run (stored as logp_f64), so the test bar is that measured floor, not 1e-5.
STRESS = {
    "stress": {"mnist": (64, -2.0, 8.0), "cifar10": (64, -2.0, 8.0), "smap": (64, 0.0, 8.0)},
    "extreme": {"mnist": (16, -4.0, 15.0), "cifar10": (16, -4.0, 15.0), "smap": (8, -4.0, 15.0)},
}


def end_to_end_stress(name, tag="stress", seed=7):
    """Stress fixture: the reference's own output at BASELINE config 1's batch (64) on trained-like parameters
    (oracle.params.stress_params / stress_means): saturated coupling log-scales, ActNorm log-scales of both signs spread
    over several units, ill-conditioned Conv1x1 (cond 1e3), mixture scales far from 1 with the component means sitting on
    latent samples.  Stores inputs, captured noise, the LAPACK- / init- / data-dependent parameters, per-layer log-dets,
    z and logp (fp32 and the reference run in fp64)."""
    B, sg_lo, raw_target = STRESS[tag][name]
    ops, prior_size, M = fo.program(name)
    spec = op.param_spec(ops, prior_size, M)
    x = synth_input(name, B, seed)

    def run(gain, fit):
        flow, nctx = build_reference(name)
        flow.load_state_dict(op.stress_params(op.gen_params(spec, seed), spec, seed, gain, sg_lo), strict=True)
        ctx = torch.zeros(B, nctx, dtype=torch.long)
        raws, gmm_in, hooks = [], {}, []
        for i, m in enumerate(flow.sequence_modules):
            if type(m).__name__ in ("Coupling", "TransCoupling"):
                hooks.append(m.NN.register_forward_hook(lambda mod, i, o: raws.append(float(o[:, o.shape[1] // 2:].abs().max()))))
            if type(m).__name__ == "SplitPrior":
                hooks.append(m.register_forward_hook(lambda mod, inp, o, i=i: gmm_in.__setitem__("%d.dist." % i, inp[0][:, inp[0].shape[1] // 2:].clone())))
        with torch.no_grad():
            torch.manual_seed(4321 + seed)
            z, logp = flow(x, ctx)                               # first call: ActNorm data-dependent init
            gmm_in["dist."] = z.clone()
            if fit:                                              # fitted mixtures: component means on the latent samples
                sd = flow.state_dict()
                for pre, zin in gmm_in.items():
                    sd[pre + "mG"].copy_(op.stress_means(zin, sd[pre + "sG"], seed, pre + "mG"))
        for h in hooks:
            h.remove()
        return flow, ctx, raws

    _, _, raws = run(1.0, False)
    gain = float(np.float32(raw_target / np.median(raws)))       # median layer reaches the target; others spread around it
    flow, ctx, raws = run(gain, True)
    post = {k: v.clone() for k, v in flow.state_dict().items()}
    with torch.no_grad():                                        # second pass with the same RNG stream: capture the noise
        torch.manual_seed(4321 + seed)
        h, logdet, ldjs, noise_u, noise_eps = x, torch.zeros(B, M), [], None, []
        for i, m in enumerate(flow.sequence_modules):
            out, ldj = m(h, ctx)
            if ops[i][0] == "dequant":
                noise_u = out - h
                assert torch.equal(h + noise_u, out)
            if ops[i][0] == "augment":
                noise_eps.append(out[:, h.shape[1]:].clone())
            logdet += ldj if ldj.dim() == 2 else ldj.unsqueeze(-1)
            ldjs.append(ldj.clone())
            h = out
        logp = flow.dist.log_prob(h, ctx) + logdet
    assert torch.isfinite(logp).all()
    fx = dict(x=x.numpy().astype(np.uint8) if name != "smap" else x.numpy(), logp=logp.numpy(), z=h.numpy(),
              seed=np.int64(seed), raw_gain=np.float32(gain), sg_lo=np.float32(sg_lo), raw_absmax=np.asarray(raws, dtype=np.float32))
    if noise_u is not None:
        fx["u"] = noise_u.numpy()
    for j, e in enumerate(noise_eps):
        fx["eps%d" % j] = e.numpy()
    for k in op.stored_keys(spec) + [k for k in spec if k.endswith("mG")]:
        fx["param:" + k] = post[k].numpy()
    for i, l in enumerate(ldjs):
        fx["ldj%d" % i] = l.numpy()
    flow64 = flow.double()
    with torch.no_grad():
        h, logdet, eps_iter = x.double(), torch.zeros(B, M, dtype=torch.float64), iter(noise_eps)
        for i, m in enumerate(flow64.sequence_modules):
            if ops[i][0] == "dequant":
                out, ldj = h + noise_u.double(), torch.zeros(B, dtype=torch.float64)
            elif ops[i][0] == "augment":
                e = next(eps_iter).double()
                out, ldj = torch.cat([h, e], 1), -m.distribution.log_prob(e)
            else:
                out, ldj = m(h, ctx)
            logdet += ldj if ldj.dim() == 2 else ldj.unsqueeze(-1)
            h = out
        logp64 = flow64.dist.log_prob(h, ctx) + logdet
    fx["logp_f64"] = logp64.numpy()
    D = int(np.prod(fo.CONFIGS[name][0]))
    floor = (fo.bits_per_dim(logp, D) - fo.bits_per_dim(logp64, D)).abs().max().item()
    fx["floor_bpd"] = np.float64(floor)                           # the reference's own fp32 result vs its fp64 run
    np.savez(os.path.join(HERE, "e2e_%s_%s.npz" % (name, tag)), **fx)
    logs = torch.cat([v.flatten() for k, v in post.items() if k.endswith("NN_logs")])
    conds = [float(torch.linalg.cond(v.double())) for k, v in post.items() if k.endswith(".NN") and v.dim() == 2]
    print("%s %-8s B %d gain %.2f  max|raw| per coupling %s" % (tag, name, B, gain, np.round(raws, 1)))
    print("   ActNorm logs in [%.2f, %.2f]  Conv1x1 cond %.0f..%.0f  bits/dim %.3f..%.3f  fp32 reference vs its fp64 run: %.2e bits/dim"
          % (logs.min(), logs.max(), min(conds), max(conds), fo.bits_per_dim(logp, D).min(), fo.bits_per_dim(logp, D).max(), floor))


def end_to_end_wide(name="smap", B=512, seed=11):
    """Wide fixture: the reference's fp32 and fp64 answers on B fresh samples with the parameters of e2e_<name>.npz (post
    ActNorm init).  Only inputs, captured augment noise and logp are stored.  Its purpose is statistics: the one-kernel
    transformer step the benchmark runs (cf_vit_step_fwd, batches > TransCoupling.STEP_RS_MAX_BATCH) is compared per sample
    with the reference on hundreds of distinct samples, tiled past the dispatch threshold."""
    from tests.helpers import load_e2e
    ops, prior_size, M, params, _ = load_e2e(name)
    flow, nctx = build_reference(name)
    flow.load_state_dict(params, strict=True)
    x = synth_input(name, B, seed)
    ctx = torch.zeros(B, nctx, dtype=torch.long)
    with torch.no_grad():
        torch.manual_seed(99 + seed)
        h, logdet, noise_eps = x, torch.zeros(B, M), []
        for i, m in enumerate(flow.sequence_modules):
            out, ldj = m(h, ctx)
            assert ops[i][0] != "dequant"
            if ops[i][0] == "augment":
                noise_eps.append(out[:, h.shape[1]:].clone())
            logdet += ldj if ldj.dim() == 2 else ldj.unsqueeze(-1)
            h = out
        logp = flow.dist.log_prob(h, ctx) + logdet
        flow64 = flow.double()
        h, logdet, eps_iter = x.double(), torch.zeros(B, M, dtype=torch.float64), iter(noise_eps)
        for i, m in enumerate(flow64.sequence_modules):
            if ops[i][0] == "augment":
                e = next(eps_iter).double()
                out, ldj = torch.cat([h, e], 1), -m.distribution.log_prob(e)
            else:
                out, ldj = m(h, ctx)
            logdet += ldj if ldj.dim() == 2 else ldj.unsqueeze(-1)
            h = out
        logp64 = flow64.dist.log_prob(h, ctx) + logdet
    D = int(np.prod(fo.CONFIGS[name][0]))
    d = (fo.bits_per_dim(logp, D) - fo.bits_per_dim(logp64, D)).abs()
    fx = dict(x=x.numpy(), logp=logp.numpy(), logp_f64=logp64.numpy(), seed=np.int64(seed), floor_bpd=np.float64(d.max().item()))
    for j, e in enumerate(noise_eps):
        fx["eps%d" % j] = e.numpy()
    np.savez(os.path.join(HERE, "e2e_%s_wide.npz" % name), **fx)
    print("wide %s B %d: fp32 reference vs its fp64 run: max %.2e rms %.2e bits/dim" % (name, B, d.max().item(), d.pow(2).mean().sqrt().item()))


if __name__ == "__main__":
    for name in ("mnist", "cifar10", "smap"):
        end_to_end(name)
    end_to_end("atm")
    for tag in ("stress", "extreme"):
        for name in ("mnist", "cifar10", "smap"):
            end_to_end_stress(name, tag)
    end_to_end_wide("smap")
    sample_inverse_mnist()
    unit_layers()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-24s %8.1f KB" % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))
