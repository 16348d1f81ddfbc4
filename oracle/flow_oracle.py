"""CPU oracle for the ContextFlow coupling-layer density path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-torch (CPU, fp32 or fp64) *restatement* of the reference's
forward / inverse flow arithmetic.  It is NOT part of the product: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
The product path (`contextflow_amd/`) never does and fails loudly without its HIP
library.

Parity status: PINNED.  The reference holds no golden vectors of its own
(SURVEY.md §4), so the pins are outputs of the reference itself, run in the build
container by `tests/golden/make_golden.py` and committed as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against them.

Every function cites the reference lines (relative to /root/reference/contextflow)
whose arithmetic it restates.  The model is described by a flat *program* — a list
of op tuples produced by `program()` — and a flat dict of tensors keyed with the
reference's `state_dict` names, so no nn.Module is involved.
"""
import math

import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)
ALPHA = 1e-4          # model.py:96
K_COMPONENTS = 8      # model.py:115


# --------------------------------------------------------------------------------------
# topology  (model.py:95-163 restated as data)
# --------------------------------------------------------------------------------------
CONFIGS = {
    # name: (data_size, mixtures, num_blocks, block_size, split_prior, coupling)   model.py:173-220
    "mnist": ((1, 32, 32), 10, 2, 2, False, "conv"),
    "cifar10": ((3, 32, 32), 10, 3, 4, True, "conv"),
    "smap": ((25, 8, 1), 1, 2, 4, False, "trans"),
    "atm": ((38, 144, 1), 2, 3, 4, True, "trans"),          # model.py:189-198,281
}


def program(dataset, data_size=None, mixtures=None, num_blocks=None, block_size=None,
            split_prior=None, coupling=None):
    """Layer list of `create_model` (model.py:95-163) for the generalist / context-free case.

    Returns (ops, prior_size, mixtures).  Each op is a tuple whose 2nd element is the index
    of the layer inside the reference's FlowSequential (= its state_dict key prefix).
    """
    d = CONFIGS[dataset]
    data_size = data_size or d[0]
    mixtures = mixtures or d[1]
    num_blocks = num_blocks or d[2]
    block_size = block_size or d[3]
    split_prior = d[4] if split_prior is None else split_prior
    coupling = coupling or d[5]
    ts = dataset in ("atm", "msl", "smd", "smap")            # model.py:113
    patch, krn, pad = ((2, 1), (3, 1), (1, 0)) if ts else ((2, 2), (3, 3), (1, 1))  # model.py:114
    ops = []
    if dataset in ("mnist", "cifar10"):                       # model.py:97-100
        ops += [("dequant", 0), ("affine", 1, 0.0, 256.0),
                ("affine", 2, ALPHA, 1.0 / (1.0 - 2.0 * ALPHA)), ("logit", 3)]
    sz = tuple(data_size)
    for blk in range(num_blocks):
        if sz[0] % 2:                                         # model.py:121-123
            ops.append(("augment", len(ops), 1)); sz = (sz[0] + 1, sz[1], sz[2])
        if dataset not in ("msl", "smd", "smap"):             # model.py:125-127
            ops.append(("squeeze", len(ops), patch))
            sz = (sz[0] * patch[0] * patch[1], sz[1] // patch[0], sz[2] // patch[1])
        for _ in range(block_size):
            ops.append(("conv1x1", len(ops), sz))
            ops.append(("actnorm", len(ops), sz))
            if coupling == "trans" and sz[1] % patch[0] == 0 and sz[2] % patch[1] == 0:  # model.py:139
                ops.append(("transcoupling", len(ops), sz, patch))
            elif coupling == "conv":
                ops.append(("coupling", len(ops), sz, krn, pad))
            if dataset == "atm":                              # model.py:149-151
                ops.append(("permute", len(ops), (0, 2, 1, 3)))
                sz = (sz[1], sz[0], sz[2])
        if split_prior and blk < num_blocks - 1:              # model.py:153-158
            sz = (sz[0] // 2, sz[1], sz[2])
            ops.append(("split", len(ops), sz))
    return ops, sz, mixtures


# --------------------------------------------------------------------------------------
# index-only ops
# --------------------------------------------------------------------------------------
def squeeze_fwd(x, p):
    """squeeze.py:10-11  'b c (h p1) (w p2) -> b (c p1 p2) h w'."""
    B, C, H, W = x.shape
    v = x.reshape(B, C, H // p[0], p[0], W // p[1], p[1])
    return v.permute(0, 1, 3, 5, 2, 4).reshape(B, C * p[0] * p[1], H // p[0], W // p[1])


def squeeze_inv(z, p):
    """squeeze.py:13-14."""
    B, C, H, W = z.shape
    c = C // (p[0] * p[1])
    v = z.reshape(B, c, p[0], p[1], H, W)
    return v.permute(0, 1, 4, 2, 5, 3).reshape(B, c, H * p[0], W * p[1])


# --------------------------------------------------------------------------------------
# pre-processing (ldj is part of bits/dim)
# --------------------------------------------------------------------------------------
def affine_fwd(x, translation, scale):
    """normalize.py:27-34,42-49 with scalar scale: out = x/scale + translation,
    ldj = -C*(H*W)*log(scale) for every sample."""
    s = torch.tensor([scale], dtype=torch.float32).to(x.dtype)      # torch.Tensor([scale]) is fp32
    t = torch.tensor([translation], dtype=torch.float32).to(x.dtype)
    B, C = x.shape[:2]
    n = x.numel() / B / C
    ldj = C * (-1.0 * n * torch.log(s).sum())
    return x / s + t, ldj.expand(B)


def affine_inv(y, translation, scale):
    """normalize.py:36-40."""
    s = torch.tensor([scale], dtype=torch.float32).to(y.dtype)
    t = torch.tensor([translation], dtype=torch.float32).to(y.dtype)
    return (y - t) * s


def logit_fwd(x):
    """transforms.py:11-18."""
    l0, l1 = torch.log(x), torch.log(1 - x)
    return l0 - l1, (-l0 - l1).flatten(1).sum(-1)


def std_normal_neg_logq(eps):
    """augment.py:14-18 + gaussian.py:50-54: ldj of Augment = -log N(eps;0,1), shape (B,1)."""
    lp = (-0.5 * LOG_2PI - 0.5 * eps ** 2).flatten(1).sum(-1)
    return -lp.unsqueeze(-1)


# --------------------------------------------------------------------------------------
# flow layers
# --------------------------------------------------------------------------------------
def conv1x1_fwd(x, W):
    """conv1x1.py:52-57 (context-free): z = W x per pixel, ldj = slogdet(W)*H*W."""
    B, C, H, Wd = x.shape
    z = torch.einsum("oi,bihw->bohw", W, x)
    ldj = torch.linalg.slogdet(W)[1] * H * Wd
    return z, ldj.expand(B)


def conv1x1_inv(z, W):
    """conv1x1.py:72."""
    return torch.einsum("oi,bihw->bohw", torch.inverse(W), z)


def actnorm_stats(x):
    """actnorm.py:28-35: data-dependent init, unbiased std over (B,H,W), log(std + 1e-8)."""
    dims = [0, 2, 3]
    return torch.mean(x, dim=dims), torch.log(torch.std(x, dim=dims) + 1e-8)


def actnorm_fwd(x, t, logs):
    """actnorm.py:53-60: z = (x-t)*exp(-logs); ldj = +sum_c logs (no H*W factor — reference quirk)."""
    z = (x - t.view(1, -1, 1, 1)) * torch.exp(-logs.view(1, -1, 1, 1))
    return z, logs.sum().expand(x.shape[0])


def actnorm_inv(z, t, logs):
    """actnorm.py:78."""
    return z * torch.exp(logs.view(1, -1, 1, 1)) + t.view(1, -1, 1, 1)


def coupling_net(x0, p, prefix, pad):
    """coupling.py:26-29: 1x1 -> ReLU -> k x k (reflect pad) -> ReLU -> 1x1."""
    h = F.relu(F.conv2d(x0, p[prefix + "NN.0.weight"], p[prefix + "NN.0.bias"]))
    if pad[0] or pad[1]:
        h = F.pad(h, (pad[1], pad[1], pad[0], pad[0]), mode="reflect")
    h = F.relu(F.conv2d(h, p[prefix + "NN.2.weight"], p[prefix + "NN.2.bias"]))
    return F.conv2d(h, p[prefix + "NN.4.weight"], p[prefix + "NN.4.bias"])


def affine_from_net(h):
    """coupling.py:52-57: t = first half of h, log_s = 2*tanh(second half / 2)."""
    c = h.shape[1] // 2
    return h[:, :c], 2.0 * torch.tanh(h[:, c:] / 2.0)


def coupling_apply_fwd(x, h):
    """coupling.py:60-66: second channel half transformed, first half is the conditioner."""
    c = x.shape[1] // 2
    t, log_s = affine_from_net(h)
    z1 = x[:, c:] * torch.exp(log_s) + t
    return torch.cat([x[:, :c], z1], 1), log_s.flatten(1).sum(-1)


def coupling_apply_inv(z, h):
    """coupling.py:68-73."""
    c = z.shape[1] // 2
    t, log_s = affine_from_net(h)
    return torch.cat([z[:, :c], (z[:, c:] - t) / torch.exp(log_s)], 1)


def coupling_fwd(x, p, prefix, pad):
    return coupling_apply_fwd(x, coupling_net(x[:, : x.shape[1] // 2], p, prefix, pad))


def coupling_inv(z, p, prefix, pad):
    return coupling_apply_inv(z, coupling_net(z[:, : z.shape[1] // 2], p, prefix, pad))


# ---- SimpleViT inner net of TransCoupling ------------------------------------------------
def posemb_sincos_2d(h, w, dim, temperature=10000.0):
    """simple_vit.py:18-27 (note omega = arange(dim/4)/(dim/4 - 1))."""
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    omega = torch.arange(dim // 4) / (dim // 4 - 1)
    omega = 1.0 / (temperature ** omega)
    y = y.flatten()[:, None] * omega[None, :]
    x = x.flatten()[:, None] * omega[None, :]
    return torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1).float()


def vit_dims(sz, patch):
    """coupling.py:103-114: dim = O*p1*p2, heads 1, dim_head 64, depth 6, mlp_dim = dim."""
    C, H, W = sz
    return dict(cin=C // 2, dim=C * patch[0] * patch[1], gh=H // patch[0], gw=W // patch[1],
                depth=6, dim_head=64, patch_dim=(C // 2) * patch[0] * patch[1])


def vit_net(x0, p, prefix, sz, patch, concat=False):
    """simple_vit.py:117-127 with Attention (:56-68), FeedForward (:30-40), Transformer (:71-88).
    concat: the specialist non-contextflow variant (coupling.py:114-115): input = [x0 ; CN(c) broadcast], C/2 + C
    channels, and the ViT is a direct child (`NN.` instead of `NN.0.`)."""
    d = dict(vit_dims(sz, patch))
    B = x0.shape[0]
    p1, p2, gh, gw, dim = patch[0], patch[1], d["gh"], d["gw"], d["dim"]
    q = prefix + ("NN." if concat else "NN.0.")
    if concat:
        d["cin"] = x0.shape[1]
        d["patch_dim"] = d["cin"] * p1 * p2
    # 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)'
    tok = x0.reshape(B, d["cin"], gh, p1, gw, p2).permute(0, 2, 4, 3, 5, 1).reshape(B, gh * gw, d["patch_dim"])
    tok = F.layer_norm(tok, (d["patch_dim"],), p[q + "to_patch_embedding.1.weight"], p[q + "to_patch_embedding.1.bias"])
    tok = F.linear(tok, p[q + "to_patch_embedding.2.weight"], p[q + "to_patch_embedding.2.bias"])
    tok = F.layer_norm(tok, (dim,), p[q + "to_patch_embedding.3.weight"], p[q + "to_patch_embedding.3.bias"])
    tok = tok + posemb_sincos_2d(gh, gw, dim).to(tok.dtype)
    for l in range(d["depth"]):
        a = q + "transformer.layers.%d.0." % l
        f = q + "transformer.layers.%d.1.net." % l
        y = F.layer_norm(tok, (dim,), p[a + "norm.weight"], p[a + "norm.bias"])
        qkv = F.linear(y, p[a + "to_qkv.weight"])
        qq, kk, vv = qkv.chunk(3, dim=-1)
        att = torch.softmax(torch.matmul(qq, kk.transpose(-1, -2)) * d["dim_head"] ** -0.5, dim=-1)
        tok = F.linear(torch.matmul(att, vv), p[a + "to_out.weight"]) + tok
        y = F.layer_norm(tok, (dim,), p[f + "0.weight"], p[f + "0.bias"])
        y = F.linear(F.gelu(F.linear(y, p[f + "1.weight"], p[f + "1.bias"])), p[f + "3.weight"], p[f + "3.bias"])
        tok = y + tok
    tok = F.layer_norm(tok, (dim,), p[q + "transformer.norm.weight"], p[q + "transformer.norm.bias"])
    # 'b (h w) (p1 p2 c) -> b c (h p1) (w p2)'  with c = dim/(p1 p2) = full channel count
    cout = dim // (p1 * p2)
    return tok.reshape(B, gh, gw, p1, p2, cout).permute(0, 5, 1, 3, 2, 4).reshape(B, cout, gh * p1, gw * p2)


def transcoupling_fwd(x, p, prefix, sz, patch):
    """coupling.py:123-147."""
    return coupling_apply_fwd(x, vit_net(x[:, : x.shape[1] // 2], p, prefix, sz, patch))


def transcoupling_inv(z, p, prefix, sz, patch):
    """coupling.py:149-155."""
    return coupling_apply_inv(z, vit_net(z[:, : z.shape[1] // 2], p, prefix, sz, patch))


# ---- rational-quadratic spline activation (activations.py:120-211, splines/rational_quadratic.py) ----
def rq_spline_tables(uw, uh, ud, tail_bound, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """Knot tables (cumwidths, cumheights, derivatives), each (..., K+1).  rational_quadratic.py:36-48,98-118:
    the (K-1) inner derivative parameters are zero-padded to K+1 and the constant log(exp(1-min_d)-1) is added
    to ALL of them, so the two boundary derivatives are exactly 1 (linear tails)."""
    K = uw.shape[-1]
    const = math.log(math.exp(1 - min_d) - 1)
    ud = F.pad(ud, (1, 1)) + const

    def knots(u, lo, hi, m):
        v = m + (1 - m * K) * torch.softmax(u, dim=-1)
        c = F.pad(torch.cumsum(v, dim=-1), (1, 0))
        c = (hi - lo) * c + lo
        c[..., 0], c[..., -1] = lo, hi
        return c

    return knots(uw, -tail_bound, tail_bound, min_w), knots(uh, -tail_bound, tail_bound, min_h), min_d + F.softplus(ud)


def rq_spline(x, uw, uh, ud, tail_bound=10.0, inverse=False):
    """Elementwise spline with linear tails; parameters broadcast against x with a trailing knot axis.
    Returns (y, logabsdet) with logabsdet = 0 and y = x outside [-tail_bound, tail_bound]."""
    cw, ch, dv = rq_spline_tables(uw, uh, ud, tail_bound)
    cw, ch, dv = (t.expand(x.shape + (t.shape[-1],)) for t in (cw, ch, dv))
    K = cw.shape[-1] - 1
    inside = (x >= -tail_bound) & (x <= tail_bound)
    loc = (ch if inverse else cw).clone()
    loc[..., -1] += 1e-6                                               # searchsorted eps (rational_quadratic.py:13-18)
    idx = ((x[..., None] >= loc).sum(-1) - 1).clamp(0, K - 1)[..., None]
    g = lambda t: t.gather(-1, idx)[..., 0]
    w0, h0 = g(cw), g(ch)
    w = g(cw[..., 1:]) - w0
    h = g(ch[..., 1:]) - h0
    d0, d1 = g(dv), g(dv[..., 1:])
    delta = h / w
    if inverse:
        a = (x - h0) * (d0 + d1 - 2 * delta) + h * (delta - d0)
        b = h * d0 - (x - h0) * (d0 + d1 - 2 * delta)
        c = -delta * (x - h0)
        root = (2 * c) / (-b - torch.sqrt(b * b - 4 * a * c))
        y = root * w + w0
        th = root
    else:
        th = (x - w0) / w
        y = h0 + h * (delta * th * th + d0 * th * (1 - th)) / (delta + (d0 + d1 - 2 * delta) * th * (1 - th))
    den = delta + (d0 + d1 - 2 * delta) * th * (1 - th)
    lad = torch.log(delta * delta * (d1 * th * th + 2 * delta * th * (1 - th) + d0 * (1 - th) ** 2)) - 2 * torch.log(den)
    if inverse:
        lad = -lad
    return torch.where(inside, y, x), torch.where(inside, lad, torch.zeros_like(lad))


def spline_activation_fwd(x, uw, uh, ud, tail_bound=10.0):
    """SplineActivation.forward (activations.py:166-180): (act, ldj (B,)).  uw/uh: (K,) or (1,C,H,W,K)."""
    y, lad = rq_spline(x, uw, uh, ud, tail_bound, False)
    return y, lad.flatten(1).sum(-1)


def spline_activation_inv(y, uw, uh, ud, tail_bound=10.0):
    """SplineActivation.reverse (activations.py:182-194)."""
    return rq_spline(y, uw, uh, ud, tail_bound, True)[0]



def maf_mask(mask_type, cin, cout, kh, kw, data_channels):
    """autoregressive/utils.py:27-91: causal spatial mask, channel-autoregressive centre tap."""
    base = torch.ones(data_channels, data_channels).tril(-1 if mask_type == "A" else 0)
    ch = base.repeat(cout // data_channels + 1, cin // data_channels + 1)[:cout, :cin]
    mask = torch.ones(cout, cin, kh, kw)
    mask[:, :, kh // 2, kw // 2] = ch
    mask[:, :, kh // 2, kw // 2 + 1:] = 0
    mask[:, :, kh // 2 + 1:] = 0
    return mask


def masked_coupling_fwd(x, p, prefix, pad):
    """MaskedCoupling.forward (ar.py:33-57) over MaskedResidualBlock2d (masked_conv_2d.py:81-98): ReLU in FRONT of every
    masked conv, identity = x repeated twice; all channels transformed; ldj = sum log_s."""
    D = x.shape[1]

    def conv(name, h, padding):
        w = p[prefix + name + ".weight"] * p[prefix + name + ".mask"].to(h.dtype)
        if padding[0] or padding[1]:
            h = F.pad(h, (padding[1], padding[1], padding[0], padding[0]), mode="reflect")
        return F.conv2d(h, w, p[prefix + name + ".bias"])
    h = conv("NN.conv1", F.relu(x), (0, 0))
    h = conv("NN.conv2", F.relu(h), pad)
    h = conv("NN.conv3", F.relu(h), (0, 0)) + x.repeat(1, 2, 1, 1)
    t, log_s = h[:, :D], 2.0 * torch.tanh(h[:, D:] / 2.0)
    return x * torch.exp(log_s) + t, log_s.flatten(1).sum(-1)


# --------------------------------------------------------------------------------------
# specialist (context-conditioned) branches          SURVEY 8(f) rank 2
# --------------------------------------------------------------------------------------
# `ctx` = dict(contexts=[K_0, K_1, ...], enc_emb='eye'|'onehot', contextflow=bool) describes the reference's
# `ContextEncoder(contexts, enc_emb, 'uniform', ...)` (model.py:30-90) that every Conv1x1 / ActNorm / Coupling owns in
# the specialist mode (model.py:117,130-143), and the 'embed' + 'eyesample' lookup of the priors (model.py:157,162).
def ctx_width(ctx, data_dim=None):
    """ContextEncoder.C (model.py:33-47,87): one-hot width, number of context variables, (even) number of code bits
    for argmax, or data_dim * number of variables for the embedding encoders."""
    if ctx["enc_emb"] == "onehot":
        return sum(ctx["contexts"])
    if ctx["enc_emb"] == "embed":
        return data_dim * len(ctx["contexts"])
    if ctx.get("enc_type") == "argmax":
        n = sum(argmax_bits(ctx["contexts"]))
        return n + n % 2
    return len(ctx["contexts"])


def ctx_encode(context, ctx, u):
    """OneHotEncoder / EyeEncoder (rtdl/nn/_embeddings.py:76-150) followed by UniformCatDequantization
    (dequantize.py:55-64): z = (x + u) / qbins, ldj = sum_d(-log(qbins_d) * n_dims) for every sample."""
    K = ctx["contexts"]
    if ctx["enc_emb"] == "onehot":
        x = torch.cat([F.one_hot(context[:, i], K[i]) for i in range(len(K))], 1).to(u.dtype)
        qbins = torch.ones(sum(K), dtype=torch.float32)
    else:
        x = context.to(u.dtype)
        qbins = torch.tensor(K, dtype=torch.float32)
    z = (x + u) / qbins.to(u.dtype)
    ldj = ((-torch.log(qbins)) * x.shape[1]).sum(-1).to(u.dtype)
    return z, ldj.repeat(x.shape[0])


def _ctx_code(context, ctx, dtype):
    K = ctx["contexts"]
    if ctx["enc_emb"] == "onehot":
        return (torch.cat([F.one_hot(context[:, i], K[i]) for i in range(len(K))], 1).to(dtype),
                torch.ones(sum(K), dtype=torch.float32))
    return context.to(dtype), torch.tensor(K, dtype=torch.float32)


def _enc_flow_sample(context, ctx, params, e, n, eps):
    """FlowInvSequential(ConditionalGaussianDistribution, 2 x [FC, ActNormFC, CouplingFC]).sample
    (flowsequential.py:58-68, gaussian.py:263-270, model.py:52-66): returns (u, log q(u))."""
    c = torch.cat([params[e + "dist.context_net._embeddings.%d.weight" % i][context[:, i]] for i in range(len(ctx["contexts"]))], 1)
    mean, ls = c[:, :n], c[:, n:]
    u = mean + ls.exp() * eps
    logq = (-0.5 * LOG_2PI - ls - 0.5 * torch.exp(-2 * ls) * (u - mean) ** 2).sum(-1)
    for l in range(2):
        W = params[e + "%d.NN" % (3 * l)]                                          # FC (conv1x1.py:80-96)
        u = u @ W.t()
        logq = logq - torch.linalg.slogdet(W)[1]
        t, logs = params[e + "%d.NN_t" % (3 * l + 1)], params[e + "%d.NN_logs" % (3 * l + 1)]      # ActNormFC
        u = (u - t) * torch.exp(-logs)
        logq = logq - logs.sum()
        q = e + "%d." % (3 * l + 2)                                                # CouplingFC (1x1 convs)
        h = F.relu(u[:, : n // 2] @ params[q + "NN.0.weight"].flatten(1).t() + params[q + "NN.0.bias"])
        h = F.relu(h @ params[q + "NN.2.weight"].flatten(1).t() + params[q + "NN.2.bias"])
        h = h @ params[q + "NN.4.weight"].flatten(1).t() + params[q + "NN.4.bias"]
        tt, lsc = h[:, : n // 2], 2.0 * torch.tanh(h[:, n // 2:] / 2.0)
        u = torch.cat([u[:, : n // 2], u[:, n // 2:] * torch.exp(lsc) + tt], 1)
        logq = logq - lsc.sum(-1)
    return u, logq


def argmax_bits(contexts):
    """ArgmaxCatDequantization.cats2bits (dequantize.py:189-194)."""
    return [int(math.ceil(math.log2(k))) for k in contexts]


def ctx_encode_argmax(context, ctx, params, prefix, eps):
    """ArgmaxCatDequantization.forward (dequantize.py:236-262): z = sigmoid(u) * (2 bits - 1), ldj = ldj_sigmoid - log q."""
    bits = argmax_bits(ctx["contexts"])
    cols = []
    for i, nb in enumerate(bits):
        powers = 2 ** torch.arange(nb - 1, -1, -1)
        cols.append((context[:, i, None] // powers) % 2)
    code = torch.cat(cols, -1).to(eps.dtype)
    if code.shape[1] % 2:
        code = torch.cat([code, torch.zeros(code.shape[0], 1, dtype=eps.dtype)], -1)
    u, logq = _enc_flow_sample(context, ctx, params, prefix + "encoder.", code.shape[1], eps)
    act_ldj = (-F.softplus(-u) - F.softplus(u)).sum(-1)
    return torch.sigmoid(u) * (code * 2 - 1), act_ldj - logq


def ctx_encode_probsample(context, ctx, params, prefix, eps):
    """ProbSampling.forward (dequantize.py:152-161): the code IS the sigmoid of the flow sample; ldj = ldj_sigmoid + log q
    (the reference's sign)."""
    u, logq = _enc_flow_sample(context, ctx, params, prefix + "encoder.", eps.shape[1], eps)
    act_ldj = (-F.softplus(-u) - F.softplus(u)).sum(-1)
    return torch.sigmoid(u), act_ldj + logq


def ctx_encode_vardeq(context, ctx, params, prefix, eps):
    """VariationalCatDequantization (dequantize.py:104-118) over the reference's encoder flow (model.py:52-79):
    u ~ FlowInvSequential(ConditionalGaussianDistribution(embedding lookup), 2 x [FC, ActNormFC, CouplingFC])
    (flowsequential.py:58-68, gaussian.py:263-270), then Sigmoid (activations.py:234-238):
    z = (x + sigmoid(u)) / qbins, ldj = sum_d(-log qbins_d * n_dims) + ldj_sigmoid - log q(u).
    `prefix` = '<layer>.context_net.1.'; ActNormFC parameters must be initialised (post first call)."""
    x, qbins = _ctx_code(context, ctx, eps.dtype)
    n = x.shape[1]
    u, logq = _enc_flow_sample(context, ctx, params, prefix + "encoder.", n, eps)
    act_ldj = (-F.softplus(-u) - F.softplus(u)).sum(-1)                          # temperature 1
    z = (x + torch.sigmoid(u)) / qbins.to(eps.dtype)
    ldj = ((-torch.log(qbins)) * n).sum(-1).to(eps.dtype)
    return z, ldj + act_ldj - logq


def conv1x1_ctx_fwd(x, W, cn_w, cn_b, c, logp_c, contextflow):
    """conv1x1.py:34-50: per-sample triangular matrix from CN(c); log-det from its diagonal."""
    B, D, H, Wd = x.shape
    m = (c @ cn_w.t() + cn_b).reshape(B, D, D)
    diag = torch.diagonal(torch.tril(m), dim1=-2, dim2=-1)
    c_ldj = diag.sum(-1)
    tri = torch.tril(m, diagonal=-1) + torch.diag_embed(torch.exp(diag))
    if contextflow:
        tri = tri - torch.eye(D, dtype=x.dtype) + W
        ldj = H * Wd * (torch.linalg.slogdet(W)[1] + c_ldj)
    else:
        ldj = H * Wd * c_ldj
    z = torch.einsum("boi,bihw->bohw", tri, x)
    return z, ldj + logp_c * H * Wd


def actnorm_ctx_fwd(x, t, logs, cn_w, cn_b, c, logp_c, contextflow):
    """actnorm.py:40-60: per-sample shift / log-scale from CN(c), added to the shared ones under contextflow."""
    B, D, H, Wd = x.shape
    m = c @ cn_w.t() + cn_b
    tb, lb = m[:, :D], m[:, D:]
    if contextflow:
        tb, lb = tb + t.view(1, -1), lb + logs.view(1, -1)
    z = (x - tb.view(B, D, 1, 1)) * torch.exp(-lb.view(B, D, 1, 1))
    return z, lb.sum(-1) + logp_c * H * Wd


def coupling_cn(c, p, prefix):
    """coupling.py:37: CN = Linear -> ReLU -> Linear -> ReLU -> Linear."""
    h = F.relu(c @ p[prefix + "CN.0.weight"].t() + p[prefix + "CN.0.bias"])
    h = F.relu(h @ p[prefix + "CN.2.weight"].t() + p[prefix + "CN.2.bias"])
    return h @ p[prefix + "CN.4.weight"].t() + p[prefix + "CN.4.bias"]


def coupling_ctx_fwd(x, p, prefix, pad, c, logp_c, contextflow):
    """coupling.py:39-66 with a context net: additive CN(c) on the net output (contextflow) or CN(c) broadcast over
    the image and concatenated to the conditioner input."""
    B, C, H, Wd = x.shape
    x0 = x[:, : C // 2]
    cn = coupling_cn(c, p, prefix)
    if contextflow:
        h = coupling_net(x0, p, prefix, pad) + cn.view(B, -1, 1, 1)
    else:
        h = coupling_net(torch.cat([x0, cn.view(B, -1, 1, 1).expand(B, cn.shape[1], H, Wd)], 1), p, prefix, pad)
    z, ldj = coupling_apply_fwd(x, h)
    return z, ldj + logp_c * H * Wd


def transcoupling_ctx_fwd(x, p, prefix, sz, patch, c, logp_c, contextflow):
    """coupling.py:123-147 with a context net; note: logp_c is NOT multiplied by H*W here (reference quirk)."""
    B, C, H, Wd = x.shape
    x0 = x[:, : C // 2]
    cn = coupling_cn(c, p, prefix)
    if contextflow:
        h = vit_net(x0, p, prefix, sz, patch) + cn.view(B, -1, 1, 1)
    else:
        h = vit_net(torch.cat([x0, cn.view(B, -1, 1, 1).expand(B, cn.shape[1], H, Wd)], 1), p, prefix, sz, patch, concat=True)
    z, ldj = coupling_apply_fwd(x, h)
    return z, ldj + logp_c


def gmm_ctx_logprob(x, mG, sG, wG, emb, context, chunk=16):
    """gaussian.py:142-158 with the 'embed' + 'eyesample' context net (model.py:157,162): per-sample additive
    shifts of the component means and pre-softplus scales, constant over (h, w); logp_c = 0."""
    B = x.shape[0]
    M, K, D = mG.shape[:3]
    c = torch.cat([emb[i][context[:, i]] for i in range(len(emb))], 1).reshape(B, 2, M, K, D)
    logw = torch.log_softmax(wG, dim=-1)
    out = []
    for b0 in range(0, B, chunk):
        cm = c[b0:b0 + chunk, 0].reshape(-1, M, K, D, 1, 1)
        cs = c[b0:b0 + chunk, 1].reshape(-1, M, K, D, 1, 1)
        mu, sig = mG.unsqueeze(0) + cm, F.softplus(sG.unsqueeze(0) + cs)
        xv = x[b0:b0 + chunk].reshape(-1, 1, 1, *x.shape[1:])
        lp = (-0.5 * ((xv - mu) / sig) ** 2 - torch.log(sig) - 0.5 * LOG_2PI).flatten(3).sum(-1)
        out.append(torch.logsumexp(lp + logw, dim=-1))
    return torch.cat(out, 0)


# ---- prior ------------------------------------------------------------------------------
def gmm_logprob(x, mG, sG, wG, chunk=64):
    """gaussian.py:138-161 (context-free): for each class-mixture m, logsumexp over K diagonal
    components of log_softmax(wG[m]) + sum_d N(x; mG, softplus(sG)).  Returns (B, M)."""
    B = x.shape[0]
    M, K = wG.shape
    mu = mG.reshape(1, M * K, -1)
    sig = F.softplus(sG).reshape(1, M * K, -1)
    cst = (-torch.log(sig) - 0.5 * LOG_2PI)
    logw = torch.log_softmax(wG, dim=-1)
    out = []
    for b0 in range(0, B, chunk):
        xv = x[b0:b0 + chunk].reshape(-1, 1, mu.shape[-1])
        lp = (-0.5 * ((xv - mu) / sig) ** 2 + cst).sum(-1).reshape(-1, M, K)
        out.append(torch.logsumexp(lp + logw, dim=-1))
    return torch.cat(out, 0)


# --------------------------------------------------------------------------------------
# whole flow  (flowsequential.py:18-30)
# --------------------------------------------------------------------------------------
def flow_forward(ops, params, x, u=None, eps=(), init_actnorm=False, trace=None, ctx=None, context=None, cnoise=()):
    """log p(x) for every class-mixture: returns (z, logp (B,M)).

    `u` is the dequantisation noise (uniform.py:31-34) and `eps` the list of Augment noises
    (gaussian.py:68-72), passed in so that runs are reproducible.  With `init_actnorm` the
    ActNorm parameters are (re)computed from this batch as on the reference's first call and
    written into `params`.  `trace`, if a list, receives (op_name, index, z, ldj) per layer.
    """
    B = x.shape[0]
    M = params["dist.wG"].shape[0]
    logdet = torch.zeros((B, M), dtype=x.dtype)
    eps = list(eps)
    cnoise = list(cnoise)          # specialist mode: one uniform noise tensor per context encoder, in layer order

    def enc(prefix=None):
        et = ctx.get("enc_type", "uniform")
        if et == "vardeq":
            return ctx_encode_vardeq(context, ctx, params, prefix + "context_net.1.", cnoise.pop(0))
        if et == "argmax":
            return ctx_encode_argmax(context, ctx, params, prefix + "context_net.1.", cnoise.pop(0))
        if et == "probsample":
            return ctx_encode_probsample(context, ctx, params, prefix + "context_net.1.", cnoise.pop(0))
        if et == "eyesample":               # 'embed' + EyeSampling: the embedding rows themselves, no noise, no density
            w = [params[prefix + "context_net.0._embeddings.%d.weight" % i] for i in range(len(ctx["contexts"]))]
            return torch.cat([w[i][context[:, i]] for i in range(len(w))], 1), torch.zeros(context.shape[0], dtype=x.dtype)
        return ctx_encode(context, ctx, cnoise.pop(0))

    def emb(prefix):
        return [params[prefix + "context_net.0._embeddings.%d.weight" % i] for i in range(len(ctx["contexts"]))]
    for op in ops:
        kind, idx = op[0], op[1]
        pre = "%d." % idx
        if kind == "dequant":
            x, ldj = x + u, torch.zeros(B, dtype=x.dtype)      # dequantize.py:14-17, log q = 0
        elif kind == "affine":
            x, ldj = affine_fwd(x, op[2], op[3])
        elif kind == "logit":
            x, ldj = logit_fwd(x)
        elif kind == "augment":
            e = eps.pop(0)
            x, ldj = torch.cat([x, e], 1), std_normal_neg_logq(e)
        elif kind == "squeeze":
            x, ldj = squeeze_fwd(x, op[2]), torch.zeros(B, dtype=x.dtype)
        elif kind == "permute":                                   # permute_axes.py:14-15
            x, ldj = x.permute(op[2]).contiguous(), torch.zeros(B, dtype=x.dtype)
        elif kind == "conv1x1" and ctx is not None:
            c, lc = enc(pre)
            x, ldj = conv1x1_ctx_fwd(x, params[pre + "NN"], params[pre + "CN.weight"], params[pre + "CN.bias"], c, lc,
                                     ctx["contextflow"])
        elif kind == "conv1x1":
            x, ldj = conv1x1_fwd(x, params[pre + "NN"])
        elif kind == "actnorm" and ctx is not None:
            c, lc = enc(pre)
            if init_actnorm and ctx["contextflow"]:                # actnorm.py:46: only the contextflow branch initialises
                t, logs = actnorm_stats(x)
                params[pre + "NN_t"], params[pre + "NN_logs"] = t, logs
                params[pre + "initialized"] = torch.tensor(1)
            x, ldj = actnorm_ctx_fwd(x, params[pre + "NN_t"], params[pre + "NN_logs"], params[pre + "CN.weight"],
                                     params[pre + "CN.bias"], c, lc, ctx["contextflow"])
        elif kind == "coupling" and ctx is not None:
            c, lc = enc(pre)
            x, ldj = coupling_ctx_fwd(x, params, pre, op[4], c, lc, ctx["contextflow"])
        elif kind == "transcoupling" and ctx is not None:
            c, lc = enc(pre)
            x, ldj = transcoupling_ctx_fwd(x, params, pre, op[2], op[3], c, lc, ctx["contextflow"])
        elif kind == "split" and ctx is not None:
            cc = x.shape[1] // 2
            ldj = gmm_ctx_logprob(x[:, cc:], params[pre + "dist.mG"], params[pre + "dist.sG"], params[pre + "dist.wG"],
                                  emb(pre + "dist."), context)
            x = x[:, :cc]
        elif kind == "actnorm":
            if init_actnorm:
                t, logs = actnorm_stats(x)
                params[pre + "NN_t"], params[pre + "NN_logs"] = t, logs
                params[pre + "initialized"] = torch.tensor(1)
            x, ldj = actnorm_fwd(x, params[pre + "NN_t"], params[pre + "NN_logs"])
        elif kind == "coupling":
            x, ldj = coupling_fwd(x, params, pre, op[4])
        elif kind == "transcoupling":
            x, ldj = transcoupling_fwd(x, params, pre, op[2], op[3])
        elif kind == "split":                                     # splitprior.py:12-15
            c = x.shape[1] // 2
            ldj = gmm_logprob(x[:, c:], params[pre + "dist.mG"], params[pre + "dist.sG"], params[pre + "dist.wG"])
            x = x[:, :c]
        else:
            raise ValueError(kind)
        logdet = logdet + (ldj if ldj.dim() == 2 else ldj.unsqueeze(-1))   # flowsequential.py:23
        if trace is not None:
            trace.append((kind, idx, x, ldj))
    if ctx is not None:
        logp = gmm_ctx_logprob(x, params["dist.mG"], params["dist.sG"], params["dist.wG"], emb("dist."), context)
    else:
        logp = gmm_logprob(x, params["dist.mG"], params["dist.sG"], params["dist.wG"])
    return x, logp + logdet


def flow_inverse_layers(ops, params, z, upto=None):
    """Per-layer `reverse` chain (flowsequential.py:32-39) over the invertible, context-free layers.
    Stops at SplitPrior/Augment boundaries being handled by the caller (the reference's own
    `SplitPrior.reverse` is broken), so `ops` must not contain them."""
    for op in reversed(ops[:upto]):
        kind, idx = op[0], op[1]
        pre = "%d." % idx
        if kind == "squeeze":
            z = squeeze_inv(z, op[2])
        elif kind == "conv1x1":
            z = conv1x1_inv(z, params[pre + "NN"])
        elif kind == "actnorm":
            z = actnorm_inv(z, params[pre + "NN_t"], params[pre + "NN_logs"])
        elif kind == "coupling":
            z = coupling_inv(z, params, pre, op[4])
        elif kind == "transcoupling":
            z = transcoupling_inv(z, params, pre, op[2], op[3])
        elif kind == "affine":
            z = affine_inv(z, op[2], op[3])
        elif kind == "logit":
            z = torch.sigmoid(z)                                  # transforms.py:14-15
        elif kind == "dequant":
            z = z.floor()                                         # dequantize.py:19-20
        else:
            raise ValueError("no inverse for %s" % kind)
    return z


def bits_per_dim(logp, dims):
    """-logsumexp_m logp / (D ln 2), D = prod of the un-augmented data size."""
    return -torch.logsumexp(logp, dim=-1) / (dims * math.log(2.0))


# ---- elementwise flow activations (activations.py:34-118, 213-245) --------------------------------------------------
def activation_fwd(kind, x, a=0.0, b=0.0):
    """(y, ldj).  kind: identity | leaky(alpha=a) | smooth_leaky(alpha=a) | smooth_tanh(alpha=a, beta=b) |
    sigmoid(temperature=a).  ldj sums log|f'| over every non-batch element (activations.py:19-22); the Sigmoid layer sums
    over the LAST dim only (activations.py:234-238)."""
    if kind == "sigmoid":
        t = torch.as_tensor(a, dtype=x.dtype)
        xs = t * x
        return torch.sigmoid(xs), (torch.log(t) - F.softplus(-xs) - F.softplus(xs)).sum(-1)
    if kind == "identity":
        y, d = x, torch.ones_like(x)
    elif kind == "leaky":
        y, d = torch.where(x < 0, a * x, x), torch.where(x < 0, torch.full_like(x, a), torch.ones_like(x))
    elif kind == "smooth_leaky":
        y = a * x + (1 - a) * torch.logsumexp(torch.stack((torch.zeros_like(x), x)), dim=0)
        d = a + (1 - a) * torch.sigmoid(x)
    elif kind == "smooth_tanh":
        y, d = torch.tanh(a * x) + b * x, b + a / torch.cosh(a * x) ** 2
    else:
        raise ValueError(kind)
    return y, torch.log(torch.abs(d)).flatten(1).sum(-1)


def activation_inv(kind, y, a=0.0, b=0.0, eps=0.0):
    """Inverse: closed form for the piecewise-linear ones and the sigmoid, the reference's Newton iteration
    (activations.py:25-31: 100 steps from x0 = y, derivative clamped at 1e-2) for the smooth ones."""
    if kind == "identity":
        return y
    if kind == "leaky":
        return torch.where(y < 0, y / a, y)
    if kind == "sigmoid":
        z = torch.clamp(y, eps, 1 - eps)
        return (torch.log(z) - torch.log1p(-z)) / a
    x = y
    for _ in range(100):
        f, _ = activation_fwd(kind, x, a, b)
        if kind == "smooth_leaky":
            d = a + (1 - a) * torch.sigmoid(x)
        else:
            d = b + a / torch.cosh(a * x) ** 2
        x = x - (f - y) / torch.clamp(d, min=1e-2)
    return x
