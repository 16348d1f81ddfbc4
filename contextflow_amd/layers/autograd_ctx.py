"""Training step of SPECIALIST models under contextflow (`create_model(generalist=False, contextflow=True)`:
coupling.py:36, conv1x1.py:27, actnorm.py:23, gaussian.py:134 freeze the generalist's own parameters; the CN nets of
every Conv1x1 / ActNorm / Coupling and the priors' embedding tables are what trains).

One autograd.Function for the whole flow, layer by layer: the forward keeps each layer's input and the (stochastic)
context-encoder output; the backward runs the data-gradient chain with HIP kernels (cf_conv1x1_ctx_bwd,
cf_actnorm_ctx_bwd, the fused MFMA step-backward kernel with the per-sample CN(c) bias for the Coupling layers,
cf_gmm_ctx_bwd) and turns the per-sample parameter gradients into CN / embedding gradients with library GEMMs.
Built for the conv couplings with the encoders that have no trainable parameters of their own (eye | onehot + uniform)
and for embedding lookups (embed + eyesample)."""
import os

import torch

from . import _hip
from .actnorm import ActNorm
from .context import (ArgmaxCatDequantization, CatEmbeddings, EyeSampling, ProbSampling, UniformCatDequantization,
                      VariationalCatDequantization)
from .conv1x1 import Conv1x1
from .coupling import Coupling, TransCoupling
from .permute_axes import PermuteAxes
from .splitprior import SplitPrior
from .squeeze import Squeeze, squeeze_op


def _new(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


def trainable(flow):
    """True when this specialist flow can be trained here: every context encoder create_model builds, conv and
    transformer couplings, with contextflow (frozen generalist) or without (every parameter trains)."""
    ok = (UniformCatDequantization, EyeSampling, VariationalCatDequantization, ArgmaxCatDequantization, ProbSampling)
    for m in list(flow.sequence_modules) + [flow.dist]:
        cn = getattr(m, "context_net", None) or getattr(getattr(m, "dist", None), "context_net", None)
        if not cn:
            continue
        if not isinstance(cn[1], ok):
            return False
    return True


def _check_encoder(enc):
    if not isinstance(enc[1], (UniformCatDequantization, EyeSampling, VariationalCatDequantization, ArgmaxCatDequantization,
                               ProbSampling)):
        raise NotImplementedError("specialist training with a %s context encoder" % type(enc[1]).__name__)


def _encoder_backward(enc, context, gc, grads, glq=None, eps=None):
    """d/d of the encoder's outputs: gc = d/dc (B, width), glq = d/d logp_c (B).  uniform / eyesample encoders have no
    parameters of their own (only the embedding lookup in front has); vardeq / argmax / probsample draw their noise from a
    small conditional flow whose parameters train (model.py:52-79)."""
    if isinstance(enc[1], (UniformCatDequantization, EyeSampling)):
        if isinstance(enc[0], CatEmbeddings):
            _embedding_grads(enc[0], context, gc, grads)
        return
    _flow_encoder_backward(enc[1], context, gc, glq, grads, eps)


def _dense_bwd(x_in, W2d, gy, need_gx=True):
    """y = x W^T (+ b): (gx, gW, gb) through cf_linear / cf_linear_wgrad."""
    N, K = W2d.shape
    rows = x_in.shape[0]
    gy, x_in, st = gy.contiguous(), x_in.contiguous(), _hip.stream()
    gW, gb = _new(N, K, like=gy), _new(N, like=gy)
    ws = torch.empty(_hip.lib().cf_linear_wgrad_ws_bytes(rows, K, N), device=gy.device, dtype=torch.uint8)
    _hip.call("cf_linear_wgrad", _hip.p(x_in), _hip.p(gy), _hip.p(gW), _hip.p(gb), _hip.p(ws), rows, K, N, st)
    gx = None
    if need_gx:
        gx = _new(rows, K, like=gy)
        Wt = W2d.t().contiguous()
        _hip.call("cf_linear", _hip.p(gy), _hip.p(Wt), None, None, _hip.p(gx), rows, N, K, 0, st)
    return gx, gW, gb


def _dense(x_in, W2d, b, act):
    x_in = x_in.contiguous()
    y = _new(x_in.shape[0], W2d.shape[0], like=x_in)
    _hip.call("cf_linear", _hip.p(x_in), _hip.p(W2d), _hip.p(b), None, _hip.p(y), x_in.shape[0], W2d.shape[1],
              W2d.shape[0], act, _hip.stream())
    return y


def _couplingfc_backward(m, x_in, gz, gld, grads):
    """CouplingFC (coupling.py:76-98: the affine coupling on (B, n) with 1x1 "convolutions" = Linear layers)."""
    B, n = x_in.shape
    half, st = n // 2, _hip.stream()
    convs = (m.NN[0], m.NN[2], m.NN[4])
    Ws = [_hip.f32(c.weight.detach()).flatten(1).contiguous() for c in convs]
    bs = [_hip.f32(c.bias.detach()) for c in convs]
    x0 = x_in[:, :half].contiguous()
    a1 = _dense(x0, Ws[0], bs[0], 2)
    a2 = _dense(a1, Ws[1], bs[1], 2)
    h = _dense(a2, Ws[2], bs[2], 0)
    gx, gh = _new(B, n, like=x_in), _new(B, n, like=x_in)
    xc, gzc, gldc = x_in.contiguous(), _hip.f32(gz).contiguous(), _hip.f32(gld).contiguous()
    _hip.call("cf_coupling_apply_bwd", _hip.p(xc), _hip.p(h), _hip.p(gzc), _hip.p(gldc), _hip.p(gx), _hip.p(gh), B, n, 1, n, n,
              st)
    g2, gW3, gb3 = _dense_bwd(a2, Ws[2], gh)
    g1, gW2, gb2 = _dense_bwd(a1, Ws[1], _relu_bwd(a2, g2))
    g0, gW1, gb1 = _dense_bwd(x0, Ws[0], _relu_bwd(a1, g1))
    for c, gW, gb in zip(convs, (gW1, gW2, gW3), (gb1, gb2, gb3)):
        grads[c.weight], grads[c.bias] = gW.view_as(c.weight), gb
    gx[:, :half] += g0
    return gx


def _flow_encoder_backward(encoder, context, gc, glq, grads, eps):
    """Backward of VariationalCatDequantization / ArgmaxCatDequantization / ProbSampling (dequantize.py:104-118, 236-262,
    152-161) and of the FlowInvSequential they sample from (flowsequential.py:58-68: Gaussian draw, then the layers'
    forward, log q = log N - sum ldj).  The draw is replayed from the kept noise; the (B, n) intermediates are recomputed."""
    from .actnorm import ActNormFC
    from .autograd_layers import actnorm_backward, conv1x1_backward
    from .conv1x1 import FC
    from .coupling import CouplingFC
    flow = encoder.encoder
    dist = flow.dist
    dev = gc.device
    ctx = context.to(device=dev, dtype=torch.int64).contiguous()
    B, n, st = ctx.shape[0], dist.D, _hip.stream()
    glq = _hip.f32(glq).contiguous()
    # ---- replay: Gaussian draw, layers (inputs kept), sigmoid
    cemb = _hip.f32(dist.context_net(ctx)[0]).contiguous()   # (B, 2n) = [mean | log_scale]
    if eps is None:
        raise RuntimeError("encoder backward without the taped noise of its forward")
    eps = _hip.f32(eps).contiguous()
    u = _new(B, n, like=gc)
    lq0 = _new(B, like=gc)
    _hip.call("cf_cond_gauss_sample", _hip.p(cemb), _hip.p(eps), _hip.p(u), _hip.p(lq0), B, n, st)
    inputs = []
    with torch.no_grad():
        for mod in flow.sequence_modules:
            inputs.append(u)
            u, _ = mod(u, ctx)
    u = u.contiguous()
    su = torch.empty_like(u)
    _hip.call("cf_sigmoid_ldj", _hip.p(u), _hip.p(su), _hip.p(_new(B, like=gc)), B, n, st)     # only sigmoid(u) is needed here
    # ---- heads: d/d sigmoid(u) from d/dc, sign of log q in the encoder's log-density
    if isinstance(encoder, VariationalCatDequantization):    # z = (x + su) / K ; ldj = const + act_ldj - log q
        gsu, sq = gc / encoder.qbins, -1.0
    elif isinstance(encoder, ArgmaxCatDequantization):       # z = su * (2 bits - 1) ; ldj = act_ldj - log q
        bits = torch.tensor(encoder.num_bits, device=dev, dtype=torch.int64)
        ones = torch.ones(B, n, device=dev, dtype=torch.float32)
        sign = _new(B, n, like=gc)
        _hip.call("cf_ctx_encode", _hip.p(ctx), _hip.p(ones), None, _hip.p(bits), _hip.p(sign), B, ctx.shape[1], n, 2, st)
        gsu, sq = gc * sign, -1.0
    else:                                                    # ProbSampling: z = su ; ldj = act_ldj + log q (reference sign)
        gsu, sq = gc, 1.0
    gu = gsu * su * (1.0 - su) + glq.unsqueeze(1) * (1.0 - 2.0 * su)      # d act_ldj / du = 1 - 2 sigmoid(u)
    glogq = sq * glq                                         # d/d log q(u);  log q = log N(u0) - sum_layers ldj
    for mod, xin in zip(reversed(flow.sequence_modules), reversed(inputs)):
        if isinstance(mod, FC):
            g4, gp = conv1x1_backward(mod, xin.view(B, n, 1, 1), gu.contiguous().view(B, n, 1, 1), -glogq)
            gu = g4.view(B, n)
        elif isinstance(mod, ActNormFC):
            g4, gp = actnorm_backward(mod, xin.view(B, n, 1, 1), gu.contiguous().view(B, n, 1, 1), -glogq)
            gu = g4.view(B, n)
        elif isinstance(mod, CouplingFC):
            gp = {}
            gu = _couplingfc_backward(mod, xin, gu, -glogq, gp)
        else:
            raise NotImplementedError("encoder flow layer %s" % type(mod).__name__)
        grads.update(gp)
    # ---- the draw u0 = mean + exp(ls) eps, log N(u0) = sum(-1/2 log 2pi - ls - eps^2 / 2)
    ls = cemb[:, n:]
    gcemb = torch.cat([gu, gu * torch.exp(ls) * eps - glogq.unsqueeze(1)], dim=1)
    _embedding_grads(dist.context_net, ctx, gcemb, grads)


def _embedding_grads(emb, context, gc, grads):
    o = 0
    for i, m in enumerate(emb._embeddings):
        d = m.weight.shape[1]
        g = torch.zeros_like(m.weight, dtype=torch.float32)
        g.index_add_(0, context[:, i].to(g.device), gc[:, o:o + d])           # scatter by context id (index op)
        grads[m.weight] = g
        o += d


DEFER_LINEAR_WGRADS = os.environ.get("CONTEXTFLOW_SPEC_DEFER_WGRADS", "1") != "0"      # A/B switch (tools/specialist_train_bench.py)
_DEFER = "_deferred_linear_wgrads"      # key of `grads`: [(x_in, gy, lin)] whose weight gradients are formed at the end of the backward


def _linear_bwd(x_in, lin, gy, grads, need_gx=True):
    """nn.Linear of a CN net: weight / bias gradients and the data gradient through cf_linear_wgrad / cf_linear.  With a deferral
    list in `grads` (the specialist backward) only the data gradient is formed here; the weight gradients of all CN Linears of the
    flow - 60 operand pairs in the cifar10 flow, 420 launches of 7 - 10 us one by one - leave in grouped launches at the end
    (_flush_linear_wgrads: cf_linear_wgrad_group, the transformer step's LDS-free kernel)."""
    defer = grads.get(_DEFER)
    if defer is None:
        gx, gW, gb = _dense_bwd(_hip.f32(x_in), _hip.f32(lin.weight.detach()), _hip.f32(gy), need_gx)
        grads[lin.weight], grads[lin.bias] = gW, gb
        return gx
    x_in, gy = _hip.f32(x_in).contiguous(), _hip.f32(gy).contiguous()
    defer.append((x_in, gy, lin))
    if not need_gx:
        return None
    W2d = _hip.f32(lin.weight.detach())
    gx = _new(x_in.shape[0], W2d.shape[1], like=gy)
    _hip.call("cf_linear", _hip.p(gy), _hip.p(W2d.t().contiguous()), None, None, _hip.p(gx), x_in.shape[0], W2d.shape[0], W2d.shape[1], 0,
              _hip.stream())
    return gx


def _flush_linear_wgrads(grads, dev):
    from .autograd import wgrad_group
    todo = grads.pop(_DEFER, None) or []
    for i0 in range(0, len(todo), 32):                  # cf_linear_wgrad_group takes up to 32 members per launch pair
        part = todo[i0:i0 + 32]
        res = wgrad_group([(x_in, gy, True) for x_in, gy, _ in part], dev)
        for (_, _, lin), (gW, gb) in zip(part, res):
            grads[lin.weight], grads[lin.bias] = gW.view_as(lin.weight), gb


def _encoder_needs_gc(enc):
    """False when nothing behind the encoder's output trains (uniform dequantisation / pass-through of a code that is not an
    embedding lookup): the data gradient of the first CN Linear - a (B, N) x (N, width) product and a transposed copy of its
    weight per layer and step - is then never read (_encoder_backward returns at once)."""
    if isinstance(enc[1], (UniformCatDequantization, EyeSampling)):
        return isinstance(enc[0], CatEmbeddings)
    return True


def _relu_bwd(act, gy):
    out = torch.empty_like(gy)
    _hip.call("cf_relu_bwd", _hip.p(act), _hip.p(gy), _hip.p(out), gy.numel(), _hip.stream())
    return out


def conv1x1_ctx_backward(m, rec, context, gz, gld, grads):
    x, xbs = _hip.bview(rec["x"])
    gzv, gzbs = _hip.bview(gz)
    B, C, H, W = x.shape
    gx = _new(B, C, H, W, like=x)
    gm = _new(B, C * C, like=x)
    Wm = _hip.f32(m.NN.detach()) if m.contextflow else None
    _hip.call("cf_conv1x1_ctx_bwd", _hip.p(x), _hip.p(rec["m"]), _hip.p(Wm), _hip.p(gzv), _hip.p(gld), _hip.p(gx), _hip.p(gm),
              B, C, H * W, xbs, gzbs, _hip.stream())
    need = _encoder_needs_gc(m.context_net)
    gc = _linear_bwd(rec["c"], m.CN, gm, grads, need)
    if need:
        _encoder_backward(m.context_net, context, gc, grads, _hip.f32(gld) * float(H * W), rec.get("eps"))   # ldj += H W logp_c
    return gx


def actnorm_ctx_backward(m, rec, context, gz, gld, grads):
    x, xbs = _hip.bview(rec["x"])
    gzv, gzbs = _hip.bview(gz)
    B, C, H, W = x.shape
    gx = _new(B, C, H, W, like=x)
    gm = _new(B, 2 * C, like=x)
    t = _hip.f32(m.NN_t.detach()) if m.contextflow else None
    logs = _hip.f32(m.NN_logs.detach()) if m.contextflow else None
    _hip.call("cf_actnorm_ctx_bwd", _hip.p(x), _hip.p(rec["m"]), _hip.p(t), _hip.p(logs), _hip.p(gzv), _hip.p(gld), _hip.p(gx),
              _hip.p(gm), B, C, H * W, xbs, gzbs, _hip.stream())
    need = _encoder_needs_gc(m.context_net)
    gc = _linear_bwd(rec["c"], m.CN, gm, grads, need)
    if need:
        _encoder_backward(m.context_net, context, gc, grads, _hip.f32(gld) * float(H * W), rec.get("eps"))   # ldj += H W logp_c
    return gx


def _cn_chain_backward(m, rec, context, gcn, grads, glq):
    """CN = Linear -> ReLU -> Linear -> ReLU -> Linear (coupling.py:37) and the context encoder behind it; glq = d/d of the
    encoder's log-density term of the layer's log-det."""
    ga2 = _relu_bwd(rec["a2"], _linear_bwd(rec["a2"], m.CN[4], gcn, grads))
    ga1 = _relu_bwd(rec["a1"], _linear_bwd(rec["a1"], m.CN[2], ga2, grads))
    need = _encoder_needs_gc(m.context_net)
    gc = _linear_bwd(rec["c"], m.CN[0], ga1, grads, need)
    if need:
        _encoder_backward(m.context_net, context, gc, grads, glq() if callable(glq) else glq, rec.get("eps"))


def coupling_ctx_backward(m, rec, context, gz, gld, grads):
    """Coupling with a context net through the fused step-backward kernel (identity 1x1 / ActNorm in front).
    contextflow (mode 1): the conditioner is frozen, CN(c) is a bias on its output - d/d CN(c) = per-sample row sums of the
    conditioner-output gradient plane.  Without contextflow (mode 2): CN(c) enters through the extra input channels of
    the first 1x1 (a per-sample bias before its ReLU) and every parameter trains - the backward kernel loads the planes the
    forward taped, the weight gradients are the generalist's GEMMs plus two small products for the context columns."""
    x, xbs = _hip.bview(rec["x"])
    B, C, H, W = x.shape
    HW, D, HID = H * W, C // 2, 2 * C
    dev, st, f, pp, L = x.device, _hip.stream(), _hip.f32, _hip.p, _hip.lib()
    c1, c2, c3 = m.NN[0], m.NN[2], m.NN[4]
    w1 = f(c1.weight.detach())
    # transposed fragments of the backward kernel: kept while the three weights are unchanged (frozen under contextflow)
    key = (rec["mode"], C, H, W, str(dev)) + tuple((t._version, t.data_ptr()) for t in (c1.weight, c2.weight, c3.weight))
    hit = m.__dict__.get("_ctx_wsb")
    capturing = torch.cuda.is_current_stream_capturing()
    if hit is not None and hit[0] == key and not capturing:
        wsb = hit[1]
        torch.cuda.current_stream(dev).wait_event(hit[2])
    else:
        eye = torch.eye(C, device=dev, dtype=torch.float32)
        zero = torch.zeros(C, device=dev, dtype=torch.float32)
        wsb = torch.empty(L.cf_flow_step_bwd_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
        w1x = w1 if rec["mode"] == 1 else w1[:, :D].contiguous()
        _hip.call("cf_flow_step_bwd_prepare", pp(eye), pp(zero), pp(w1x), pp(f(c2.weight.detach())),
                  pp(f(c3.weight.detach())), pp(wsb), C, H, W, st)
        if not capturing:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            m.__dict__["_ctx_wsb"] = (key, wsb, ev)
    gx = torch.empty(B, C, H, W, device=dev, dtype=torch.float32)
    s_gh = torch.empty(B, C, HW, device=dev, dtype=torch.float32)
    gzc = f(gz).contiguous()
    if rec["mode"] == 1:                                                  # s_gh is the only plane this mode writes
        _hip.call("cf_flow_step_bwd_ctx", pp(x), pp(gzc), pp(f(gld)), pp(rec["ws"]), pp(wsb), pp(rec["cn"]), pp(gx), None,
                  None, None, pp(s_gh), None, None, None, B, C, H, W, xbs, st)
        gcn = _new(B, C, like=x)
        _hip.call("cf_sample_channel_sums", pp(s_gh), pp(gcn), B, C, HW, st)
        _cn_chain_backward(m, rec, context, gcn, grads, lambda: f(gld) * float(HW))      # ldj += H W logp_c (coupling.py:43)
        return gx
    new = lambda rows: torch.empty(B, rows, HW, device=dev, dtype=torch.float32)
    y0, h1, h2, aux = rec["planes"]
    s_gh2, s_gh1, s_gy = new(HID), new(HID), new(C)
    _hip.call("cf_flow_step_bwd_taped", pp(gzc), pp(f(gld)), pp(wsb), pp(aux), pp(gx), pp(s_gh), pp(s_gh2), pp(s_gh1), pp(s_gy),
              B, C, H, W, 0, st)

    def wgrad(A, Bm, taps):
        MR, NR = A.shape[1], Bm.shape[1]
        gw = torch.empty(taps, MR, NR, device=dev, dtype=torch.float32)
        gb = torch.empty(MR, device=dev, dtype=torch.float32)
        wsw = torch.empty(L.cf_wgrad_ws_bytes(B, MR, NR, H, W, taps), device=dev, dtype=torch.uint8)
        _hip.call("cf_wgrad", pp(A), pp(Bm), pp(gw), pp(gb), pp(wsw), B, MR, NR, H, W, taps, st)
        return gw, gb

    gw3, gb3 = wgrad(s_gh, h2, 1)
    gw2, gb2 = wgrad(s_gh2, h1, 9)
    gw1, gb1 = wgrad(s_gh1, y0, 1)
    # context columns of the first 1x1: its input there is CN(c), constant over the pixels
    s1 = _new(B, HID, like=x)
    _hip.call("cf_sample_channel_sums", pp(s_gh1), pp(s1), B, HID, HW, st)
    wc = w1[:, D:, 0, 0]                                                  # (HID, O)
    _, gwc, _ = _dense_bwd(_hip.f32(rec["cn"]), wc.contiguous(), s1, need_gx=False)      # (HID, O) = s1^T CN(c)
    grads[c1.weight] = torch.cat([gw1[0], gwc], dim=1).reshape(c1.weight.shape)
    grads[c1.bias] = gb1
    grads[c2.weight], grads[c2.bias] = gw2.permute(1, 2, 0).reshape(c2.weight.shape), gb2
    grads[c3.weight], grads[c3.bias] = gw3[0].reshape(c3.weight.shape), gb3
    gcn = _new(B, wc.shape[1], like=x)
    _hip.call("cf_linear", pp(s1), pp(wc.t().contiguous()), None, None, pp(gcn), B, HID, wc.shape[1], 0, st)   # s1 wc
    _cn_chain_backward(m, rec, context, gcn, grads, lambda: f(gld) * float(HW))
    return gx


def transcoupling_ctx_backward(m, rec, context, gz, gld, grads):
    """TransCoupling with a context net (coupling.py:123-133).  contextflow: h = ViT(x0) + CN(c) with the ViT frozen -
    the ViT is re-run with a tape (as in the generalist backward), d/d CN(c) = per-sample row sums of d/dh, d/dx0 through
    the ViT (data gradient only).  Without contextflow: h = ViT([x0 ; CN(c) broadcast over the window]) and every
    parameter trains - d/d CN(c) = the row sums of the ViT's input gradient over its context channels."""
    from .autograd_layers import vit_backward, vit_forward_taped
    x, xbs = _hip.bview(rec["x"])
    gzv, gzbs = _hip.bview(gz)
    B, C, H, W = x.shape
    half, st = C // 2, _hip.stream()
    if m.contextflow:
        h, vtape = vit_forward_taped(m.NN[0], x[:, :half])
        _hip.call("cf_add_sample_bias", _hip.p(h), _hip.p(rec["cn"]), B, C, H * W, 0, st)
    else:
        cn = rec["cn"]
        xin = torch.cat([_hip.f32(x[:, :half]), cn.view(B, -1, 1, 1).expand(B, cn.shape[1], H, W)], dim=1)   # index op
        h, vtape = vit_forward_taped(m.NN, xin)
    gx = _new(B, C, H, W, like=x)
    ghd = _new(B, C, H, W, like=x)
    _hip.call("cf_coupling_apply_bwd", _hip.p(x), _hip.p(h), _hip.p(gzv), _hip.p(_hip.f32(gld)), _hip.p(gx), _hip.p(ghd),
              B, C, H * W, xbs, gzbs, st)
    if m.contextflow:
        gcn = _new(B, C, like=x)
        _hip.call("cf_sample_channel_sums", _hip.p(ghd), _hip.p(gcn), B, C, H * W, st)
        gx[:, :half] += vit_backward(m.NN[0], vtape, ghd, None)       # grads = None: the ViT is frozen, data gradient only
    else:
        gxin = vit_backward(m.NN, vtape, ghd, grads)
        gx[:, :half] += gxin[:, :half]
        gctx = gxin[:, half:].contiguous()
        gcn = _new(B, gctx.shape[1], like=x)
        _hip.call("cf_sample_channel_sums", _hip.p(gctx), _hip.p(gcn), B, gctx.shape[1], H * W, st)
    _cn_chain_backward(m, rec, context, gcn, grads, _hip.f32(gld))          # quirk: no H W factor here (coupling.py:126)
    return gx


def gmm_ctx_backward(dist, rec, g, grads):
    """Context-shifted GMM: d/dx and the embedding-table gradients (mG / sG / wG are frozen under contextflow)."""
    x, xbs = _hip.bview(rec["x"])
    B, D, H, W = x.shape
    M, K = dist.M, dist.K
    gx = _new(B, D, H, W, like=x)
    gc = _new(B, 2 * M * K * D, like=x)
    tab = rec.get("tab")
    if tab is not None:                       # scale shifts by table lookup (embedding context net)
        key, inv, dsig, lsum = tab
        _hip.call("cf_gmm_ctx_bwd_tab", _hip.p(x), _hip.p(_hip.f32(dist.mG.detach())), _hip.p(inv), _hip.p(dsig), _hip.p(lsum),
                  _hip.p(rec["logw"]), _hip.p(rec["c"]), _hip.p(key), _hip.p(_hip.f32(g)), _hip.p(rec.get("lp")), _hip.p(gx),
                  _hip.p(gc), B, M, K, D, H * W, xbs, _hip.stream())
    else:
        _hip.call("cf_gmm_ctx_bwd", _hip.p(x), _hip.p(_hip.f32(dist.mG.detach())), _hip.p(_hip.f32(dist.sG.detach())),
                  _hip.p(rec["logw"]), _hip.p(rec["c"]), _hip.p(_hip.f32(g)), _hip.p(rec.get("lp")), _hip.p(gx), _hip.p(gc), B, M,
                  K, D, H * W, xbs, _hip.stream())
    _embedding_grads(dist.context_net[0], rec["context"], gc, grads)
    if not dist.contextflow:                  # the prior's own parameters train too (gaussian.py:130-137)
        if tab is None or rec.get("lp") is None:
            raise NotImplementedError("prior parameter gradients need the embedding-lookup context net (table form)")
        gf = _hip.f32(g)
        r = (torch.softmax(rec["lp"].view(B, M, K), dim=-1) * gf.unsqueeze(-1)).reshape(B, M * K).contiguous()
        slab = 256
        nb = (B + slab - 1) // slab
        pgm = _new(nb, M * K, D * H * W, like=x)
        pgs = _new(nb, M * K, D * H * W, like=x)
        _hip.call("cf_gmm_ctx_pgrad_tab", _hip.p(x), _hip.p(_hip.f32(dist.mG.detach())), _hip.p(inv), _hip.p(dsig),
                  _hip.p(rec["c"]), _hip.p(key), _hip.p(r), _hip.p(pgm), _hip.p(pgs), B, M, K, D, H * W, xbs, slab, _hip.stream())
        grads[dist.mG] = pgm.sum(0).view_as(dist.mG)
        grads[dist.sG] = pgs.sum(0).view_as(dist.sG)
        R = r.sum(0).view(M, K)
        grads[dist.wG] = R - R.sum(-1, keepdim=True) * torch.softmax(_hip.f32(dist.wG.detach()), dim=-1)
    return gx


class SpecialistLogProb(torch.autograd.Function):
    """logp (B, M) of a specialist FlowSequential with gradients for its trainable (context) parameters."""

    @staticmethod
    def forward(ctx, flow, x, context, *params):
        B, M = x.shape[0], flow.mixtures
        tape = []
        logdet = torch.zeros(B, M, device=x.device, dtype=torch.float32)
        # everything the CN nets compute from the context alone, for all layers at once (the uniform encoders' codes are formed
        # inside the first grouped launch and kept for the backward): 36 encodings + 60 Linears one by one otherwise
        from . import specialist
        specialist._draw_encoder_noise(flow, B, x.device)
        pre = specialist._front_end(flow, context, B, x.device, train=True)
        for mod in flow.sequence_modules:
            rec = []
            if isinstance(mod, Conv1x1) and mod.context_net:
                _check_encoder(mod.context_net)
                x, ldj = mod._forward_ctx(x, context, rec, pre.get(id(mod)))
            elif isinstance(mod, ActNorm) and mod.context_net:
                _check_encoder(mod.context_net)
                x, ldj = mod._forward_ctx(x, context, rec, pre.get(id(mod)))
            elif type(mod) is Coupling and mod.context_net:
                _check_encoder(mod.context_net)
                if not mod._fused_ctx_ok(x):
                    raise NotImplementedError("specialist training needs the fused coupling geometry (3x3, C in 8..64)")
                x, ldj = mod._fused_ctx(x, context, rec, pre.get(id(mod)))
            elif isinstance(mod, TransCoupling) and mod.context_net:
                _check_encoder(mod.context_net)
                x, ldj = mod._forward_ctx(x, context, rec)
            elif isinstance(mod, SplitPrior) and getattr(mod.dist, "context_net", None):
                c = x.shape[1] // 2
                ldj = mod.dist._log_prob_ctx(x[:, c:], context, rec)
                rec[0]["full"] = x
                x = x[:, :c]
            else:
                if any(p.requires_grad for p in mod.parameters()):
                    raise NotImplementedError("specialist training: %s has trainable parameters" % type(mod).__name__)
                x, ldj = mod(x, context)
            tape.append((mod, rec[0] if rec else None))
            logdet += ldj if ldj.dim() == 2 else ldj.unsqueeze(-1)
        rec = []
        logp = flow.dist._log_prob_ctx(x, context, rec) + logdet
        ctx.flow, ctx.tape, ctx.prior, ctx.params, ctx.context = flow, tape, rec[0], params, context
        ctx.mark_non_differentiable(x)
        return x, logp

    @staticmethod
    def backward(ctx, _gz_unused, glogp):
        flow, tape, params, context = ctx.flow, ctx.tape, ctx.params, ctx.context
        glogp = _hip.f32(glogp)
        gld = glogp.sum(1).contiguous()
        grads = {_DEFER: []} if DEFER_LINEAR_WGRADS else {}
        gz = gmm_ctx_backward(flow.dist, ctx.prior, glogp, grads)
        for mod, rec in reversed(tape):
            if rec is None:
                if isinstance(mod, Squeeze):
                    gz = squeeze_op(gz, mod.p, True)
                    continue
                if isinstance(mod, PermuteAxes):
                    gz = gz.permute(mod.inverse_permutation).contiguous()
                    continue
                break                                    # pre-processing: nothing trainable upstream
            if isinstance(mod, SplitPrior):
                g2 = gmm_ctx_backward(mod.dist, rec, glogp, grads)
                gz = torch.cat([gz, g2], dim=1)
            elif isinstance(mod, Conv1x1):
                gz = conv1x1_ctx_backward(mod, rec, context, gz, gld, grads)
            elif isinstance(mod, ActNorm):
                gz = actnorm_ctx_backward(mod, rec, context, gz, gld, grads)
            elif isinstance(mod, TransCoupling):
                gz = transcoupling_ctx_backward(mod, rec, context, gz, gld, grads)
            else:
                gz = coupling_ctx_backward(mod, rec, context, gz, gld, grads)
        _flush_linear_wgrads(grads, glogp.device)
        return (None, None, None) + tuple(grads.get(p) for p in params)
