"""Evaluation plan of the specialist (context-conditioned) conv flows: `create_model(generalist=False)`, model.py:117-162.

Layer by layer, a specialist step costs three context encoders, five CN Linears, the per-sample Conv1x1, the per-sample
ActNorm, the coupling step and a dozen tiny bookkeeping kernels (`profiles/r3_spec_fwd_b32768_kstats.txt`).  Here the
layer list is grouped the way `FlowSequential._build_plan` groups the generalist's:

  pre-processing          Dequantization -> Normalization x2 -> LogitTransform [-> Augment]: cf_preprocess_rng_fwd
  [Squeeze ->] Conv1x1(c) -> ActNorm(c')    one pass over the sample: cf_affine_ctx_fwd (Squeeze folded into its reads)
  Coupling(c'')           the fused step kernel with the CN(c'') bias, its packed weights kept while the parameters are
                          unchanged, the log-det accumulated in place
  everything else         the layer's own forward

and the per-sample log-dets accumulate in ONE (B,) buffer inside the kernels instead of a (B, M) add per layer.
Evaluation only (no tape); `FlowSequential.forward` uses it under no_grad and falls back to the layer loop when a
model does not have this shape."""
import math
import os

import torch

from . import _hip
from .actnorm import ActNorm
from .conv1x1 import Conv1x1, slogdet_inverse
from .coupling import Coupling
from .dequantize import Dequantization
from .distributions.gaussian import StandardNormal
from .distributions.uniform import UniformDistribution
from .augment import Augment
from .normalize import Normalization
from .squeeze import Squeeze
from .transforms import LogitTransform


def _const_logp(net):
    """Host constant of an encoder's log-density, or None when it depends on the sample."""
    return getattr(net, "const_logp", None)


class _Acc:
    """Running log-dets: ld1 (B,) per-sample terms, ldM (B, M) per-mixture terms; kernels accumulate into ld1 in place."""

    def __init__(self, B, dev):
        self.B, self.dev, self.ld1, self.ldM, self.cadd = B, dev, None, None, 0.0

    def add(self, ldj):
        if ldj.dim() == 2:
            self.ldM = ldj if self.ldM is None else self.ldM + ldj
            return
        if ldj.dim() == 0:
            ldj = ldj.reshape(1)
        if self.ld1 is None:         # the first term becomes the buffer the kernels accumulate into: never a view / a broadcast
            self.ld1 = ldj.expand(self.B).clone() if (ldj.shape[0] != self.B or ldj.stride(0) == 0 or ldj._base is not None) else ldj
        else:
            self.ld1 += ldj

    def buffer(self):
        """(ld1 buffer, accumulate flag) for a kernel that assigns or accumulates its per-sample log-det."""
        if self.ld1 is None:
            self.ld1 = torch.empty(self.B, device=self.dev, dtype=torch.float32)
            return self.ld1, 0
        return self.ld1, 1

    def zeroed(self):
        if self.ld1 is None:
            self.ld1 = torch.zeros(self.B, device=self.dev, dtype=torch.float32)
        return self.ld1


def _affine_ok(conv, act, shape):
    return (isinstance(conv, Conv1x1) and isinstance(act, ActNorm) and bool(conv.context_net) and bool(act.context_net)
            and len(shape) == 3 and conv.D == shape[0] and act.D == shape[0] and shape[0] <= 128
            and (not act.contextflow or act.is_initialized()))


def _coupling_ws(flow, cpl, C, H, W, dev):
    """Packed weights of the coupling step (identity 1x1 / ActNorm in front), kept while the conditioner is unchanged."""
    convs = (cpl.NN[0], cpl.NN[2], cpl.NN[4])
    srcs = tuple(p for c in convs for p in (c.weight, c.bias))
    ver = tuple(t._version for t in srcs) + tuple(t.data_ptr() for t in srcs) + (C, H, W, str(dev))
    cache = flow.__dict__.setdefault("_spec_ws", {})
    hit = cache.get(id(cpl))
    if hit is not None and hit[0] == ver:
        return hit[1]
    f, pp = _hip.f32, _hip.p
    eye = torch.eye(C, device=dev, dtype=torch.float32)
    zero = torch.zeros(C, device=dev, dtype=torch.float32)
    ws = torch.empty(_hip.lib().cf_flow_step_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
    _hip.call("cf_flow_step_prepare", pp(eye), pp(zero), pp(zero), pp(f(convs[0].weight.detach())), pp(f(convs[0].bias.detach())),
              pp(f(convs[1].weight.detach())), pp(f(convs[1].bias.detach())), pp(f(convs[2].weight.detach())),
              pp(f(convs[2].bias.detach())), pp(ws), C, H, W, _hip.stream())
    cache[id(cpl)] = (ver, ws)
    return ws


def _lad(flow, conv, dev):
    """log|det NN| of a frozen Conv1x1 (device scalar), kept while NN is unchanged."""
    cache = flow.__dict__.setdefault("_spec_lad", {})
    ver = (conv.NN._version, conv.NN.data_ptr(), str(dev))
    hit = cache.get(id(conv))
    if hit is not None and hit[0] == ver:
        return hit[1]
    lad, _ = slogdet_inverse(_hip.f32(conv.NN.detach()), False)
    cache[id(conv)] = (ver, lad)
    return lad


def blocked_rows(C):
    """Row order of Conv1x1.CN for the blocked form of the per-sample matrix (cf_affine_ctx_fwd, m1_blocked): the 16 x 16
    blocks (rt, g <= rt) on and below the diagonal, row-major inside - entry (16 rt + n, 16 g + c) of the (C, C) matrix."""
    idx = []
    for rt in range(C // 16):
        for g in range(rt + 1):
            for n in range(16):
                idx.extend((16 * rt + n) * C + 16 * g + c for c in range(16))
    return idx


def _cn_blocked(flow, conv, C, dev):
    """Conv1x1.CN with its rows in blocked order (weight (Nc, width), bias (Nc)), kept while the CN parameters are unchanged:
    the blocks above the diagonal of the per-sample matrix are never computed, written or read."""
    cache = flow.__dict__.setdefault("_spec_cnb", {})
    w, b = conv.CN.weight, conv.CN.bias
    ver = (w._version, b._version, w.data_ptr(), b.data_ptr(), str(dev))
    hit = cache.get(id(conv))
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    idx = torch.tensor(blocked_rows(C), device=dev, dtype=torch.long)
    wp, bp = _hip.f32(w.detach())[idx].contiguous(), _hip.f32(b.detach())[idx].contiguous()
    cache[id(conv)] = (ver, wp, bp)
    return wp, bp


def supported(flow):
    """True when the plan below has something to fuse: a specialist model with at least one Conv1x1(c) -> ActNorm(c') pair."""
    mods = flow.sequence_modules
    return any(isinstance(m, Conv1x1) and m.context_net and i + 1 < len(mods) and isinstance(mods[i + 1], ActNorm)
               and mods[i + 1].context_net for i, m in enumerate(mods))


def _draw_encoder_noise(flow, B, dev):
    """One torch.rand for the dequantisation noise of ALL uniform context encoders of the flow (three per step: 36 launches
    of a few microseconds each in the cifar10 flow); every encoder picks up its (B, width) slice at its next call.  The
    stream of torch's generator is consumed in module order, (n, B, width) at once instead of n times (B, width)."""
    from .context import ContextEncoder, UniformCatDequantization
    encs = flow.__dict__.get("_spec_encs")
    if encs is None:
        encs = flow.__dict__["_spec_encs"] = [m[1] for m in flow.modules()
                                              if isinstance(m, ContextEncoder) and isinstance(m[1], UniformCatDequantization)]
    live = [e for e in encs if e.fixed_noise is None]
    if len(live) < 2:
        return
    width = max(e.D for e in live)
    u = torch.rand(len(live), B, width, device=dev, dtype=torch.float32)
    for i, e in enumerate(live):
        e.__dict__["_noise_once"] = u[i] if e.D == width else u[i, :, :e.D].contiguous()


FRONT_END = os.environ.get("CONTEXTFLOW_SPEC_FRONT", "1") != "0"      # A/B switch of _front_end (tools/specialist_bench.py)


def _uniform_encoder(net):
    """(encoder, card, onehot) when `net` is a ContextEncoder whose code is the uniform dequantisation of the one-hot code / of the
    context itself (the form cf_linear_group can build while it stages its input), else None."""
    from .context import ContextEncoder, EyeEncoder, OneHotEncoder, UniformCatDequantization
    if not isinstance(net, ContextEncoder) or not isinstance(net[1], UniformCatDequantization):
        return None
    if isinstance(net[0], OneHotEncoder):
        return net[1], net[0].cardinalities, 1
    if isinstance(net[0], EyeEncoder):
        return net[1], None, 0
    return None


def _front_end(flow, context, B, dev, train=False):
    """Everything a specialist flow computes from the CONTEXT ALONE, ahead of the data path and in three launches instead of ~96:
    the code of every uniform context encoder is formed inside the first Linear that consumes it (Conv1x1.CN, ActNorm.CN,
    Coupling.CN[0]: one grouped launch, cf_linear_group with ctx), then the second and third Linears of the coupling CN nets
    (one grouped launch each).  Returns {id(layer): tensor}: Conv1x1 -> (m1, blocked), ActNorm -> m2, Coupling -> CN(c).
    train (layers/autograd_ctx.py): the training forward's form - every intermediate the backward needs is kept and returned as
    {id(layer): dict(c=code, logp=log-density of the code, m=CN(c) | a1=.., a2=.., cn=..)}, the per-sample matrix in full (C, C) form,
    couplings with and without contextflow."""
    import ctypes
    mods, n = flow.sequence_modules, len(flow.sequence_modules)
    if not torch.is_tensor(context) or context.dim() != 2 or not FRONT_END:
        return {}
    f = _hip.f32
    first, second, third, out = [], [], [], {}
    enc0 = None
    for i, m in enumerate(mods):
        net = getattr(m, "context_net", None)
        ue = _uniform_encoder(net) if net else None
        if ue is None:
            continue
        enc, card, onehot = ue
        key = (enc.D, onehot, tuple(net.contexts), context.shape[1])   # the code layout: width, kind, cardinalities of the variables
        if enc0 is None:
            enc0 = (key, enc, card, onehot)
        if isinstance(m, (Conv1x1, ActNorm)):
            first_lin = m.CN
        elif type(m) is Coupling and (m.contextflow or train):
            first_lin = m.CN[0]
        else:
            continue
        if (key != enc0[0] or enc.qbins.device != dev or first_lin.weight.device != dev or context.shape[1] != len(net.contexts)
                or first_lin.weight.shape[1] != enc.D or (not onehot and enc.D != context.shape[1])):
            continue                                       # another code layout / device, or a context of another width: the layer's own path
        u = enc.fixed_noise
        if u is not None and tuple(u.shape) != (B, enc.D):
            continue                                       # (the layer's own path reports it)
        if u is None:
            u = enc.__dict__.pop("_noise_once", None)
            if u is None or u.shape != (B, enc.D):
                u = torch.rand((B, enc.D), device=dev, dtype=torch.float32)
        u = f(u.to(dev))                                   # raw pointers travel in host arrays below: everything on `dev`, fp32, dense
        cbuf, lc = None, None
        if train:
            cbuf = torch.empty(B, enc.D, device=dev, dtype=torch.float32)
            lc = getattr(enc, "_lc", None)
            if lc is None or lc.device != dev:
                lc = enc._lc = (enc.ldj_per_dim * enc.D).sum(-1).reshape(1)             # dequantize.py:62, as UniformCatDequantization.encode
            lc = lc.expand(B)
        if isinstance(m, Conv1x1):
            C = m.D
            nblk = 0 if train else _hip.lib().cf_affine_ctx_blocked_floats(C, m.H, m.W)
            if nblk and nblk < C * C:
                w, b = _cn_blocked(flow, m, C, dev)
                blocked = 1
            else:
                w, b, blocked = f(m.CN.weight.detach()), f(m.CN.bias.detach()), 0
            y = torch.empty(B, w.shape[0], device=dev, dtype=torch.float32)
            first.append((u, enc.qbins, w, b, y, 0, cbuf))
            out[id(m)] = dict(c=cbuf, logp=lc, m=y) if train else (y, blocked)
        elif isinstance(m, ActNorm):
            w, b = f(m.CN.weight.detach()), f(m.CN.bias.detach())
            y = torch.empty(B, w.shape[0], device=dev, dtype=torch.float32)
            first.append((u, enc.qbins, w, b, y, 0, cbuf))
            out[id(m)] = dict(c=cbuf, logp=lc, m=y) if train else y
        elif type(m) is Coupling and (m.contextflow or train):
            l0, l1, l2 = m.CN[0], m.CN[2], m.CN[4]
            a1 = torch.empty(B, l0.weight.shape[0], device=dev, dtype=torch.float32)
            a2 = torch.empty(B, l1.weight.shape[0], device=dev, dtype=torch.float32)
            cn = torch.empty(B, l2.weight.shape[0], device=dev, dtype=torch.float32)
            first.append((u, enc.qbins, f(l0.weight.detach()), f(l0.bias.detach()), a1, 2, cbuf))
            second.append((a1, None, f(l1.weight.detach()), f(l1.bias.detach()), a2, 2, None))
            third.append((a2, None, f(l2.weight.detach()), f(l2.bias.detach()), cn, 0, None))
            out[id(m)] = dict(c=cbuf, logp=lc, a1=a1, a2=a2, cn=cn) if train else cn
    if not first:
        return {}
    _, enc, card, onehot = enc0
    ctx = context.to(device=dev, dtype=torch.int64).contiguous()
    st = _hip.stream()

    def launch(probs, K, with_ctx):
        # problems of one launch share K; the coupling nets of different levels have different hidden widths: one launch per K
        byk = {}
        for pr in probs:
            byk.setdefault(pr[2].shape[1], []).append(pr)
        for k, ps in byk.items():
            nn_ = len(ps)
            arr = lambda j: (ctypes.c_void_p * nn_)(*[(pr[j].data_ptr() if pr[j] is not None else None) for pr in ps])
            Ns = (ctypes.c_int * nn_)(*[pr[2].shape[0] for pr in ps])
            acts = (ctypes.c_int * nn_)(*[pr[5] for pr in ps])
            _hip.call("cf_linear_group", nn_, arr(0), arr(1), arr(2), arr(3), arr(4), arr(6) if (with_ctx and train) else None, Ns, acts,
                      _hip.p(ctx) if with_ctx else None, _hip.p(card) if (with_ctx and card is not None) else None,
                      ctx.shape[1] if with_ctx else 0, onehot if with_ctx else 0, B, k, st)
    launch(first, enc.D, True)
    launch(second, 0, False)
    launch(third, 0, False)
    return out


def forward_eval(flow, x, context):
    """(z, logp (B, M)) of a specialist flow, evaluation.  Same results as FlowSequential._forward_layers to fp32 rounding
    (the per-sample log-dets are summed in another order)."""
    from .simple_vit import _linear
    mods, n = flow.sequence_modules, len(flow.sequence_modules)
    B, dev = x.shape[0], x.device
    _draw_encoder_noise(flow, B, dev)
    pre = _front_end(flow, context, B, dev)
    acc = _Acc(B, dev)
    st, pp, f = _hip.stream(), _hip.p, _hip.f32
    i = 0
    while i < n:
        m = mods[i]
        shape = tuple(x.shape[1:])
        # ---- pre-processing in one kernel, noise drawn inside it (as the generalist's plan does)
        if (isinstance(m, Dequantization) and isinstance(m.dist, UniformDistribution) and i + 3 < n and len(shape) == 3
                and isinstance(mods[i + 1], Normalization) and isinstance(mods[i + 2], Normalization)
                and isinstance(mods[i + 3], LogitTransform) and m.dist.fixed_noise is None):
            aug = mods[i + 4] if (i + 4 < n and isinstance(mods[i + 4], Augment) and isinstance(mods[i + 4].distribution, StandardNormal)
                                  and mods[i + 4].split_dim == 1) else None
            C, H, W = shape
            N, ca = C * H * W, (aug.aug_size if aug is not None else 0)
            if (aug is None or aug.distribution.fixed_noise is None) and N % 4 == 0 and (ca * H * W) % 4 == 0:
                n1, n2 = mods[i + 1], mods[i + 2]
                xin = f(x)
                y = torch.empty(B, C + ca, H, W, device=dev, dtype=torch.float32)
                cst = -N * math.log(n1._s) - N * math.log(n2._s)
                ldp = torch.empty(B, device=dev, dtype=torch.float32)
                nonce = flow._noise_nonce(dev)
                _hip.call("cf_preprocess_rng_fwd", pp(xin), pp(y), pp(ldp), pp(nonce), flow._rng_key, B, N, ca * H * W,
                          (C + ca) * H * W, n1._t, n1._s, n2._t, n2._s, cst, 0, st)
                acc.add(ldp)
                x = y
                i += 5 if aug is not None else 4
                continue
        # ---- [Squeeze ->] Conv1x1(c) -> ActNorm(c')
        sq = (isinstance(m, Squeeze) and tuple(m.p) == (2, 2) and len(shape) == 3 and shape[1] % 2 == 0 and shape[2] % 2 == 0)
        j = i + 1 if sq else i
        sshape = (shape[0] * 4, shape[1] // 2, shape[2] // 2) if sq else shape
        if j + 1 < n and _affine_ok(mods[j], mods[j + 1], sshape):
            conv, act = mods[j], mods[j + 1]
            C, H, W = sshape
            xv, xbs = _hip.bview(x)
            aligned = xbs % 4 == 0 and xv.data_ptr() % 16 == 0
            lp1 = lp2 = None
            if id(conv) in pre and (aligned or not pre[id(conv)][1]):      # formed by the front end (grouped launch)
                m1, blocked = pre[id(conv)]
            else:
                c1, lp1 = conv.context_net(context)
                nblk = _hip.lib().cf_affine_ctx_blocked_floats(C, H, W) if aligned else 0
                if nblk and nblk < C * C:                          # (B, 10/16 C*C) at C = 64, 3/4 at C = 32: lower blocks only
                    wp, bp = _cn_blocked(flow, conv, C, dev)
                    c1f = f(c1)
                    m1 = torch.empty(B, nblk, device=dev, dtype=torch.float32)
                    _hip.call("cf_linear", pp(c1f), pp(wp), pp(bp), None, pp(m1), B, c1f.shape[1], nblk, 0, st)
                    blocked = 1
                else:
                    m1 = _linear(f(c1), conv.CN)                   # (B, C*C)
                    blocked = 0
            if id(act) in pre:
                m2 = pre[id(act)]
            else:
                c2, lp2 = act.context_net(context)
                m2 = _linear(f(c2), act.CN)                        # (B, 2C)
            cadd = 0.0
            for net, lp in ((conv.context_net, lp1), (act.context_net, lp2)):
                k = _const_logp(net)
                if k is not None:
                    cadd += k * float(H * W)
                else:
                    acc.add(lp * float(H * W))
            Wm = f(conv.NN.detach()) if conv.contextflow else None
            t = f(act.NN_t.detach()) if act.contextflow else None
            logs = f(act.NN_logs.detach()) if act.contextflow else None
            lad = _lad(flow, conv, dev) if conv.contextflow else None
            z = torch.empty(B, C, H, W, device=dev, dtype=torch.float32)
            buf, accum = acc.buffer()
            _hip.call("cf_affine_ctx_fwd", pp(xv), pp(m1), pp(Wm), pp(m2), pp(t), pp(logs), pp(lad), cadd, pp(z), pp(buf), B, C, H, W,
                      xbs, int(sq), accum, blocked, st)
            x = z
            i = j + 2
            continue
        # ---- Coupling(c''): the fused step kernel, CN bias on the conditioner output (contextflow)
        if (type(m) is Coupling and m.context_net and m.contextflow and len(shape) == 3 and m._fused_ctx_ok(x)):
            C, H, W = shape
            lp = None
            if id(m) in pre:
                cn = pre[id(m)]
            else:
                c, lp = m.context_net(context)
                a1 = _linear(f(c), m.CN[0], act=2)
                a2 = _linear(a1, m.CN[2], act=2)
                cn = _linear(a2, m.CN[4])
            xv, xbs = _hip.bview(x)
            ws = _coupling_ws(flow, m, C, H, W, dev)
            z = torch.empty(B, C, H, W, device=dev, dtype=torch.float32)
            _hip.call("cf_flow_step_fwd_ctx", pp(xv), pp(z), pp(acc.zeroed()), pp(ws), pp(cn), 1, B, C, H, W, xbs, st)
            k = _const_logp(m.context_net)
            if k is not None:
                acc.cadd += k * float(H * W)
            else:
                acc.add(lp * float(H * W))
            x = z
            i += 1
            continue
        x, ldj = m(x, context)
        acc.add(ldj)
        i += 1
    logp = flow.dist.log_prob(x, context)
    if acc.ldM is not None:
        logp = logp + acc.ldM
    if acc.ld1 is not None:
        logp = logp + (acc.ld1 + acc.cadd if acc.cadd else acc.ld1).unsqueeze(-1)
    elif acc.cadd:
        logp = logp + acc.cadd
    return x, logp
