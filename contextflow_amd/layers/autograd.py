"""Backward of the fused density path: d log p(x)(B,M) / d parameters, for the training step that follows
`model.log_prob` in the reference (contextflow/experiment_cl.py:127-136: loss(logp).backward(); optimizer.step()).

Forward = the fused plan of FlowSequential (same kernels, same numbers) with a tape of each group's input.
Backward walks the tape in reverse:
  * flow step  : ONE kernel (cf_flow_step_bwd_taped) reads the lean tape of the taping forward kernel and runs the
                 data-gradient chain on the fp32 matrix cores; it also writes the operand planes of the weight gradients,
                 which are split-K MFMA GEMMs over (batch x pixels) with the 3x3 tap shifts (cf_wgrad).  Without a kept
                 tape (TAPE_PLANES = False / huge batches) the taping forward kernel is re-run per step at backward time;
  * GMM priors : component responsibilities from the HIP quadratic-form kernel (cf_gmm_resp), the remaining
                 contractions ((B x 80)(80 x D) and (80 x B)(B x D)) on cf_linear / cf_linear_wgrad (fp32 MFMA) + small
                 elementwise kernels;
  * Squeeze / SplitPrior / Augment : index maps (squeeze kernel with `inverse`, concatenation).
Reference quirks carried into the gradients: ActNorm's ldj = +sum(logs) (d/dlogs gets sum_b g_ld), Conv1x1's
ldj = H*W*log|det W| (d/dW gets sum_b g_ld * H*W * W^-T).  Layers that run layer by layer in the plan (TransCoupling + its ViT,
Conv1x1 / ActNorm of other shapes, Augment: the SMAP topology) go through autograd_layers.py.  Weight-gradient partial sums are combined in a fixed order (no float atomics)."""
import torch

from . import _hip
from .autograd_layers import layer_backward
from .squeeze import squeeze_op



def _param_part_on(side, keep, alive, fn, defer, dev):
    """Run fn() - the parameter half of a record's backward - on the stream `side`, behind what the current stream has been
    given so far; `alive` (what it reads) goes to `keep`.  With `defer` (a list) the launch is POSTPONED: a thunk that does it
    is appended and None returned - the caller runs the thunk once it has issued the next record's kernels on the main
    stream.  (A captured HIP graph runs its nodes on a few hardware queues in creation order, and a node that waits for a node of
    another queue waits for everything created on that queue before the waiter: created right behind its backward kernel, a
    step's weight-gradient chain held up the NEXT step's backward kernel - tools/dev/step_timeline.py.)"""
    import contextlib
    if side is None:
        return fn()
    main = torch.cuda.current_stream(dev)
    keep.append(alive)
    if defer is None:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            return fn()
    ev = torch.cuda.Event()
    ev.record(main)

    def thunk():
        side.wait_event(ev)
        with torch.cuda.stream(side):
            return fn()
    defer.append(thunk)
    return None


# ------------------------------------------------------------------------------------------------ GMM prior
def gmm_backward(x, dist, prepared, g, gcol=None, side=None, keep=None, sink=None, defer=None):
    """x: (B, D...) possibly a channel slice; g: (B, M) upstream; gcol: its column sums (M,) if the caller has them (the
    priors of one backward pass share g).  Returns (gx like x, {param: grad})."""
    a, nm, cst, M, K, D = prepared
    xv, xbs = _hip.bview(x)
    B = xv.shape[0]
    r = torch.empty(B, M * K, device=xv.device, dtype=torch.float32)          # responsibilities x upstream
    ws = torch.empty(_hip.lib().cf_gmm_resp_ws_bytes(B, M, K, D), device=xv.device, dtype=torch.uint8)
    _hip.call("cf_gmm_resp", _hip.p(xv), _hip.p(a), _hip.p(nm), _hip.p(cst), _hip.p(_hip.f32(g)), _hip.p(r), _hip.p(ws), B, M, K,
              D, xbs, _hip.stream())
    dev, st, pp = xv.device, _hip.stream(), _hip.p
    MK = M * K
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    # d/dx = -sum_mk r a^2 (x + nm) = -(x (r A2) + r AB), nm = -mu: two (B x MK)(MK x D) products on cf_linear (fp32 MFMA; the
    # coefficient kernel writes A2 / AB transposed, as cf_linear's weight operand) and one combining kernel
    A2t, ABt = new(D, MK), new(D, MK)
    _hip.call("cf_gmm_bwd_coeffs", pp(a), pp(nm), pp(A2t), pp(ABt), MK, D, 1, st)
    G1, G2 = new(B, D), new(B, D)
    _hip.call("cf_linear", pp(r), pp(A2t), None, None, pp(G1), B, MK, D, 0, st)
    _hip.call("cf_linear", pp(r), pp(ABt), None, None, pp(G2), B, MK, D, 0, st)
    gx = new(B, D)
    _hip.call("cf_gmm_bwd_gx", pp(xv), pp(G1), pp(G2), pp(gx), B, D, xbs, st)
    # parameter sums over the batch: S1 = r^T x, S2 = r^T x^2 (MK x D) and S0 = column sums of r, as split-K MFMA GEMMs
    # over the samples (cf_linear_wgrad: the bias-gradient column gives S0; x is squared while it is staged for S2)
    # (small batches: on the side stream - the chain to the previous layer needs gx only)
    return gx.view(xv.shape), _param_part_on(side, keep, (xv, r, a, nm, gcol),
                                             lambda: _gmm_param_part(xv, xbs, r, dist, a, nm, g, gcol, B, M, K, D, dev, sink), defer, dev)


def _out(sink, p, shape, dev):
    """Gradient buffer of parameter p with the kernel's own shape: the parameter's slice of the data-parallel bucket
    (dist.GradBucket) when one is active - the kernel then writes p.grad's storage directly - or a fresh tensor."""
    v = sink(p) if sink is not None else None
    return v.view(shape) if v is not None else torch.empty(shape, device=dev, dtype=torch.float32)


def _gmm_param_part(xv, xbs, r, dist, a, nm, g, gcol, B, M, K, D, dev, sink=None):
    """Second half of gmm_backward: the parameter sums and the gradients of mG / sG / wG."""
    st, pp = _hip.stream(), _hip.p
    MK = M * K
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    S0, S1, S2 = new(MK), new(MK, D), new(MK, D)
    L = _hip.lib()
    if L.cf_gmm_bwd_sums_supported(MK, D) and xbs < (1 << 20):
        # one product with two right-hand sides (x, x^2) on 16-row MFMA tiles, x read in place through its batch stride
        wsw = torch.empty(L.cf_gmm_bwd_sums_ws_bytes(B, MK, D), device=dev, dtype=torch.uint8)
        _hip.call("cf_gmm_bwd_sums", pp(xv), pp(r), pp(S0), pp(S1), pp(S2), pp(wsw), B, MK, D, xbs, st)
    else:
        xf = xv.reshape(B, -1) if xv.is_contiguous() else xv.contiguous().reshape(B, -1)
        wsw = torch.empty(L.cf_linear_wgrad_ws_bytes(B, D, MK), device=dev, dtype=torch.uint8)
        _hip.call("cf_linear_wgrad", pp(xf), pp(r), pp(S1), pp(S0), pp(wsw), B, D, MK, st)
        _hip.call("cf_linear_wgrad_x2", pp(xf), pp(r), pp(S2), pp(wsw), B, D, MK, st)
    g_mu, g_sigma = _out(sink, dist.mG, (MK, D), dev), _out(sink, dist.sG, (MK, D), dev)
    sG = _hip.f32(dist.sG.detach()).reshape(MK, D)
    if gcol is None:
        gcol = _hip.f32(g).sum(0)
    g_w = _out(sink, dist.wG, (M, K), dev)
    wG = _hip.f32(dist.wG.detach()).reshape(M, K)
    _hip.call("cf_gmm_bwd_params_w", pp(a), pp(nm), pp(sG), pp(S0), pp(S1), pp(S2), pp(wG), pp(_hip.f32(gcol)), pp(g_mu), pp(g_sigma),
              pp(g_w), M, K, D, st)
    return {dist.mG: g_mu.view_as(dist.mG), dist.sG: g_sigma.view_as(dist.sG), dist.wG: g_w.view_as(dist.wG)}


# ------------------------------------------------------------------------------------------------ flow step
WGRAD_SIDE_MAX_BATCH = 1024      # below: the weight gradients of a step run on a side stream, next to the data-gradient chain
WGRAD_SIDE_STREAMS = int(__import__("os").environ.get("CONTEXTFLOW_WGRAD_STREAMS", "4"))
WGRAD_DEFER = __import__("os").environ.get("CONTEXTFLOW_WGRAD_DEFER", "0") == "1"    # see _param_part_on (measured: slower)
WGRAD_BATCH = __import__("os").environ.get("CONTEXTFLOW_WGRAD_BATCH", "1") == "1"    # see _step_param_part_batch


def step_backward(x, squeeze, conv, act, cpl, shape, ws, gz, gld, winv=None, planes=None, gsum=None, side=None, keep=None, wsb=None,
                  sink=None, defer=None, collect=None):
    """x: saved step input (un-squeezed when `squeeze`), gz: dL/dz (B,C,H,W), gld: dL/d(ld1) (B,).
    planes: the step tape (y0, h1, h2, aux) written by cf_flow_step_fwd_taped, or None = rebuild it from x with the same kernel.
    gsum: 1-element tensor sum(gld) (the same for every step of a backward pass), or None.
    side / keep (small batches): the weight gradients and the parameter chain - which the data-gradient chain of the steps
    before does not wait for - are launched on the stream `side` once the backward kernel has written its planes; what they
    read is appended to `keep` (the caller holds it until it has joined the streams: a block handed back to the allocator could
    be given out again on the main stream while the side stream still reads it).
    Returns (dL/dx in the layout of x, {param: grad})."""
    C, H, W = shape
    HW, HALF, HID = H * W, C // 2, 2 * C
    xv, xbs = _hip.bview(x)
    B, dev = xv.shape[0], xv.device
    L = _hip.lib()
    f, pp, st = _hip.f32, _hip.p, _hip.stream()
    c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
    Wm, t, logs = f(conv.NN.detach()), f(act.NN_t.detach()), f(act.NN_logs.detach())
    if wsb is None:                  # (the training forward packs the backward kernel's fragments on its side stream)
        wsb = torch.empty(L.cf_flow_step_bwd_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
        _hip.call("cf_flow_step_bwd_prepare", pp(Wm), pp(logs), pp(f(c1.weight.detach())), pp(f(c2.weight.detach())),
                  pp(f(c3.weight.detach())), pp(wsb), C, H, W, st)
    new = lambda rows: torch.empty(B, rows, HW, device=dev, dtype=torch.float32)
    # with a Squeeze in front of the step, dL/dx leaves the kernel in the un-squeezed layout of x (index map folded into its stores)
    gx = torch.empty((B, C // 4, 2 * H, 2 * W) if squeeze else (B, C, H, W), device=dev, dtype=torch.float32)
    s_gh, s_gh2, s_gh1, s_gy = new(C), new(HID), new(HID), new(C)
    gzc = f(gz)
    if planes is None:
        # recompute form (the forward kept only the step input): run the SAME taping forward kernel again into a scratch
        # tape that lives for this step only - the ReLU masks and log-scales are then bit for bit the ones of the forward
        # that produced the loss, whichever form (Winograd / direct, by batch size) the dispatch chose for the 3x3.
        # (Round 2 rebuilt h2 inside the backward kernel in the direct form: under the Winograd forward, units within
        # rounding of zero got another mask bit and a tensor's gradient could differ by 1e-3..1e-2 of its largest entry.)
        from .flowsequential import step_tape
        planes = step_tape(B, C, H, W, dev)
        zs = torch.empty(B, C, H, W, device=dev, dtype=torch.float32)
        lds = torch.zeros(B, device=dev, dtype=torch.float32)
        _hip.call("cf_flow_step_fwd_taped", pp(xv), pp(zs), pp(lds), pp(ws), pp(planes[0]), pp(planes[1]), pp(planes[2]),
                  pp(planes[3]), B, C, H, W, xbs, int(squeeze), st)
        del zs, lds
    s_y0, s_h1, s_h2, aux = planes
    _hip.call("cf_flow_step_bwd_taped", pp(gzc), pp(f(gld)), pp(wsb), pp(aux), pp(gx), pp(s_gh), pp(s_gh2), pp(s_gh1),
              pp(s_gy), B, C, H, W, int(bool(squeeze)), st)
    # ---- weight gradients: split-K MFMA GEMMs over (batch, pixel) with the 3x3 tap shifts, the four of a step in one call
    # (cf_step_wgrads: four k_wgrad launches, ONE reduce launch)
    if collect is not None:
        # small batches: the parameter halves of the steps of a resolution level are launched TOGETHER once the level's
        # backward kernels are in the queue (_step_param_part_batch); what they read travels in `collect`
        collect.append((conv, act, cpl, (C, H, W), xv, xbs, squeeze, s_gh, s_gh2, s_gh1, s_gy, s_h2, s_h1, s_y0, Wm, t, logs, winv,
                        (planes, wsb, gzc)))
        return gx, None
    return gx, _param_part_on(side, keep, (s_gh, s_gh2, s_gh1, s_gy, planes, xv, wsb, gzc, winv, gsum),
                              lambda: _step_param_part(conv, act, cpl, (C, H, W), xv, xbs, squeeze, s_gh, s_gh2, s_gh1, s_gy, s_h2, s_h1,
                                                       s_y0, Wm, t, logs, winv, gsum, gld, B, dev, sink), defer, dev)


def _step_param_part(conv, act, cpl, shape, xv, xbs, squeeze, s_gh, s_gh2, s_gh1, s_gy, s_h2, s_h1, s_y0, Wm, t, logs, winv, gsum,
                     gld, B, dev, sink=None):
    """Second half of step_backward: the four weight gradients of the step and the Conv1x1 / ActNorm parameter chain."""
    C, H, W = shape
    HW, HALF, HID = H * W, C // 2, 2 * C
    L = _hip.lib()
    f, pp, st = _hip.f32, _hip.p, _hip.stream()
    c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
    e = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
    o = lambda p, *sh: _out(sink, p, sh, dev)
    gw3, gb3, gw2, gb2 = o(c3.weight, 1, C, HID), o(c3.bias, C), o(c2.weight, HID, HID, 3, 3), o(c2.bias, HID)
    gw1, gb1, gWp, gbp = o(c1.weight, 1, HID, HALF), o(c1.bias, HID), e(1, C, C), e(C)
    # the step input is read in place by the Conv1x1 weight gradient: through its batch stride (a channel slice after a
    # SplitPrior) and, behind a Squeeze, through the squeeze index map - no squeezed / contiguous copy
    wsw = torch.empty(L.cf_step_wgrads_ws_bytes(B, C, H, W), device=dev, dtype=torch.uint8)
    _hip.call("cf_step_wgrads", pp(s_gh), pp(s_gh2), pp(s_gh1), pp(s_gy), pp(s_h2), pp(s_h1), pp(s_y0), pp(xv), pp(gw3), pp(gb3),
              pp(gw2), pp(gb2), pp(gw1), pp(gb1), pp(gWp), pp(gbp), pp(wsw), B, C, H, W, xbs, int(bool(squeeze)), st)
    gw3, gw1, gWp = gw3[0], gw1[0], gWp[0]
    # ---- chain to Conv1x1 / ActNorm parameters (W' = diag(s) Wm, b' = -t s, s = exp(-logs)): one small kernel
    if gsum is None:
        gsum = gld.sum().reshape(1)
    if winv is None:
        lad = torch.empty(1, device=dev, dtype=torch.float32)
        winv = torch.empty(C, C, device=dev, dtype=torch.float32)
        _hip.call("cf_slogdet_inverse", pp(Wm), C, pp(lad), pp(winv), st)
    gNN, gt, glogs = o(conv.NN, C, C), o(act.NN_t, C), o(act.NN_logs, C)
    gWpc = gWp.contiguous()
    _hip.call("cf_step_param_grads", pp(gWpc), pp(gbp), pp(Wm), pp(t), pp(logs), pp(f(winv)), pp(f(gsum)), HW, pp(gNN), pp(gt),
              pp(glogs), C, st)
    grads = {
        conv.NN: gNN.view_as(conv.NN), act.NN_t: gt.view_as(act.NN_t), act.NN_logs: glogs.view_as(act.NN_logs),   # quirk: ldj = +sum(logs)
        c1.weight: gw1.reshape(c1.weight.shape), c1.bias: gb1,
        c2.weight: gw2, c2.bias: gb2,
        c3.weight: gw3.reshape(c3.weight.shape), c3.bias: gb3,
    }
    return grads



def _step_param_part_batch(items, gsum, gld, B, dev, sink=None):
    """_step_param_part for the n steps of one shape in 6 - 7 launches instead of 6 n (cf_step_wgrads_batch: one launch per
    product over all steps + one reduce; cf_step_param_grads_batch: one workgroup per step): at the reference's batch of 256
    the parameter work of a backward pass is latency, not arithmetic.  Same kernels on the same operands: bitwise equal.
    items: what step_backward collected.  Returns one {param: grad} per item."""
    import ctypes
    L = _hip.lib()
    f, pp, st, A = _hip.f32, _hip.p, _hip.stream(), _hip.ptr_array
    C, H, W = items[0][3]
    HW, HALF, HID = H * W, C // 2, 2 * C
    e = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)
    o = lambda p, *sh: _out(sink, p, sh, dev)
    n = len(items)
    outs = []
    for (conv, act, cpl, _, xv, xbs, squeeze, s_gh, s_gh2, s_gh1, s_gy, s_h2, s_h1, s_y0, Wm, t, logs, winv, _alive) in items:
        c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
        outs.append(dict(gw3=o(c3.weight, 1, C, HID), gb3=o(c3.bias, C), gw2=o(c2.weight, HID, HID, 3, 3), gb2=o(c2.bias, HID),
                         gw1=o(c1.weight, 1, HID, HALF), gb1=o(c1.bias, HID), gWp=e(1, C, C), gbp=e(C),
                         gNN=o(conv.NN, C, C), gt=o(act.NN_t, C), glogs=o(act.NN_logs, C),
                         wsw=torch.empty(L.cf_step_wgrads_ws_bytes(B, C, H, W), device=dev, dtype=torch.uint8)))
    col = lambda k: [it[k] for it in items]
    oc = lambda k: [d[k] for d in outs]
    xbs_arr = (ctypes.c_int64 * n)(*[int(it[5]) for it in items])
    sq_arr = (ctypes.c_int * n)(*[int(bool(it[6])) for it in items])
    _hip.call("cf_step_wgrads_batch", n, A(col(7)), A(col(8)), A(col(9)), A(col(10)), A(col(11)), A(col(12)), A(col(13)), A(col(4)),
              A(oc("gw3")), A(oc("gb3")), A(oc("gw2")), A(oc("gb2")), A(oc("gw1")), A(oc("gb1")), A(oc("gWp")), A(oc("gbp")),
              A(oc("wsw")), B, C, H, W, ctypes.cast(xbs_arr, ctypes.c_void_p), ctypes.cast(sq_arr, ctypes.c_void_p), st)
    if gsum is None:
        gsum = gld.sum().reshape(1)
    winvs = []
    for it in items:
        winv = it[17]
        if winv is None:
            lad = torch.empty(1, device=dev, dtype=torch.float32)
            winv = torch.empty(C, C, device=dev, dtype=torch.float32)
            _hip.call("cf_slogdet_inverse", pp(it[14]), C, pp(lad), pp(winv), st)
        winvs.append(f(winv))
    _hip.call("cf_step_param_grads_batch", n, A(oc("gWp")), A(oc("gbp")), A(col(14)), A(col(15)), A(col(16)), A(winvs), pp(f(gsum)), HW,
              A(oc("gNN")), A(oc("gt")), A(oc("glogs")), C, st)
    res = []
    for it, d in zip(items, outs):
        conv, act, cpl = it[0], it[1], it[2]
        c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
        res.append({
            conv.NN: d["gNN"].view_as(conv.NN), act.NN_t: d["gt"].view_as(act.NN_t), act.NN_logs: d["glogs"].view_as(act.NN_logs),
            c1.weight: d["gw1"][0].reshape(c1.weight.shape), c1.bias: d["gb1"],
            c2.weight: d["gw2"], c2.bias: d["gb2"],
            c3.weight: d["gw3"][0].reshape(c3.weight.shape), c3.bias: d["gb3"],
        })
    return res


# ------------------------------------------------------------------------------------------------ transformer flow step
def wgrad_group(members, dev, dests=None):
    """members: [(x (rows, K), gy (rows, N), has_bias)], all dense fp32 on `dev`.  One cf_linear_wgrad_group launch pair;
    returns [(gW (N, K), gb (N,) | None)].  dests (data-parallel bucket): per member (gW | None, gb | None) - tensors the kernel
    writes instead of slices of a fresh buffer."""
    import ctypes
    L = _hip.lib()
    n = len(members)
    rows = [m[0].shape[0] for m in members]
    Ks = [m[0].shape[1] for m in members]
    Ns = [m[1].shape[1] for m in members]
    tot = sum(N * K + (N if m[2] else 0) for m, K, N in zip(members, Ks, Ns))
    out = torch.empty(tot, device=dev, dtype=torch.float32)
    res, o = [], 0
    for i, (m, K, N) in enumerate(zip(members, Ks, Ns)):
        dW, db = dests[i] if dests is not None else (None, None)
        gW = out[o:o + N * K].view(N, K) if dW is None else dW.view(N, K); o += N * K
        gb = None
        if m[2]:
            gb = out[o:o + N] if db is None else db.view(N); o += N
        res.append((gW, gb))
    iarr = lambda v: (ctypes.c_int * n)(*v)
    parr = lambda ts: (ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts])
    r_, k_, n_ = iarr(rows), iarr(Ks), iarr(Ns)
    ws = torch.empty(L.cf_linear_wgrad_group_ws_bytes(r_, k_, n_, n), device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _hip.check(L.cf_linear_wgrad_group(parr([m[0] for m in members]), parr([m[1] for m in members]), parr([r[0] for r in res]),
                                           parr([r[1] for r in res]), r_, k_, n_, n, _hip.p(ws), _hip.stream(dev)), "cf_linear_wgrad_group")
    return res


def vstep_backward(x, conv, act, cpl, gz, gld, gsum=None, ws=None, xtape=None, side=None, keep=None, sink=None, winv=None, wsb=None,
                   defer=None):
    """Conv1x1 -> ActNorm -> TransCoupling (one fused step of the transformer flows) backwards: ONE kernel re-runs the step
    from its input and walks back (cf_vit_step_bwd), ONE grouped launch contracts the 26 weight-gradient operand pairs it
    leaves (cf_linear_wgrad_group), the LayerNorm gradients are column sums of its per-workgroup partials, and the Conv1x1 /
    ActNorm chain is cf_step_param_grads - as for the conv flows.  Returns (dL/dx, {param: grad})."""
    xv, xbs = _hip.bview(x)
    B, C = xv.shape[0], xv.shape[1]
    dev = xv.device
    vit = cpl.NN[0]
    depth = len(vit.transformer.layers)
    L = _hip.lib()
    f, pp, st = _hip.f32, _hip.p, _hip.stream()
    Wm, t, logs = f(conv.NN.detach()), f(act.NN_t.detach()), f(act.NN_logs.detach())
    if ws is None:                                        # the forward ran the wave form: pack the row-split table now
        ws = cpl.step_prepare(conv.NN, act.NN_t, act.NN_logs, dev, "rs")
    if wsb is None:              # (the training forward at small batches packs these with its own tables: cf_vit_step_rs_prepare_batch)
        flat = cpl._flat_params()
        wsb = torch.empty(L.cf_vit_step_bwd_ws_bytes(C, depth), device=dev, dtype=torch.uint8)
        _hip.call("cf_vit_step_bwd_prepare", pp(Wm), pp(logs), pp(flat), pp(wsb), C, depth, st)
    nwg = (B + 3) // 4
    planes = torch.empty(L.cf_vit_step_bwd_plane_floats(B, C, depth), device=dev, dtype=torch.float32)
    lnp = torch.empty(nwg, L.cf_vit_step_bwd_ln_floats(B, C, depth) // nwg, device=dev, dtype=torch.float32)
    gx = torch.empty(B, C, xv.shape[2], xv.shape[3], device=dev, dtype=torch.float32)
    if xtape is not None:        # the forward taped the residual stream at the layer boundaries: the layers are not run again
        _hip.call("cf_vit_step_bwd_taped", pp(xv), pp(f(gz).contiguous()), pp(f(gld)), pp(gx), pp(ws), pp(wsb), pp(planes), pp(lnp),
                  pp(xtape), B, C, depth, xbs, st)
    else:
        _hip.call("cf_vit_step_bwd", pp(xv), pp(f(gz).contiguous()), pp(f(gld)), pp(gx), pp(ws), pp(wsb), pp(planes), pp(lnp), B, C,
                  depth, xbs, st)
    # ---- weight gradients: the planes as (rows, width) matrices (layout: include/contextflow_hip.h, cf_vit_step_bwd)
    # (small batches: on the side stream, as in step_backward)
    HWv = xv.shape[2] * xv.shape[3]
    return gx, _param_part_on(side, keep, (planes, lnp, xv, ws, wsb, gsum, winv),
                              lambda: _vstep_param_part(conv, act, cpl, vit, depth, planes, lnp, nwg, C, HWv, Wm, t, logs, gsum, gld, dev,
                                                        sink, winv), defer, dev)


def _vstep_param_part(conv, act, cpl, vit, depth, planes, lnp, nwg, C, HW, Wm, t, logs, gsum, gld, dev, sink=None, winv=None):
    """Second half of vstep_backward: grouped weight gradients, LayerNorm sums, Conv1x1 / ActNorm parameter chain."""
    f, pp, st = _hip.f32, _hip.p, _hip.stream()
    Bp = nwg * 4
    R4, P8, PD, DIM = 4 * Bp, 8 * Bp, C, 2 * C
    o = [0]

    def take(rows, width):
        v = planes[o[0]:o[0] + rows * width].view(rows, width)
        o[0] += rows * width
        return v
    xT, gyT, u0, ge = take(P8, C), take(P8, C), take(R4, PD), take(R4, DIM)
    members = [(xT, gyT, True), (u0, ge, True)]
    for _ in range(depth):
        u1, gqkv, oo, gxm, u2, ghp, h, gxo = (take(R4, DIM), take(R4, 192), take(R4, 64), take(R4, DIM), take(R4, DIM), take(R4, DIM),
                                               take(R4, DIM), take(R4, DIM))
        members += [(u1, gqkv, False), (oo, gxm, False), (u2, ghp, True), (h, gxo, True)]
    tpe = vit.to_patch_embedding
    dests = None
    if sink is not None:             # data-parallel bucket: the grouped kernel writes the Linear gradients into p.grad's storage
        dests = [(None, None), (sink(tpe[2].weight), sink(tpe[2].bias))]
        for attn, ff in vit.transformer.layers:
            dests += [(sink(attn.to_qkv.weight), None), (sink(attn.to_out.weight), None), (sink(ff.net[1].weight), sink(ff.net[1].bias)),
                      (sink(ff.net[3].weight), sink(ff.net[3].bias))]
    wg = wgrad_group(members, dev, dests)
    ln = lnp.sum(0)                                       # fixed-order column sums of the per-workgroup partials
    grads = {}
    grads[tpe[1].weight], grads[tpe[1].bias] = ln[0:PD], ln[32:32 + PD]
    grads[tpe[2].weight], grads[tpe[2].bias] = wg[1]
    grads[tpe[3].weight], grads[tpe[3].bias] = ln[64:64 + DIM], ln[128:128 + DIM]
    for l, (attn, ff) in enumerate(vit.transformer.layers):
        b = 192 + 256 * l
        grads[attn.norm.weight], grads[attn.norm.bias] = ln[b:b + DIM], ln[b + 64:b + 64 + DIM]
        grads[ff.net[0].weight], grads[ff.net[0].bias] = ln[b + 128:b + 128 + DIM], ln[b + 192:b + 192 + DIM]
        q = 2 + 4 * l
        grads[attn.to_qkv.weight] = wg[q][0]
        grads[attn.to_out.weight] = wg[q + 1][0]
        grads[ff.net[1].weight], grads[ff.net[1].bias] = wg[q + 2]
        grads[ff.net[3].weight], grads[ff.net[3].bias] = wg[q + 3]
    b = 192 + 256 * depth
    grads[vit.transformer.norm.weight], grads[vit.transformer.norm.bias] = ln[b:b + DIM], ln[b + 64:b + 64 + DIM]
    # ---- Conv1x1 / ActNorm chain from the folded matrix / bias gradient (W' = diag(s) Wm, b' = -t s)
    gWp, gbp = wg[0]
    if gsum is None:
        gsum = gld.sum().reshape(1)
    if winv is None:
        lad = torch.empty(1, device=dev, dtype=torch.float32)
        winv = torch.empty(C, C, device=dev, dtype=torch.float32)
        _hip.call("cf_slogdet_inverse", pp(Wm), C, pp(lad), pp(winv), st)
    gNN, gt, glogs = _out(sink, conv.NN, (C, C), dev), _out(sink, act.NN_t, (C,), dev), _out(sink, act.NN_logs, (C,), dev)
    _hip.call("cf_step_param_grads", pp(gWp.contiguous()), pp(gbp.contiguous()), pp(Wm), pp(t), pp(logs), pp(winv), pp(f(gsum)),
              HW, pp(gNN), pp(gt), pp(glogs), C, st)
    grads[conv.NN] = gNN.view_as(conv.NN)
    grads[act.NN_t], grads[act.NN_logs] = gt.view_as(act.NN_t), glogs.view_as(act.NN_logs)
    return grads


# ------------------------------------------------------------------------------------------------ the Function
class FlowLogProb(torch.autograd.Function):
    """logp (B,M) of a FlowSequential with parameter gradients.  `params` are passed positionally only so that autograd
    tracks them; the arithmetic reads them from the modules."""

    @staticmethod
    def forward(ctx, flow, x, *params):
        tape = []
        z, logp = flow._forward_fused(x, None, tape=tape)
        ctx.flow, ctx.tape, ctx.params = flow, tape, params
        ctx.mark_non_differentiable(z)
        return z, logp

    @staticmethod
    def backward(ctx, _gz_unused, glogp):
        flow, tape, params = ctx.flow, ctx.tape, ctx.params
        glogp = _hip.f32(glogp)
        gld = glogp.sum(1).contiguous()                     # d/d ld1[b]: logp = ldM + ld1[:, None]
        gsum = gld.sum().reshape(1)                         # shared by the parameter chains of all steps
        gcol = glogp.sum(0)                                 # ... and by the mixture-weight gradients of all priors
        acc = {}

        # data-parallel training (FlowSequential.data_parallel): gradients are written straight into the flat bucket that
        # p.grad views (dist.GradBucket); each segment is all-reduced as soon as its last gradient kernel has been launched
        bucket = _bucket_for(flow, tape, params) if getattr(flow, "data_parallel", False) else None
        sink = bucket.view if bucket is not None else None
        pending, written = ([], []), set()

        def add(d):
            for p, g in d.items():
                v = sink(p) if sink is not None else None
                if v is None:
                    acc[p] = g if p not in acc else acc[p] + g
                elif p in written:
                    v.add_(g.view_as(v))
                else:
                    written.add(p)
                    if g.data_ptr() != v.data_ptr():         # produced elsewhere (transformer steps, layer-by-layer records)
                        pending[0].append(v)
                        pending[1].append(g.reshape(v.shape))

        def flush(seg_done):
            """copy what did not land in the bucket by itself (one multi-tensor launch), then start the collectives"""
            if pending[0]:
                torch._foreach_copy_(pending[0], pending[1])
                del pending[0][:], pending[1][:]
            for i in seg_done:
                bucket.reduce(i)

        # small batches: the kernels of a step fill a fraction of the chip, and the weight gradients of step k are off the chain
        # that leads to step k - 1 - they run on the flow's side stream and meet the main stream once, at the end
        B0 = glogp.shape[0]
        dev = glogp.device
        # (WGRAD_SIDE_STREAMS of them, taken in turn by the records: the parameter work of one step is a chain of 6 - 8 small
        # launches, longer than the step's backward kernel, so a single side stream becomes the critical path)
        sides = ([flow._side_stream(dev, k) for k in range(WGRAD_SIDE_STREAMS)]
                 if (glogp.is_cuda and B0 <= WGRAD_SIDE_MAX_BATCH) else [])
        side, owner = None, {}
        keep = []

        def add_on(d, ri, side):     # parameter gradients produced on a side stream are accumulated there
            import contextlib
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                if side is not None:
                    for p in d:                                  # a parameter shared by two records: behind its first producer
                        o = owner.setdefault(p, side)
                        if o is not side:
                            side.wait_stream(o)
                add(d)
                if bucket is not None:
                    flush(())
            seg_done = bucket.closes.get(ri, ()) if bucket is not None else ()
            if seg_done:
                # the collectives of all segments are issued from ONE stream (the first side stream, behind the others: the
                # segment's gradients came from all of them) - RCCL's own stream then follows a single launching stream
                red = sides[0] if sides else None
                with (torch.cuda.stream(red) if red is not None else contextlib.nullcontext()):
                    for s_ in sides[1:]:
                        red.wait_stream(s_)
                    flush(seg_done)
        gz, turn = None, -1
        held = []                    # [(thunk, record index, side)]: the previous record's parameter half, not launched yet
        group, group_ri, group_side = [], [], None      # the steps of the current resolution level (WGRAD_BATCH)

        def run_pending():
            while held:
                thunk, ri_p, side_p = held.pop(0)
                add_on(thunk(), ri_p, side_p)

        def finish(gp, ri, defer):
            """this record's kernels of the data-gradient chain are in the queue: launch the previous record's parameter half,
            then hold this one's back in turn (WGRAD_DEFER) or account for its gradients at once"""
            run_pending()
            if defer:
                held.append((defer[0], ri, side))
            else:
                add_on(gp, ri, side)
        for ri in range(len(tape) - 1, -1, -1):
            rec = tape[ri]
            kind = rec[0]
            if sides and kind in ("prior", "split", "step", "vstep"):
                turn += 1
                side = sides[turn % len(sides)]
            defer = [] if (sides and WGRAD_DEFER) else None
            if kind == "prior":
                _, xin, dist, prep = rec
                gz, gp = gmm_backward(xin, dist, prep, glogp, gcol, side, keep, sink, defer)
                finish(gp, ri, defer)
            elif kind == "split":
                _, xin, dist, prep = rec                      # xin: full tensor before the split
                c = xin.shape[1] // 2
                g2, gp = gmm_backward(xin[:, c:], dist, prep, glogp, gcol, side, keep, sink, defer)
                gz = torch.cat([gz, g2], dim=1)
                finish(gp, ri, defer)
            elif kind == "step":
                _, xin, sq, conv, act, cpl, shape, ws, winv, planes, wsb = rec
                if sides and WGRAD_BATCH:
                    # the level's steps share ONE side stream and ONE set of batched launches, issued behind its last backward kernel
                    if not group:
                        group_side = side
                    gz, _ = step_backward(xin, sq, conv, act, cpl, shape, ws, gz, gld, winv, planes, gsum, None, None, wsb, sink,
                                          None, group)
                    group_ri.append(ri)
                    nxt = tape[ri - 1] if ri > 0 else None
                    if nxt is None or nxt[0] != "step" or tuple(nxt[6]) != tuple(shape):
                        run_pending()
                        items, ris = list(group), list(group_ri)
                        del group[:], group_ri[:]
                        gps = _param_part_on(group_side, keep, items, lambda: _step_param_part_batch(items, gsum, gld, B0, dev, sink),
                                             None, dev)
                        for gp_i, ri_i in zip(gps, ris):
                            add_on(gp_i, ri_i, group_side)
                else:
                    gz, gp = step_backward(xin, sq, conv, act, cpl, shape, ws, gz, gld, winv, planes, gsum, side, keep, wsb, sink, defer)
                    finish(gp, ri, defer)
            elif kind == "vstep":
                _, xin, conv, act, cpl, ws_rs, xtape = rec[:7]
                winv_v, wsb_v = rec[7:9] if len(rec) >= 9 else (None, None)
                gz, gp = vstep_backward(xin, conv, act, cpl, gz, gld, gsum, ws_rs, xtape, side, keep, sink, winv_v, wsb_v, defer)
                finish(gp, ri, defer)
            elif kind == "squeeze":
                gz = squeeze_op(gz, rec[1], True)
            elif kind == "pre":
                break                                        # nothing trainable upstream of the pre-processing
            elif kind == "layer":
                _, mod, xin = rec
                gz, gp = layer_backward(mod, xin, gz, gld)
                run_pending()
                add(gp)
                if bucket is not None:
                    flush(bucket.closes.get(ri, ()))
            else:
                raise NotImplementedError("no backward for tape record %r" % (kind,))
        run_pending()
        if sides:
            main = torch.cuda.current_stream(dev)
            for s_ in sides:
                main.wait_stream(s_)
            for g in acc.values():
                g.record_stream(main)
            del keep
        if bucket is not None:
            bucket.finish()                                  # this stream (the optimizer's) waits for the collectives
            for p in params:
                v = bucket.view(p)
                if v is not None:
                    p.grad = v                               # (assigned, not accumulated: one backward per optimizer step)
        return (None, None) + tuple(acc.get(p) for p in params)


SEGMENT_MIN_BYTES = 1 << 20      # data-parallel bucket: a segment is closed at the first record boundary past this size, and at every SplitPrior


def _record_params(rec):
    """trainable tensors of a tape record, in the order the backward lists them"""
    kind = rec[0]
    if kind in ("prior", "split"):
        d = rec[2]
        return [d.mG, d.sG, d.wG]
    if kind == "step":
        conv, act, cpl = rec[3], rec[4], rec[5]
        c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
        return [conv.NN, act.NN_t, act.NN_logs, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias]
    if kind == "vstep":
        return [p for m in rec[2:5] for p in m.parameters()]
    if kind == "layer":
        return list(rec[1].parameters())
    return []


def _bucket_for(flow, tape, params):
    """The flow's gradient bucket for this tape (built on the first backward of a parameter set, then kept: its storage is
    what p.grad views, also across replays of a captured step).  Segments follow the backward: the final prior and the
    steps of the last level first; a segment closes at a SplitPrior (= a resolution level is done) or once it holds
    SEGMENT_MIN_BYTES; bucket.closes[record index] = segments whose last gradient that record produces."""
    from .. import dist as cdist
    want = {id(p) for p in params if p.requires_grad}
    groups, closes, cur, size = [], {}, [], 0
    for ri in range(len(tape) - 1, -1, -1):
        rec = tape[ri]
        if rec[0] == "pre":
            break
        if rec[0] == "split" and cur:                        # the level above is complete
            closes.setdefault(last, []).append(len(groups))
            groups.append(cur)
            cur, size = [], 0
        ps = [p for p in _record_params(rec) if id(p) in want]
        if not ps:
            continue
        cur += ps
        size += 4 * sum(p.numel() for p in ps)
        last = ri
        if size >= SEGMENT_MIN_BYTES:
            closes.setdefault(ri, []).append(len(groups))
            groups.append(cur)
            cur, size = [], 0
    if cur:
        closes.setdefault(last, []).append(len(groups))
        groups.append(cur)
    key = tuple(id(p) for g in groups for p in g)
    b = getattr(flow, "_grad_bucket", None)
    dev = params[0].device if params else torch.device("cpu")
    if b is None or b.key != key or b.flat.device != dev:
        b = flow._grad_bucket = cdist.GradBucket(groups, dev)
    b.closes = closes
    return b
