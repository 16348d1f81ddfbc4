"""Affine coupling layers (reference: contextflow/layers/coupling.py:14-159), context-free branch.

Coupling      — conditioner = Conv2d 1x1 -> ReLU -> Conv2d kxk (reflect) -> ReLU -> Conv2d 1x1
TransCoupling — conditioner = SimpleViT
Both transform the SECOND channel half given the first: t = h[:, :C/2], log_s = 2 tanh(h[:, C/2:]/2),
z1 = x1*exp(log_s) + t, ldj = sum log_s (coupling.py:52-66).  The nn.Conv2d / nn.Linear objects are
kept purely as parameter containers so that `state_dict` keys equal the reference's; the arithmetic
runs in the HIP kernels."""
import torch
import torch.nn as nn

from . import _hip
from .flowlayer import FlowLayer, encoder_noise

VIT_EVENTS = None        # bench.py: list collecting (start, end, batch) HIP events per fused ViT-coupling launch
from .simple_vit import SimpleViT


def conv2d_reflect(x, conv, relu):
    """x: (B,Cin,H,W) possibly a channel slice; conv: nn.Conv2d holding weight/bias."""
    x, xbs = _hip.bview(x)
    B, Cin, H, W = x.shape
    w = _hip.f32(conv.weight.detach())
    b = _hip.f32(conv.bias.detach()) if conv.bias is not None else None
    Cout, _, kh, kw = w.shape
    ph, pw = conv.padding if isinstance(conv.padding, tuple) else (conv.padding, conv.padding)
    out = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
    _hip.call("cf_conv2d_reflect", _hip.p(x), _hip.p(w), _hip.p(b), _hip.p(out), B, Cin, Cout, H, W, kh, kw, ph, pw,
              int(relu), xbs, _hip.stream())
    return out


def coupling_apply(x, h, inverse):
    x, h = _hip.f32(x), _hip.f32(h)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(B * C, 1) if B else 1
    z = torch.empty_like(x)
    ldj = None if inverse else torch.empty(B, device=x.device, dtype=torch.float32)
    _hip.call("cf_coupling_apply", _hip.p(x), _hip.p(h), _hip.p(z), _hip.p(ldj), B, C, HW, int(inverse), _hip.stream())
    return z, ldj


class _AffineCoupling(FlowLayer):
    def net(self, x0):
        raise NotImplementedError

    def forward(self, x, context=None):
        _hip.require_device(x)
        h = self.net(x[:, : x.shape[1] // 2])
        return coupling_apply(x, h, False)

    def reverse(self, z, context=None):
        _hip.require_device(z)
        h = self.net(z[:, : z.shape[1] // 2])
        return coupling_apply(z, h, True)[0]

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]


class Coupling(_AffineCoupling):
    def __init__(self, data_channels, kernel_size=(1, 1), padding=(0, 0), context_net=None, contextflow=False):
        super().__init__()
        D, Hd, O = data_channels // 2, data_channels * 2, data_channels
        self.context_net = context_net                     # registered before NN, as in the reference (key order)
        self.contextflow = contextflow
        concat = bool(context_net) and not contextflow       # coupling.py:33-34: CN(c) concatenated to the net input
        self.NN = nn.Sequential(
            nn.Conv2d(D + O if concat else D, Hd, 1), nn.ReLU(),
            nn.Conv2d(Hd, Hd, kernel_size, padding=padding, padding_mode="reflect"), nn.ReLU(),
            nn.Conv2d(Hd, O, 1))
        if self.context_net:                               # coupling.py:31-37
            if self.contextflow:
                for p in self.NN.parameters():
                    p.requires_grad_(False)
            self.C = self.context_net.C
            self.CN = nn.Sequential(nn.Linear(self.C, Hd), nn.ReLU(), nn.Linear(Hd, Hd), nn.ReLU(), nn.Linear(Hd, O))
        self.fused = True                                  # specialist: one-kernel path when the geometry allows it

    def net(self, x0):
        h = conv2d_reflect(x0, self.NN[0], True)
        h = conv2d_reflect(h, self.NN[2], True)
        return conv2d_reflect(h, self.NN[4], False)

    def _net_ctx(self, x0, context):
        """coupling.py:39-47: h = NN(x0) + CN(c) (contextflow) or NN([x0 ; CN(c) broadcast]) — the broadcast part of the
        first 1x1 convolution is a per-sample bias W[:, D:] CN(c) + b."""
        from .simple_vit import _linear
        c, logp_c = self.context_net(context)
        cn = _linear(_linear(_linear(_hip.f32(c), self.CN[0], act=2), self.CN[2], act=2), self.CN[4])    # (B, O)
        x0, xbs = _hip.bview(x0)
        B, D, H, W = x0.shape
        st = _hip.stream()
        if self.contextflow:
            h = self.net(x0)
            _hip.call("cf_add_sample_bias", _hip.p(h), _hip.p(cn), B, h.shape[1], H * W, 0, st)
            return h, logp_c
        c1 = self.NN[0]
        w = _hip.f32(c1.weight.detach())
        Hd = w.shape[0]
        wx = w[:, :D].contiguous()
        wc = w[:, D:, 0, 0].contiguous()
        bias_b = torch.empty(B, Hd, device=x0.device, dtype=torch.float32)       # W[:, D:] cn_b + b
        _hip.call("cf_linear", _hip.p(cn), _hip.p(wc), _hip.p(_hip.f32(c1.bias.detach())), None, _hip.p(bias_b), B,
                  wc.shape[1], Hd, 0, st)
        h1 = torch.empty(B, Hd, H, W, device=x0.device, dtype=torch.float32)
        _hip.call("cf_conv2d_reflect", _hip.p(x0), _hip.p(wx), None, _hip.p(h1), B, D, Hd, H, W, 1, 1, 0, 0, 0, xbs, st)
        _hip.call("cf_add_sample_bias", _hip.p(h1), _hip.p(bias_b), B, Hd, H * W, 1, st)
        h = conv2d_reflect(h1, self.NN[2], True)
        return conv2d_reflect(h, self.NN[4], False), logp_c

    def _fused_ctx(self, x, context, tape=None, pre=None):
        """The Coupling layer as ONE fp32-MFMA kernel (the fused flow-step kernel with an identity 1x1 / ActNorm in
        front) with the CN(c) term as a per-sample bias: on the conditioner output (contextflow) or before its first
        ReLU (CN(c) concatenated to the conditioner input: W[:, D:] CN(c))."""
        from .simple_vit import _linear
        if pre is not None:          # code, log-density and the CN chain from the grouped front end (layers/specialist.py, train form)
            c, logp_c, a1, a2, cn = pre["c"], pre["logp"], pre["a1"], pre["a2"], pre["cn"]
        else:
            c, logp_c = self.context_net(context)
            c = _hip.f32(c)
            a1 = _linear(c, self.CN[0], act=2)
            a2 = _linear(a1, self.CN[2], act=2)
            cn = _linear(a2, self.CN[4])                                                          # (B, O)
        x, xbs = _hip.bview(x)
        B, C, H, W = x.shape
        D = C // 2
        dev, st, f, pp = x.device, _hip.stream(), _hip.f32, _hip.p
        c1, c2, c3 = self.NN[0], self.NN[2], self.NN[4]
        w1 = f(c1.weight.detach())
        if self.contextflow:
            mode, sbias = 1, cn
        else:
            wc = w1[:, D:, 0, 0].contiguous()
            sbias = torch.empty(B, wc.shape[0], device=dev, dtype=torch.float32)
            _hip.call("cf_linear", pp(cn), pp(wc), None, None, pp(sbias), B, wc.shape[1], wc.shape[0], 0, st)
            mode, w1 = 2, w1[:, :D].contiguous()
        # packed tables of the step: kept while the conditioner is unchanged (version counter + storage of its six tensors) - under
        # contextflow it is frozen, and a training step otherwise factorises an identity and packs the same tables for every coupling
        srcs = (c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias)
        key = (mode, C, H, W, str(dev)) + tuple((t._version, t.data_ptr()) for t in srcs)
        hit = self.__dict__.get("_ctx_ws")
        capturing = torch.cuda.is_current_stream_capturing()
        if hit is not None and hit[0] == key and not capturing:
            ws = hit[1]
            torch.cuda.current_stream(dev).wait_event(hit[2])
        else:
            eye = torch.eye(C, device=dev, dtype=torch.float32)
            zero = torch.zeros(C, device=dev, dtype=torch.float32)
            ws = torch.empty(_hip.lib().cf_flow_step_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
            _hip.call("cf_flow_step_prepare", pp(eye), pp(zero), pp(zero), pp(w1), pp(f(c1.bias.detach())),
                      pp(f(c2.weight.detach())), pp(f(c2.bias.detach())), pp(f(c3.weight.detach())), pp(f(c3.bias.detach())),
                      pp(ws), C, H, W, st)
            if not capturing:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                self.__dict__["_ctx_ws"] = (key, ws, ev)
        z = torch.empty(B, C, H, W, device=dev, dtype=torch.float32)
        ldj = torch.zeros(B, device=dev, dtype=torch.float32)
        planes = None
        if tape is not None and mode == 2:
            # every parameter trains: the forward kernel also writes the step tape for the backward kernel and the weight gradients
            from .flowsequential import step_tape
            planes = step_tape(B, C, H, W, dev)
            _hip.call("cf_flow_step_fwd_ctx_taped", pp(x), pp(z), pp(ldj), pp(ws), pp(sbias), pp(planes[0]), pp(planes[1]),
                      pp(planes[2]), pp(planes[3]), B, C, H, W, xbs, st)
        else:
            # training under contextflow (tape, mode 1): the backward kernel rebuilds the conditioner in the direct form of the 3x3,
            # so the forward that produces the loss keeps that form too (flag 4) - same ReLU masks on both sides
            _hip.call("cf_flow_step_fwd_ctx", pp(x), pp(z), pp(ldj), pp(ws), pp(sbias), mode | (4 if tape is not None else 0), B, C, H, W,
                      xbs, st)
        if tape is not None:
            tape.append(dict(x=x, c=c, a1=a1, a2=a2, cn=cn, ws=ws, mode=mode, planes=planes, eps=encoder_noise(self.context_net)))
        return z, ldj + logp_c * float(H * W)

    def _fused_ctx_ok(self, x):
        k = self.NN[2]
        return (self.fused and x.dim() == 4 and tuple(k.kernel_size) == (3, 3) and tuple(k.padding) == (1, 1)
                and bool(_hip.lib().cf_flow_step_supported(x.shape[1], x.shape[2], x.shape[3], 3, 3)))

    def forward(self, x, context=None):
        if not self.context_net:
            return super().forward(x, context)
        _hip.require_device(x)
        if self._fused_ctx_ok(x):
            return self._fused_ctx(x, context)
        h, logp_c = self._net_ctx(x[:, : x.shape[1] // 2], context)
        z, ldj = coupling_apply(x, h, False)
        return z, ldj + logp_c * float(x.shape[2] * x.shape[3])

    def reverse(self, z, context=None):
        if not self.context_net:
            return super().reverse(z, context)
        _hip.require_device(z)
        h, _ = self._net_ctx(z[:, : z.shape[1] // 2], context)
        return coupling_apply(z, h, True)[0]


class CouplingFC(Coupling):
    def __init__(self, data_channels, kernel_size=(1, 1), padding=(0, 0), context_net=None, contextflow=False):
        super().__init__(data_channels, kernel_size=(1, 1), padding=(0, 0), context_net=None, contextflow=False)
        self.D = data_channels

    def forward(self, x, context=None):
        out, ldj = super().forward(x.view(-1, self.D, 1, 1), context)
        return out.view(-1, self.D), ldj

    def reverse(self, z, context=None):
        return super().reverse(z.view(-1, self.D, 1, 1), context).view(-1, self.D)

    def logdet(self, x, context=None):
        return super().logdet(x.view(-1, self.D, 1, 1))


class TransCoupling(_AffineCoupling):
    def __init__(self, in_sz, p_sz, context_net=None, contextflow=False):
        super().__init__()
        D, Hd, O = in_sz[0] // 2, in_sz[0] * 2, in_sz[0]
        T = O * p_sz[0] * p_sz[1]                       # transformer width (coupling.py:108)
        self.context_net = context_net
        self.contextflow = contextflow
        self.in_sz, self.p_sz = tuple(in_sz), tuple(p_sz)
        vit = dict(image_size=(in_sz[1], in_sz[2]), patch_size=p_sz, dim=T, depth=6, heads=1, mlp_dim=T)
        self.NN = nn.Sequential(SimpleViT(channels=D, **vit))
        if self.context_net:                               # coupling.py:113-119
            if not self.contextflow:
                self.NN = SimpleViT(channels=D + O, **vit)  # input = [x0 ; CN(c) broadcast]; a direct child (key names)
            else:
                for p in self.NN.parameters():
                    p.requires_grad_(False)
            self.C = self.context_net.C
            self.CN = nn.Sequential(nn.Linear(self.C, Hd), nn.ReLU(), nn.Linear(Hd, Hd), nn.ReLU(), nn.Linear(Hd, O))
        self.fused = True                                # one-kernel path when the geometry allows it

    def net(self, x0):
        """Conditioner output h (layer-by-layer kernels; also the fallback for unsupported geometries)."""
        return self.NN[0](x0)

    def _net_ctx(self, x0, context, tape=None):
        """coupling.py:123-133 with a context net: h = ViT(x0) + CN(c) (contextflow) or ViT([x0 ; CN(c) broadcast]).
        Quirk kept: logp_c is not multiplied by H*W here (unlike Coupling)."""
        from .simple_vit import _linear
        c, logp_c = self.context_net(context)
        c = _hip.f32(c)
        a1 = _linear(c, self.CN[0], act=2)
        a2 = _linear(a1, self.CN[2], act=2)
        cn = _linear(a2, self.CN[4])                                                              # (B, O)
        if tape is not None:                  # training (contextflow): what the CN-net backward needs
            tape.append(dict(c=c, a1=a1, a2=a2, cn=cn, eps=encoder_noise(self.context_net)))
        B, _, H, W = x0.shape
        if self.contextflow:
            h = self.NN[0](x0)
            _hip.call("cf_add_sample_bias", _hip.p(h), _hip.p(cn), B, h.shape[1], H * W, 0, _hip.stream())
            return h, logp_c
        xin = torch.cat([_hip.f32(x0), cn.view(B, -1, 1, 1).expand(B, cn.shape[1], H, W)], dim=1)   # concatenation: index op
        return self.NN(xin), logp_c

    # ---- fused path: patchify -> ViT -> un-patchify -> affine map -> log-det in one kernel
    def _fused_ok(self, x):
        if not self.fused or x.dim() != 4:
            return False
        vit = self.NN[0]
        C, H, W = x.shape[1:]
        att = vit.transformer.layers[0][0] if len(vit.transformer.layers) else None
        if att is None or (C, H, W) != self.in_sz:
            return False
        return bool(_hip.lib().cf_vit_supported(C, H, W, self.p_sz[0], self.p_sz[1], vit.dim, att.dim_head, att.heads))

    def step_supported(self, shape):
        """True when Conv1x1 -> ActNorm -> this layer can run as ONE kernel (cf_vit_step_fwd)."""
        if not self.fused or self.context_net or tuple(shape) != self.in_sz:
            return False
        vit = self.NN[0]
        att = vit.transformer.layers[0][0] if len(vit.transformer.layers) else None
        if att is None:
            return False
        C, H, W = shape
        L = _hip.lib()
        a = (C, H, W, self.p_sz[0], self.p_sz[1], vit.dim, att.dim_head, att.heads)
        # both one-kernel forms must cover the geometry (the dispatch picks by batch size, the training step runs the
        # row-split backward kernel, depth <= STEP_MAX_DEPTH); anything else keeps the layer path
        return bool(L.cf_vit_step_supported(*a)) and bool(L.cf_vit_step_rs_supported(*a)) and len(vit.transformer.layers) <= self.STEP_MAX_DEPTH

    def step_sources(self):
        """Parameters the packed step workspace derives from (cache key of FlowSequential).  The tuple is built once: walking
        the module tree costs more host time per call than the whole step kernel takes at a batch of 256."""
        src = getattr(self, "_step_src", None)
        if src is None:
            src = self._step_src = tuple(self.NN[0].parameters())
        return src

    # batches up to this size take the row-split step kernel (cf_vit_step_rs_fwd: 4 samples per workgroup, an eighth of the
    # serial chain); larger ones the one-wave-per-8-samples kernel (cf_vit_step_fwd)
    STEP_RS_MAX_BATCH = 3584          # measured cross-over (tools/dev/vit_variants.py, round 4, both forms with the fused attention tables): 36 vs 61 us at 2048, 46 vs 61 at 3072, 64 vs 61 at 4096

    STEP_MAX_DEPTH = 6                # cf_vit_step_bwd parks the residual stream of <= 6 layer boundaries in LDS

    def step_variant(self, B):
        """'rs' | 'wave': which one-kernel form of the step a batch of B samples takes (FlowSequential keys its packed
        workspaces on it: the two kernels have their own fragment layouts)."""
        return "rs" if B <= self.STEP_RS_MAX_BATCH else "wave"

    def step_prepare(self, Wm, t, logs, dev, variant="wave"):
        """Pack Conv1x1 / ActNorm / ViT parameters into the fragment order of the one-kernel step."""
        vit = self.NN[0]
        C, depth = self.in_sz[0], len(vit.transformer.layers)
        rs = variant == "rs"
        L = _hip.lib()
        ws = torch.empty((L.cf_vit_step_rs_ws_bytes if rs else L.cf_vit_step_ws_bytes)(C, depth), device=dev, dtype=torch.uint8)
        if vit.pos_embedding.device != dev:
            vit.pos_embedding = vit.pos_embedding.to(dev).contiguous()
        flat = self._flat_params()
        _hip.call("cf_vit_step_rs_prepare" if rs else "cf_vit_step_prepare", _hip.p(_hip.f32(Wm.detach())), _hip.p(_hip.f32(t.detach())),
                  _hip.p(_hip.f32(logs.detach())), _hip.p(flat), _hip.p(_hip.f32(vit.pos_embedding)), _hip.p(ws), C, depth, _hip.stream())
        return ws

    @staticmethod
    def step_prepare_rs_batch(items, dev, train=False):
        """The row-split tables of SEVERAL steps - items: [(coupling, Wm, t, logs)] - in one factorisation, one fuse and one
        packing launch (cf_vit_step_rs_prepare_batch).  train: also Wm^-1 and the backward kernel's tables per step.  Returns
        [ws] or [(ws, winv, wsb)]."""
        L = _hip.lib()
        cpl0 = items[0][0]
        vit = cpl0.NN[0]
        C, depth, n = cpl0.in_sz[0], len(vit.transformer.layers), len(items)
        if vit.pos_embedding.device != dev:
            vit.pos_embedding = vit.pos_embedding.to(dev).contiguous()
        f, A = _hip.f32, _hip.ptr_array
        ws = [torch.empty(L.cf_vit_step_rs_ws_bytes(C, depth), device=dev, dtype=torch.uint8) for _ in range(n)]
        Wm, t, logs = [f(i[1].detach()) for i in items], [f(i[2].detach()) for i in items], [f(i[3].detach()) for i in items]
        flat = [i[0]._flat_params() for i in items]
        winv = [torch.empty(C, C, device=dev, dtype=torch.float32) for _ in range(n)] if train else None
        wsb = [torch.empty(L.cf_vit_step_bwd_ws_bytes(C, depth), device=dev, dtype=torch.uint8) for _ in range(n)] if train else None
        _hip.call("cf_vit_step_rs_prepare_batch", n, A(Wm), A(t), A(logs), A(flat), _hip.p(_hip.f32(vit.pos_embedding)), A(ws),
                  A(winv) if train else None, A(wsb) if train else None, C, depth, _hip.stream())
        return list(zip(ws, winv, wsb)) if train else ws

    def step_forward(self, x, ws, ld1, h_out=None, variant="wave", xtape=None):
        """z = TransCoupling(ActNorm(Conv1x1(x))) and ld1 += the step's log-det, one launch.  xtape (training, wave form):
        receives the residual stream at the layer boundaries (cf_vit_step_fwd_taped) for cf_vit_step_bwd_taped."""
        x, xbs = _hip.bview(x)
        B, C = x.shape[0], x.shape[1]
        depth = len(self.NN[0].transformer.layers)
        z = torch.empty(B, C, x.shape[2], x.shape[3], device=x.device, dtype=torch.float32)
        if xtape is not None:
            assert h_out is None
            _hip.call("cf_vit_step_rs_fwd_taped" if variant == "rs" else "cf_vit_step_fwd_taped", _hip.p(x), _hip.p(z), _hip.p(ld1), _hip.p(ws), _hip.p(xtape), B, C, depth, xbs,
                      _hip.stream())
            return z
        events = VIT_EVENTS
        if events is not None:               # bench.py: HIP events on the launch stream around exactly this kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(x.device))
        _hip.call("cf_vit_step_rs_fwd" if variant == "rs" else "cf_vit_step_fwd", _hip.p(x), _hip.p(z), _hip.p(ld1), _hip.p(ws),
                  _hip.p(h_out), B, C, depth, xbs, _hip.stream())
        if events is not None:
            e1.record(torch.cuda.current_stream(x.device))
            events.append((e0, e1, B))
        return z

    def _flat_params(self):
        """The ViT parameters as one flat fp32 tensor in the order the pack kernels read them; kept until a parameter's
        version counter moves (a training step asks for it twice: forward and backward)."""
        src = self.step_sources()
        ver = tuple(p._version for p in src) + tuple(p.data_ptr() for p in src)
        hit = getattr(self, "_flat_cache", None)
        if hit is not None and hit[0] == ver:
            return hit[1]
        flat = self._flat_params_build()
        self._flat_cache = (ver, flat)
        return flat

    def _flat_params_build(self):
        vit = self.NN[0]
        tpe = vit.to_patch_embedding
        parts = [tpe[1].weight, tpe[1].bias, tpe[2].weight, tpe[2].bias, tpe[3].weight, tpe[3].bias]
        for attn, ff in vit.transformer.layers:
            parts += [attn.norm.weight, attn.norm.bias, attn.to_qkv.weight, attn.to_out.weight,
                      ff.net[0].weight, ff.net[0].bias, ff.net[1].weight, ff.net[1].bias, ff.net[3].weight, ff.net[3].bias]
        parts += [vit.transformer.norm.weight, vit.transformer.norm.bias]
        return torch.cat([t.detach().reshape(-1).float() for t in parts])

    def _fused(self, x, inverse):
        vit = self.NN[0]
        x, xbs = _hip.bview(x)
        B, C, H, W = x.shape
        p1, p2 = self.p_sz
        pd, dim, depth = (C // 2) * p1 * p2, vit.dim, len(vit.transformer.layers)
        L = _hip.lib()
        flat = self._flat_params()
        assert flat.numel() == L.cf_vit_flat_params(pd, dim, depth)
        st = _hip.stream()
        # the packed table follows the flat parameter tensor (rebuilt when a version counter moves): packed once per parameter version,
        # not once per call (`sample` walks eight such layers per call)
        hit = self.__dict__.get("_fused_ws")
        capturing = torch.cuda.is_current_stream_capturing()
        if hit is not None and hit[0] is flat and hit[1] == (pd, dim, depth, str(x.device)) and not capturing:
            ws = hit[2]
            torch.cuda.current_stream(x.device).wait_event(hit[3])
        else:
            ws = torch.empty(L.cf_vit_ws_bytes(pd, dim, depth), device=x.device, dtype=torch.uint8)
            _hip.call("cf_vit_prepare", _hip.p(flat), _hip.p(ws), pd, dim, depth, st)
            if not capturing:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(x.device))
                self.__dict__["_fused_ws"] = (flat, (pd, dim, depth, str(x.device)), ws, ev)
        if vit.pos_embedding.device != x.device:
            vit.pos_embedding = vit.pos_embedding.to(x.device).contiguous()
        z = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
        ldj = None if inverse else torch.empty(B, device=x.device, dtype=torch.float32)
        events = VIT_EVENTS
        if events is not None:               # bench.py: HIP events on the launch stream around exactly this kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(x.device))
        _hip.call("cf_vit_coupling", _hip.p(x), _hip.p(z), _hip.p(ldj), _hip.p(ws), _hip.p(vit.pos_embedding),
                  B, C, H, W, p1, p2, dim, depth, xbs, int(inverse), st)
        if events is not None:
            e1.record(torch.cuda.current_stream(x.device))
            events.append((e0, e1, B))
        return z, ldj

    def forward(self, x, context=None):
        _hip.require_device(x)
        if self.context_net:
            return self._forward_ctx(x, context)
        if self._fused_ok(x):
            return self._fused(x, False)
        return super().forward(x, context)

    def _forward_ctx(self, x, context, tape=None):
        h, logp_c = self._net_ctx(x[:, : x.shape[1] // 2], context, tape)
        if tape is not None:
            tape[-1]["x"] = x
        z, ldj = coupling_apply(x, h, False)
        return z, ldj + logp_c

    def reverse(self, z, context=None):
        _hip.require_device(z)
        if self.context_net:
            h, _ = self._net_ctx(z[:, : z.shape[1] // 2], context)
            return coupling_apply(z, h, True)[0]
        if self._fused_ok(z):
            return self._fused(z, True)[0]
        return super().reverse(z, context)
