"""Invertible 1x1 convolution (reference: contextflow/layers/conv1x1.py:9-77): the shared matrix of the generalist
and the per-sample triangular matrix CN(c) of the specialist mode (conv1x1.py:34-50)."""
import torch
import torch.nn as nn

from . import _hip
from .flowlayer import FlowLayer, encoder_noise


def slogdet_inverse(W, want_inverse):
    """Device-side log|det W| (and W^-1): one tiny fp64 Gauss-Jordan workgroup, no host sync."""
    C = W.shape[0]
    lad = torch.empty(1, device=W.device, dtype=torch.float32)
    inv = torch.empty(C, C, device=W.device, dtype=torch.float32) if want_inverse else None
    _hip.call("cf_slogdet_inverse", _hip.p(W), C, _hip.p(lad), _hip.p(inv), _hip.stream())
    return lad, inv


def conv1x1_apply(x, W, bias=None):
    x, xbs = _hip.bview(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(B * C, 1) if B else 1
    out = torch.empty((B,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    _hip.call("cf_conv1x1_fwd", _hip.p(x), _hip.p(W), _hip.p(bias), _hip.p(out), B, C, HW, xbs, C * HW, _hip.stream())
    return out


class Conv1x1(FlowLayer):
    def __init__(self, data_size, context_net=None, contextflow=False):
        super().__init__()
        D, H, W = data_size if len(data_size) == 3 else (data_size[0], 1, 1)
        self.D, self.H, self.W = D, H, W
        self.NN = nn.Parameter(torch.empty(D, D))          # same parameter name as the reference
        nn.init.orthogonal_(self.NN)
        self.context_net = context_net
        self.contextflow = contextflow
        if self.context_net:                               # conv1x1.py:20-26
            self.C = self.context_net.C
            self.CN = nn.Linear(self.C, D * D)
            nn.init.zeros_(self.CN.weight)
            nn.init.zeros_(self.CN.bias)
            if self.contextflow:
                self.NN.requires_grad_(False)

    def _forward_ctx(self, x, context, tape=None, pre=None):
        """conv1x1.py:34-50: per-sample triangular matrix from CN(c).  `tape` (training): receives what the backward
        needs - the encoder is stochastic, so its output must be kept, not recomputed.  pre: the code, its log-density and CN(c)
        already formed by the grouped front end (layers/specialist.py::_front_end, train form)."""
        from .simple_vit import _linear
        x, xbs = _hip.bview(x)
        B, C, H, W = x.shape
        if pre is not None:
            c, logp_c, m = pre["c"], pre["logp"], pre["m"]
        else:
            c, logp_c = self.context_net(context)
            m = _linear(_hip.f32(c), self.CN)              # (B, C*C)
        Wm = _hip.f32(self.NN.detach()) if self.contextflow else None
        z = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
        ldj = torch.empty(B, device=x.device, dtype=torch.float32)
        _hip.call("cf_conv1x1_ctx", _hip.p(x), _hip.p(m), _hip.p(Wm), _hip.p(z), _hip.p(ldj), B, C, H * W, xbs, _hip.stream())
        if self.contextflow:
            # H W log|det NN| of the frozen shared matrix: kept while NN is unchanged (one factorisation per layer and call otherwise)
            key = (self.NN._version, self.NN.data_ptr(), H * W, str(x.device))
            hit = self.__dict__.get("_lad_cache")
            if hit is None or hit[0] != key or torch.cuda.is_current_stream_capturing():
                lad, _ = slogdet_inverse(Wm, False)
                hit = (key, lad * float(H * W))
                if not torch.cuda.is_current_stream_capturing():
                    self.__dict__["_lad_cache"] = hit
            ldj = ldj + hit[1]
        if tape is not None:
            tape.append(dict(x=x, c=_hip.f32(c), m=m, eps=encoder_noise(self.context_net)))
        return z, ldj + logp_c * float(H * W)

    def forward(self, x, context=None):
        _hip.require_device(x, self.NN)
        if self.context_net:
            return self._forward_ctx(x, context)
        B, _, H, W = x.shape
        Wm = _hip.f32(self.NN.detach())
        z = conv1x1_apply(x, Wm)
        lad, _ = slogdet_inverse(Wm, False)
        return z, (lad * float(H * W)).expand(B)           # conv1x1.py:53

    def reverse(self, z, context=None):
        _hip.require_device(z, self.NN)
        if self.context_net:
            raise NotImplementedError("Conv1x1.reverse with a context net (the reference's own is marked 'to update')")
        # W^-1 follows NN's version counter and storage (`sample` inverts every layer's matrix per call otherwise)
        key = (self.NN._version, self.NN.data_ptr(), str(z.device))
        hit = self.__dict__.get("_winv_cache")
        capturing = torch.cuda.is_current_stream_capturing()
        if hit is None or hit[0] != key or capturing:
            _, inv = slogdet_inverse(_hip.f32(self.NN.detach()), True)
            if not capturing:
                self.__dict__["_winv_cache"] = (key, inv)
        else:
            inv = hit[1]
        return conv1x1_apply(z, inv)                       # conv1x1.py:72

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]


class FC(Conv1x1):
    """Flat variant (conv1x1.py:80-96)."""

    def __init__(self, data_size, context_net=None, contextflow=False):
        super().__init__(data_size, context_net=None, contextflow=False)

    def forward(self, x, context=None):
        out, ldj = super().forward(x.view(-1, self.D, 1, 1), context)
        return out.view(-1, self.D), ldj

    def reverse(self, z, context=None):
        return super().reverse(z.view(-1, self.D, 1, 1), context).view(-1, self.D)

    def logdet(self, x, context=None):
        return super().logdet(x.view(-1, self.D, 1, 1))
