"""Invertible 1x1 convolution (reference: contextflow/layers/conv1x1.py:9-77), context-free branch."""
import torch
import torch.nn as nn

from . import _hip
from .flowlayer import FlowLayer, no_context


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
        no_context("Conv1x1", context_net)
        D, H, W = data_size if len(data_size) == 3 else (data_size[0], 1, 1)
        self.D, self.H, self.W = D, H, W
        self.NN = nn.Parameter(torch.empty(D, D))          # same parameter name as the reference
        nn.init.orthogonal_(self.NN)
        self.context_net = context_net
        self.contextflow = contextflow

    def forward(self, x, context=None):
        _hip.require_device(x, self.NN)
        B, _, H, W = x.shape
        Wm = _hip.f32(self.NN.detach())
        z = conv1x1_apply(x, Wm)
        lad, _ = slogdet_inverse(Wm, False)
        return z, (lad * float(H * W)).expand(B)           # conv1x1.py:53

    def reverse(self, z, context=None):
        _hip.require_device(z, self.NN)
        _, inv = slogdet_inverse(_hip.f32(self.NN.detach()), True)
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
