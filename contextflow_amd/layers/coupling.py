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
from .flowlayer import FlowLayer, no_context
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
        no_context("Coupling", context_net)
        D, Hd, O = data_channels // 2, data_channels * 2, data_channels
        self.context_net = context_net
        self.contextflow = contextflow
        self.NN = nn.Sequential(
            nn.Conv2d(D, Hd, 1), nn.ReLU(),
            nn.Conv2d(Hd, Hd, kernel_size, padding=padding, padding_mode="reflect"), nn.ReLU(),
            nn.Conv2d(Hd, O, 1))

    def net(self, x0):
        h = conv2d_reflect(x0, self.NN[0], True)
        h = conv2d_reflect(h, self.NN[2], True)
        return conv2d_reflect(h, self.NN[4], False)


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
        no_context("TransCoupling", context_net)
        D, O = in_sz[0] // 2, in_sz[0]
        T = O * p_sz[0] * p_sz[1]                       # transformer width (coupling.py:108)
        self.context_net = context_net
        self.contextflow = contextflow
        self.NN = nn.Sequential(SimpleViT(image_size=(in_sz[1], in_sz[2]), patch_size=p_sz, dim=T, depth=6, heads=1,
                                          mlp_dim=T, channels=D))

    def net(self, x0):
        return self.NN[0](x0)
