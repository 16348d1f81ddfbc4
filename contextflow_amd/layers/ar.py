"""MaskedCoupling (`--coupling maf`; reference: contextflow/layers/ar.py:15-69, layers/autoregressive/
masked_conv_2d.py:7-98, utils.py:27-91): an affine transform of ALL channels whose shift / log-scale come from a
masked residual block of three convolutions (autoregressive over channels at the centre tap, causal in space).

    h = conv3(relu(conv2(relu(conv1(relu(x)))))) + [x ; x]        t = h[:, :D],  log_s = 2 tanh(h[:, D:] / 2)
    z = x exp(log_s) + t,   ldj = sum log_s

As in the reference the masks multiply the weights IN PLACE on every forward and `reverse` is not implemented (upstream
returns zeros).  Training: autograd_layers.masked_coupling_backward (the plain convolution gradients, as torch.autograd
gives them in the reference for in-place masked weights).  Context-conditioned variants are not built: upstream's cannot run
(ar.py:23-30 - the 2D-input residual block adds a 4D-channel identity to a 2D-channel output, masked_conv_2d.py:93-98; the
contextflow branch adds a (B, C) code to a (B, 2D) output, masked_linear.py:121-129).
The convolutions run in the generic implicit-GEMM conv kernel (cf_conv2d_reflect, fp32 MFMA)."""
import torch
import torch.nn as nn

from . import _hip
from .coupling import conv2d_reflect, coupling_apply
from .flowlayer import FlowLayer, no_context


def mask_channels(mask_type, in_channels, out_channels, data_channels):
    """Channel mask of the centre tap: (out, in) tiling of a lower-triangular (data_channels x data_channels) block,
    strictly lower for type 'A' (utils.py:27-58)."""
    base = torch.ones(data_channels, data_channels).tril(-1 if mask_type == "A" else 0)
    reps_in, reps_out = in_channels // data_channels + 1, out_channels // data_channels + 1
    return base.repeat(reps_out, reps_in)[:out_channels, :in_channels]


def mask_conv2d(mask_type, in_channels, out_channels, height, width, data_channels):
    """utils.py:61-91: everything above the centre row and left of the centre in its row passes, the centre tap is
    channel-masked, the rest is cut."""
    mask = torch.ones(out_channels, in_channels, height, width)
    mask[:, :, height // 2, width // 2] = mask_channels(mask_type, in_channels, out_channels, data_channels)
    mask[:, :, height // 2, width // 2 + 1:] = 0
    mask[:, :, height // 2 + 1:] = 0
    return mask


class MaskedConv2d(nn.Conv2d):
    """Parameter + mask container (masked_conv_2d.py:7-24,46-79); the arithmetic is conv2d_reflect."""

    def __init__(self, *args, mask_type, data_channels=3, **kwargs):
        super().__init__(*args, **kwargs)
        o, i, h, w = self.weight.shape
        self.register_buffer("mask", mask_conv2d(mask_type, i, o, h, w, data_channels))

    def apply_mask_(self):
        with torch.no_grad():
            self.weight.mul_(self.mask)                      # masked_conv_2d.py:22: in place, every forward


class MaskedResidualBlock2d(nn.Module):
    def __init__(self, I, O, kernel_size=(1, 1), padding=(0, 0), D=0, mask_type="B"):
        super().__init__()
        self.conv1 = MaskedConv2d(I, 2 * I, 1, mask_type=mask_type, data_channels=D)
        self.conv2 = MaskedConv2d(2 * I, 2 * I, kernel_size, padding=padding, padding_mode="reflect", mask_type=mask_type,
                                  data_channels=D)
        self.conv3 = MaskedConv2d(2 * I, 2 * O, 1, mask_type=mask_type, data_channels=D)

    def forward(self, x):
        for c in (self.conv1, self.conv2, self.conv3):
            c.apply_mask_()
        x = _hip.f32(x)
        B, C, H, W = x.shape
        a0 = torch.empty_like(x)
        _hip.call("cf_relu_bwd", _hip.p(x), _hip.p(x), _hip.p(a0), x.numel(), _hip.stream())     # relu(x) = x * (x > 0)
        h = conv2d_reflect(a0, self.conv1, True)             # relu(conv1(relu x)): the ReLU in front of conv2
        h = conv2d_reflect(h, self.conv2, True)
        h = conv2d_reflect(h, self.conv3, False)
        _hip.call("cf_add_repeat", _hip.p(h), _hip.p(x), B, h.shape[1], C, H * W, _hip.stream())   # + [x ; x]
        return h


class MaskedCoupling(FlowLayer):
    def __init__(self, data_channels, kernel_size=(1, 1), padding=(0, 0), context_net=None, contextflow=False, mask_type="B"):
        super().__init__()
        no_context("MaskedCoupling", context_net)
        D = data_channels
        self.context_net, self.contextflow = context_net, contextflow
        self.NN = MaskedResidualBlock2d(D, D, kernel_size=kernel_size, padding=padding, D=D, mask_type=mask_type)

    def forward(self, x, context=None):
        _hip.require_device(x)
        x = _hip.f32(x)
        h = self.NN(x)                                       # (B, 2D, H, W) = [t ; raw]
        # the affine-map kernel transforms the SECOND channel half of its input: feed it [0 ; x]
        xx = torch.cat([torch.zeros_like(x), x], dim=1)
        z2, ldj = coupling_apply(xx, h, False)
        return z2[:, x.shape[1]:].contiguous(), ldj

    def reverse(self, z, context=None):
        raise NotImplementedError("MaskedCoupling.reverse: not implemented upstream either (ar.py:59-66 returns zeros)")

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]
