"""Flow activation layers (reference: contextflow/layers/activations.py).

`FlowActivationLayer` is the (trivial) base class ActNorm derives from upstream; `SplineActivation`
(activations.py:120-211) is the elementwise rational-quadratic spline with linear tails — disabled in every
shipped config (model.py:137) but part of the `layers` API and named by the hot-path description."""
import torch
import torch.nn as nn

from . import _hip
from .flowlayer import FlowLayer


class FlowActivationLayer(FlowLayer):
    def forward(self, input, context=None):
        return self.activation(input, context), self.logdet(input, context)

    def activation(self, input, context=None):
        raise NotImplementedError

    def act_prime(self, input, context=None):
        raise NotImplementedError

    def reverse(self, input, context=None):
        raise NotImplementedError

    def logdet(self, input, context=None):
        return torch.log(torch.abs(self.act_prime(input, context))).flatten(start_dim=1).sum(dim=-1)


class SplineActivation(FlowActivationLayer):
    def __init__(self, input_size, n_bins=5, tail_bound=10., individual_weights=False):
        super().__init__()
        self.n_bins = n_bins
        self.tail_bound = tail_bound
        self.individual_weights = individual_weights
        shape = (1, *input_size) if individual_weights else ()
        self.unnormalized_widths = nn.Parameter(torch.randn(*shape, n_bins) * 0.01)
        self.unnormalized_heights = nn.Parameter(torch.randn(*shape, n_bins) * 0.01)
        self.unnormalized_derivatives = nn.Parameter(torch.randn(*shape, n_bins - 1) * 0.01)

    def _run(self, input, inverse):
        _hip.require_device(input, self.unnormalized_widths)
        x = _hip.f32(input)
        B = x.shape[0]
        N = x.numel() // max(B, 1) if B else 1
        K = self.n_bins
        P = self.unnormalized_widths.numel() // K
        if P != 1 and P != N:
            raise RuntimeError("SplineActivation: parameter shape %s does not match input %s" %
                               (tuple(self.unnormalized_widths.shape), tuple(x.shape)))
        L = _hip.lib()
        table = torch.empty(L.cf_spline_table_floats(P, K), device=x.device, dtype=torch.float32)
        st = _hip.stream()
        _hip.call("cf_spline_prepare", _hip.p(_hip.f32(self.unnormalized_widths.detach())),
                  _hip.p(_hip.f32(self.unnormalized_heights.detach())), _hip.p(_hip.f32(self.unnormalized_derivatives.detach())),
                  _hip.p(table), P, K, float(self.tail_bound), st)
        y = torch.empty_like(x)
        ldj = None if inverse else torch.empty(B, device=x.device, dtype=torch.float32)
        _hip.call("cf_spline", _hip.p(x), _hip.p(table), _hip.p(y), _hip.p(ldj), B, N, P, K, float(self.tail_bound),
                  int(inverse), st)
        return y, ldj

    def forward(self, input, context=None):
        return self._run(input, False)

    def activation_and_logdet(self, input, context=None):
        return self._run(input, False)

    def reverse(self, input, context=None):
        return self._run(input, True)[0]

    def logdet(self, input, context=None):
        return self._run(input, False)[1]
