"""Flow activation layers (reference: contextflow/layers/activations.py).

`FlowActivationLayer` is the (trivial) base class ActNorm derives from upstream; `SplineActivation`
(activations.py:120-211) is the elementwise rational-quadratic spline with linear tails — disabled in every
shipped config (model.py:137) but part of the `layers` API and named by the hot-path description.  The simple
elementwise activations (activations.py:34-118, 213-245) run through one kernel, `cf_activation`."""
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


class _Elementwise(FlowActivationLayer):
    """y = f(x) elementwise with ldj = sum over every non-batch element of log|f'(x)| (activations.py:13-22)."""
    _mode = 0

    def _params(self):
        return 0.0, 0.0, None

    def _run(self, input, inverse, per_last_dim=False):
        _hip.require_device(input)
        x = _hip.f32(input).contiguous()
        if x.dim() < 2:
            raise ValueError("activation layers take (batch, ...) tensors")
        rows = x.shape[0] if not per_last_dim else x.numel() // max(x.shape[-1], 1)
        D = x.numel() // rows if rows else 1
        a, b, ptr = self._params()
        y = torch.empty_like(x)
        ldj = None if inverse else torch.empty(rows, device=x.device, dtype=torch.float32)
        if rows:
            _hip.call("cf_activation", _hip.p(x), _hip.p(y), _hip.p(ldj), rows, D, self._mode, float(a), float(b), _hip.p(ptr),
                      int(inverse), _hip.stream())
        return y, ldj

    def forward(self, input, context=None):
        return self._run(input, False)

    def activation(self, input, context=None):
        return self._run(input, False)[0]

    def logdet(self, input, context=None):
        return self._run(input, False)[1]

    def reverse(self, input, context=None):
        return self._run(input, True)[0]


class Identity(_Elementwise):
    _mode = 0


class LeakyRelu(_Elementwise):
    _mode = 1

    def __init__(self, alpha=0.1):
        super().__init__()
        self.alpha = alpha

    def _params(self):
        return self.alpha, 0.0, None


class SmoothLeakyRelu(_Elementwise):
    _mode = 2

    def __init__(self, alpha=0.3):
        super().__init__()
        self.alpha = alpha

    def _params(self):
        return self.alpha, 0.0, None


class SmoothTanh(_Elementwise):
    _mode = 3

    def __init__(self, alpha=1.0, beta=0.1):
        super().__init__()
        self.alpha, self.beta = alpha, beta

    def _params(self):
        return self.alpha, self.beta, None


class LearnableLeakyRelu(_Elementwise):
    """activations.py:78-100: slope sigmoid(alpha_logit) + 0.5 on the negative side (evaluation; no backward here)."""
    _mode = 5

    def __init__(self):
        super().__init__()
        self.alpha_logit = nn.Parameter(torch.zeros([1]))

    def get_alpha(self):
        return torch.sigmoid(self.alpha_logit.detach()) + .5

    def _params(self):
        return 0.0, 0.0, _hip.f32(self.alpha_logit.detach())


class Sigmoid(_Elementwise):
    """activations.py:227-245: z = sigmoid(T x); the reference sums the log-det over the LAST dim only."""
    _mode = 4

    def __init__(self, temperature=1, eps=0.0):
        super().__init__()
        self.eps = eps
        self._t = float(temperature)
        self.register_buffer("temperature", torch.Tensor([temperature]))

    def _params(self):
        return self._t, self.eps, None

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        super()._load_from_state_dict(state_dict, prefix, *a, **k)
        self._t = float(self.temperature.detach().cpu()[0])

    def forward(self, x, context=None):
        z, ldj = self._run(x, False, per_last_dim=True)
        return z, ldj.view(x.shape[:-1])

    def logdet(self, x, context=None):
        return self.forward(x)[1]

    def reverse(self, z, context=None):
        if z.numel():
            lo, hi = torch.aminmax(z)
            assert float(lo) >= 0 and float(hi) <= 1, "input must be in [0,1]"
        return self._run(z, True, per_last_dim=True)[0]
