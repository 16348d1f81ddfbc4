"""Context encoders of the specialist mode (reference: contextflow/model.py:30-90, layers/dequantize.py:26-143,
layers/rtdl/nn/_embeddings.py:76-285): a discrete context (B, n) of integer codes becomes the continuous vector c (B, C)
that the CN nets of Conv1x1 / ActNorm / Coupling consume, plus a log-density term.

Built: every combination `create_model` can produce (model.py:32-85) with an even code width: `enc_emb` eye | onehot
with `enc_type` uniform (uniform dequantisation of the code) or vardeq (variational dequantisation: the noise comes
from a small conditional flow over the context); eye + argmax (argmax surjection over binary codes); embed +
eyesample (embedding rows, also the priors' lookup) and embed + probsample (sigmoid of a flow sample).

Module and buffer names follow the reference so that checkpoints load: ContextEncoder = Sequential(emb, encoder);
OneHotEncoder.cardinalities, UniformCatDequantization.{qbins, ldj_per_dim}, CatEmbeddings._embeddings.N.weight."""
import torch
import torch.nn as nn

from . import _hip


class EyeEncoder(nn.Module):
    """rtdl/nn/_embeddings.py:76-109: the integer context itself."""

    def forward(self, x):
        if x.ndim != 2:
            raise ValueError("The input must have two dimensions")
        return x, x


class OneHotEncoder(nn.Module):
    """rtdl/nn/_embeddings.py:112-150: concatenated one-hot codes of the context variables (an index op)."""

    def __init__(self, cardinalities):
        super().__init__()
        self.register_buffer("cardinalities", torch.tensor(cardinalities))

    def forward(self, x):
        if x.ndim != 2:
            raise ValueError("The input must have two dimensions")
        cols = [torch.nn.functional.one_hot(x[:, i], int(k)) for i, k in enumerate(self.cardinalities.tolist())]
        return torch.cat(cols, 1), x


class CatEmbeddings(nn.Module):
    """rtdl/nn/_embeddings.py:153-285 (stack=False, bias=False): concatenated embedding-table rows."""

    def __init__(self, cardinalities, d_embedding, stack=False, bias=False, init="zeros"):
        super().__init__()
        if stack or bias:
            raise NotImplementedError("CatEmbeddings: stack / bias variants are not used by create_model")
        self._embeddings = nn.ModuleList([nn.Embedding(k, d_embedding) for k in cardinalities])
        for m in self._embeddings:
            if init == "zeros":
                nn.init.zeros_(m.weight)
            else:
                nn.init.uniform_(m.weight, -d_embedding ** -0.5, d_embedding ** -0.5)

    def forward(self, x):
        if x.ndim != 2 or x.shape[1] != len(self._embeddings):
            raise ValueError("x must be (batch, %d)" % len(self._embeddings))
        out = [m.weight.detach()[x[:, i]] for i, m in enumerate(self._embeddings)]      # row gather (index op)
        return torch.cat(out, 1), x


class UniformCatDequantization(nn.Module):
    """dequantize.py:26-70: z = (x + u) / K, u ~ U[0,1); ldj = sum_d(-log K_d * n_dims) for every sample.
    `fixed_noise` (tests) replaces the draw."""

    def __init__(self, num_cats=(1,)):
        super().__init__()
        self.D = len(num_cats)
        self.register_buffer("qbins", torch.tensor(num_cats, dtype=torch.float))
        self.register_buffer("ldj_per_dim", -torch.log(torch.tensor(num_cats, dtype=torch.float)))
        self.fixed_noise = None

    def forward(self, input):
        x, context = input                                   # x: integer code (B, D) - the context itself or its one-hot
        dev = self.qbins.device
        B, width = x.shape[0], self.D
        u = self.fixed_noise if self.fixed_noise is not None else torch.rand((B, width), device=dev, dtype=torch.float32)
        z = torch.empty(B, width, device=dev, dtype=torch.float32)
        code = x.to(device=dev, dtype=torch.int64).contiguous()
        _hip.call("cf_ctx_encode", _hip.p(code), _hip.p(_hip.f32(u)), _hip.p(self.qbins), None, _hip.p(z), B, width, width, 0,
                  _hip.stream())
        ldj = (self.ldj_per_dim * width).sum(-1).repeat(B)           # dequantize.py:62 (num_dims = width)
        return z, ldj

    def encode(self, context, card=None):
        """OneHotEncoder / EyeEncoder + this layer in one launch from the integer context (B, n): the one-hot code
        (card: the cardinalities, int64 on the device) or the context itself (card None) never exists as a tensor.
        Returns (z, ldj) as forward does - ldj as a stride-0 view of the constant (`const_logp` is its host value).
        An out-of-range code gives an all-zero one-hot block (torch's one_hot would raise)."""
        dev = self.qbins.device
        B, width = context.shape[0], self.D
        u = self.fixed_noise
        if u is None:
            u = self.__dict__.pop("_noise_once", None)          # a slice of one draw for all encoders of a forward (specialist.py)
            if u is None or u.shape != (B, width):
                u = torch.rand((B, width), device=dev, dtype=torch.float32)
        z = torch.empty(B, width, device=dev, dtype=torch.float32)
        code = context.to(device=dev, dtype=torch.int64).contiguous()
        _hip.call("cf_ctx_encode", _hip.p(code), _hip.p(_hip.f32(u)), _hip.p(self.qbins), _hip.p(card), _hip.p(z), B,
                  code.shape[1], width, 0 if card is None else 1, _hip.stream())
        lc = getattr(self, "_lc", None)
        if lc is None or lc.device != dev:
            lc = self._lc = (self.ldj_per_dim * width).sum(-1).reshape(1)
        return z, lc.expand(B)

    @property
    def const_logp(self):
        """Host value of the (sample-independent) log-density term, dequantize.py:62."""
        v = getattr(self, "_lc_host", None)
        if v is None:
            v = self._lc_host = float((self.ldj_per_dim.detach().cpu() * self.D).sum())
        return v

    def reverse(self, z, context=None):
        return (z * self.qbins).floor().clamp(min=0).minimum(self.qbins - 1).long()


class ConditionalGaussianDistribution(nn.Module):
    """gaussian.py:234-270: diagonal Gaussian whose mean / log-scale are looked up from the context."""

    def __init__(self, size, mixtures=1, context_net=None, contextflow=False):
        super().__init__()
        assert mixtures == 1, "mixtures should be 1 in GaussianDistribution"
        self.size, self.D, self.M = tuple(size), size[0], mixtures
        self.context_net = context_net
        self.contextflow = contextflow
        self.fixed_noise = None

    def sample(self, n_samples, context=None):
        c, _ = self.context_net(context)                     # (B, 2D) = [mean | log_scale]
        c = _hip.f32(c)
        D = self.D
        eps = self.fixed_noise if self.fixed_noise is not None else torch.randn(n_samples, D, device=c.device)
        self.last_eps = eps                                  # picked up by the caller's tape record (encoder_noise)
        x = torch.empty(n_samples, D, device=c.device, dtype=torch.float32)
        logp = torch.empty(n_samples, device=c.device, dtype=torch.float32)
        _hip.call("cf_cond_gauss_sample", _hip.p(c), _hip.p(_hip.f32(eps)), _hip.p(x), _hip.p(logp), n_samples, D,
                  _hip.stream())
        return x, logp


class VariationalCatDequantization(nn.Module):
    """dequantize.py:73-123: z = (x + sigmoid(u)) / K with u from the encoder flow;
    ldj = sum_d(-log K_d * n_dims) + ldj_sigmoid - log q(u)."""

    def __init__(self, encoder, num_cats=(1,)):
        super().__init__()
        self.D = len(num_cats)
        self.register_buffer("qbins", torch.tensor(num_cats, dtype=torch.float))
        self.register_buffer("ldj_per_dim", -torch.log(torch.tensor(num_cats, dtype=torch.float)))
        self.encoder = encoder
        self.sigmoid = _Sigmoid()

    def forward(self, input):
        x, context = input
        dev = self.qbins.device
        B, width = x.shape[0], self.D
        ctx = context.to(device=dev, dtype=torch.int64).contiguous()
        u, qu = self.encoder.sample(ctx, ctx)
        su = torch.empty_like(u)
        act_ldj = torch.empty(B, device=dev, dtype=torch.float32)
        _hip.call("cf_sigmoid_ldj", _hip.p(u), _hip.p(su), _hip.p(act_ldj), B, width, _hip.stream())
        z = torch.empty(B, width, device=dev, dtype=torch.float32)
        code = x.to(device=dev, dtype=torch.int64).contiguous()
        _hip.call("cf_ctx_encode", _hip.p(code), _hip.p(su), _hip.p(self.qbins), None, _hip.p(z), B, width, width, 0,
                  _hip.stream())
        ldj = (self.ldj_per_dim * width).sum(-1).repeat(B)
        return z, ldj + act_ldj - qu


class ArgmaxCatDequantization(nn.Module):
    """dequantize.py:170-270: z = sigmoid(u) * (2 bits(context) - 1) with u from the encoder flow; ldj = ldj_sigmoid - log q."""

    def __init__(self, encoder, num_cats=(1,)):
        super().__init__()
        self.encoder = encoder
        self.num_bits = self.cats2bits(list(num_cats))
        self.sigmoid = _Sigmoid()

    @staticmethod
    def cats2bits(num_cats):
        import math
        if isinstance(num_cats, (list, tuple)):
            return [int(math.ceil(math.log2(c))) for c in num_cats]
        return int(math.ceil(math.log2(num_cats)))

    def forward(self, input):
        x, context = input
        dev = self.sigmoid.temperature.device
        ctx = context.to(device=dev, dtype=torch.int64).contiguous()
        B = ctx.shape[0]
        u, qu = self.encoder.sample(ctx, ctx)
        width = u.shape[1]
        su = torch.empty_like(u)
        act_ldj = torch.empty(B, device=dev, dtype=torch.float32)
        _hip.call("cf_sigmoid_ldj", _hip.p(u), _hip.p(su), _hip.p(act_ldj), B, width, _hip.stream())
        bits = torch.tensor(self.num_bits, device=dev, dtype=torch.int64)
        z = torch.empty(B, width, device=dev, dtype=torch.float32)
        _hip.call("cf_ctx_encode", _hip.p(ctx), _hip.p(su), None, _hip.p(bits), _hip.p(z), B, ctx.shape[1], width, 2,
                  _hip.stream())
        return z, act_ldj - qu


class ProbSampling(nn.Module):
    """dequantize.py:145-167: the code is sigmoid(u), u from the encoder flow; ldj = ldj_sigmoid + log q (reference sign)."""

    def __init__(self, encoder):
        super().__init__()
        self.encoder = encoder
        self.sigmoid = _Sigmoid()

    def forward(self, input):
        x, context = input
        dev = self.sigmoid.temperature.device
        ctx = context.to(device=dev, dtype=torch.int64).contiguous()
        B = ctx.shape[0]
        u, qu = self.encoder.sample(ctx, ctx)
        su = torch.empty_like(u)
        act_ldj = torch.empty(B, device=dev, dtype=torch.float32)
        _hip.call("cf_sigmoid_ldj", _hip.p(u), _hip.p(su), _hip.p(act_ldj), B, u.shape[1], _hip.stream())
        return su, act_ldj + qu


class _Sigmoid(nn.Module):
    """Parameter holder of activations.Sigmoid (buffer `temperature` = 1); the arithmetic is cf_sigmoid_ldj."""

    def __init__(self):
        super().__init__()
        self.register_buffer("temperature", torch.Tensor([1.0]))


class EyeSampling(nn.Module):
    """dequantize.py:129-142: pass-through, zero log-density."""

    def forward(self, input):
        x, _ = input
        return x, torch.zeros(x.shape[0], device=x.device)

    def reverse(self, z, context=None):
        return z.long()


class ContextEncoder(nn.Sequential):
    """model.py:30-90."""

    def __init__(self, contexts, enc_emb, enc_type, data_size, init="orthogonal"):
        contexts = list(contexts)
        if enc_emb == "onehot":
            sz, emb, num_cats = sum(contexts), OneHotEncoder(contexts), sum(contexts) * [1]
        elif enc_emb == "eye" and enc_type == "argmax":
            sz = sum(ArgmaxCatDequantization.cats2bits(contexts))
            sz, emb, num_cats = sz + sz % 2, EyeEncoder(), contexts
        elif enc_emb == "eye":
            sz, emb, num_cats = len(contexts), EyeEncoder(), contexts
        elif enc_emb == "embed":
            sz, emb, num_cats = data_size[0] * len(contexts), CatEmbeddings(contexts, data_size[0], init=init), None
        else:
            raise NotImplementedError("contextflow_amd ContextEncoder: enc-emb=%s" % enc_emb)

        def enc_flow():                                                       # model.py:52-66
            from .actnorm import ActNormFC
            from .conv1x1 import FC
            from .coupling import CouplingFC
            from .flowsequential import FlowInvSequential
            if sz % 2:
                raise NotImplementedError("context encoder flow with an odd code width (the reference's Augment path)")
            layers = []
            for _ in range(2):
                layers += [FC((sz,)), ActNormFC((sz,)), CouplingFC(sz)]
            cnet = CatEmbeddings(contexts, 2 * sz // len(contexts), init="zeros")
            return FlowInvSequential(ConditionalGaussianDistribution(size=(sz,), context_net=cnet), *layers)
        if enc_type == "eyesample":
            encoder = EyeSampling()
        elif enc_type == "probsample":
            encoder = ProbSampling(enc_flow())
        elif enc_type == "uniform" and num_cats is not None:
            encoder = UniformCatDequantization(num_cats=num_cats)
        elif enc_type == "vardeq" and num_cats is not None:
            encoder = VariationalCatDequantization(enc_flow(), num_cats=num_cats)
        elif enc_type == "argmax" and num_cats is not None:
            encoder = ArgmaxCatDequantization(enc_flow(), num_cats=num_cats)
        else:
            raise NotImplementedError("contextflow_amd ContextEncoder: enc-emb=%s with enc-type=%s" % (enc_emb, enc_type))
        super().__init__(emb, encoder)
        self.C = sz
        self.contexts = contexts

    @property
    def const_logp(self):
        """The encoder's log-density when it is the same constant for every sample (uniform dequantisation), else None."""
        return self[1].const_logp if isinstance(self[1], UniformCatDequantization) else None

    def forward(self, input):
        emb, enc = self[0], self[1]
        if isinstance(enc, UniformCatDequantization) and input.ndim == 2:
            if isinstance(emb, OneHotEncoder):
                return enc.encode(input, emb.cardinalities)
            if isinstance(emb, EyeEncoder):
                return enc.encode(input)
        return super().forward(input)
