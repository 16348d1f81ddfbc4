"""Base densities on the hot path (reference: contextflow/layers/distributions/gaussian.py).

StandardNormal (:10-72) is the Augment noise model; GaussianMixtureDistribution (:118-169) is the
class-conditional prior used by SplitPrior and as the final base density.  The reference broadcasts
(B, M, K, D, H, W); here one register-tiled kernel streams x once per 80 components."""
import math

import torch
import torch.nn as nn

from .. import _hip
from ..flowlayer import no_context


class StandardNormal(nn.Module):
    def __init__(self, size, mixtures=1, context_net=None, contextflow=False):
        super().__init__()
        assert mixtures == 1, "mixtures should be 1 in StandardNormal"
        no_context("StandardNormal", context_net)
        self.size = torch.Size(size)
        self.M = mixtures
        self.K = 8
        self.register_buffer("buffer", torch.zeros(1))
        self.context_net = context_net
        self.contextflow = contextflow
        self.fixed_noise = None      # tests inject the reference's captured noise here

    def forward(self, input, context=None):
        return self.log_prob(input, context)

    def log_prob(self, input, context=None):
        """(B,1): sum(-0.5 log 2pi - 0.5 x^2)   gaussian.py:50-54."""
        _hip.require_device(input)
        x, xbs = _hip.bview(input)
        B = x.shape[0]
        N = x.numel() // max(B, 1) if B else 1
        nll = torch.empty(B, device=x.device, dtype=torch.float32)
        _hip.call("cf_std_normal_nll", _hip.p(x), _hip.p(nll), B, N, xbs, _hip.stream())
        return -nll.unsqueeze(-1)

    def sample(self, n_samples, context=None):
        dev = self.buffer.device
        if self.fixed_noise is not None:
            x = self.fixed_noise.to(dev)
        else:
            x = torch.randn(n_samples, *self.size, device=dev, dtype=self.buffer.dtype)
        return x, self.log_prob(x, context)


def gmm_prepare(mG, sG, wG):
    """Per-call parameter transform on the device: a = 1/softplus(sG), nm = -mG, cst (M,K)."""
    M, K = wG.shape
    D = mG.numel() // (M * K)
    dev = mG.device
    a = torch.empty(M * K, D, device=dev, dtype=torch.float32)
    nm = torch.empty(M * K, D, device=dev, dtype=torch.float32)
    cst = torch.empty(M * K, device=dev, dtype=torch.float32)
    _hip.call("cf_gmm_prepare", _hip.p(_hip.f32(mG)), _hip.p(_hip.f32(sG)), _hip.p(_hip.f32(wG)),
              _hip.p(a), _hip.p(nm), _hip.p(cst), M, K, D, _hip.stream())
    return a, nm, cst, M, K, D


def gmm_logprob(x, prepared, out=None, accumulate=False):
    a, nm, cst, M, K, D = prepared
    x, xbs = _hip.bview(x)
    B = x.shape[0]
    assert x.numel() // max(B, 1) == D or B == 0
    if out is None:
        out = torch.empty(B, M, device=x.device, dtype=torch.float32)
        accumulate = False
    nbytes = _hip.lib().cf_gmm_ws_bytes(B, M, K, D)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    _hip.call("cf_gmm_logprob", _hip.p(x), _hip.p(a), _hip.p(nm), _hip.p(cst), _hip.p(out), _hip.p(ws),
              B, M, K, D, xbs, int(accumulate), _hip.stream())
    return out


GMM_LEVELS_MAX_BATCH = 1024       # below: all mixtures of a flow in one launch pair (the launches are what they cost there)


def gmm_levels_ok(levels):
    """levels: [(x, prepared)].  True when cf_gmm_logprob_levels takes them as they are (no copies made here)."""
    if not 1 <= len(levels) <= 4:
        return False
    M, K = levels[0][1][3], levels[0][1][4]
    if M * K > 256 or not ((16 % K == 0 or M == 1) if M * K <= 16 else 80 % K == 0):
        return False
    for x, (a, nm, cst, m, k, D) in levels:
        xv, xbs = _hip.bview(x)
        if (m, k) != (M, K) or xv is not x or D % 4 or xbs % 4 or (x.data_ptr() | a.data_ptr() | nm.data_ptr()) & 15:
            return False
    return True


def gmm_logprob_levels(levels, ldM=None, ld1=None):
    """logp (B, M) = (ldM +) sum over the levels' mixture log-probs (+ ld1[:, None]) in one launch pair; equals the chain of
    gmm_logprob(..., accumulate=True) + cf_logdet_combine bit for bit.  levels as for gmm_levels_ok (which must hold)."""
    import ctypes
    n = len(levels)
    a0, _, _, M, K, _ = levels[0][1]
    B, dev = levels[0][0].shape[0], levels[0][0].device
    parr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    Ds = (ctypes.c_int * n)(*[lv[1][5] for lv in levels])
    xbs = (ctypes.c_int64 * n)(*[_hip.bview(lv[0])[1] for lv in levels])
    L = _hip.lib()
    ws = torch.empty(max(L.cf_gmm_levels_ws_bytes(n, Ds, B, M, K), 1), device=dev, dtype=torch.uint8)
    out = torch.empty(B, M, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _hip.check(L.cf_gmm_logprob_levels(n, parr([lv[0] for lv in levels]), parr([lv[1][0] for lv in levels]),
                                           parr([lv[1][1] for lv in levels]), parr([lv[1][2] for lv in levels]), Ds, xbs,
                                           _hip.p(ldM), _hip.p(ld1), _hip.p(out), _hip.p(ws), B, M, K, _hip.stream(dev)),
                   "cf_gmm_logprob_levels")
    return out


KEYED_MIN_PER_KEY = 16             # cf_gmm_logprob_keyed from 16 samples per (mean key, scale key) value on average
_bucket_cache = [None, None]
_key_cache = []                    # [(context tensor, signature, key)]: GaussianMixtureDistribution._scale_tables


def _key_buckets(context, key_s, key_m, Us, Um, tag=None, TB=128):
    """Samples grouped by (scale key, mean key) for cf_gmm_logprob_keyed, all on the device (no host sync):
    order (B) int32 = sample indices sorted by key, tiles (T, 4) int32 = [ks, km, first position in order, count] with
    T = ceil(B / TB) + Us Um (an upper bound; unused tiles have count 0).  The last result is kept with the context tensor
    it was made from (the object itself, so its storage cannot be recycled under the cache; in-place writes bump its
    version): the mixtures of all levels of a flow see the same context."""
    sig = (context._version, Us, Um, tag)       # tag: what the keys were formed from (equal tag => equal keys)
    if _bucket_cache[0] is not None and _bucket_cache[0][0] is context and _bucket_cache[0][1] == sig:
        return _bucket_cache[1]
    B, U, dev = key_s.shape[0], Us * Um, key_s.device
    kc = key_s.long() * Um + key_m.long()
    order = torch.argsort(kc).to(torch.int32)
    counts = torch.bincount(kc, minlength=U)
    per = (counts + TB - 1) // TB
    tile_end = torch.cumsum(per, 0)
    row_start = torch.cumsum(counts, 0) - counts
    T = (B + TB - 1) // TB + U
    t = torch.arange(T, device=dev)
    u = torch.searchsorted(tile_end, t, right=True)
    uc = u.clamp(max=U - 1)
    j = t - (tile_end[uc] - per[uc])
    n = torch.where(u < U, (counts[uc] - j * TB).clamp(max=TB), torch.zeros_like(j))
    tiles = torch.stack([uc // Um, uc % Um, row_start[uc] + j * TB, n], 1).to(torch.int32).contiguous()
    _bucket_cache[0], _bucket_cache[1] = (context, sig), (order, tiles)
    return order, tiles


class GaussianDistribution(nn.Module):
    """gaussian.py:75-115: diagonal Gaussian with one (frozen) mean / pre-softplus scale per channel; log_prob (B,) sums
    over (C, H, W).  Evaluated by the mixture kernel with M = K = 1 (the parameters broadcast over the pixels)."""

    def __init__(self, size, mixtures=1, context_net=None, contextflow=False):
        super().__init__()
        assert mixtures == 1, "mixtures should be 1 in GaussianDistribution"
        self.size = size
        self.D = D = size[0]
        self.M = mixtures
        self.m = nn.Parameter(torch.zeros(D, 1, 1), requires_grad=False)
        self.s = nn.Parameter(torch.ones(D, 1, 1), requires_grad=False)
        self.context_net = context_net
        self.contextflow = contextflow

    def forward(self, input, context=None):
        return self.log_prob(input, context)

    def log_prob(self, input, context=None, sum=True):
        _hip.require_device(input, self.m)
        B, C, H, W = input.shape
        mG = _hip.f32(self.m.detach()).expand(C, H, W).contiguous().view(1, 1, C, H, W)
        sG = _hip.f32(self.s.detach()).expand(C, H, W).contiguous().view(1, 1, C, H, W)
        wG = torch.zeros(1, 1, device=input.device, dtype=torch.float32)
        return gmm_logprob(input, gmm_prepare(mG, sG, wG)).view(B)

    def sample(self, n_samples, context=None):
        eps = torch.randn(n_samples, *self.size, device=self.m.device, dtype=torch.float32)
        x = _hip.f32(self.m.detach()) + torch.nn.functional.softplus(_hip.f32(self.s.detach())) * eps
        return x, self.log_prob(x, context)


class GaussianMixtureDistribution(nn.Module):
    def __init__(self, size, mixtures=2, components=8, context_net=None, contextflow=False):
        super().__init__()
        self.size = size
        D, H, W = size
        self.D = D
        self.M = M = mixtures
        self.K = K = components
        self.mG = nn.Parameter(torch.randn(M, K, D, H, W))
        self.sG = nn.Parameter(torch.ones(M, K, D, H, W))
        self.wG = nn.Parameter(torch.randn(M, K))
        self.context_net = context_net
        self.contextflow = contextflow
        if self.context_net:                                # gaussian.py:130-137
            self.C = self.context_net.C
            if self.contextflow:
                for p in (self.mG, self.sG, self.wG):
                    p.requires_grad_(False)

    def prepared(self):
        return gmm_prepare(self.mG.detach(), self.sG.detach(), self.wG.detach())

    def _scale_tables(self, context, need_dsig):
        """Table form of the context shifts when the context net is the embedding lookup of model.py:157,162 (CatEmbeddings +
        EyeSampling): the concatenated embedding row is [mean shifts | pre-softplus scale shifts], so each half depends
        only on the context variables whose embedding columns fall in it, i.e. on a key with few values.  Returns
        (key (B) int32, inv_sig, dsig | None, lsum, ckey (B) int32, cm_tab (Um, M*K*D)) or None when the context net is
        anything else (per-sample kernel then)."""
        from ..context import CatEmbeddings, EyeSampling
        cn = self.context_net
        if not (len(cn) == 2 and isinstance(cn[0], CatEmbeddings) and isinstance(cn[1], EyeSampling)):
            return None
        embs = list(cn[0]._embeddings)
        D, H, W = self.size
        half, dE = self.M * self.K * D, embs[0].embedding_dim
        if dE * len(embs) != 2 * half:
            return None

        def keyed(rel):                       # mixed-radix key over the relevant variables, and the grid of all its values
            U, strides = 1, {}
            for i in reversed(rel):
                strides[i] = U
                U *= embs[i].num_embeddings
            return U, strides
        rel_s = [i for i in range(len(embs)) if (i + 1) * dE > half]        # variables that reach the scale half
        rel_m = [i for i in range(len(embs)) if i * dE < half]              # ... the mean half
        Us, st_s = keyed(rel_s)
        Um, st_m = keyed(rel_m)
        if Us * half * H * W * 4 > (256 << 20) or Um * half * 4 > (256 << 20):
            return None
        dev = self.sG.device
        # (storage, version) of every tensor the tables are built from; writes through `.data` bump no version counter:
        # callers that do that (none in this package) must drop `_tab_cache` themselves (FlowSequential.invalidate_caches)
        ver = (need_dsig, self.sG._version, self.sG.data_ptr(), str(dev)) + tuple((e.weight._version, e.weight.data_ptr()) for e in embs)
        cache = getattr(self, "_tab_cache", None)
        if cache is None or cache[0] != ver:
            def rows(U, rel, strides):
                grid = torch.zeros(U, len(embs), dtype=torch.long, device=dev)
                u = torch.arange(U, device=dev)
                for i in rel:
                    grid[:, i] = (u // strides[i]) % embs[i].num_embeddings
                return cn[0](grid)[0]
            cs_tab = _hip.f32(rows(Us, rel_s, st_s)[:, half:])
            cm_tab = _hip.f32(rows(Um, rel_m, st_m)[:, :half])
            inv = torch.empty(Us, half * H * W, device=dev, dtype=torch.float32)
            dsig = torch.empty_like(inv) if need_dsig else None
            lsum = torch.empty(Us, self.M * self.K, device=dev, dtype=torch.float32)
            _hip.call("cf_gmm_ctx_tables", _hip.p(_hip.f32(self.sG.detach())), _hip.p(cs_tab), _hip.p(inv), _hip.p(dsig),
                      _hip.p(lsum), Us, self.M * self.K, D, H * W, _hip.stream())
            cache = self._tab_cache = (ver, inv, dsig, lsum, cm_tab)
        ctx = context.to(device=dev, dtype=torch.long)

        def key_of(rel, strides):
            # the mixtures of all levels of a flow see the same context and, with model.py's embedding widths, form the same keys:
            # kept with the context tensor they were made from (the object itself: its storage cannot be recycled under the cache;
            # in-place writes bump its version) - 11 tiny launches per mixture and call otherwise
            sig = (context._version, tuple(rel), tuple(sorted(strides.items())), tuple(embs[i].num_embeddings for i in rel), str(dev))
            for ent in _key_cache:
                if ent[0] is context and ent[1] == sig:
                    return ent[2]
            key = torch.zeros(ctx.shape[0], dtype=torch.long, device=dev)
            for i in rel:        # clamped: an out-of-range code must not index past the tables (the lookup would raise)
                key += ctx[:, i].clamp(0, embs[i].num_embeddings - 1) * strides[i]
            key = key.to(torch.int32)
            _key_cache.append((context, sig, key))
            del _key_cache[:-4]               # the two keys of the current context (+ those of the one before)
            return key
        return (key_of(rel_s, st_s),) + cache[1:4] + (key_of(rel_m, st_m), cache[4])

    def _keyed_tables(self, tab, logw):
        """Per-key parameter rows of cf_gmm_logprob_keyed, cached with the scale tables: nm (Um, M*K*N) = -(mG + mean shift)
        - the reference's `self.mG + cond_mean` rounded once (gaussian.py:143) - and cst (Us, M*K)."""
        cache = self._tab_cache
        if len(cache) == 5 or cache[5][0] is not tab[1]:
            D, H, W = self.size
            MK = self.M * self.K
            nm = -(_hip.f32(self.mG.detach()).view(1, MK, D, H * W) + tab[5].view(-1, MK, D, 1))
            cst = logw.reshape(1, MK) - tab[3] - 0.5 * D * H * W * math.log(2 * math.pi)
            cache = self._tab_cache = cache[:5] + ((tab[1], nm.reshape(nm.shape[0], -1).contiguous(), cst.contiguous()),)
        return cache[5][1:]

    def _log_prob_ctx(self, input, context, tape=None):
        """gaussian.py:146-158: per-sample shifts (B, 2, M, K, D) of the component means / pre-softplus scales."""
        if isinstance(context, list):
            context = context[0]
        x, xbs = _hip.bview(input)
        B, D, H, W = x.shape
        M, K = self.M, self.K
        logw = torch.log_softmax(_hip.f32(self.wG.detach()), dim=-1).contiguous()
        out = torch.empty(B, M, device=x.device, dtype=torch.float32)
        tab = self._scale_tables(context, tape is not None)
        if tab is not None and tape is None:
            # evaluation with the embedding-lookup context net: both halves of the shift come from tables indexed by a
            # key per sample - the (B, 2 M K D) rows of embeddings are never gathered (logp_c of the lookup is 0)
            key, inv, _, lsum, ckey, cm_tab = tab
            Um, Us = cm_tab.shape[0], inv.shape[0]
            N = D * H * W
            if (B >= KEYED_MIN_PER_KEY * Um * Us and 16 < M * K <= 256 and 80 % K == 0 and N % 4 == 0 and xbs % 4 == 0
                    and x.data_ptr() % 16 == 0):
                # saturating batches: samples bucketed by (scale key, mean key) - a workgroup's 128 samples then share their
                # parameter rows and the register-tiled mixture kernel applies (cf_gmm_logprob_keyed)
                nm_tab, cst_tab = self._keyed_tables(tab, logw)
                order, tiles = _key_buckets(context, key, ckey, Us, Um, tuple(e.num_embeddings for e in self.context_net[0]._embeddings))
                T = tiles.shape[0]
                ws = torch.empty(_hip.lib().cf_gmm_keyed_ws_bytes(T, B, M, K, N), device=x.device, dtype=torch.uint8)
                _hip.call("cf_gmm_logprob_keyed", _hip.p(x), _hip.p(inv), _hip.p(nm_tab), _hip.p(cst_tab), _hip.p(key),
                          _hip.p(tiles), _hip.p(order), _hip.p(out), _hip.p(ws), T, B, M, K, N, xbs, 0, _hip.stream())
                return out
            _hip.call("cf_gmm_ctx_logprob_tab", _hip.p(x), _hip.p(_hip.f32(self.mG.detach())), _hip.p(inv), _hip.p(lsum),
                      _hip.p(logw), _hip.p(cm_tab), _hip.p(ckey), _hip.p(key), _hip.p(out), None, B, M, K, D, H * W, xbs, 0,
                      _hip.stream())
            return out
        c, logp_c = self.context_net(context)
        # training: keep the per-component log-joints for the backward (it then skips their recompute)
        lp = torch.empty(B, M * K, device=x.device, dtype=torch.float32) if tape is not None else None
        if tab is not None:
            key, inv, dsig, lsum = tab[:4]
            _hip.call("cf_gmm_ctx_logprob_tab", _hip.p(x), _hip.p(_hip.f32(self.mG.detach())), _hip.p(inv), _hip.p(lsum),
                      _hip.p(logw), _hip.p(_hip.f32(c)), None, _hip.p(key), _hip.p(out), _hip.p(lp), B, M, K, D, H * W, xbs, 0,
                      _hip.stream())
            tab = tab[:4]
        else:
            _hip.call("cf_gmm_ctx_logprob", _hip.p(x), _hip.p(_hip.f32(self.mG.detach())), _hip.p(_hip.f32(self.sG.detach())),
                      _hip.p(logw), _hip.p(_hip.f32(c)), _hip.p(out), _hip.p(lp), B, M, K, D, H * W, xbs, 0, _hip.stream())
        if tape is not None:
            tape.append(dict(x=x, c=_hip.f32(c), logw=logw, context=context, lp=lp, tab=tab))
        return out + (logp_c * float(H * W)).unsqueeze(-1)

    def log_prob(self, input, context=None):
        """(B, M) class-mixture log-densities   gaussian.py:142-161."""
        _hip.require_device(input, self.mG)
        if self.context_net:
            return self._log_prob_ctx(input, context)
        return gmm_logprob(input, self.prepared())

    def sample(self, n_samples, context=None, need_log_prob=True):
        """gaussian.py:163-169: draw from the mixture of class-mixture index 1 (the reference hard-codes `x[:, 1]`;
        with a single mixture, where the reference raises, index 0 is used) and return (x, log_prob(x))."""
        _hip.require_device(self.mG)
        dev = self.mG.device
        m = 1 if self.M > 1 else 0
        w = torch.softmax(self.wG.detach()[m].float(), dim=-1)
        k = torch.multinomial(w, n_samples, replacement=True)              # component per sample (RNG: plumbing)
        rows = (m * self.K + k).to(torch.int64).contiguous()
        D = self.mG[0, 0].numel()
        eps = torch.randn(n_samples, D, device=dev, dtype=torch.float32)
        x = torch.empty(n_samples, *self.mG.shape[2:], device=dev, dtype=torch.float32)
        _hip.call("cf_gmm_sample", _hip.p(_hip.f32(self.mG.detach())), _hip.p(_hip.f32(self.sG.detach())), _hip.p(rows),
                  _hip.p(eps), _hip.p(x), n_samples, D, _hip.stream())
        return x, (self.log_prob(x, context) if need_log_prob else None)
