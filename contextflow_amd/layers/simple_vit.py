"""SimpleViT image->image transformer, the conditioner of TransCoupling
(reference: contextflow/layers/simple_vit.py:18-127; 1 head x 64, pre-LN, sincos position table).

The nn.LayerNorm / nn.Linear children exist only to hold parameters under the reference's
state_dict names (to_patch_embedding.{1,2,3}, transformer.layers.L.{0.norm,0.to_qkv,0.to_out,
1.net.{0,1,3}}, transformer.norm); every contraction runs in cf_linear (fp32 MFMA) and the
normalisations / attention in their HIP kernels."""
import math

import torch
import torch.nn as nn

from . import _hip


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def posemb_sincos_2d(h, w, dim, temperature=10000, dtype=torch.float32):
    """Fixed 2-D sin/cos table (simple_vit.py:18-27; note omega = arange(dim/4)/(dim/4 - 1))."""
    assert dim % 4 == 0, "feature dimension must be multiple of 4 for sincos emb"
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    omega = 1.0 / (temperature ** (torch.arange(dim // 4) / (dim // 4 - 1)))
    ang_y = gy.flatten()[:, None] * omega[None, :]
    ang_x = gx.flatten()[:, None] * omega[None, :]
    return torch.cat((ang_x.sin(), ang_x.cos(), ang_y.sin(), ang_y.cos()), dim=1).type(dtype)


def _linear(x, lin, res=None, act=0):
    rows, K = x.shape
    N = lin.weight.shape[0]
    b = _hip.f32(lin.bias.detach()) if lin.bias is not None else None
    w = _hip.f32(lin.weight.detach())
    y = torch.empty(rows, N, device=x.device, dtype=torch.float32)
    _hip.call("cf_linear", _hip.p(x), _hip.p(w), _hip.p(b), _hip.p(res), _hip.p(y), rows, K, N, act, _hip.stream())
    return y


def _layernorm(x, ln, pos=None, ntok=1):
    rows, dim = x.shape
    y = torch.empty_like(x)
    _hip.call("cf_layernorm", _hip.p(x), _hip.p(_hip.f32(ln.weight.detach())), _hip.p(_hip.f32(ln.bias.detach())),
              _hip.p(pos), _hip.p(y), rows, dim, ntok, float(ln.eps), _hip.stream())
    return y


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, dim))

    def forward(self, x):                       # x: (rows, dim) ; returns FF(x) + x
        y = _layernorm(x, self.net[0])
        y = _linear(y, self.net[1], act=1)
        return _linear(y, self.net[3], res=x)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("contextflow_amd Attention: heads=1 only (coupling.py:109)")
        inner = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x, ntok):                 # returns Attn(x) + x
        rows = x.shape[0]
        qkv = _linear(_layernorm(x, self.norm), self.to_qkv)
        o = torch.empty(rows, self.dim_head, device=x.device, dtype=torch.float32)
        _hip.call("cf_attention", _hip.p(qkv), _hip.p(o), rows // ntok, ntok, self.dim_head, float(self.scale), _hip.stream())
        return _linear(o, self.to_out, res=x)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.layers = nn.ModuleList(
            [nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head), FeedForward(dim, mlp_dim)]) for _ in range(depth)])

    def forward(self, x, ntok):
        for attn, ff in self.layers:
            x = attn(x, ntok)
            x = ff(x)
        return _layernorm(x, self.norm)


class SimpleViT(nn.Module):
    def __init__(self, *, image_size, patch_size, dim, depth, heads, mlp_dim, channels=3, dim_head=64):
        super().__init__()
        ih, iw = pair(image_size)
        ph, pw = pair(patch_size)
        assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
        self.image_size, self.patch_size, self.channels, self.dim = (ih, iw), (ph, pw), channels, dim
        patch_dim = channels * ph * pw
        # index 0 is the (parameter-free) patch rearrangement in the reference
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.LayerNorm(patch_dim), nn.Linear(patch_dim, dim),
                                                nn.LayerNorm(dim))
        self.grid = (ih // ph, iw // pw)
        self.pos_embedding = posemb_sincos_2d(self.grid[0], self.grid[1], dim)   # plain attribute, as in the reference
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)

    def forward(self, img):
        _hip.require_device(img)
        x, xbs = _hip.bview(img)
        B, C, H, W = x.shape
        ph, pw = self.patch_size
        ntok = self.grid[0] * self.grid[1]
        if self.pos_embedding.device != x.device:
            self.pos_embedding = self.pos_embedding.to(x.device).contiguous()
        tok = torch.empty(B * ntok, C * ph * pw, device=x.device, dtype=torch.float32)
        _hip.call("cf_patchify", _hip.p(x), _hip.p(tok), B, C, H, W, ph, pw, xbs, 0, _hip.stream())
        t = _layernorm(tok, self.to_patch_embedding[1])
        t = _linear(t, self.to_patch_embedding[2])
        t = _layernorm(t, self.to_patch_embedding[3], pos=self.pos_embedding, ntok=ntok)
        t = self.transformer(t, ntok)
        cout = self.dim // (ph * pw)
        out = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
        _hip.call("cf_patchify", _hip.p(t), _hip.p(out), B, cout, H, W, ph, pw, cout * H * W, 1, _hip.stream())
        return out
