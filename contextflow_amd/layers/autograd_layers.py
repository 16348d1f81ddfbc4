"""Backward of the layers that run layer by layer inside the fused plan (everything that is not a fused conv flow
step): TransCoupling with its SimpleViT conditioner, and Conv1x1 / ActNorm / Augment of shapes the step kernel does
not cover (the SMAP topology: 26 channels, 8x1 windows; reference training step experiment_ad.py:204-213).

Each `*_backward(module, x_in, gz, gld)` takes the layer's saved input, the gradient w.r.t. its output and the
gradient w.r.t. the per-sample log-det (B,), and returns (gradient w.r.t. the input, {parameter: gradient}).
Nothing but the layer input is kept from the forward: the conditioner is recomputed (token-major, with its
intermediates) inside the backward.  Elementwise / normalisation / attention pieces are HIP kernels
(csrc/cf_layers_bwd.hip); the Linear layers' two backward products are plain library GEMMs."""
import torch

from . import _hip
from .simple_vit import _layernorm, _linear


def _new(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------ small pieces
def _linear_bwd(x_in, lin, gy, grads, need_gx=True):
    """y = x W^T + b: gx = gy W (cf_linear with the transposed weight), gW = gy^T x and gb = column sums (cf_linear_wgrad:
    split-K MFMA GEMM over the rows).  grads = None: the layer is frozen, only gx is computed."""
    W = _hip.f32(lin.weight.detach())
    N, K = W.shape
    rows = x_in.shape[0]
    gy = gy.contiguous()
    st = _hip.stream()
    if grads is not None:
        x_in = x_in.contiguous()
        gW = _new(N, K, like=gy)
        gb = _new(N, like=gy) if lin.bias is not None else None
        ws = torch.empty(_hip.lib().cf_linear_wgrad_ws_bytes(rows, K, N), device=gy.device, dtype=torch.uint8)
        _hip.call("cf_linear_wgrad", _hip.p(x_in), _hip.p(gy), _hip.p(gW), _hip.p(gb), _hip.p(ws), rows, K, N, st)
        grads[lin.weight] = gW
        if gb is not None:
            grads[lin.bias] = gb
    if not need_gx:
        return None
    gx = _new(rows, K, like=gy)
    _hip.call("cf_linear_tn", _hip.p(gy), _hip.p(W.contiguous()), _hip.p(gx), rows, N, K, st)      # gy W, the weight as stored
    return gx


def _layernorm_bwd(x_in, ln, gy, grads):
    rows, dim = x_in.shape
    gx = _new(rows, dim, like=x_in)
    nparts = _hip.lib().cf_layernorm_bwd_parts()
    part = _new(nparts, 2 * dim, like=x_in)
    _hip.call("cf_layernorm_bwd", _hip.p(x_in), _hip.p(_hip.f32(ln.weight.detach())), _hip.p(gy), _hip.p(gx), _hip.p(part),
              rows, dim, float(ln.eps), _hip.stream())
    if grads is not None:
        s = part.sum(0)
        grads[ln.weight], grads[ln.bias] = s[:dim], s[dim:]
    return gx


def _gelu(x, gy=None):
    out = torch.empty_like(x)
    _hip.call("cf_gelu", _hip.p(x), _hip.p(gy), _hip.p(out), x.numel(), 0 if gy is None else 1, _hip.stream())
    return out


# ------------------------------------------------------------------------------------------------ SimpleViT
def vit_forward_taped(vit, x0):
    """SimpleViT.forward (simple_vit.py:117-127) on x0 (B, C, H, W), keeping what the backward needs."""
    x, xbs = _hip.bview(x0)
    B, C, H, W = x.shape
    ph, pw = vit.patch_size
    ntok = vit.grid[0] * vit.grid[1]
    if vit.pos_embedding.device != x.device:
        vit.pos_embedding = vit.pos_embedding.to(x.device).contiguous()
    tpe = vit.to_patch_embedding
    tok = _new(B * ntok, C * ph * pw, like=x)
    _hip.call("cf_patchify", _hip.p(x), _hip.p(tok), B, C, H, W, ph, pw, xbs, 0, _hip.stream())
    a0 = _layernorm(tok, tpe[1])
    e = _linear(a0, tpe[2])
    t = _layernorm(e, tpe[3], pos=vit.pos_embedding, ntok=ntok)
    layers = []
    for attn, ff in vit.transformer.layers:
        n1 = _layernorm(t, attn.norm)
        qkv = _linear(n1, attn.to_qkv)
        o = _new(t.shape[0], attn.dim_head, like=t)
        _hip.call("cf_attention", _hip.p(qkv), _hip.p(o), B, ntok, attn.dim_head, float(attn.scale), _hip.stream())
        u = _linear(o, attn.to_out, res=t)
        n2 = _layernorm(u, ff.net[0])
        hp = _linear(n2, ff.net[1])
        hg = _gelu(hp)
        t_next = _linear(hg, ff.net[3], res=u)
        layers.append((t, n1, qkv, o, u, n2, hp, hg))
        t = t_next
    f = _layernorm(t, vit.transformer.norm)
    cout = vit.dim // (ph * pw)
    h = _new(B, cout, H, W, like=x)
    _hip.call("cf_patchify", _hip.p(f), _hip.p(h), B, cout, H, W, ph, pw, cout * H * W, 1, _hip.stream())
    return h, (tok, a0, e, layers, t, (B, C, H, W, ntok, cout))


def vit_backward(vit, tape, gh, grads):
    """gh: d/d h (B, cout, H, W) dense.  Returns d/d x0 (B, C, H, W) dense; parameter gradients go into `grads`."""
    tok, a0, e, layers, t_last, (B, C, H, W, ntok, cout) = tape
    ph, pw = vit.patch_size
    tpe = vit.to_patch_embedding
    st = _hip.stream()
    gf = _new(B * ntok, vit.dim, like=gh)
    _hip.call("cf_patchify", _hip.p(gh), _hip.p(gf), B, cout, H, W, ph, pw, cout * H * W, 0, st)
    gt = _layernorm_bwd(t_last, vit.transformer.norm, gf, grads)
    for (attn, ff), (t, n1, qkv, o, u, n2, hp, hg) in zip(reversed(list(vit.transformer.layers)), reversed(layers)):
        # t_next = u + W2 gelu(W1 LN2(u) + b1) + b2
        ghg = _linear_bwd(hg, ff.net[3], gt, grads)
        ghp = _gelu(hp, ghg)
        gn2 = _linear_bwd(n2, ff.net[1], ghp, grads)
        gu = gt + _layernorm_bwd(u, ff.net[0], gn2, grads)
        # u = t + Wo attn(Wqkv LN1(t))
        go = _linear_bwd(o, attn.to_out, gu, grads)
        gqkv = torch.empty_like(qkv)
        _hip.call("cf_attention_bwd", _hip.p(qkv), _hip.p(go), _hip.p(gqkv), B, ntok, attn.dim_head, float(attn.scale), st)
        gn1 = _linear_bwd(n1, attn.to_qkv, gqkv, grads)
        gt = gu + _layernorm_bwd(t, attn.norm, gn1, grads)
    # t0 = LN_b(e) + pos ; e = We LN_a(tok) + be
    ge = _layernorm_bwd(e, tpe[3], gt, grads)
    ga0 = _linear_bwd(a0, tpe[2], ge, grads)
    gtok = _layernorm_bwd(tok, tpe[1], ga0, grads)
    gx0 = _new(B, C, H, W, like=gh)
    _hip.call("cf_patchify", _hip.p(gtok), _hip.p(gx0), B, C, H, W, ph, pw, C * H * W, 1, st)
    return gx0


# ------------------------------------------------------------------------------------------------ layers
def transcoupling_backward(m, x_in, gz, gld):
    """TransCoupling (coupling.py:123-155): z = [x0 | x1 exp(log_s) + t], ldj = sum log_s, [t | raw] = ViT(x0)."""
    x, xbs = _hip.bview(x_in)
    gzv, gzbs = _hip.bview(gz)
    B, C, H, W = x.shape
    half = C // 2
    h, tape = vit_forward_taped(m.NN[0], x[:, :half])
    gx = _new(B, C, H, W, like=x)
    ghd = _new(B, C, H, W, like=x)
    _hip.call("cf_coupling_apply_bwd", _hip.p(x), _hip.p(h), _hip.p(gzv), _hip.p(_hip.f32(gld)), _hip.p(gx), _hip.p(ghd),
              B, C, H * W, xbs, gzbs, _hip.stream())
    grads = {}
    gx0 = vit_backward(m.NN[0], tape, ghd, grads)
    gx[:, :half] += gx0
    return gx, grads


def _relu_mask(act, gy):
    out = torch.empty_like(gy)
    _hip.call("cf_relu_bwd", _hip.p(act), _hip.p(gy), _hip.p(out), gy.numel(), _hip.stream())
    return out


def conv_backward(x_in, conv, gy, grads, need_gx=True):
    """k x k stride-1 convolution with reflect padding (cf_conv2d_reflect; nn.Conv2d of coupling.py:26-29 / the masked
    convolutions of ar.py).  x_in (B, Cin, H, W), gy (B, Cout, H, W) -> gx; grads[weight], grads[bias].
      * data gradient: zero-padded convolution of gy with the spatially flipped, channel-transposed weights on the MFMA
        conv kernel (cf_conv2d_zero, padding k - 1), then the adjoint of the reflect padding folds the border ring back;
      * weight / bias gradient: gW (Cout, Cin kh kw) = gy_rows^T cols, the split-K MFMA GEMM over the (sample, pixel)
        rows (cf_linear_wgrad) of the unfolded, reflect-padded input (index ops)."""
    x = _hip.f32(x_in).contiguous()
    gy = _hip.f32(gy).contiguous()
    B, Cin, H, W = x.shape
    w = _hip.f32(conv.weight.detach())
    Cout, _, kh, kw = w.shape
    ph, pw = conv.padding if isinstance(conv.padding, tuple) else (conv.padding, conv.padding)
    st = _hip.stream()
    if grads is not None:
        xp = torch.nn.functional.pad(x, (pw, pw, ph, ph), mode="reflect") if (ph or pw) else x
        cols = xp.unfold(2, kh, 1).unfold(3, kw, 1).permute(0, 2, 3, 1, 4, 5).reshape(B * H * W, Cin * kh * kw).contiguous()
        rows_g = gy.permute(0, 2, 3, 1).reshape(B * H * W, Cout).contiguous()
        K = Cin * kh * kw
        gW = _new(Cout, K, like=gy)
        gb = _new(Cout, like=gy) if conv.bias is not None else None
        ws = torch.empty(_hip.lib().cf_linear_wgrad_ws_bytes(B * H * W, K, Cout), device=gy.device, dtype=torch.uint8)
        _hip.call("cf_linear_wgrad", _hip.p(cols), _hip.p(rows_g), _hip.p(gW), _hip.p(gb), _hip.p(ws), B * H * W, K, Cout, st)
        grads[conv.weight] = gW.view(Cout, Cin, kh, kw)
        if gb is not None:
            grads[conv.bias] = gb
    if not need_gx:
        return None
    wt = w.flip(2, 3).transpose(0, 1).contiguous()                     # (Cin, Cout, kh, kw)
    gpad = _new(B, Cin, H + 2 * ph, W + 2 * pw, like=gy)
    _hip.call("cf_conv2d_zero", _hip.p(gy), _hip.p(wt), None, _hip.p(gpad), B, Cout, Cin, H, W, kh, kw, kh - 1, kw - 1, 0,
              Cout * H * W, st)
    if not (ph or pw):
        return gpad
    gx = _new(B, Cin, H, W, like=gy)
    _hip.call("cf_reflect_pad_adjoint", _hip.p(gpad), _hip.p(gx), B * Cin, H, W, ph, pw, st)
    return gx


def coupling_conv_backward(m, x_in, gz, gld):
    """Coupling with a conv conditioner of any shape (coupling.py:39-66; the fused step kernels cover only the image
    shapes): the conditioner is re-run layer by layer with its activations kept, then the chain back."""
    from .coupling import conv2d_reflect
    x, xbs = _hip.bview(x_in)
    gzv, gzbs = _hip.bview(gz)
    B, C, H, W = x.shape
    half = C // 2
    c1, c2, c3 = m.NN[0], m.NN[2], m.NN[4]
    x0 = x[:, :half].contiguous()
    a1 = conv2d_reflect(x0, c1, True)
    a2 = conv2d_reflect(a1, c2, True)
    h = conv2d_reflect(a2, c3, False)
    gx, gh = _new(B, C, H, W, like=x), _new(B, C, H, W, like=x)
    _hip.call("cf_coupling_apply_bwd", _hip.p(x), _hip.p(h), _hip.p(gzv), _hip.p(_hip.f32(gld)), _hip.p(gx), _hip.p(gh),
              B, C, H * W, xbs, gzbs, _hip.stream())
    grads = {}
    g2 = _relu_mask(a2, conv_backward(a2, c3, gh, grads))
    g1 = _relu_mask(a1, conv_backward(a1, c2, g2, grads))
    gx[:, :half] += conv_backward(x0, c1, g1, grads)
    return gx, grads


def masked_coupling_backward(m, x_in, gz, gld):
    """MaskedCoupling (ar.py:33-57): z = x exp(log_s) + t with [t ; raw] = conv3(relu(conv2(relu(conv1(relu x))))) + [x ; x].
    The weights are masked in place by the forward (masked_conv_2d.py:22), so - as under torch.autograd in the reference -
    the weight gradients are the plain convolution gradients (entries at masked positions are wiped by the next forward)."""
    from .coupling import conv2d_reflect
    x = _hip.f32(x_in).contiguous()
    gzc = _hip.f32(gz).contiguous()
    B, D, H, W = x.shape
    nn_ = m.NN
    for c in (nn_.conv1, nn_.conv2, nn_.conv3):
        c.apply_mask_()
    a0 = _relu_mask(x, x)                                               # relu(x)
    a1 = conv2d_reflect(a0, nn_.conv1, True)
    a2 = conv2d_reflect(a1, nn_.conv2, True)
    h = conv2d_reflect(a2, nn_.conv3, False)
    _hip.call("cf_add_repeat", _hip.p(h), _hip.p(x), B, 2 * D, D, H * W, _hip.stream())       # + [x ; x]
    # the affine-map kernels transform the SECOND channel half of their input: feed them [0 ; x] / [0 ; gz]
    zeros = torch.zeros_like(x)
    xx, gzz = torch.cat([zeros, x], dim=1), torch.cat([zeros, gzc], dim=1)
    gxx, gh = _new(B, 2 * D, H, W, like=x), _new(B, 2 * D, H, W, like=x)
    _hip.call("cf_coupling_apply_bwd", _hip.p(xx), _hip.p(h), _hip.p(gzz), _hip.p(_hip.f32(gld)), _hip.p(gxx), _hip.p(gh),
              B, 2 * D, H * W, 2 * D * H * W, 2 * D * H * W, _hip.stream())
    grads = {}
    g2 = _relu_mask(a2, conv_backward(a2, nn_.conv3, gh, grads))
    g1 = _relu_mask(a1, conv_backward(a1, nn_.conv2, g2, grads))
    g0 = _relu_mask(x, conv_backward(a0, nn_.conv1, g1, grads))
    gx = gxx[:, D:] + g0 + gh[:, :D] + gh[:, D:]                       # through x * s, the residual block, and + [x ; x]
    return gx.contiguous(), grads


def conv1x1_backward(m, x_in, gz, gld):
    """Conv1x1 (conv1x1.py:52-57): z = W x per pixel, ldj = H W log|det W|."""
    x = _hip.f32(x_in)
    B, C, H, W = x.shape
    Wm = _hip.f32(m.NN.detach())
    gzc = _hip.f32(gz).contiguous()
    gx = torch.empty_like(gzc)
    _hip.call("cf_conv1x1_fwd", _hip.p(gzc), _hip.p(Wm.t().contiguous()), None, _hip.p(gx), B, C, H * W, C * H * W, C * H * W,
              _hip.stream())
    lad = _new(1, like=gzc)
    winv = _new(C, C, like=gzc)
    _hip.call("cf_slogdet_inverse", _hip.p(Wm), C, _hip.p(lad), _hip.p(winv), _hip.stream())
    # gW[o][i] = sum over (b, pixel) of gz[b,o,p] x[b,i,p]: the split-K MFMA GEMM over rows = (sample, pixel) (cf_linear_wgrad)
    # on channel-last copies (index ops)
    rows = B * H * W
    xs = x.permute(0, 2, 3, 1).reshape(rows, C).contiguous()
    gs = gzc.permute(0, 2, 3, 1).reshape(rows, C).contiguous()
    gW = _new(C, C, like=gzc)
    ws = torch.empty(_hip.lib().cf_linear_wgrad_ws_bytes(rows, C, C), device=gzc.device, dtype=torch.uint8)
    _hip.call("cf_linear_wgrad", _hip.p(xs), _hip.p(gs), _hip.p(gW), None, _hip.p(ws), rows, C, C, _hip.stream())
    gW = gW + (gld.sum() * (H * W)) * winv.t()
    return gx, {m.NN: gW}


def actnorm_backward(m, x_in, gz, gld):
    """ActNorm (actnorm.py:53-60): z = (x - t) exp(-logs), ldj = +sum(logs) (reference quirk: no H W factor)."""
    x = _hip.f32(x_in).contiguous()
    B, C, H, W = x.shape
    t, logs = _hip.f32(m.NN_t.detach()), _hip.f32(m.NN_logs.detach())
    gzc = _hip.f32(gz).contiguous()
    st = _hip.stream()
    z = _new(B, C, H, W, like=gzc)
    _hip.call("cf_actnorm", _hip.p(x), _hip.p(t), _hip.p(logs), _hip.p(z), None, B, C, H * W, 0, st)
    sums = _new(2 * C, like=gzc)
    wsum = torch.empty(_hip.lib().cf_channel_sums_ws_bytes(B, C), device=gzc.device, dtype=torch.uint8)
    _hip.call("cf_channel_sums", _hip.p(gzc), _hip.p(z), _hip.p(sums), _hip.p(wsum), B, C, H * W, C * H * W, C * H * W, st)
    s = torch.exp(-logs)
    gx = gzc * s.view(1, C, 1, 1)
    return gx, {m.NN_t: -s * sums[:C], m.NN_logs: gld.sum() - sums[C:]}


def layer_backward(m, x_in, gz, gld):
    """Dispatch on the layer type; raises for layers without a hand-written backward."""
    from .actnorm import ActNorm
    from .augment import Augment
    from .conv1x1 import Conv1x1
    from .ar import MaskedCoupling
    from .coupling import Coupling, TransCoupling
    from .permute_axes import PermuteAxes
    if isinstance(m, TransCoupling):
        return transcoupling_backward(m, x_in, gz, gld)
    if type(m) is Coupling and not m.context_net:
        return coupling_conv_backward(m, x_in, gz, gld)
    if isinstance(m, MaskedCoupling):
        return masked_coupling_backward(m, x_in, gz, gld)
    if type(m) is Conv1x1:
        return conv1x1_backward(m, x_in, gz, gld)
    if type(m) is ActNorm:
        return actnorm_backward(m, x_in, gz, gld)
    if isinstance(m, Augment) and m.split_dim == 1:
        return gz[:, : x_in.shape[1]], {}
    if isinstance(m, PermuteAxes):
        return gz.permute(m.inverse_permutation).contiguous(), {}
    raise NotImplementedError("no backward for layer %s in the fused plan" % type(m).__name__)
