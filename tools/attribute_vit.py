#!/usr/bin/env python3
"""Stage-by-stage rounding of the SimpleViT conditioner of ONE TransCoupling layer of a fixture: every stage is run on the
fp64 trace's stage input rounded to fp32 - by the HIP layer kernels and by torch fp32 on the CPU - and its output compared
with the fp64 stage output (max abs error / max |output|).  Shows which HIP op rounds worse than the reference's.
Then the whole conditioner (propagated) through the layer-by-layer kernels, the fused layer kernel's path and torch fp32.
usage: attribute_vit.py [name=smap] [tag=extreme] [layer index=3]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tests.gpu_util import build_model, set_noise
from tests.helpers import load_e2e, e2e_inputs
import oracle.flow_oracle as fo
from contextflow_amd.layers import _hip
from contextflow_amd.layers.simple_vit import _linear, _layernorm

name = sys.argv[1] if len(sys.argv) > 1 else "smap"
tag = sys.argv[2] if len(sys.argv) > 2 else "extreme"
li = int(sys.argv[3]) if len(sys.argv) > 3 else 3
DEV = "cuda:0"
ops, _, M, params, fx = load_e2e(name, None if tag == "none" else tag)
x, u, eps = e2e_inputs(name, fx)
p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
tr64 = []
fo.flow_forward(ops, p64, x.double(), None if u is None else u.double(), [e.double() for e in eps], trace=tr64)
op = ops[li]
assert op[0] == "transcoupling", op
xin64 = tr64[li - 1][2]
x0 = xin64[:, : xin64.shape[1] // 2]
model = build_model(name, params)
cpl = model.sequence_modules[li]
vit = cpl.NN[0]
sz, patch = op[2], op[3]
d = fo.vit_dims(sz, patch)
B = x0.shape[0]
p1, p2, gh, gw, dim = patch[0], patch[1], d["gh"], d["gw"], d["dim"]
q = "%d.NN.0." % li
ntok = gh * gw
pos = fo.posemb_sincos_2d(gh, gw, dim)


def err(a, b):
    return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


rows = []


def stage(label, f64, f32, hip):
    """f64(): exact output; f32(inp32) torch fp32 on rounded exact inputs; hip(inp dev) same through the HIP kernels"""
    ref = f64()
    rows.append((label, err(f32(), ref), err(hip(), ref), ref.abs().max().item()))
    return ref


P = lambda k, dt=torch.float64: p64[q + k].to(dt)
tok = x0.reshape(B, d["cin"], gh, p1, gw, p2).permute(0, 2, 4, 3, 5, 1).reshape(B * ntok, d["patch_dim"])
r32 = lambda t: t.float()
g = lambda t: t.float().to(DEV).contiguous()
with torch.no_grad():
    tpe = vit.to_patch_embedding
    a = stage("LN patch", lambda: F.layer_norm(tok, (d["patch_dim"],), P("to_patch_embedding.1.weight"), P("to_patch_embedding.1.bias")),
              lambda: F.layer_norm(r32(tok), (d["patch_dim"],), P("to_patch_embedding.1.weight", torch.float32), P("to_patch_embedding.1.bias", torch.float32)),
              lambda: _layernorm(g(tok), tpe[1]))
    b = stage("Linear embed", lambda: F.linear(a, P("to_patch_embedding.2.weight"), P("to_patch_embedding.2.bias")),
              lambda: F.linear(r32(a), P("to_patch_embedding.2.weight", torch.float32), P("to_patch_embedding.2.bias", torch.float32)),
              lambda: _linear(g(a), tpe[2]))
    posr = pos.double().repeat(B, 1)
    t = stage("LN + pos", lambda: F.layer_norm(b, (dim,), P("to_patch_embedding.3.weight"), P("to_patch_embedding.3.bias")) + posr,
              lambda: F.layer_norm(r32(b), (dim,), P("to_patch_embedding.3.weight", torch.float32), P("to_patch_embedding.3.bias", torch.float32)) + posr.float(),
              lambda: _layernorm(g(b), tpe[3], pos=pos.to(DEV).contiguous(), ntok=ntok))
    for l in range(d["depth"]):
        A, Fq = "transformer.layers.%d.0." % l, "transformer.layers.%d.1.net." % l
        attn, ff = vit.transformer.layers[l]
        y = stage("L%d LN1" % l, lambda: F.layer_norm(t, (dim,), P(A + "norm.weight"), P(A + "norm.bias")),
                  lambda: F.layer_norm(r32(t), (dim,), P(A + "norm.weight", torch.float32), P(A + "norm.bias", torch.float32)),
                  lambda: _layernorm(g(t), attn.norm))
        qkv = stage("L%d qkv" % l, lambda: F.linear(y, P(A + "to_qkv.weight")), lambda: F.linear(r32(y), P(A + "to_qkv.weight", torch.float32)),
                    lambda: _linear(g(y), attn.to_qkv))

        def att_ref(v, dt):
            qq, kk, vv = v.to(dt).reshape(B, ntok, -1).chunk(3, dim=-1)
            w = torch.softmax(torch.matmul(qq, kk.transpose(-1, -2)) * d["dim_head"] ** -0.5, dim=-1)
            return torch.matmul(w, vv).reshape(B * ntok, -1)

        def att_hip():
            o = torch.empty(B * ntok, attn.dim_head, device=DEV, dtype=torch.float32)
            _hip.call("cf_attention", _hip.p(g(qkv)), _hip.p(o), B, ntok, attn.dim_head, float(attn.scale), _hip.stream())
            return o
        o = stage("L%d attention" % l, lambda: att_ref(qkv, torch.float64), lambda: att_ref(qkv, torch.float32), att_hip)
        t2 = stage("L%d out + res" % l, lambda: F.linear(o, P(A + "to_out.weight")) + t,
                   lambda: F.linear(r32(o), P(A + "to_out.weight", torch.float32)) + r32(t),
                   lambda: _linear(g(o), attn.to_out, res=g(t)))
        y = stage("L%d LN2" % l, lambda: F.layer_norm(t2, (dim,), P(Fq + "0.weight"), P(Fq + "0.bias")),
                  lambda: F.layer_norm(r32(t2), (dim,), P(Fq + "0.weight", torch.float32), P(Fq + "0.bias", torch.float32)),
                  lambda: _layernorm(g(t2), ff.net[0]))
        hcd = stage("L%d fc1 + GELU" % l, lambda: F.gelu(F.linear(y, P(Fq + "1.weight"), P(Fq + "1.bias"))),
                    lambda: F.gelu(F.linear(r32(y), P(Fq + "1.weight", torch.float32), P(Fq + "1.bias", torch.float32))),
                    lambda: _linear(g(y), ff.net[1], act=1))
        t = stage("L%d fc2 + res" % l, lambda: F.linear(hcd, P(Fq + "3.weight"), P(Fq + "3.bias")) + t2,
                  lambda: F.linear(r32(hcd), P(Fq + "3.weight", torch.float32), P(Fq + "3.bias", torch.float32)) + r32(t2),
                  lambda: _linear(g(hcd), ff.net[3], res=g(t2)))
    fin = stage("final LN", lambda: F.layer_norm(t, (dim,), P("transformer.norm.weight"), P("transformer.norm.bias")),
                lambda: F.layer_norm(r32(t), (dim,), P("transformer.norm.weight", torch.float32), P("transformer.norm.bias", torch.float32)),
                lambda: _layernorm(g(t), vit.transformer.norm))
    print("%-16s  torch fp32   HIP        max|out|" % "stage (local)")
    for r in rows:
        print("%-16s  %.2e   %.2e   %.3g" % r)
    # whole conditioner, propagated
    h64 = fo.vit_net(x0, p64, "%d." % li, sz, patch)
    h32 = fo.vit_net(x0.float(), params, "%d." % li, sz, patch)
    hl = cpl.net(g(x0))
    print("conditioner output h (propagated): torch fp32 %.2e | HIP layer-by-layer %.2e   (max|h| %.3g)" % (err(h32, h64), err(hl, h64), h64.abs().max().item()))
    # t and raw halves separately (t adds to z directly; raw goes through 2 tanh(raw/2), saturated in the stress regimes)
    c2 = h64.shape[1] // 2
    print("   t half: torch %.2e HIP %.2e | raw half: torch %.2e HIP %.2e" % (err(h32[:, :c2], h64[:, :c2]), err(hl[:, :c2], h64[:, :c2]),
                                                                             err(h32[:, c2:], h64[:, c2:]), err(hl[:, c2:], h64[:, c2:])))
    zf, _ = cpl(g(xin64), None)
    z64 = tr64[li][2]
    z32, _ = fo.transcoupling_fwd(xin64.float(), params, "%d." % li, sz, patch)
    print("layer output z: torch fp32 %.2e | HIP fused layer %.2e" % (err(z32, z64), err(zf, z64)))
