// One transformer-coupling flow step — Conv1x1 -> ActNorm -> TransCoupling (patchify, SimpleViT, un-patchify, affine
// map, log-det) — as ONE gfx950 kernel for the time-series topologies (H x 1 windows, patch (2,1), 4 tokens per sample:
// SMAP).  Reference: contextflow/model.py:129-147 (the per-step triple), layers/conv1x1.py:52-57, layers/actnorm.py:
// 53-60, layers/coupling.py:100-159, layers/simple_vit.py:18-127.
//
// Design (MI355X).  Everything between the input x and the output z stays in REGISTERS:
//  * every Linear runs on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) with the weights as the A operand (row =
//    output feature) and the activations as the B operand (column = token).  A result tile has the TOKEN on the lane and
//    the FEATURES in the 16 accumulator registers - and a following product that sums over those features can take the
//    accumulator registers AS its B operand, with no lane movement and no LDS: k-step s of the next Linear is simply
//    "register s", i.e. lane half 0 supplies feature row(s, 0) and lane half 1 feature row(s, 1) of its own token; the
//    packed weights are stored in that k order (k_vit_step_pack).  The residual stream, LayerNorm outputs, q / k / v, the
//    attention output and the MLP hidden layer never leave the register file;
//  * LayerNorm is lane-local over the registers (+ one exchange between the two lane halves that share a token), GELU is
//    elementwise on the accumulators (erf by Abramowitz-Stegun 7.1.26 on the hardware exp / rcp: |err| <= 2e-7);
//  * a sample's 4 tokens sit in 4 consecutive lanes: q.k^T and p.v use DPP quad permutations fused into the FMAs, the
//    softmax over 4 scores is exact (no online rescaling);
//  * Conv1x1 + ActNorm run as a first product with rows (position-in-patch, channel): its two result tiles ARE the
//    patchified conditioner input x0 (tile 0) and the half that gets transformed, x1 (tile 1).  The model's 2C features
//    are laid out so that t (tile 0) and raw log-scale (tile 1) of one (position, channel) share lane AND register index
//    with x1: the affine epilogue is lane-local;
//  * a wave owns 32 token columns = 8 samples; nothing crosses a wave: no LDS, no barrier.  One launch = one flow step:
//    HBM traffic = read x + write z.
// Geometry covered: C even, C <= 32 (SMAP: 26), H = 8, W = 1, patch (2, 1), dim = 2C, 1 head x 64, any depth.  Other
// transformer couplings keep the LDS-plane kernel (cf_vit_fused.hip) or the layer-by-layer kernels (cf_vit.hip).
#include "cf_common.h"
#include <math.h>

extern "C" int cf_slogdet_inverse(const float* W, int C, float* logabsdet, float* Winv, cf_stream_t stream);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__host__ __device__ constexpr int trow(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
__host__ __device__ constexpr int ngrp(int ks) { return (ks + 3) / 4; }

// ---- compile-time geometry ------------------------------------------------------------------------------------
template <int C_> struct VS {
    static constexpr int C = C_, CIN = C / 2, HW = 8, NTOK = 4, DIM = 2 * C;
    // registers of a residual tile that can hold a valid row: rows trow(r, lk) < C.  r < KPT for some lane half
    static constexpr int KPT = C > 28 ? 16 : (C > 24 ? 14 : (C > 20 ? 12 : (C > 16 ? 10 : 8)));
    static constexpr int KS_RES = 2 * KPT;            // k-steps when the B operand is the two residual tiles
    static constexpr int KS_IN = KPT;                 // ... tile 0 only (patch features)
    static constexpr int KS0 = C;                     // Conv1x1: one k-step per input channel (lane half = position in patch)
    static constexpr int KS_HEAD = 32;                // attention output: 64 features, all registers
    // workspace (floats).  "vec" = 64 floats in register order [lk][tile][r]
    static constexpr int OFF_B0 = 4, OFF_A0 = OFF_B0 + 64;
    static constexpr int OFF_LN0 = OFF_A0 + ngrp(KS0) * 2 * 256;              // [w | b]
    static constexpr int OFF_WE = OFF_LN0 + 128, OFF_BE = OFF_WE + ngrp(KS_IN) * 2 * 256;
    static constexpr int OFF_LN1 = OFF_BE + 64, OFF_POS = OFF_LN1 + 128, OFF_LAYER = OFF_POS + NTOK * 64;
    // per layer
    static constexpr int L_LNA = 0, L_WQKV = 128, L_WOUT = L_WQKV + ngrp(KS_RES) * 6 * 256;
    static constexpr int L_LNF = L_WOUT + ngrp(KS_HEAD) * 2 * 256, L_W1 = L_LNF + 128;
    static constexpr int L_B1 = L_W1 + ngrp(KS_RES) * 2 * 256, L_W2 = L_B1 + 64, L_B2 = L_W2 + ngrp(KS_RES) * 2 * 256;
    static constexpr int L_STRIDE = L_B2 + 64;
    static_assert(C % 2 == 0 && C >= 4 && C <= 32, "C even, <= 32");
};
template <class V> __host__ __device__ constexpr int off_lno(int depth) { return V::OFF_LAYER + depth * V::L_STRIDE; }
template <class V> __host__ __device__ constexpr int ws_floats(int depth) { return off_lno<V>(depth) + 128; }

// physical row p (0..63: tile = p >> 5) of the residual layout -> model feature f = ii * C + ch, or -1 for padding.
// tile 0 holds channels ch < C/2 (t after the last LayerNorm), tile 1 channels ch >= C/2 (raw log-scale), both at row
// ii * C/2 + c: the same (lane, register) in the two tiles belongs to one (position ii, channel c).
template <class V> __host__ __device__ inline int feat_of_phys(int p) {
    const int q = p & 31;
    if (q >= V::C) return -1;
    return (q / V::CIN) * V::C + (q % V::CIN) + V::CIN * (p >> 5);
}
// k-step s of a product whose B operand is the residual tile pair -> physical row supplied by lane half lk
template <class V> __host__ __device__ inline int phys_of_kstep(int s, int lk) { return trow(s % V::KPT, lk) + 32 * (s / V::KPT); }

// ---- packing (cf_vit_step_prepare) ----------------------------------------------------------------------------------
// fragment element ((g * RT + rt) * 64 + lane) * 4 + e = A[row = rt * 32 + (lane & 31)][k-step 4 g + e, lane half]
template <class V>
__global__ __launch_bounds__(256) void k_vit_step_pack(const float* __restrict__ Wm, const float* __restrict__ t,
                                                       const float* __restrict__ logs, const float* __restrict__ flat,
                                                       const float* __restrict__ pos, float* __restrict__ ws, int depth) {
    constexpr int C = V::C, CIN = V::CIN, DIM = V::DIM;
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    auto vec = [&](float* dst, const float* src, bool tile0_only) {          // register-order vector of a residual-feature vector
        for (int i = gtid; i < 64; i += gsz) {
            const int lk = i >> 5, rt = (i >> 4) & 1, r = i & 15;
            const int p = trow(r, lk) + 32 * rt;
            int f = tile0_only ? ((rt == 0 && (p & 31) < C) ? (p & 31) : -1) : feat_of_phys<V>(p);
            dst[i] = f >= 0 ? src[f] : 0.f;
        }
    };
    // A fragments of a Linear.  out_tiles x 32 physical output rows -> model rows through rowmap; k-steps through kmap
    auto frags = [&](float* dst, const float* W, int K, int out_tiles, int nks, auto rowmap, auto kmap) {
        const int n = ngrp(nks) * out_tiles * 256;
        for (int i = gtid; i < n; i += gsz) {
            const int e = i & 3, lane = (i >> 2) & 63, q = i >> 8, rt = q % out_tiles, g = q / out_tiles;
            const int s = 4 * g + e;
            const int row = rowmap(rt * 32 + (lane & 31));
            const int k = s < nks ? kmap(s, lane >> 5) : -1;
            dst[i] = (row >= 0 && k >= 0) ? W[row * K + k] : 0.f;
        }
    };
    auto res_row = [](int p) { return feat_of_phys<V>(p); };
    auto nat_row = [](int p) { return p; };
    auto res_k = [](int s, int lk) { return feat_of_phys<V>(phys_of_kstep<V>(s, lk)); };
    auto in_k = [](int s, int lk) { const int p = trow(s, lk); return p < C ? p : -1; };        // patch feature = tile-0 row
    auto head_k = [](int s, int lk) { return trow(s & 15, lk) + 32 * (s >> 4); };               // 64 head features, natural

    if (gtid == 0) {       // ldj_const = H*W*log|det Wm| + sum_c logs  (ws[1] holds log|det| from cf_slogdet_inverse)
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += logs[c];
        ws[0] = (float)V::HW * ws[1] + s;                                   // conv1x1.py:53 + actnorm.py:58
    }
    // Conv1x1 + ActNorm as a product over (position-in-patch, channel): rows = physical rows, k-step s = input channel s,
    // lane half = position ii'; block diagonal in the position
    for (int i = gtid; i < 64; i += gsz) {
        const int lk = i >> 5, rt = (i >> 4) & 1, r = i & 15, p = trow(r, lk) + 32 * rt, q = p & 31;
        const int ch = q < C ? (q % CIN) + CIN * rt : -1;
        ws[V::OFF_B0 + i] = ch >= 0 ? -t[ch] * expf(-logs[ch]) : 0.f;        // (x - t) e^{-logs}
    }
    for (int i = gtid; i < ngrp(V::KS0) * 2 * 256; i += gsz) {
        const int e = i & 3, lane = (i >> 2) & 63, qq = i >> 8, rt = qq % 2, g = qq / 2, s = 4 * g + e;
        const int q = lane & 31, ii = q / CIN, ch = q < C ? (q % CIN) + CIN * rt : -1;
        ws[V::OFF_A0 + i] = (ch >= 0 && s < C && ii == (lane >> 5)) ? expf(-logs[ch]) * Wm[ch * C + s] : 0.f;
    }
    const float* p = flat;       // order of TransCoupling._flat_params(): see cf_vit_fused.hip
    vec(ws + V::OFF_LN0, p, true); vec(ws + V::OFF_LN0 + 64, p + C, true); p += 2 * C;          // to_patch_embedding.1 (pd = C)
    frags(ws + V::OFF_WE, p, C, 2, V::KS_IN, res_row, in_k); p += DIM * C;
    vec(ws + V::OFF_BE, p, false); p += DIM;
    vec(ws + V::OFF_LN1, p, false); vec(ws + V::OFF_LN1 + 64, p + DIM, false); p += 2 * DIM;
    for (int n = 0; n < V::NTOK; ++n) vec(ws + V::OFF_POS + 64 * n, pos + n * DIM, false);
    for (int l = 0; l < depth; ++l) {
        float* w = ws + V::OFF_LAYER + l * V::L_STRIDE;
        vec(w + V::L_LNA, p, false); vec(w + V::L_LNA + 64, p + DIM, false); p += 2 * DIM;
        frags(w + V::L_WQKV, p, DIM, 6, V::KS_RES, nat_row, res_k); p += 192 * DIM;
        frags(w + V::L_WOUT, p, 64, 2, V::KS_HEAD, res_row, head_k); p += DIM * 64;
        vec(w + V::L_LNF, p, false); vec(w + V::L_LNF + 64, p + DIM, false); p += 2 * DIM;
        frags(w + V::L_W1, p, DIM, 2, V::KS_RES, res_row, res_k); p += DIM * DIM;
        vec(w + V::L_B1, p, false); p += DIM;
        frags(w + V::L_W2, p, DIM, 2, V::KS_RES, res_row, res_k); p += DIM * DIM;
        vec(w + V::L_B2, p, false); p += DIM;
    }
    vec(ws + off_lno<V>(depth), p, false); vec(ws + off_lno<V>(depth) + 64, p + DIM, false);
}

// ---- device helpers ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ rsrc_t make_rsrc(const float* ws, int floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, floats * 4, 0x00020000);
}
__device__ __forceinline__ float4 frag(rsrc_t rs, int lane, int foff) {
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, foff * 4, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// register-order vector (64 floats [lk][tile][r]) -> this lane's two tiles
__device__ __forceinline__ void load_vec(f32x16 (&v)[2], const float* __restrict__ base, int lk) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = *reinterpret_cast<const float4*>(base + lk * 32 + rt * 16 + 4 * q);
            v[rt][4 * q + 0] = x.x; v[rt][4 * q + 1] = x.y; v[rt][4 * q + 2] = x.z; v[rt][4 * q + 3] = x.w;
        }
}

// acc[rt] += sum over k-steps A(ws frags at foff) * B, B operand = registers: k-step s -> bop(s).
// Fragments of group g+1 are requested before the MFMAs of group g (pinned: hipcc otherwise sinks the loads).
// ZERO: the accumulators start from zero - passed to the first MFMA as the inline constant instead of 16 v_mov per tile.
template <int RT, int NKS, bool ZERO = false, class BOP>
__device__ __forceinline__ void gemm_regs(f32x16 (&acc)[RT], rsrc_t rs, int lane, int foff, BOP bop) {
    constexpr int NG = (NKS + 3) / 4;
    float4 a[2][RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[0][rt] = frag(rs, lane, foff + rt * 256);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[(g + 1) & 1][rt] = frag(rs, lane, foff + ((g + 1) * RT + rt) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g + e < NKS) {
                const float b = bop(4 * g + e);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (ZERO && g == 0 && e == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a[g & 1][rt], e), b, zero, 0, 0, 0);
                    } else {
                        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a[g & 1][rt], e), b, acc[rt], 0, 0, 0);
                    }
                }
            }
    }
}

// LayerNorm of the token over its valid features (NT tiles of C valid rows each; the two lane halves hold different rows
// of the same token): y = (x - mean) rstd w + b (+ extra).  Biased variance, eps 1e-5 (torch.nn.LayerNorm).
template <class V, int NT>
__device__ __forceinline__ void layernorm(const f32x16 (&x)[2], f32x16 (&y)[2], const float* __restrict__ ln, int lk,
                                          const float* __restrict__ extra) {
    constexpr int C = V::C;
    const float m12 = (trow(12, 1) < C || lk == 0) ? 1.f : 0.f;      // registers 12..15: rows 24..27 (lk 0) / 28..31 (lk 1)
    auto valid = [&](int r) -> float {                               // compile-time for every r except the lk-dependent ones
        if (trow(r, 1) < C) return 1.f;                              // valid in both halves
        if (trow(r, 0) >= C) return 0.f;                             // valid in neither
        return m12;                                                  // valid for lk = 0 only
    };
    float s = 0.f;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (trow(r, 0) < C) s += x[rt][r];                       // padded rows hold exact zeros
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / (float)(NT * C));
    float v = 0.f;
    f32x16 d[2];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (trow(r, 0) < C) {
                d[rt][r] = (x[rt][r] - mean) * valid(r);
                v = fmaf(d[rt][r], d[rt][r], v);
            }
    v += __shfl_xor(v, 32, 64);
    const float rstd = 1.0f / sqrtf(v * (1.0f / (float)(NT * C)) + 1e-5f);
    f32x16 w[2], b[2];
    load_vec(w, ln, lk);
    load_vec(b, ln + 64, lk);
    f32x16 ex[2];
    if (extra) load_vec(ex, extra, lk);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (rt < NT && trow(r, 0) < C) {
                float o = fmaf(d[rt][r] * rstd, w[rt][r], b[rt][r]);
                if (extra) o += ex[rt][r];
                y[rt][r] = o;
            } else {
                y[rt][r] = 0.f;
            }
        }
}

// exact GELU 0.5 v (1 + erf(v / sqrt 2)), erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = v * 0.70710678118654752f, ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);
    const float er = copysignf(fmaf(-p * t, e, 1.0f), x);
    return 0.5f * v * (1.0f + er);
}

template <int CTRL> __device__ __forceinline__ float quad(float v) {          // DPP quad permutation of the 4 tokens of a sample
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int M> __device__ __forceinline__ float tok_xor(float v) {
    if constexpr (M == 0) return v;
    else if constexpr (M == 1) return quad<0xB1>(v);      // [1,0,3,2]
    else if constexpr (M == 2) return quad<0x4E>(v);      // [2,3,0,1]
    else return quad<0x1B>(v);                            // [3,2,1,0]
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// x, z: (B, C, 8, 1).  ldj_acc[b] += H*W*log|det W| + sum logs + sum log_s.  hout (optional, tests): the conditioner's
// output un-patchified, (B, C, 8, 1) = [t | raw].
// DUMP (training forward at saturating batches): the residual stream at the depth + 1 layer boundaries goes to `xtape`,
// feature-major [boundary][feature < DIM][token < T] (token = 4 sample + n: 128 contiguous bytes per feature and half wave) -
// cf_vit_step_bwd_taped then walks back from these instead of running the six layers again.
template <class V, bool DUMP = false>
__global__ __launch_bounds__(256, 2) void k_vit_step(const float* __restrict__ x, float* __restrict__ z,
                                                     float* __restrict__ ldj_acc, const float* __restrict__ ws, int B,
                                                     int64_t xbs, int depth, float* __restrict__ hout,
                                                     float* __restrict__ xtape = nullptr, int64_t T = 0) {
    constexpr int C = V::C, CIN = V::CIN, HW = V::HW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int n = li & 3;                                   // token of the sample: positions 2n, 2n + 1
    const int smp = (blockIdx.x * 4 + wave) * 8 + (li >> 2);
    const bool live = smp < B;
    const float* xb = x + (int64_t)min(smp, B - 1) * xbs;
    const rsrc_t rs = make_rsrc(ws, ws_floats<V>(depth));

    // ================= Conv1x1 + ActNorm: [x0' | x1'] rows (ii, c), one k-step per input channel, lane half = position
    f32x16 y[2];
    load_vec(y, ws + V::OFF_B0, lk);
    {
        float xv[C];
#pragma unroll
        for (int k = 0; k < C; ++k) xv[k] = xb[k * HW + 2 * n + lk];
        gemm_regs<2, V::KS0>(y, rs, lane, V::OFF_A0, [&](int s) { return xv[s]; });
    }
    float* zb = z + (int64_t)smp * C * HW;
    if (live) {                                             // first half passes through (coupling.py:154)
#pragma unroll
        for (int r = 0; r < V::KPT; ++r) {
            const int p = trow(r, lk), ii = p / CIN, c = p - ii * CIN;
            if (p < C) zb[c * HW + 2 * n + ii] = y[0][r];
        }
    }
    // ================= patch embedding: LN(pd) -> Linear -> LN(dim) + pos        (simple_vit.py:100-105,122)
    f32x16 X[2];
    {
        f32x16 u[2];
        layernorm<V, 1>(y, u, ws + V::OFF_LN0, lk, nullptr);
        load_vec(X, ws + V::OFF_BE, lk);
        gemm_regs<2, V::KS_IN>(X, rs, lane, V::OFF_WE, [&](int s) { return u[0][s]; });
        layernorm<V, 2>(X, X, ws + V::OFF_LN1, lk, ws + V::OFF_POS + 64 * n);
    }
    auto dump = [&](int bnd) {
        if constexpr (DUMP) {
            float* tb = xtape + (int64_t)bnd * V::DIM * T + (int64_t)(blockIdx.x * 4 + wave) * 32 + li;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < V::KPT; ++r) {
                    const int f = feat_of_phys<V>(trow(r, lk) + 32 * t);
                    if (f >= 0) tb[(int64_t)f * T] = X[t][r];
                }
        }
    };
    // (boundary 0, the embedding output, is rebuilt by the backward kernel together with the statistics it needs: not taped)
    // ================= transformer                                                 (simple_vit.py:56-88)
#pragma unroll 1
    for (int l = 0; l < depth; ++l) {
        const int wl = V::OFF_LAYER + l * V::L_STRIDE;
        f32x16 o[2];
        {
            f32x16 u[2];
            layernorm<V, 2>(X, u, ws + wl + V::L_LNA, lk, nullptr);
            f32x16 qkv[6];
            gemm_regs<6, V::KS_RES, true>(qkv, rs, lane, wl + V::L_WQKV, [&](int s) { return u[s / V::KPT][s % V::KPT]; });
            // scores of this token against the 4 tokens of its sample (partner = token ^ m), exact softmax
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float q = qkv[t][r], k = qkv[2 + t][r];
                    d0 = fmaf(q, k, d0);
                    d1 = fmaf(q, tok_xor<1>(k), d1);
                    d2 = fmaf(q, tok_xor<2>(k), d2);
                    d3 = fmaf(q, tok_xor<3>(k), d3);
                }
            d0 += __shfl_xor(d0, 32, 64); d1 += __shfl_xor(d1, 32, 64);           // the lane halves hold different features
            d2 += __shfl_xor(d2, 32, 64); d3 += __shfl_xor(d3, 32, 64);
            const float scale = 0.125f;                                            // dim_head ** -0.5, dim_head = 64
            d0 *= scale; d1 *= scale; d2 *= scale; d3 *= scale;
            const float mx = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3));
            float p0 = __expf(d0 - mx), p1 = __expf(d1 - mx), p2 = __expf(d2 - mx), p3 = __expf(d3 - mx);
            const float inv = 1.0f / ((p0 + p1) + (p2 + p3));
            p0 *= inv; p1 *= inv; p2 *= inv; p3 *= inv;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = qkv[4 + t][r];
                    o[t][r] = fmaf(p3, tok_xor<3>(v), fmaf(p2, tok_xor<2>(v), fmaf(p1, tok_xor<1>(v), p0 * v)));
                }
        }
        {
            // the product is summed on its own and meets the residual stream in ONE addition per element, as in the
            // reference (x = to_out(...) + x, simple_vit.py:84).  Accumulating the k-steps on top of X rounds every partial
            // sum at the magnitude of the residual stream (|X| ~ 5 against |product| < 1): 56-64 roundings of ulp(X) per
            // element and block, which was 2-3x the reference's own fp32 error on the layer's output and - amplified by
            // 1 / sigma of a fitted prior - the whole distance of the SMAP "extreme" fixtures (tools/attribute_vit.py)
            f32x16 a[2];
            gemm_regs<2, V::KS_HEAD, true>(a, rs, lane, wl + V::L_WOUT, [&](int s) { return o[s >> 4][s & 15]; });
#pragma unroll
            for (int t = 0; t < 2; ++t) X[t] += a[t];
        }
        {
            f32x16 u[2], h[2];
            layernorm<V, 2>(X, u, ws + wl + V::L_LNF, lk, nullptr);
            load_vec(h, ws + wl + V::L_B1, lk);
            gemm_regs<2, V::KS_RES>(h, rs, lane, wl + V::L_W1, [&](int s) { return u[s / V::KPT][s % V::KPT]; });
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = (r < V::KPT) ? gelu_erf(h[t][r]) : 0.f;     // gelu(0) = 0 on padded rows anyway
            f32x16 a[2];
            load_vec(a, ws + wl + V::L_B2, lk);                                   // W2 h + b2 on its own, then one add into the residual
            gemm_regs<2, V::KS_RES>(a, rs, lane, wl + V::L_W2, [&](int s) { return h[s / V::KPT][s % V::KPT]; });
#pragma unroll
            for (int t = 0; t < 2; ++t) X[t] += a[t];
        }
        dump(l + 1);
    }
    f32x16 hn[2];
    layernorm<V, 2>(X, hn, ws + off_lno<V>(depth), lk, nullptr);                  // transformer.norm: tile 0 = t, tile 1 = raw

    // ================= affine map, log-det, stores                                 (coupling.py:139-155)
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < V::KPT; ++r) {
        const int p = trow(r, lk);                           // physical row of both tiles -> (position ii, channel c)
        const int ii = p / CIN, c = p - ii * CIN;
        if (p < C) {
            const float ls = 2.0f * tanhf(0.5f * hn[1][r]);
            const float z1 = fmaf(y[1][r], expf(ls), hn[0][r]);
            lsum += ls;
            if (live) {
                zb[(CIN + c) * HW + 2 * n + ii] = z1;
                if (hout) {
                    float* hb = hout + (int64_t)smp * C * HW;
                    hb[c * HW + 2 * n + ii] = hn[0][r];
                    hb[(CIN + c) * HW + 2 * n + ii] = hn[1][r];
                }
            }
        }
    }
    lsum += tok_xor<1>(lsum);
    lsum += tok_xor<2>(lsum);
    lsum += __shfl_xor(lsum, 32, 64);
    if (live && lk == 0 && n == 0) ldj_acc[smp] += ws[0] + lsum;
}

using VS26 = VS<26>;

bool step_ok(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads) {
    return C == 26 && H == 8 && W == 1 && p1 == 2 && p2 == 1 && dim == 2 * C && dim_head == 64 && heads == 1;
}

}  // namespace

extern "C" {

int cf_vit_step_supported(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads) {
    return step_ok(C, H, W, p1, p2, dim, dim_head, heads) ? 1 : 0;
}

int64_t cf_vit_step_ws_bytes(int C, int depth) { return C == 26 ? (int64_t)ws_floats<VS26>(depth) * 4 : 0; }

int cf_vit_step_prepare(const float* Wm, const float* t, const float* logs, const float* flat_vit_params, const float* pos,
                        void* ws, int C, int depth, cf_stream_t stream) {
    CF_REQUIRE(Wm && t && logs && flat_vit_params && pos && ws && depth >= 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0);
    if (C != 26) { cf_set_error("cf_vit_step_prepare: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    float* w = (float*)ws;
    int rc = cf_slogdet_inverse(Wm, C, w + 1, nullptr, stream);
    if (rc) return rc;
    k_vit_step_pack<VS26><<<dim3(64), dim3(256), 0, cf_s(stream)>>>(Wm, t, logs, flat_vit_params, pos, w, depth);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_vit_step_fwd(const float* x, float* z, float* ldj_acc, const void* ws, float* h_out, int B, int C, int depth,
                    int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && B > 0 && x_bstride >= (int64_t)C * 8);
    if (C != 26) { cf_set_error("cf_vit_step_fwd: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    k_vit_step<VS26><<<dim3((unsigned)((B + 31) / 32)), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, (const float*)ws, B,
                                                                                    x_bstride, depth, h_out);
    CF_LAUNCH_CHECK();
    return 0;
}

// training forward at saturating batches: cf_vit_step_fwd that also writes the residual stream at the layer boundaries -
// xtape: cf_vit_step_tape_floats(B, C, depth) floats, [depth + 1][2C][T] with T = 4 * (B rounded up to 32) tokens.
int64_t cf_vit_step_tape_tokens(int B) { return 4ll * ((B + 31) / 32 * 32); }
int64_t cf_vit_step_tape_floats(int B, int C, int depth) { return C == 26 ? (int64_t)(depth + 1) * 2 * C * cf_vit_step_tape_tokens(B) : 0; }

int cf_vit_step_fwd_taped(const float* x, float* z, float* ldj_acc, const void* ws, float* xtape, int B, int C, int depth,
                          int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && xtape && B > 0 && x_bstride >= (int64_t)C * 8);
    if (C != 26) { cf_set_error("cf_vit_step_fwd_taped: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    k_vit_step<VS26, true><<<dim3((unsigned)((B + 31) / 32)), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, (const float*)ws, B,
                                                                                          x_bstride, depth, nullptr, xtape,
                                                                                          cf_vit_step_tape_tokens(B));
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
