// Shared device code of the row-split transformer-coupling step kernels (cf_vit_rs.hip: forward for small batches,
// cf_vit_rs_bwd.hip: the training backward): geometry, workspace layout, packing helpers, 16x16x4 product helpers, token
// statistics.  Everything lives in an anonymous namespace of the including translation unit.
#pragma once
#include "cf_common.h"
#include "cf_vit_fuse.h"
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__host__ __device__ constexpr int ngrp(int ks) { return (ks + 3) / 4; }

template <int C_> struct RS {
    static constexpr int C = C_, CIN = C / 2, HW = 8, NTOK = 4, DIM = 2 * C, PD = 2 * CIN, HEAD = 64;
    static constexpr int TOK = 16, SPW = 4, POSC = SPW * HW;              // token / position columns of a workgroup
    static constexpr int KS_C = (C + 3) / 4, KS_PD = (PD + 3) / 4, KS_D = (DIM + 3) / 4, KS_H = HEAD / 4;
    static constexpr int NG_C = ngrp(KS_C), NG_PD = ngrp(KS_PD), NG_D = ngrp(KS_D), NG_H = ngrp(KS_H);
    // workspace (floats)
    static constexpr int OFF_B0 = 4, OFF_A0 = OFF_B0 + 32;                                  // conv bias' (32 rows), frags 2 tiles
    static constexpr int OFF_WE = OFF_A0 + 2 * NG_C * 256, OFF_BE = OFF_WE + 4 * NG_PD * 256;   // embed (LN0 folded), bias' 64
    static constexpr int OFF_LN1 = OFF_BE + 64, OFF_POS = OFF_LN1 + 128, OFF_LAYER = OFF_POS + NTOK * 64;
    static constexpr int L_WQKV = 0, L_CQKV = L_WQKV + 12 * NG_D * 256, L_WOUT = L_CQKV + 192;
    static constexpr int L_W1 = L_WOUT + 4 * NG_H * 256, L_B1 = L_W1 + 4 * NG_D * 256, L_W2 = L_B1 + 64, L_B2 = L_W2 + 4 * NG_D * 256;
    static constexpr int L_STRIDE = L_B2 + 64;
    // fused attention tables of the FORWARD kernel (round 4, as in cf_vit_step.hip): per layer A1 = diag(g) Wk^T Wq diag(g) / 8,
    // c1 = diag(g) Wk^T Wq b / 8 (scores s_ij = (A1 n_i + c1) . n_j) and A2 = Wout Wv diag(g), c2 = Wout Wv b (value / output pair),
    // g / b = the attention pre-norm's affine part; they sit BEHIND the tables above, which the backward kernel keeps using
    static constexpr int F_A1 = 0, F_C1 = F_A1 + 4 * NG_D * 256, F_A2 = F_C1 + 64, F_C2 = F_A2 + 4 * NG_D * 256, F_STRIDE = F_C2 + 64;
    // LDS planes (floats)
    static constexpr int P_XIN = 0, P_Y = P_XIN + 4 * KS_C * POSC, P_X0 = P_Y + 32 * POSC, P_X1 = P_X0 + 64 * TOK;
    static constexpr int P_O = P_X1 + 64 * TOK, P_H = P_O + 64 * TOK, P_SC = P_H + 64 * TOK, P_LS = P_SC + 4 * 4 * TOK;
    static constexpr int LDS_FLOATS = P_LS + CIN * 2 * TOK;
    static_assert(C % 2 == 0 && C <= 32 && DIM <= 64 && KS_D <= 16, "C even, <= 32");
};
template <class V> __host__ __device__ constexpr int off_lno(int depth) { return V::OFF_LAYER + depth * V::L_STRIDE; }
template <class V> __host__ __device__ constexpr int off_fused(int depth) { return off_lno<V>(depth) + 128; }
template <class V> __host__ __device__ constexpr int off_fuse_scratch(int depth) { return off_fused<V>(depth) + depth * V::F_STRIDE; }     // k_vit_fuse output
template <class V> __host__ __device__ constexpr int ws_floats(int depth) { return off_fuse_scratch<V>(depth) + depth * VitFuse<V::DIM, V::HEAD>::LAYER_FLOATS; }

// ---- packing ----------------------------------------------------------------------------------------------------------
// 16x16x4 A fragments of a Linear W (N x K, row-major): element ((rt * NG + gi) * 64 + lane) * 4 + e =
// W[16 rt + (lane & 15)][4 (4 gi + e) + (lane >> 4)] * gamma[k]   (gamma: the preceding LayerNorm's weight, or none)
struct VitRsPackBatch {                                       // per flow step of a batch (blockIdx.y)
    const float *Wm[kVitPrepBatch], *t[kVitPrepBatch], *logs[kVitPrepBatch], *flat[kVitPrepBatch];
    float* ws[kVitPrepBatch];
};
template <class V>
__global__ __launch_bounds__(256) void k_vit_rs_pack(const VitRsPackBatch pb, const float* __restrict__ pos, int depth) {
    const float* __restrict__ Wm = pb.Wm[blockIdx.y]; const float* __restrict__ t = pb.t[blockIdx.y];
    const float* __restrict__ logs = pb.logs[blockIdx.y]; const float* __restrict__ flat = pb.flat[blockIdx.y];
    float* __restrict__ ws = pb.ws[blockIdx.y];
    constexpr int C = V::C, DIM = V::DIM, PD = V::PD, HEAD = V::HEAD;
    using F = VitFuse<DIM, HEAD>;
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    auto frags_fn = [&](float* dst, int N, int K, int tiles, int ng, auto val) {             // as frags, element (row, k) = val(row, k), fp64
        for (int i = gtid; i < tiles * ng * 256; i += gsz) {
            const int e = i & 3, lane = (i >> 2) & 63, q = i >> 8, gi = q % ng, rt = q / ng;
            const int row = 16 * rt + (lane & 15), k = 4 * (4 * gi + e) + (lane >> 4);
            dst[i] = (row < N && k < K) ? (float)val(row, k) : 0.f;
        }
    };
    auto frags = [&](float* dst, const float* W, int N, int K, int tiles, int ng, const float* gamma) {
        for (int i = gtid; i < tiles * ng * 256; i += gsz) {
            const int e = i & 3, lane = (i >> 2) & 63, q = i >> 8, gi = q % ng, rt = q / ng;
            const int row = 16 * rt + (lane & 15), k = 4 * (4 * gi + e) + (lane >> 4);
            dst[i] = (row < N && k < K) ? W[row * K + k] * (gamma ? gamma[k] : 1.0f) : 0.f;
        }
    };
    // bias'[row] = bias[row] + sum_k W[row][k] beta[k]  (beta: the preceding LayerNorm's bias), rows >= N: 0.  A wave per
    // row, lanes along k (coalesced), fp64 partial sums combined in a fixed order
    auto fold_bias = [&](float* dst, const float* W, const float* bias, const float* beta, int N, int K, int rows) {
        const int lane = threadIdx.x & 63, gw = gtid >> 6, nw = gsz >> 6;
        for (int r = gw; r < rows; r += nw) {
            double s = 0.0;
            if (r < N && beta) for (int k = lane; k < K; k += 64) s += (double)W[r * K + k] * (double)beta[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) dst[r] = (float)(s + ((r < N && bias) ? (double)bias[r] : 0.0));
        }
    };
    auto vec = [&](float* dst, const float* src, int n, int rows) {
        for (int r = gtid; r < rows; r += gsz) dst[r] = r < n ? src[r] : 0.f;
    };
    if (gtid == 0) {       // ldj_const = H*W*log|det Wm| + sum_c logs  (ws[1] holds log|det| from cf_slogdet_inverse)
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += logs[c];
        ws[0] = (float)V::HW * ws[1] + s;                                   // conv1x1.py:53 + actnorm.py:58
    }
    // Conv1x1 + ActNorm: y = e^{-logs} (Wm x - t)                          (conv1x1.py:54, actnorm.py:59)
    for (int r = gtid; r < 32; r += gsz) ws[V::OFF_B0 + r] = r < C ? -t[r] * expf(-logs[r]) : 0.f;
    for (int i = gtid; i < 2 * V::NG_C * 256; i += gsz) {
        const int e = i & 3, lane = (i >> 2) & 63, q = i >> 8, gi = q % V::NG_C, rt = q / V::NG_C;
        const int row = 16 * rt + (lane & 15), k = 4 * (4 * gi + e) + (lane >> 4);
        ws[V::OFF_A0 + i] = (row < C && k < C) ? expf(-logs[row]) * Wm[row * C + k] : 0.f;
    }
    const float* p = flat;       // order of TransCoupling._flat_params(): see cf_vit_fused.hip
    const float *g0 = p, *b0 = p + PD; p += 2 * PD;                                          // to_patch_embedding.1 (LayerNorm(pd))
    frags(ws + V::OFF_WE, p, DIM, PD, 4, V::NG_PD, g0);
    fold_bias(ws + V::OFF_BE, p, p + DIM * PD, b0, DIM, PD, 64); p += DIM * PD + DIM;
    vec(ws + V::OFF_LN1, p, DIM, 64); vec(ws + V::OFF_LN1 + 64, p + DIM, DIM, 64); p += 2 * DIM;
    for (int n = 0; n < V::NTOK; ++n) vec(ws + V::OFF_POS + 64 * n, pos + n * DIM, DIM, 64);
    for (int l = 0; l < depth; ++l) {
        float* w = ws + V::OFF_LAYER + l * V::L_STRIDE;
        const float *ga = p, *ba = p + DIM; p += 2 * DIM;                                    // attention pre-norm
        {   // fused attention tables of the forward kernel (cf_vit_fuse.h: k_vit_fuse has formed the matrices in fp64)
            float* wf = ws + off_fused<V>(depth) + l * V::F_STRIDE;
            const float* fz = ws + off_fuse_scratch<V>(depth) + l * F::LAYER_FLOATS;
            frags_fn(wf + V::F_A1, DIM, DIM, 4, V::NG_D, [&](int a, int b) { return fz[F::M1 + a * DIM + b]; });
            frags_fn(wf + V::F_A2, DIM, DIM, 4, V::NG_D, [&](int f, int b) { return fz[F::M2 + f * DIM + b]; });
            for (int r = gtid; r < 64; r += gsz) {
                wf[V::F_C1 + r] = r < DIM ? fz[F::C1 + r] : 0.f;
                wf[V::F_C2 + r] = r < DIM ? fz[F::C2 + r] : 0.f;
            }
        }
        frags(w + V::L_WQKV, p, 192, DIM, 12, V::NG_D, ga);
        fold_bias(w + V::L_CQKV, p, nullptr, ba, 192, DIM, 192); p += 192 * DIM;
        frags(w + V::L_WOUT, p, DIM, 64, 4, V::NG_H, nullptr); p += DIM * 64;
        const float *gf = p, *bf = p + DIM; p += 2 * DIM;                                    // feed-forward pre-norm
        frags(w + V::L_W1, p, DIM, DIM, 4, V::NG_D, gf);
        fold_bias(w + V::L_B1, p, p + DIM * DIM, bf, DIM, DIM, 64); p += DIM * DIM + DIM;
        frags(w + V::L_W2, p, DIM, DIM, 4, V::NG_D, nullptr);
        vec(w + V::L_B2, p + DIM * DIM, DIM, 64); p += DIM * DIM + DIM;
    }
    vec(ws + off_lno<V>(depth), p, DIM, 64); vec(ws + off_lno<V>(depth) + 64, p + DIM, DIM, 64);
}

// ---- device helpers -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ rsrc_t make_rsrc(const float* ws, int floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, floats * 4, 0x00020000);
}
__device__ __forceinline__ float4 frag(rsrc_t rs, int lane, int foff) {
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, foff * 4, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
__device__ __forceinline__ f32x4 to4(const float4& v) { return f32x4{v.x, v.y, v.z, v.w}; }
template <int NG> __device__ __forceinline__ void load_frags(float4 (&f)[NG], rsrc_t rs, int lane, int foff) {
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) f[gi] = frag(rs, lane, foff + gi * 256);
}
// rows 16 w + 4 g .. + 3 of a 64-float vector
__device__ __forceinline__ f32x4 vec4(const float* __restrict__ v, int w, int g) {
    return to4(*reinterpret_cast<const float4*>(v + 16 * w + 4 * g));
}
// acc += A (NKS k-steps of the fragments) * B, B operand of k-step s = bop(s); two chains over even / odd k-steps
template <int NKS, int NG, class BOP>
__device__ __forceinline__ f32x4 gemm1(f32x4 init, const float4 (&fr)[NG], BOP bop) {
    f32x4 a0 = init, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const float a = f4e(fr[s >> 2], s & 3), b = bop(s);
        if (s & 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
    }
    return a0 + a1;
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
__device__ __forceinline__ float group_sum(float v) {      // over the four lane groups g (same token column)
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
// token statistics of a [rows][16] plane over its first n rows: lane group g reads rows g, g + 4, ... (kept in xv: they
// are this lane's B operands of a product over those rows).  Two passes, biased variance, eps 1e-5 (torch.nn.LayerNorm).
template <int NR>
__device__ __forceinline__ void token_stats(const float (&xv)[NR], int n, int g, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) s += (g + 4 * i < n) ? xv[i] : 0.f;
    mean = group_sum(s) * (1.0f / (float)n);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) { const float d = (g + 4 * i < n) ? xv[i] - mean : 0.f; q = fmaf(d, d, q); }
    rstd = 1.0f / sqrtf(group_sum(q) * (1.0f / (float)n) + 1e-5f);
}


}  // namespace
