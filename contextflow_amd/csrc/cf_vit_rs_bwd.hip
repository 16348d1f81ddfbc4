// Backward of one transformer-coupling flow step (Conv1x1 -> ActNorm -> TransCoupling with its SimpleViT conditioner) as
// ONE kernel + ONE grouped weight-gradient launch: the caller right after the path in the anomaly-detection experiments,
// `cost.backward()` (contextflow/experiment_ad.py:204-213; layers/coupling.py:123-155, layers/simple_vit.py:18-127,
// layers/conv1x1.py:52-57, layers/actnorm.py:53-60).  Round 2 trained these flows layer by layer: ~260 launches per step.
//
// Shape: the row-split workgroup of cf_vit_rs.hip - four waves share 16 token columns (4 samples) and split the output
// rows of every product, planes [feature][16 tokens] in LDS (52.7 KB: three workgroups per CU).  Nothing but the step INPUT is
// kept from the forward:
//   phase A  re-runs the step (the forward kernel's arithmetic) and parks the residual stream at every layer boundary in
//            LDS (the owner wave's tile only: 16 B per lane and layer);
//   phase B  walks back: epilogue (affine map, log-det), final LayerNorm, then per layer - recompute q / k / v, softmax, the
//            attention output, the mid-layer residual, the MLP pre-activation from the parked input; data gradients with
//            TRANSPOSED weight fragments (k_vit_rs_pack_bwd); LayerNorm / softmax / GELU adjoints on the owner's tile, the
//            per-token sums of the LayerNorm adjoints through a two-value exchange between the waves - then the patch
//            embedding and the Conv1x1 + ActNorm product.
// Parameter gradients: every Linear's two operands (its input u and the gradient of its output) leave the kernel as
// token-major planes, coalesced (lane = feature), and ONE launch of the grouped split-K GEMM (cf_linear_wgrad_group,
// cf_vit.hip) contracts all 26 pairs of a step over the tokens; LayerNorm weight / bias gradients are summed over a
// workgroup's 16 tokens in registers and leave as one partial per workgroup (fixed-order reduction afterwards: no float
// atomics, bitwise reproducible).
#include "cf_vit_rs_common.h"
#include <atomic>

namespace {

template <class V> struct RSB {
    static constexpr int MAXD = 6;                                   // transformer depth the LDS budget is sized for
    static constexpr int NG_Q = ngrp(48);                            // qkv^T: contraction over 192 rows
    // backward workspace (floats): transposed fragments + the LayerNorm vectors the forward workspace folds away
    static constexpr int OFF_CT = 0, OFF_G0 = OFF_CT + 2 * V::NG_C * 256, OFF_BT0 = OFF_G0 + 32;
    static constexpr int OFF_WET = OFF_BT0 + 32, OFF_LAYER = OFF_WET + 2 * V::NG_D * 256;
    static constexpr int LB_GA = 0, LB_BA = 64, LB_WQKVT = 128, LB_WOUTT = LB_WQKVT + 4 * NG_Q * 256;
    static constexpr int LB_GF = LB_WOUTT + 4 * V::NG_D * 256, LB_BF = LB_GF + 64, LB_W1T = LB_BF + 64;
    static constexpr int LB_W2T = LB_W1T + 4 * V::NG_D * 256, LB_STRIDE = LB_W2T + 4 * V::NG_D * 256;
    // LDS (floats).  Planes [feature][TS] with TS = 17: the transposed (lane = feature) global stores read them conflict-free
    static constexpr int TS = 17, PS = 33, PL = 64 * TS;
    // 52.7 KB: three workgroups per CU.  The gradient plane of q | k | v (192 rows) lives on the three planes that are dead by
    // then (layer input / LayerNorm output / MLP hidden layer), the step input on the attention-output plane (dead after
    // the Conv1x1 product); the residual stream is parked for the layer boundaries 1 .. depth - 1 only (boundary 0 is
    // rebuilt from the embedding, the last one is consumed where it is produced).
    static constexpr int P_Y = 0, P_GY = P_Y + 32 * PS;
    static constexpr int P_A = P_GY + 32 * PS, P_B = P_A + PL, P_H = P_B + PL, P_O = P_H + PL, P_G = P_O + PL;
    static constexpr int P_Q = P_A, P_XIN = P_O;
    static constexpr int P_SC = P_G + PL, P_XS = (P_SC + 2 * 256 + 3) & ~3;            // SC: two halves of [4 waves][4][16]
    static constexpr int LDS_FLOATS = P_XS + (MAXD - 1) * 1024;
    static_assert(4 * V::KS_C * PS <= PL && 192 * TS <= 3 * PL, "aliased planes fit");
    // LayerNorm partial sums of one workgroup (floats): [gamma | beta] per LayerNorm
    static constexpr int LN_0 = 0, LN_1 = 64, LN_L = 192, LN_LSTRIDE = 256;       // per layer: [ga | ba | gf | bf]
    __host__ __device__ static constexpr int ln_final(int depth) { return LN_L + depth * LN_LSTRIDE; }
    __host__ __device__ static constexpr int ln_floats(int depth) { return ln_final(depth) + 128; }
    // token-major planes (floats per token row): per layer [u1 52 | gqkv 192 | o 64 | gxm 52 | u2 52 | ghp 52 | h 52 | gxo 52]
    static constexpr int TL_U1 = 0, TL_GQKV = 52, TL_O = 244, TL_GXM = 308, TL_U2 = 360, TL_GHP = 412, TL_H = 464, TL_GXO = 516, TL = 568;
};
template <class V> __host__ __device__ constexpr int wsb_floats(int depth) { return RSB<V>::OFF_LAYER + depth * RSB<V>::LB_STRIDE; }

// transposed 16x16x4 A fragments: element ((rt * NG + gi) * 64 + lane) * 4 + e = W[n][k] with k = 16 rt + (lane & 15)
// (output row of the transposed product) and n = 4 (4 gi + e) + (lane >> 4) (its contraction index)
struct VitRsPackBwdBatch {                                    // per flow step of a batch (blockIdx.y)
    const float *Wm[kVitPrepBatch], *logs[kVitPrepBatch], *flat[kVitPrepBatch];
    float* wsb[kVitPrepBatch];
};
template <class V>
__global__ __launch_bounds__(256) void k_vit_rs_pack_bwd(const VitRsPackBwdBatch pb, int depth) {
    const float* __restrict__ Wm = pb.Wm[blockIdx.y]; const float* __restrict__ logs = pb.logs[blockIdx.y];
    const float* __restrict__ flat = pb.flat[blockIdx.y];
    float* __restrict__ wsb = pb.wsb[blockIdx.y];
    using R = RSB<V>;
    constexpr int C = V::C, DIM = V::DIM, PD = V::PD;
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    auto fragsT = [&](float* dst, const float* W, int N, int K, int tiles, int ng) {
        for (int i = gtid; i < tiles * ng * 256; i += gsz) {
            const int e = i & 3, lane = (i >> 2) & 63, q = i >> 8, gi = q % ng, rt = q / ng;
            const int k = 16 * rt + (lane & 15), n = 4 * (4 * gi + e) + (lane >> 4);
            dst[i] = (k < K && n < N) ? W[n * K + k] : 0.f;
        }
    };
    auto vec = [&](float* dst, const float* src, int n, int rows) {
        for (int r = gtid; r < rows; r += gsz) dst[r] = r < n ? src[r] : 0.f;
    };
    for (int i = gtid; i < 2 * V::NG_C * 256; i += gsz) {             // (e^{-logs} Wm)^T
        const int e = i & 3, lane = (i >> 2) & 63, q = i >> 8, gi = q % V::NG_C, rt = q / V::NG_C;
        const int k = 16 * rt + (lane & 15), ch = 4 * (4 * gi + e) + (lane >> 4);
        wsb[R::OFF_CT + i] = (k < C && ch < C) ? expf(-logs[ch]) * Wm[ch * C + k] : 0.f;
    }
    const float* p = flat;
    vec(wsb + R::OFF_G0, p, PD, 32); vec(wsb + R::OFF_BT0, p + PD, PD, 32); p += 2 * PD;
    fragsT(wsb + R::OFF_WET, p, DIM, PD, 2, V::NG_D); p += DIM * PD + DIM;
    p += 2 * DIM;                                                     // to_patch_embedding.3: in the forward workspace
    for (int l = 0; l < depth; ++l) {
        float* w = wsb + R::OFF_LAYER + l * R::LB_STRIDE;
        vec(w + R::LB_GA, p, DIM, 64); vec(w + R::LB_BA, p + DIM, DIM, 64); p += 2 * DIM;
        fragsT(w + R::LB_WQKVT, p, 192, DIM, 4, R::NG_Q); p += 192 * DIM;
        fragsT(w + R::LB_WOUTT, p, DIM, 64, 4, V::NG_D); p += DIM * 64;
        vec(w + R::LB_GF, p, DIM, 64); vec(w + R::LB_BF, p + DIM, DIM, 64); p += 2 * DIM;
        fragsT(w + R::LB_W1T, p, DIM, DIM, 4, V::NG_D); p += DIM * DIM + DIM;
        fragsT(w + R::LB_W2T, p, DIM, DIM, 4, V::NG_D); p += DIM * DIM + DIM;
    }
}

__device__ __forceinline__ float row_sum16(float v) {                 // over the 16 token columns of a lane group
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v + __shfl_xor(v, 8, 64);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float x) {                    // d/dx of the exact GELU
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// ---- the backward kernel --------------------------------------------------------------------------------------------
// x: step input (B, C, 8, 1), batch stride xbs; gz: dL/dz (B, C, 8, 1) dense; gld: dL/d(log-det) (B,); gx: dL/dx dense.
// ws: forward workspace of cf_vit_step_rs_prepare; wsb: k_vit_rs_pack_bwd.  tp: token-major planes (layout: RSB::TL_*,
// see cf_vit_step_bwd_plane_floats); lnp: one row of LayerNorm partial sums per workgroup.
// TAPED: the residual stream at the layer boundaries comes from the training forward (cf_vit_step_fwd_taped: xtape
// [depth + 1][DIM][T], feature-major) - phase A then runs the Conv1x1 / ActNorm / patch embedding only (their statistics and
// planes are needed on the way back) and skips the six layers, a quarter of this kernel's work.
template <class V, bool TAPED = false>
__global__ __launch_bounds__(256, 3) void k_vit_step_bwd_rs(const float* __restrict__ x, const float* __restrict__ gz,
                                                            const float* __restrict__ gld, float* __restrict__ gx,
                                                            const float* __restrict__ ws, const float* __restrict__ wsb,
                                                            float* __restrict__ tp, float* __restrict__ lnp, int B, int Bp,
                                                            int64_t xbs, int depth, const float* __restrict__ xtape = nullptr,
                                                            int64_t T = 0) {
    using R = RSB<V>;
    constexpr int C = V::C, CIN = V::CIN, HW = V::HW, DIM = V::DIM, PD = V::PD, TS = R::TS, PS = R::PS;
    extern __shared__ __align__(16) float lds[];
    float* XIN = lds + R::P_XIN;   // [4 KS_C][PS]  step input, channel-major
    float* YP = lds + R::P_Y;      // [32][PS]      Conv1x1 + ActNorm output
    float* GYP = lds + R::P_GY;    // [32][PS]      gradient w.r.t. it
    float* PA = lds + R::P_A;      // residual stream at the layer boundary (operand of the qkv product / statistics)
    float* PB = lds + R::P_B;      // LayerNorm outputs u (weight-gradient operands), mid-layer residual
    float* PO = lds + R::P_O;      // attention output; later the gradient of the MLP pre-activation
    float* PH = lds + R::P_H;      // MLP hidden layer; the conditioner output in the epilogue
    float* PG = lds + R::P_G;      // gradient of the residual stream
    float* PQ = lds + R::P_Q;      // [192][TS] gradient of q | k | v
    float* SC = lds + R::P_SC;     // [2][4 waves][2 .. 4][16] exchange between the waves (ping-pong)
    float* XS = lds + R::P_XS;     // [depth - 1][256 threads][4] the owner's tile of the residual stream at the layer boundaries 1 ..
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s0 = blockIdx.x * V::SPW, tok0 = blockIdx.x * V::TOK, pos0 = blockIdx.x * V::POSC;
    const int n = col & 3, sl = col >> 2, pcol = sl * HW + 2 * n;
    const int64_t R4 = 4 * (int64_t)Bp, P8 = 8 * (int64_t)Bp;
    float* tXT = tp;                                   // (P8, C)   x, position-major
    float* tGYT = tp + P8 * C;                         // (P8, C)   gradient of the Conv1x1 + ActNorm output
    float* tU0 = tp + 2 * P8 * C;                      // (R4, PD)  LayerNorm(pd) output
    float* tGE = tU0 + R4 * PD;                        // (R4, DIM) gradient of the embedding Linear's output
    float* tL = tGE + R4 * DIM;                        // per layer: R4 * TL floats
    float* lnw = lnp + (int64_t)blockIdx.x * R::ln_floats(depth);
    const rsrc_t rs = make_rsrc(ws, ws_floats<V>(depth));
    const rsrc_t rb = make_rsrc(wsb, wsb_floats<V>(depth));
    bool valid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) valid[r] = 16 * w + 4 * g + r < DIM;
    int scp = 0;                                       // ping-pong half of SC

    auto put = [&](float* P, const f32x4& v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(16 * w + 4 * g + r) * TS + col] = v[r];
    };
    auto get = [&](const float* P) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = P[(16 * w + 4 * g + r) * TS + col];
        return v;
    };
    auto rows13 = [&](const float* P, float (&xv)[V::KS_D]) {
#pragma unroll
        for (int i = 0; i < V::KS_D; ++i) xv[i] = P[(g + 4 * i) * TS + col];
    };
    // plane [nrows][TS] -> token-major rows of `dst` (row stride ld): lane = feature (coalesced, conflict-free LDS reads),
    // wave w takes tokens w, w + 4, ..; no integer divisions
    auto store_T = [&](const float* P, int nrows, float* dst, int ld) {
        for (int f0 = 0; f0 < nrows; f0 += 64) {
            const int f = f0 + lane;
            if (f < nrows) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = w + 4 * j;
                    dst[(int64_t)(tok0 + t) * ld + f] = P[f * TS + t];
                }
            }
        }
    };
    auto masked = [&](const f32x4& v) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = valid[r] ? v[r] : 0.f;
        return o;
    };
    // sums over the token's features of two per-row values held by the owner waves: lane groups, then the four waves
    auto exchange2 = [&](float& s1, float& s2) {
        s1 = group_sum(s1); s2 = group_sum(s2);
        float* sc = SC + scp * 256;
        if (g == 0) { sc[(w * 2 + 0) * 16 + col] = s1; sc[(w * 2 + 1) * 16 + col] = s2; }
        __syncthreads();
        s1 = (sc[0 * 16 + col] + sc[2 * 16 + col]) + (sc[4 * 16 + col] + sc[6 * 16 + col]);
        s2 = (sc[1 * 16 + col] + sc[3 * 16 + col]) + (sc[5 * 16 + col] + sc[7 * 16 + col]);
        scp ^= 1;
    };
    // adjoint of y = LayerNorm-normalise(x) over nf features: ghat = gradient w.r.t. the normalised values (already times
    // gamma), uh = the normalised values (both zero on padding rows)
    auto ln_bwd = [&](const f32x4& ghat, const f32x4& uh, float rstd, int nf) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1 += ghat[r]; s2 = fmaf(ghat[r], uh[r], s2); }
        exchange2(s1, s2);
        const float m1 = s1 * (1.0f / (float)nf), m2 = s2 * (1.0f / (float)nf);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (16 * w + 4 * g + r < nf) ? rstd * (ghat[r] - m1 - uh[r] * m2) : 0.f;
        return o;
    };
    // LayerNorm weight / bias gradient of this workgroup's 16 tokens: rows of the owner's tile
    auto ln_partial = [&](int off, const f32x4& gu, const f32x4& uh) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = row_sum16(gu[r] * uh[r]), b = row_sum16(gu[r]);
            if (col == 0) { lnw[off + 16 * w + 4 * g + r] = a; lnw[off + 64 + 16 * w + 4 * g + r] = b; }
        }
    };
    auto normalised = [&](const f32x4& X, float mean, float rstd) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = valid[r] ? (X[r] - mean) * rstd : 0.f;
        return o;
    };
    auto xs_put = [&](int l, const f32x4& v) { *reinterpret_cast<float4*>(&XS[((l - 1) * 256 + tid) * 4]) = make_float4(v[0], v[1], v[2], v[3]); };
    auto xs_get = [&](int l) { return to4(*reinterpret_cast<const float4*>(&XS[((l - 1) * 256 + tid) * 4])); };   // l >= 1

    // fragments of one layer's forward products (for the recompute) - requested one product ahead
    float4 fqkv[3][V::NG_D], fout[V::NG_H], ffc1[V::NG_D], ffc2[V::NG_D];
    auto load_qkv = [&](int l) {
#pragma unroll
        for (int t = 0; t < 3; ++t) load_frags(fqkv[t], rs, lane, V::OFF_LAYER + l * V::L_STRIDE + V::L_WQKV + (4 * t + w) * V::NG_D * 256);
    };
    auto layer_front = [&](int l, float& mean, float& rstd, f32x4& q, f32x4& k, f32x4& v, float (&p)[4], f32x4& o) {
        // residual plane PA -> statistics, q / k / v (owner rows), softmax over the sample's 4 tokens, attention output
        const int wl = V::OFF_LAYER + l * V::L_STRIDE;
        float xv[V::KS_D];
        rows13(PA, xv);
        token_stats(xv, DIM, g, mean, rstd);
        const float mr = -mean * rstd;
        q = vec4(ws + wl + V::L_CQKV, w, g); k = vec4(ws + wl + V::L_CQKV + 64, w, g); v = vec4(ws + wl + V::L_CQKV + 128, w, g);
#pragma unroll
        for (int s = 0; s < V::KS_D; ++s) {
            const float b = fmaf(xv[s], rstd, mr);
            q = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(fqkv[0][s >> 2], s & 3), b, q, 0, 0, 0);
            k = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(fqkv[1][s >> 2], s & 3), b, k, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(fqkv[2][s >> 2], s & 3), b, v, 0, 0, 0);
        }
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d0 = fmaf(q[r], k[r], d0);
            d1 = fmaf(q[r], tok_xor<1>(k[r]), d1);
            d2 = fmaf(q[r], tok_xor<2>(k[r]), d2);
            d3 = fmaf(q[r], tok_xor<3>(k[r]), d3);
        }
        d0 = group_sum(d0); d1 = group_sum(d1); d2 = group_sum(d2); d3 = group_sum(d3);
        float* sc = SC + scp * 256;
        if (g == 0) {
            sc[(w * 4 + 0) * 16 + col] = d0; sc[(w * 4 + 1) * 16 + col] = d1; sc[(w * 4 + 2) * 16 + col] = d2; sc[(w * 4 + 3) * 16 + col] = d3;
        }
        __syncthreads();
        auto score = [&](int m) { return ((sc[(0 + m) * 16 + col] + sc[(4 + m) * 16 + col]) + (sc[(8 + m) * 16 + col] + sc[(12 + m) * 16 + col])) * 0.125f; };
        d0 = score(0); d1 = score(1); d2 = score(2); d3 = score(3);
        scp ^= 1;
        const float mx = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3));
        p[0] = expf(d0 - mx); p[1] = expf(d1 - mx); p[2] = expf(d2 - mx); p[3] = expf(d3 - mx);
        const float inv = 1.0f / ((p[0] + p[1]) + (p[2] + p[3]));
        p[0] *= inv; p[1] *= inv; p[2] *= inv; p[3] *= inv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o[r] = fmaf(p[3], tok_xor<3>(v[r]), fmaf(p[2], tok_xor<2>(v[r]), fmaf(p[1], tok_xor<1>(v[r]), p[0] * v[r])));
    };

    // ================================================ phase A: the step again, residual stream parked per layer
    float4 fconv[V::NG_C], femb[V::NG_PD];
    load_frags(fconv, rs, lane, V::OFF_A0 + (w & 1) * V::NG_C * 256);
    load_frags(femb, rs, lane, V::OFF_WE + w * V::NG_PD * 256);
    load_qkv(0);
    load_frags(fout, rs, lane, V::OFF_LAYER + V::L_WOUT + w * V::NG_H * 256);
    load_frags(ffc1, rs, lane, V::OFF_LAYER + V::L_W1 + w * V::NG_D * 256);
    load_frags(ffc2, rs, lane, V::OFF_LAYER + V::L_W2 + w * V::NG_D * 256);
    for (int i = tid; i < 4 * V::KS_C * V::POSC; i += 256) {
        const int c = i / V::POSC, pc = i % V::POSC, b = s0 + pc / HW;
        XIN[c * PS + pc] = (c < C && b < B) ? x[(int64_t)b * xbs + c * HW + pc % HW] : 0.f;
    }
    for (int i = tid; i < 6 * PS; i += 256) GYP[26 * PS + i] = 0.f;            // channel rows past C: zero operands
    for (int i = tid; i < 12 * TS; i += 256) PG[52 * TS + i] = 0.f;
    __syncthreads();
    for (int it = tid; it < V::POSC * C; it += 256) {                          // x, position-major (operand of the Conv1x1 weight gradient)
        const int c = it % C, pc = it / C;
        tXT[(int64_t)(pos0 + pc) * C + c] = XIN[c * PS + pc];
    }
    {
        const int rt = w & 1, ct = w >> 1;
        const f32x4 y = gemm1<V::KS_C>(to4(*reinterpret_cast<const float4*>(ws + V::OFF_B0 + 16 * rt + 4 * g)), fconv,
                                       [&](int s) { return XIN[(4 * s + g) * PS + 16 * ct + col]; });
#pragma unroll
        for (int r = 0; r < 4; ++r) YP[(16 * rt + 4 * g + r) * PS + 16 * ct + col] = y[r];
    }
    __syncthreads();
    float mean0, rstd0, mean_e, rstd_e;
    f32x4 e_own;
    {
        float pv[V::KS_PD];
#pragma unroll
        for (int i = 0; i < V::KS_PD; ++i) {
            const int f = g + 4 * i, ii = f / CIN, c = f - ii * CIN;
            pv[i] = f < PD ? YP[c * PS + pcol + ii] : 0.f;
        }
        token_stats(pv, PD, g, mean0, rstd0);
        const float mr = -mean0 * rstd0;
        e_own = gemm1<V::KS_PD>(vec4(ws + V::OFF_BE, w, g), femb, [&](int s) { return (g + 4 * s < PD) ? fmaf(pv[s], rstd0, mr) : 0.f; });
        put(PB, e_own);
        __syncthreads();
        float ev[V::KS_D];
        rows13(PB, ev);
        token_stats(ev, DIM, g, mean_e, rstd_e);
        const f32x4 g1 = vec4(ws + V::OFF_LN1, w, g), b1 = vec4(ws + V::OFF_LN1 + 64, w, g), pe = vec4(ws + V::OFF_POS + 64 * n, w, g);
        f32x4 X;
#pragma unroll
        for (int r = 0; r < 4; ++r) X[r] = fmaf((e_own[r] - mean_e) * rstd_e, g1[r], b1[r]) + pe[r];
        put(PA, X);
        __syncthreads();
    }
    auto embed_out = [&]() {                                                   // the residual stream at layer boundary 0, owner's rows
        const f32x4 g1 = vec4(ws + V::OFF_LN1, w, g), b1 = vec4(ws + V::OFF_LN1 + 64, w, g), pe = vec4(ws + V::OFF_POS + 64 * n, w, g);
        f32x4 X;
#pragma unroll
        for (int r = 0; r < 4; ++r) X[r] = fmaf((e_own[r] - mean_e) * rstd_e, g1[r], b1[r]) + pe[r];
        return X;
    };
    if constexpr (TAPED) {
        // the owner's tile of boundary b: rows 16 w + 4 g + r of column tok0 + col
        auto tile = [&](int b) {
            f32x4 X;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * w + 4 * g + r;
                X[r] = f < DIM ? xtape[((int64_t)b * DIM + f) * T + tok0 + col] : 0.f;
            }
            return X;
        };
#pragma unroll 1
        for (int l = 1; l < depth; ++l) xs_put(l, tile(l));
        put(PA, tile(depth));
        __syncthreads();
    } else
#pragma unroll 1
    for (int l = 0; l < depth; ++l) {
        const int wl = V::OFF_LAYER + l * V::L_STRIDE, ln = (l + 1 < depth ? l + 1 : l), wn = V::OFF_LAYER + ln * V::L_STRIDE;
        float mean, rstd, p[4];
        f32x4 q, k, v, o;
        layer_front(l, mean, rstd, q, k, v, p, o);
        __builtin_amdgcn_sched_barrier(0);
        load_qkv(ln);
        __builtin_amdgcn_sched_barrier(0);
        put(PO, o);
        __syncthreads();
        f32x4 X = l > 0 ? xs_get(l) : embed_out();
        {
            const f32x4 a = gemm1<V::KS_H>(f32x4{0.f, 0.f, 0.f, 0.f}, fout, [&](int s) { return PO[(4 * s + g) * TS + col]; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(fout, rs, lane, wn + V::L_WOUT + w * V::NG_H * 256);
            __builtin_amdgcn_sched_barrier(0);
            X += a;
            put(PB, X);
        }
        __syncthreads();
        {
            float xv[V::KS_D];
            rows13(PB, xv);
            token_stats(xv, DIM, g, mean, rstd);
            const float mr = -mean * rstd;
            f32x4 h = gemm1<V::KS_D>(vec4(ws + wl + V::L_B1, w, g), ffc1, [&](int s) { return fmaf(xv[s], rstd, mr); });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(ffc1, rs, lane, wn + V::L_W1 + w * V::NG_D * 256);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = gelu_f(h[r]);
            put(PH, h);
        }
        __syncthreads();
        {
            const f32x4 a = gemm1<V::KS_D>(vec4(ws + wl + V::L_B2, w, g), ffc2, [&](int s) { return PH[(4 * s + g) * TS + col]; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(ffc2, rs, lane, wn + V::L_W2 + w * V::NG_D * 256);
            __builtin_amdgcn_sched_barrier(0);
            X += a;
            put(PA, X);
            if (l + 1 < depth) xs_put(l + 1, X);
        }
        __syncthreads();
    }
    // ================================================ phase B: the way back
    // ---- conditioner output, affine map and log-det                                   (coupling.py:139-155)
    float mean_n, rstd_n;
    f32x4 un;                                                                  // normalised last residual, owner's rows
    {
        float xv[V::KS_D];
        rows13(PA, xv);
        token_stats(xv, DIM, g, mean_n, rstd_n);
        un = normalised(get(PA), mean_n, rstd_n);                              // the last boundary: still in the plane, owner's rows
        const f32x4 gn = vec4(ws + off_lno<V>(depth), w, g), bn = vec4(ws + off_lno<V>(depth) + 64, w, g);
        f32x4 hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) hn[r] = fmaf(un[r], gn[r], bn[r]);
        put(PH, hn);
    }
    __syncthreads();
    for (int it = tid; it < CIN * 2 * V::TOK; it += 256) {
        const int tc = it % V::TOK, ci = it / V::TOK, ii = ci / CIN, c = ci - ii * CIN;
        const int s = tc >> 2, pos = 2 * (tc & 3) + ii, b = s0 + s;
        const float raw = PH[(ii * C + CIN + c) * TS + tc], y1 = YP[(CIN + c) * PS + s * HW + pos];
        const float ls = 2.0f * tanhf(0.5f * raw), es = expf(ls);
        float gz0 = 0.f, gz1 = 0.f, gl = 0.f;
        if (b < B) {
            gz0 = gz[(int64_t)b * C * HW + c * HW + pos];
            gz1 = gz[(int64_t)b * C * HW + (CIN + c) * HW + pos];
            gl = gld[b];
        }
        PG[(ii * C + c) * TS + tc] = gz1;                                                  // d / d t
        PG[(ii * C + CIN + c) * TS + tc] = fmaf(gz1 * y1, es, gl) * (1.0f - 0.25f * ls * ls);   // d / d raw: log_s = 2 tanh(raw / 2)
        GYP[(CIN + c) * PS + s * HW + pos] = gz1 * es;                                     // second half of y: x1 exp(log_s)
        GYP[c * PS + s * HW + pos] = gz0;                                                  // first half passes through; + the conditioner's input gradient below
    }
    __syncthreads();
    f32x4 gX;                                                                  // gradient of the residual stream, owner's rows
    {   // transformer.norm
        const f32x4 gh = get(PG), gn = vec4(ws + off_lno<V>(depth), w, g);
        ln_partial(R::ln_final(depth), gh, un);
        gX = ln_bwd(gh * gn, un, rstd_n, DIM);
        put(PG, gX);
    }
    __syncthreads();
    // transposed fragments, requested one product ahead
    float4 f2T[V::NG_D], f1T[V::NG_D], foT[V::NG_D], fqT[R::NG_Q];
    load_qkv(depth - 1);
    load_frags(fout, rs, lane, V::OFF_LAYER + (depth - 1) * V::L_STRIDE + V::L_WOUT + w * V::NG_H * 256);
#pragma unroll 1
    for (int l = depth - 1; l >= 0; --l) {
        const int wl = V::OFF_LAYER + l * V::L_STRIDE, bl = R::OFF_LAYER + l * R::LB_STRIDE, lp = l > 0 ? l - 1 : 0;
        float* tl = tL + (int64_t)l * R4 * R::TL;
        // ---- recompute the layer from its parked input
        const f32x4 Xin = l > 0 ? xs_get(l) : embed_out();
        put(PA, Xin);
        load_frags(ffc1, rs, lane, wl + V::L_W1 + w * V::NG_D * 256);
        __syncthreads();
        float mean1, rstd1, mean2, rstd2, p[4];
        f32x4 q, k, v, o;
        layer_front(l, mean1, rstd1, q, k, v, p, o);
        const f32x4 u1h = normalised(Xin, mean1, rstd1);
        {
            const f32x4 ga = vec4(wsb + bl + R::LB_GA, w, g), ba = vec4(wsb + bl + R::LB_BA, w, g);
            f32x4 u1;
#pragma unroll
            for (int r = 0; r < 4; ++r) u1[r] = fmaf(u1h[r], ga[r], ba[r]);
            put(PB, u1);
        }
        put(PO, o);
        __syncthreads();
        store_T(PB, DIM, tl + R::TL_U1, DIM);
        store_T(PO, 64, tl + R4 * R::TL_O, 64);
        f32x4 Xm;
        {
            const f32x4 a = gemm1<V::KS_H>(f32x4{0.f, 0.f, 0.f, 0.f}, fout, [&](int s) { return PO[(4 * s + g) * TS + col]; });
            Xm = Xin + a;
        }
        __syncthreads();                                                       // everyone is done with PA (qkv operands), PB / PO (stores, out-proj)
        put(PA, Xm);
        __syncthreads();
        f32x4 hp, u2h;
        {
            float xv[V::KS_D];
            rows13(PA, xv);
            token_stats(xv, DIM, g, mean2, rstd2);
            const float mr = -mean2 * rstd2;
            hp = gemm1<V::KS_D>(vec4(ws + wl + V::L_B1, w, g), ffc1, [&](int s) { return fmaf(xv[s], rstd2, mr); });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(f2T, rb, lane, bl + R::LB_W2T + w * V::NG_D * 256);          // transposed fragments: one product ahead
            __builtin_amdgcn_sched_barrier(0);
            u2h = normalised(Xm, mean2, rstd2);
            const f32x4 gf = vec4(wsb + bl + R::LB_GF, w, g), bf = vec4(wsb + bl + R::LB_BF, w, g);
            f32x4 u2, h;
#pragma unroll
            for (int r = 0; r < 4; ++r) { u2[r] = fmaf(u2h[r], gf[r], bf[r]); h[r] = gelu_f(hp[r]); }
            put(PB, u2);
            put(PH, h);
        }
        __syncthreads();
        store_T(PB, DIM, tl + R4 * R::TL_U2, DIM);
        store_T(PH, DIM, tl + R4 * R::TL_H, DIM);
        store_T(PG, DIM, tl + R4 * R::TL_GXO, DIM);                            // gradient of the layer's output = of fc2's output
        // ---- MLP block backwards: x_out = x_mid + W2 gelu(W1 LN(x_mid) + b1) + b2
        {
            const f32x4 gh = gemm1<V::KS_D>(f32x4{0.f, 0.f, 0.f, 0.f}, f2T, [&](int s) { return PG[(4 * s + g) * TS + col]; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(f1T, rb, lane, bl + R::LB_W1T + w * V::NG_D * 256);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 ghp;
#pragma unroll
            for (int r = 0; r < 4; ++r) ghp[r] = valid[r] ? gh[r] * gelu_d(hp[r]) : 0.f;
            put(PO, ghp);
        }
        __syncthreads();
        store_T(PO, DIM, tl + R4 * R::TL_GHP, DIM);
        f32x4 gXm;
        {
            const f32x4 gu2 = gemm1<V::KS_D>(f32x4{0.f, 0.f, 0.f, 0.f}, f1T, [&](int s) { return PO[(4 * s + g) * TS + col]; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(foT, rb, lane, bl + R::LB_WOUTT + w * V::NG_D * 256);
            __builtin_amdgcn_sched_barrier(0);
            ln_partial(R::LN_L + l * R::LN_LSTRIDE + 128, gu2, u2h);
            const f32x4 gf = vec4(wsb + bl + R::LB_GF, w, g);
            gXm = gX + ln_bwd(gu2 * gf, u2h, rstd2, DIM);                      // (its exchange is a workgroup barrier: fc2^T has read PG)
            put(PG, gXm);
        }
        __syncthreads();
        store_T(PG, DIM, tl + R4 * R::TL_GXM, DIM);
        // ---- attention block backwards: x_mid = x_in + Wout softmax(q k^T / 8) v
        {
            const f32x4 go = gemm1<V::KS_D>(f32x4{0.f, 0.f, 0.f, 0.f}, foT, [&](int s) { return PG[(4 * s + g) * TS + col]; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(fqT, rb, lane, bl + R::LB_WQKVT + w * R::NG_Q * 256);
            __builtin_amdgcn_sched_barrier(0);
            float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;                      // d / d p[m]: go . v(token ^ m), own rows, then everyone's
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e0 = fmaf(go[r], v[r], e0);
                e1 = fmaf(go[r], tok_xor<1>(v[r]), e1);
                e2 = fmaf(go[r], tok_xor<2>(v[r]), e2);
                e3 = fmaf(go[r], tok_xor<3>(v[r]), e3);
            }
            e0 = group_sum(e0); e1 = group_sum(e1); e2 = group_sum(e2); e3 = group_sum(e3);
            float* sc = SC + scp * 256;
            if (g == 0) {
                sc[(w * 4 + 0) * 16 + col] = e0; sc[(w * 4 + 1) * 16 + col] = e1; sc[(w * 4 + 2) * 16 + col] = e2; sc[(w * 4 + 3) * 16 + col] = e3;
            }
            __syncthreads();
            auto tot = [&](int m) { return (sc[(0 + m) * 16 + col] + sc[(4 + m) * 16 + col]) + (sc[(8 + m) * 16 + col] + sc[(12 + m) * 16 + col]); };
            e0 = tot(0); e1 = tot(1); e2 = tot(2); e3 = tot(3);
            scp ^= 1;
            const float dsum = (p[0] * e0 + p[1] * e1) + (p[2] * e2 + p[3] * e3);
            const float gd0 = p[0] * (e0 - dsum) * 0.125f, gd1 = p[1] * (e1 - dsum) * 0.125f;
            const float gd2 = p[2] * (e2 - dsum) * 0.125f, gd3 = p[3] * (e3 - dsum) * 0.125f;
            f32x4 gq, gk, gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gv[r] = p[0] * go[r] + tok_xor<1>(p[1] * go[r]) + tok_xor<2>(p[2] * go[r]) + tok_xor<3>(p[3] * go[r]);
                gq[r] = fmaf(gd3, tok_xor<3>(k[r]), fmaf(gd2, tok_xor<2>(k[r]), fmaf(gd1, tok_xor<1>(k[r]), gd0 * k[r])));
                gk[r] = gd0 * q[r] + tok_xor<1>(gd1 * q[r]) + tok_xor<2>(gd2 * q[r]) + tok_xor<3>(gd3 * q[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                PQ[(16 * w + 4 * g + r) * TS + col] = gq[r];
                PQ[(64 + 16 * w + 4 * g + r) * TS + col] = gk[r];
                PQ[(128 + 16 * w + 4 * g + r) * TS + col] = gv[r];
            }
        }
        __syncthreads();
        store_T(PQ, 192, tl + R4 * R::TL_GQKV, 192);
        {
            const f32x4 gu1 = gemm1<48>(f32x4{0.f, 0.f, 0.f, 0.f}, fqT, [&](int s) { return PQ[(4 * s + g) * TS + col]; });
            __builtin_amdgcn_sched_barrier(0);
            load_qkv(lp);                                                      // the next iteration's recompute (three workgroups per CU cover the rest of the latency)
            load_frags(fout, rs, lane, V::OFF_LAYER + lp * V::L_STRIDE + V::L_WOUT + w * V::NG_H * 256);
            __builtin_amdgcn_sched_barrier(0);
            ln_partial(R::LN_L + l * R::LN_LSTRIDE, gu1, u1h);
            const f32x4 ga = vec4(wsb + bl + R::LB_GA, w, g);
            gX = gXm + ln_bwd(gu1 * ga, u1h, rstd1, DIM);                      // (exchange = barrier: out^T has read PG)
            put(PG, gX);
        }
        __syncthreads();
    }
    // ---- patch embedding backwards: x0 = LN(e) g1 + b1 + pos, e = We (LN(patch) g0 + b0) + be
    {
        const f32x4 eh = normalised(e_own, mean_e, rstd_e), g1 = vec4(ws + V::OFF_LN1, w, g);
        ln_partial(R::LN_1, gX, eh);
        const f32x4 ge = ln_bwd(gX * g1, eh, rstd_e, DIM);
        load_frags(foT, rb, lane, R::OFF_WET + (w & 1) * V::NG_D * 256);       // (waves 2, 3: unused copies)
        load_frags(fconv, rb, lane, R::OFF_CT + (w & 1) * V::NG_C * 256);
        put(PA, ge);
        // LayerNorm(pd) output u0 (operand of the embedding weight gradient): lane group g holds features g, g + 4, .. of its token
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < V::KS_PD; ++i) {
                const int f = g + 4 * i, ii = f / CIN, c = f - ii * CIN;
                if (f < PD) PB[f * TS + col] = fmaf((YP[c * PS + pcol + ii] - mean0) * rstd0, wsb[R::OFF_G0 + f], wsb[R::OFF_BT0 + f]);
            }
        }
    }
    __syncthreads();
    store_T(PA, DIM, tGE, DIM);
    store_T(PB, PD, tU0, PD);
    {
        f32x4 gu0 = f32x4{0.f, 0.f, 0.f, 0.f}, ph = gu0;
        if (w < 2) gu0 = gemm1<V::KS_D>(gu0, foT, [&](int s) { return PA[(4 * s + g) * TS + col]; });
        bool vp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * w + 4 * g + r, ii = f / CIN, c = f - ii * CIN;
            vp[r] = f < PD;
            ph[r] = vp[r] ? (YP[c * PS + pcol + ii] - mean0) * rstd0 : 0.f;
            if (!vp[r]) gu0[r] = 0.f;
        }
        if (w < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = row_sum16(gu0[r] * ph[r]), b = row_sum16(gu0[r]);
                if (col == 0) { lnw[R::LN_0 + 16 * w + 4 * g + r] = a; lnw[R::LN_0 + 32 + 16 * w + 4 * g + r] = b; }
            }
        }
        f32x4 gh0;
#pragma unroll
        for (int r = 0; r < 4; ++r) gh0[r] = vp[r] ? gu0[r] * wsb[R::OFF_G0 + 16 * w + 4 * g + r] : 0.f;
        const f32x4 gp = ln_bwd(gh0, ph, rstd0, PD);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * w + 4 * g + r, ii = f / CIN, c = f - ii * CIN;
            if (vp[r]) GYP[c * PS + pcol + ii] += gp[r];                        // one owner per (channel, position): plain read-modify-write
        }
    }
    __syncthreads();
    // ---- Conv1x1 + ActNorm backwards: g_x = (e^{-logs} Wm)^T g_y; g_y position-major for the weight gradient
    for (int it = tid; it < V::POSC * C; it += 256) {
        const int c = it % C, pc = it / C;
        tGYT[(int64_t)(pos0 + pc) * C + c] = GYP[c * PS + pc];
    }
    {
        const int rt = w & 1, ct = w >> 1;
        const f32x4 a = gemm1<V::KS_C>(f32x4{0.f, 0.f, 0.f, 0.f}, fconv, [&](int s) { return GYP[(4 * s + g) * PS + 16 * ct + col]; });
        const int pc = 16 * ct + col, b = s0 + pc / HW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * rt + 4 * g + r;
            if (c < C && b < B) gx[(int64_t)b * C * HW + c * HW + pc % HW] = a[r];
        }
    }
}

using RS26 = RS<26>;
using RB26 = RSB<RS26>;

}  // namespace

extern "C" {

int64_t cf_vit_step_bwd_ws_bytes(int C, int depth) { return C == 26 ? (int64_t)wsb_floats<RS26>(depth) * 4 : 0; }
// floats of the token-major plane buffer / the LayerNorm partial buffer for a batch of B samples
int64_t cf_vit_step_bwd_plane_floats(int B, int C, int depth) {
    if (C != 26) return 0;
    const int64_t Bp = (B + 3) / 4 * 4, R4 = 4 * Bp, P8 = 8 * Bp;
    return 2 * P8 * C + R4 * (RS26::PD + RS26::DIM) + (int64_t)depth * R4 * RB26::TL;
}
int64_t cf_vit_step_bwd_ln_floats(int B, int C, int depth) { return C == 26 ? (int64_t)((B + 3) / 4) * RB26::ln_floats(depth) : 0; }

int cf_vit_step_bwd_prepare_batch(int n, const float* const* Wm, const float* const* logs, const float* const* flat_vit_params,
                                  void* const* wsb, int C, int depth, cf_stream_t stream) {
    CF_REQUIRE(n >= 0 && Wm && logs && flat_vit_params && wsb && depth >= 1 && depth <= RB26::MAXD);
    if (C != 26) { cf_set_error("cf_vit_step_bwd_prepare: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    for (int i0 = 0; i0 < n; i0 += kVitPrepBatch) {
        const int m = n - i0 < kVitPrepBatch ? n - i0 : kVitPrepBatch;
        VitRsPackBwdBatch pb{};
        for (int i = 0; i < m; ++i) {
            const int j = i0 + i;
            CF_REQUIRE(Wm[j] && logs[j] && flat_vit_params[j] && wsb[j] && (reinterpret_cast<uintptr_t>(wsb[j]) & 15) == 0);
            pb.Wm[i] = Wm[j]; pb.logs[i] = logs[j]; pb.flat[i] = flat_vit_params[j]; pb.wsb[i] = (float*)wsb[j];
        }
        k_vit_rs_pack_bwd<RS26><<<dim3(64, m), dim3(256), 0, cf_s(stream)>>>(pb, depth);
        CF_LAUNCH_CHECK();
    }
    return 0;
}

int cf_vit_step_bwd_prepare(const float* Wm, const float* logs, const float* flat_vit_params, void* wsb, int C, int depth,
                            cf_stream_t stream) {
    CF_REQUIRE(Wm && logs && flat_vit_params && wsb && (reinterpret_cast<uintptr_t>(wsb) & 15) == 0);
    return cf_vit_step_bwd_prepare_batch(1, &Wm, &logs, &flat_vit_params, &wsb, C, depth, stream);
}

int cf_vit_step_bwd(const float* x, const float* gz, const float* gld, float* gx, const void* ws, const void* wsb, float* planes,
                    float* ln_partials, int B, int C, int depth, int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && gz && gld && gx && ws && wsb && planes && ln_partials && B > 0 && depth >= 1 && depth <= RB26::MAXD &&
               x_bstride >= (int64_t)C * 8);
    if (C != 26) { cf_set_error("cf_vit_step_bwd: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    constexpr size_t lds_bytes = (size_t)RB26::LDS_FLOATS * sizeof(float);
    static std::atomic<uint64_t> raised{0};
    if (int rc_ = cf_raise_dynamic_lds((const void*)k_vit_step_bwd_rs<RS26>, 160 * 1024, raised, __func__)) return rc_;
    const int nwg = (B + 3) / 4;
    k_vit_step_bwd_rs<RS26><<<dim3((unsigned)nwg), dim3(256), lds_bytes, cf_s(stream)>>>(
        x, gz, gld, gx, (const float*)ws, (const float*)wsb, planes, ln_partials, B, nwg * 4, x_bstride, depth);
    CF_LAUNCH_CHECK();
    return 0;
}

// the same from the residual-stream tape of cf_vit_step_fwd_taped (xtape: cf_vit_step_tape_floats(B, C, depth) floats): the
// six layers are not run a second time before the way back.  Same outputs to fp32 rounding (the tape holds the forward's own
// residual stream, the recompute form rebuilds it in the row-split kernel's summation order).
int cf_vit_step_bwd_taped(const float* x, const float* gz, const float* gld, float* gx, const void* ws, const void* wsb,
                          float* planes, float* ln_partials, const float* xtape, int B, int C, int depth, int64_t x_bstride,
                          cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && gz && gld && gx && ws && wsb && planes && ln_partials && xtape && B > 0 && depth >= 1 && depth <= RB26::MAXD &&
               x_bstride >= (int64_t)C * 8);
    if (C != 26) { cf_set_error("cf_vit_step_bwd_taped: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    constexpr size_t lds_bytes = (size_t)RB26::LDS_FLOATS * sizeof(float);
    static std::atomic<uint64_t> raised{0};
    if (int rc_ = cf_raise_dynamic_lds((const void*)k_vit_step_bwd_rs<RS26, true>, 160 * 1024, raised, __func__)) return rc_;
    const int nwg = (B + 3) / 4;
    k_vit_step_bwd_rs<RS26, true><<<dim3((unsigned)nwg), dim3(256), lds_bytes, cf_s(stream)>>>(
        x, gz, gld, gx, (const float*)ws, (const float*)wsb, planes, ln_partials, B, nwg * 4, x_bstride, depth, xtape,
        4ll * ((B + 31) / 32 * 32));
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
