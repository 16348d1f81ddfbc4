// Row-split transformer-coupling flow step for SMALL batches: Conv1x1 -> ActNorm -> TransCoupling (patchify, SimpleViT,
// un-patchify, affine map, log-det) of the time-series topologies (H x 1 windows, patch (2,1), 4 tokens per sample: SMAP)
// with the four waves of a workgroup SHARING 16 token columns (4 samples) and splitting the OUTPUT ROWS of every Linear.
// Reference: contextflow/model.py:129-147, layers/conv1x1.py:52-57, layers/actnorm.py:53-60, layers/coupling.py:100-159,
// layers/simple_vit.py:18-127.
//
// Why a second kernel.  k_vit_step (cf_vit_step.hip) keeps a sample's whole step inside ONE wave (8 samples per wave, all
// 2 148 MFMAs + 12 K vector instructions in series): the right shape at saturating batches, but at the reference's batch of
// 256 (config.py:10) a launch is 32 waves on 8 CUs and a step takes the ~90 us of one wave's serial chain.  Here a step's
// chain is cut eight-fold: v_mfma_f32_16x16x4_f32 tiles (16 token columns = 4 samples instead of 32 = 8) and one 16-row
// tile of every 64-row result per wave.  Planes [feature][16 tokens] go through LDS (26 KB per workgroup), five
// workgroup barriers per transformer layer.
//  * LayerNorm + Linear: the 13 values a lane reads for the token statistics (features g, g + 4, ...; lane group g) ARE its
//    B operands of the following product (k-step s needs feature 4 s + g): one LDS read per MFMA, normalisation as one FMA,
//    gamma folded into the packed weights and beta into the bias (k_vit_rs_pack, fp64);
//  * residual products are summed on their own and meet the residual stream in one addition (as the reference does);
//    single-tile products run as two interleaved accumulation chains (even / odd k-steps: a dependent 16x16x4 needs 40
//    cycles, an independent one 32);
//  * attention: a sample's 4 tokens are 4 consecutive lanes - q.k^T and p.v by DPP quad permutations on the wave's own 16
//    head features, the partial scores of the four waves meet in LDS, the softmax over 4 scores is exact;
//  * weight fragments of a product are requested right after the same product of the PREVIOUS layer has used its registers:
//    a whole layer of lead time, no copies.
#include "cf_vit_rs_common.h"

extern "C" int cf_slogdet_inverse(const float* W, int C, float* logabsdet, float* Winv, cf_stream_t stream);
extern "C" int cf_slogdet_inverse_batch(int n, const float* const* Wm, int C, float* const* logabsdet, float* const* inv, cf_stream_t stream);
extern "C" int cf_vit_step_bwd_prepare_batch(int n, const float* const* Wm, const float* const* logs, const float* const* flat, void* const* wsb,
                                             int C, int depth, cf_stream_t stream);

namespace {

// ---- the forward kernel ---------------------------------------------------------------------------------------------------
// x, z: (B, C, 8, 1).  ldj_acc[b] += H*W*log|det W| + sum logs + sum log_s.  hout (optional, tests): the conditioner's
// output un-patchified, (B, C, 8, 1) = [t | raw].
#ifndef CF_VIT_RS_MINW
#define CF_VIT_RS_MINW 1
#endif
// DUMP (training forward): the residual stream at the depth + 1 layer boundaries also goes to xtape ([boundary][DIM][T],
// as cf_vit_step_fwd_taped writes it) - the owner's tile, 64 contiguous bytes per feature row.
// (x and z may be the same buffer: a workgroup reads its four samples before it writes them - the chained form below)
template <class V, bool DUMP = false>
__device__ __forceinline__ void vit_step_rs_body(const float* x, float* z, float* __restrict__ ldj_acc, const float* __restrict__ ws,
                                                 int B, int64_t xbs, int depth, float* __restrict__ hout,
                                                 float* __restrict__ xtape = nullptr, int64_t T = 0) {
    constexpr int C = V::C, CIN = V::CIN, HW = V::HW, DIM = V::DIM, PD = V::PD, TOK = V::TOK, POSC = V::POSC;
    __shared__ __align__(16) float lds[V::LDS_FLOATS];
    float* XIN = lds + V::P_XIN;     // [4 KS_C][32 positions]   step input, channel-major
    float* YP = lds + V::P_Y;        // [32][32]                 Conv1x1 + ActNorm output
    float* XA = lds + V::P_X1;       // [64][16]                 residual stream at layer boundaries
    float* XB = lds + V::P_X0;       // [64][16]                 ... after the attention block (and the embedding before its norm)
    float* OP = lds + V::P_O;        // [64][16]                 attention output
    float* HP = lds + V::P_H;        // [64][16]                 MLP hidden layer / final conditioner output
    float* SC = lds + V::P_SC;       // [4 waves][4][16]         partial scores
    float* LS = lds + V::P_LS;       // [CIN * 2][16]            log-scales of the epilogue
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform by construction: fragment offsets stay scalar
    const int s0 = blockIdx.x * V::SPW;
    const rsrc_t rs = make_rsrc(ws, ws_floats<V>(depth));

    // weight fragments of the first layer and of the front end: requested before anything else
    float4 fa1[V::NG_D], fa2[V::NG_D], ffc1[V::NG_D], ffc2[V::NG_D], fconv[V::NG_C], femb[V::NG_PD];
    load_frags(fconv, rs, lane, V::OFF_A0 + (w & 1) * V::NG_C * 256);
    load_frags(femb, rs, lane, V::OFF_WE + w * V::NG_PD * 256);
    const int wf0 = off_fused<V>(depth);
    load_frags(fa1, rs, lane, wf0 + V::F_A1 + w * V::NG_D * 256);
    load_frags(fa2, rs, lane, wf0 + V::F_A2 + w * V::NG_D * 256);
    load_frags(ffc1, rs, lane, V::OFF_LAYER + V::L_W1 + w * V::NG_D * 256);
    load_frags(ffc2, rs, lane, V::OFF_LAYER + V::L_W2 + w * V::NG_D * 256);

    for (int i = tid; i < 4 * V::KS_C * POSC; i += 256) {
        const int c = i / POSC, pc = i % POSC, b = s0 + pc / HW;
        XIN[i] = (c < C && b < B) ? x[(int64_t)b * xbs + c * HW + pc % HW] : 0.f;
    }
    __syncthreads();
    // ================= Conv1x1 + ActNorm: one (16 channels x 16 positions) tile per wave
    {
        const int rt = w & 1, ct = w >> 1;
        const f32x4 y = gemm1<V::KS_C>(to4(*reinterpret_cast<const float4*>(ws + V::OFF_B0 + 16 * rt + 4 * g)), fconv,
                                       [&](int s) { return XIN[(4 * s + g) * POSC + 16 * ct + col]; });
#pragma unroll
        for (int r = 0; r < 4; ++r) YP[(16 * rt + 4 * g + r) * POSC + 16 * ct + col] = y[r];
    }
    __syncthreads();
    // ================= patch embedding: LN(pd) -> Linear -> LN(dim) + pos          (simple_vit.py:100-105,122)
    // token column col = 4 * (sample in workgroup) + n; patch feature f = ii * CIN + c = y[c][position 2 n + ii]
    const int n = col & 3, pcol = (col >> 2) * HW + 2 * n;
    f32x4 X;                                                                   // this wave's rows 16 w + 4 g + r of the residual stream
    {
        float pv[V::KS_PD];
#pragma unroll
        for (int i = 0; i < V::KS_PD; ++i) {
            const int f = g + 4 * i, ii = f / CIN, c = f - ii * CIN;
            pv[i] = f < PD ? YP[c * POSC + pcol + ii] : 0.f;
        }
        float mean, rstd;
        token_stats(pv, PD, g, mean, rstd);
        const float mr = -mean * rstd;
        const f32x4 e = gemm1<V::KS_PD>(vec4(ws + V::OFF_BE, w, g), femb, [&](int s) { return (g + 4 * s < PD) ? fmaf(pv[s], rstd, mr) : 0.f; });
#pragma unroll
        for (int r = 0; r < 4; ++r) XB[(16 * w + 4 * g + r) * TOK + col] = e[r];
        __syncthreads();
        float ev[V::KS_D];
#pragma unroll
        for (int i = 0; i < V::KS_D; ++i) ev[i] = XB[(g + 4 * i) * TOK + col];
        token_stats(ev, DIM, g, mean, rstd);
        const f32x4 g1 = vec4(ws + V::OFF_LN1, w, g), b1 = vec4(ws + V::OFF_LN1 + 64, w, g), pe = vec4(ws + V::OFF_POS + 64 * n, w, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X[r] = fmaf((e[r] - mean) * rstd, g1[r], b1[r]) + pe[r];
            XA[(16 * w + 4 * g + r) * TOK + col] = X[r];
        }
        __syncthreads();
    }
    auto dump = [&](int bnd) {
        if constexpr (DUMP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * w + 4 * g + r;
                if (f < DIM) xtape[((int64_t)bnd * DIM + f) * T + blockIdx.x * TOK + col] = X[r];
            }
        }
    };
    // (boundary 0, the embedding output, is rebuilt by the backward kernel together with the statistics it needs: not taped)
    // ================= transformer                                                 (simple_vit.py:56-88)
#pragma unroll 1
    for (int l = 0; l < depth; ++l) {
        const int wl = V::OFF_LAYER + l * V::L_STRIDE, ln = (l + 1 < depth ? l + 1 : l), wn = V::OFF_LAYER + ln * V::L_STRIDE;
        const int wf = wf0 + l * V::F_STRIDE, wfn = wf0 + ln * V::F_STRIDE;
        float mean, rstd;
        {   // attention through the fused tables: s_ij = (A1 n_i + c1) . n_j, x += A2 (sum_j p_ij n_j) + c2   (simple_vit.py:56-68,84)
            float xv[V::KS_D];
#pragma unroll
            for (int i = 0; i < V::KS_D; ++i) xv[i] = XA[(g + 4 * i) * TOK + col];
            token_stats(xv, DIM, g, mean, rstd);
            const float mr = -mean * rstd;
            const f32x4 gq = gemm1<V::KS_D>(vec4(ws + wf + V::F_C1, w, g), fa1, [&](int s) { return (g + 4 * s < DIM) ? fmaf(xv[s], rstd, mr) : 0.f; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(fa1, rs, lane, wfn + V::F_A1 + w * V::NG_D * 256);         // next layer's fragments: a whole layer of lead time
            __builtin_amdgcn_sched_barrier(0);
            // this wave's rows f = 16 w + 4 g + r of the normalised stream (own token; the partners' by DPP), masked beyond DIM
            f32x4 nv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * w + 4 * g + r;
                nv[r] = f < DIM ? fmaf(XA[f * TOK + col], rstd, mr) : 0.f;
            }
            // scores of this token against the 4 tokens of its sample (partner = token ^ m): own 16 features, then the lane
            // groups, then the four waves
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d0 = fmaf(gq[r], nv[r], d0);
                d1 = fmaf(gq[r], tok_xor<1>(nv[r]), d1);
                d2 = fmaf(gq[r], tok_xor<2>(nv[r]), d2);
                d3 = fmaf(gq[r], tok_xor<3>(nv[r]), d3);
            }
            d0 = group_sum(d0); d1 = group_sum(d1); d2 = group_sum(d2); d3 = group_sum(d3);
            if (g == 0) {
                SC[(w * 4 + 0) * TOK + col] = d0; SC[(w * 4 + 1) * TOK + col] = d1;
                SC[(w * 4 + 2) * TOK + col] = d2; SC[(w * 4 + 3) * TOK + col] = d3;
            }
            __syncthreads();
            auto score = [&](int m) {                                          // (dim_head ** -0.5 sits in A1 / c1)
                return (SC[(0 * 4 + m) * TOK + col] + SC[(1 * 4 + m) * TOK + col]) + (SC[(2 * 4 + m) * TOK + col] + SC[(3 * 4 + m) * TOK + col]);
            };
            d0 = score(0); d1 = score(1); d2 = score(2); d3 = score(3);
            const float mx = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3));
            float p0 = expf(d0 - mx), p1 = expf(d1 - mx), p2 = expf(d2 - mx), p3 = expf(d3 - mx);
            const float inv = 1.0f / ((p0 + p1) + (p2 + p3));
            p0 *= inv; p1 *= inv; p2 *= inv; p3 *= inv;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                OP[(16 * w + 4 * g + r) * TOK + col] = fmaf(p3, tok_xor<3>(nv[r]), fmaf(p2, tok_xor<2>(nv[r]), fmaf(p1, tok_xor<1>(nv[r]), p0 * nv[r])));
        }
        __syncthreads();
        {
            const f32x4 a = gemm1<V::KS_D>(vec4(ws + wf + V::F_C2, w, g), fa2, [&](int s) { return OP[(4 * s + g) * TOK + col]; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(fa2, rs, lane, wfn + V::F_A2 + w * V::NG_D * 256);
            __builtin_amdgcn_sched_barrier(0);
            X += a;
#pragma unroll
            for (int r = 0; r < 4; ++r) XB[(16 * w + 4 * g + r) * TOK + col] = X[r];
        }
        __syncthreads();
        {   // x = W2 gelu(W1 LN(x) + b1) + b2 + x                                    (simple_vit.py:30-40,86)
            float xv[V::KS_D];
#pragma unroll
            for (int i = 0; i < V::KS_D; ++i) xv[i] = XB[(g + 4 * i) * TOK + col];
            token_stats(xv, DIM, g, mean, rstd);
            const float mr = -mean * rstd;
            f32x4 h = gemm1<V::KS_D>(vec4(ws + wl + V::L_B1, w, g), ffc1, [&](int s) { return (g + 4 * s < DIM) ? fmaf(xv[s], rstd, mr) : 0.f; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(ffc1, rs, lane, wn + V::L_W1 + w * V::NG_D * 256);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h[r] = 0.5f * h[r] * (1.0f + erff(h[r] * 0.70710678118654752f));
                HP[(16 * w + 4 * g + r) * TOK + col] = h[r];
            }
        }
        __syncthreads();
        {
            const f32x4 a = gemm1<V::KS_D>(vec4(ws + wl + V::L_B2, w, g), ffc2, [&](int s) { return HP[(4 * s + g) * TOK + col]; });
            __builtin_amdgcn_sched_barrier(0);
            load_frags(ffc2, rs, lane, wn + V::L_W2 + w * V::NG_D * 256);
            __builtin_amdgcn_sched_barrier(0);
            X += a;
#pragma unroll
            for (int r = 0; r < 4; ++r) XA[(16 * w + 4 * g + r) * TOK + col] = X[r];
            dump(l + 1);
        }
        __syncthreads();
    }
    // ================= transformer.norm -> conditioner output plane [feature ii * C + ch][token]
    {
        float xv[V::KS_D];
#pragma unroll
        for (int i = 0; i < V::KS_D; ++i) xv[i] = XA[(g + 4 * i) * TOK + col];
        float mean, rstd;
        token_stats(xv, DIM, g, mean, rstd);
        const f32x4 gn = vec4(ws + off_lno<V>(depth), w, g), bn = vec4(ws + off_lno<V>(depth) + 64, w, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) HP[(16 * w + 4 * g + r) * TOK + col] = fmaf((X[r] - mean) * rstd, gn[r], bn[r]);
    }
    __syncthreads();
    // ================= un-patchify, affine map, log-det, stores                    (coupling.py:139-155)
    // item (c, ii, token): t = h[ii C + c], raw = h[ii C + CIN + c], x1 = y[CIN + c][position 2 n + ii]
    for (int it = tid; it < CIN * 2 * TOK; it += 256) {
        const int tc = it % TOK, ci = it / TOK, ii = ci / CIN, c = ci - ii * CIN;
        const int sl = tc >> 2, pos = 2 * (tc & 3) + ii, b = s0 + sl;
        const float tt = HP[(ii * C + c) * TOK + tc], raw = HP[(ii * C + CIN + c) * TOK + tc];
        const float y1 = YP[(CIN + c) * POSC + sl * HW + pos], y0 = YP[c * POSC + sl * HW + pos];
        const float ls = 2.0f * tanhf(0.5f * raw);
        LS[it] = ls;
        if (b < B) {
            float* zb = z + (int64_t)b * C * HW;
            zb[c * HW + pos] = y0;                                             // first half passes through (coupling.py:154)
            zb[(CIN + c) * HW + pos] = fmaf(y1, expf(ls), tt);
            if (hout) {
                float* hb = hout + (int64_t)b * C * HW;
                hb[c * HW + pos] = tt;
                hb[(CIN + c) * HW + pos] = raw;
            }
        }
    }
    __syncthreads();
    {   // wave w sums the log-scales of sample w in a fixed order (no float atomics: bitwise reproducible)
        float s = 0.f;
        for (int it = lane; it < CIN * 2 * 4; it += 64) {                      // the sample's items: (c, ii) x its 4 tokens
            const int ci = it >> 2, tc = 4 * w + (it & 3);
            s += LS[ci * TOK + tc];
        }
        s = cf_wave_sum(s);
        if (lane == 0 && s0 + w < B) ldj_acc[s0 + w] += ws[0] + s;
    }
}

template <class V, bool DUMP = false>
__global__ __launch_bounds__(256, CF_VIT_RS_MINW) void k_vit_step_rs(const float* __restrict__ x, float* __restrict__ z,
                                                     float* __restrict__ ldj_acc, const float* __restrict__ ws, int B,
                                                     int64_t xbs, int depth, float* __restrict__ hout,
                                                     float* __restrict__ xtape = nullptr, int64_t T = 0) {
    vit_step_rs_body<V, DUMP>(x, z, ldj_acc, ws, B, xbs, depth, hout, xtape, T);
}

// Consecutive transformer flow steps in ONE launch (evaluation at small batches: a step is latency - 28 us on 64 workgroups at a
// batch of 256 - and so is the gap between two launches): a workgroup owns its four samples end to end, steps 2.. run in place
// on z behind a workgroup barrier.  Same code per step: bit for bit the per-step launches.
constexpr int kVitChain = 8;
struct VitWsChain { const float* ws[kVitChain]; };
template <class V>
__global__ __launch_bounds__(256, CF_VIT_RS_MINW) void k_vit_step_rs_chain(const float* x, float* z, float* __restrict__ ldj_acc,
                                                                           const VitWsChain wc, int nsteps, int B, int64_t xbs, int depth) {
    vit_step_rs_body<V>(x, z, ldj_acc, wc.ws[0], B, xbs, depth, nullptr);
    for (int st = 1; st < nsteps; ++st) {
        __syncthreads();
        vit_step_rs_body<V>(z, z, ldj_acc, wc.ws[st], B, (int64_t)V::C * V::HW, depth, nullptr);
    }
}

using RS26 = RS<26>;

bool rs_ok(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads) {
    return C == 26 && H == 8 && W == 1 && p1 == 2 && p2 == 1 && dim == 2 * C && dim_head == 64 && heads == 1;
}

}  // namespace

extern "C" {

int cf_vit_step_rs_supported(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads) {
    return rs_ok(C, H, W, p1, p2, dim, dim_head, heads) ? 1 : 0;
}

int64_t cf_vit_step_rs_ws_bytes(int C, int depth) { return C == 26 ? (int64_t)ws_floats<RS26>(depth) * 4 : 0; }

// The row-split tables of n flow steps in ONE factorisation launch, ONE k_vit_fuse launch and ONE packing launch (a training
// step at a batch of 256 packs all 8 steps of the SMAP flow per update: 24 serial launches of 14-27 us before).  HOST arrays
// of device pointers.  winv (NULL or n): also Wm^-1 (the Conv1x1 log-det gradient); wsb (NULL or n): also the backward
// kernel's transposed fragments (cf_vit_step_bwd_prepare).
int cf_vit_step_rs_prepare_batch(int n, const float* const* Wm, const float* const* t, const float* const* logs,
                                 const float* const* flat_vit_params, const float* pos, void* const* ws, float* const* winv,
                                 void* const* wsb, int C, int depth, cf_stream_t stream) {
    CF_REQUIRE(n >= 0 && Wm && t && logs && flat_vit_params && pos && ws && depth >= 1);
    if (C != 26) { cf_set_error("cf_vit_step_rs_prepare: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    for (int i0 = 0; i0 < n; i0 += kVitPrepBatch) {
        const int m = n - i0 < kVitPrepBatch ? n - i0 : kVitPrepBatch;
        VitRsPackBatch pb{};
        VitFuseBatch fb{};
        float* lad[kVitPrepBatch];
        for (int i = 0; i < m; ++i) {
            const int j = i0 + i;
            CF_REQUIRE(Wm[j] && t[j] && logs[j] && flat_vit_params[j] && ws[j] && (reinterpret_cast<uintptr_t>(ws[j]) & 15) == 0);
            float* w = (float*)ws[j];
            pb.Wm[i] = Wm[j]; pb.t[i] = t[j]; pb.logs[i] = logs[j]; pb.flat[i] = flat_vit_params[j]; pb.ws[i] = w;
            fb.layers[i] = flat_vit_params[j] + 2 * RS26::PD + RS26::DIM * RS26::PD + RS26::DIM + 2 * RS26::DIM;
            fb.scratch[i] = w + off_fuse_scratch<RS26>(depth);
            lad[i] = w + 1;
        }
        int rc = cf_slogdet_inverse_batch(m, pb.Wm, C, lad, winv ? winv + i0 : nullptr, stream);
        if (rc) return rc;
        k_vit_fuse<RS26::DIM, RS26::HEAD><<<dim3(depth * m, 2, FUSE_SPLIT), dim3(256), 0, cf_s(stream)>>>(fb, depth);
        k_vit_rs_pack<RS26><<<dim3(64, m), dim3(256), 0, cf_s(stream)>>>(pb, pos, depth);
        CF_LAUNCH_CHECK();
    }
    if (wsb) return cf_vit_step_bwd_prepare_batch(n, Wm, logs, flat_vit_params, wsb, C, depth, stream);
    return 0;
}

int cf_vit_step_rs_prepare(const float* Wm, const float* t, const float* logs, const float* flat_vit_params, const float* pos,
                           void* ws, int C, int depth, cf_stream_t stream) {
    CF_REQUIRE(Wm && t && logs && flat_vit_params && pos && ws && depth >= 1 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0);
    return cf_vit_step_rs_prepare_batch(1, &Wm, &t, &logs, &flat_vit_params, pos, &ws, nullptr, nullptr, C, depth, stream);
}

int cf_vit_step_rs_fwd(const float* x, float* z, float* ldj_acc, const void* ws, float* h_out, int B, int C, int depth,
                       int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && B > 0 && depth >= 1 && x_bstride >= (int64_t)C * 8);
    if (C != 26) { cf_set_error("cf_vit_step_rs_fwd: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    k_vit_step_rs<RS26><<<dim3((unsigned)((B + 3) / 4)), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, (const float*)ws, B,
                                                                                   x_bstride, depth, h_out);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_vit_step_rs_chain_max_steps(void) { return kVitChain; }

// n <= cf_vit_step_rs_chain_max_steps() consecutive steps (ws: host array of their packed tables); z may not alias x
int cf_vit_step_rs_fwd_chain(const float* x, float* z, float* ldj_acc, const void* const* ws, int n, int B, int C, int depth,
                             int64_t x_bstride, cf_stream_t stream) {
    if (B == 0 || n == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && n >= 1 && n <= kVitChain && B > 0 && depth >= 1 && x_bstride >= (int64_t)C * 8 && (const float*)z != x);
    if (C != 26) { cf_set_error("cf_vit_step_rs_fwd_chain: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    VitWsChain wc{};
    for (int i = 0; i < n; ++i) { CF_REQUIRE(ws[i]); wc.ws[i] = (const float*)ws[i]; }
    k_vit_step_rs_chain<RS26><<<dim3((unsigned)((B + 3) / 4)), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, wc, n, B, x_bstride, depth);
    CF_LAUNCH_CHECK();
    return 0;
}

// training forward: cf_vit_step_rs_fwd that also writes the residual-stream tape of cf_vit_step_fwd_taped (same layout and size)
int cf_vit_step_rs_fwd_taped(const float* x, float* z, float* ldj_acc, const void* ws, float* xtape, int B, int C, int depth,
                             int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && xtape && B > 0 && depth >= 1 && x_bstride >= (int64_t)C * 8);
    if (C != 26) { cf_set_error("cf_vit_step_rs_fwd_taped: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    k_vit_step_rs<RS26, true><<<dim3((unsigned)((B + 3) / 4)), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, (const float*)ws, B,
                                                                                         x_bstride, depth, nullptr, xtape,
                                                                                         4ll * ((B + 31) / 32 * 32));
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
