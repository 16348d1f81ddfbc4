// Context-conditioned (specialist) branches of the flow layers: every Conv1x1 / ActNorm /
// Coupling of a specialist model owns a context encoder and a small CN net whose output turns the layer's parameters
// into PER-SAMPLE parameters; the priors get per-sample shifts of their component means / scales.
//
// All of it is HBM- / VALU-bound work over (sample, channel, pixel); the CN nets themselves are cf_linear (MFMA).
// One workgroup owns one sample in the three per-sample kernels, so the per-sample parameters live in LDS / registers.
#include "cf_common.h"
#include <math.h>

namespace {


// ContextEncoder = (OneHotEncoder | EyeEncoder) -> UniformCatDequantization (rtdl/nn/_embeddings.py:76-150,
// dequantize.py:55-64): out[b, j] = (x[b, j] + u[b, j]) / qbins[j]; x = the integer context itself (eye) or its
// one-hot code, column j belonging to the context variable whose cardinality range contains j.
__global__ __launch_bounds__(256) void k_ctx_encode(const int64_t* __restrict__ ctx, const float* __restrict__ u,
                                                    const float* __restrict__ qbins, const int64_t* __restrict__ card,
                                                    float* __restrict__ out, int B, int nctx, int width, int onehot) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * width) return;
    const int b = (int)(e / width), j = (int)(e - (int64_t)b * width);
    float x;
    if (onehot) {                                     // modes 1 (one-hot) and 2 (binary sign code): find the variable
        int i = 0, off = 0;
        while (i < nctx - 1 && j >= off + (int)card[i]) { off += (int)card[i]; ++i; }
        if (onehot == 2) {
            // ArgmaxCatDequantization (dequantize.py:236-262): big-endian binary code of every context variable
            // (card[i] = its number of bits), one zero pad column when the total is odd; z = u * (2 bit - 1)
            const int nb = (int)card[i], k = j - off;
            const int bit = (k < nb) ? (int)((ctx[(int64_t)b * nctx + i] >> (nb - 1 - k)) & 1) : 0;
            out[e] = u[e] * (bit ? 1.f : -1.f);
            return;
        }
        x = (ctx[(int64_t)b * nctx + i] == (int64_t)(j - off)) ? 1.f : 0.f;
    } else {
        x = (float)ctx[(int64_t)b * nctx + j];
    }
    out[e] = (x + u[e]) / qbins[j];
}

// Conv1x1 with a context net (conv1x1.py:34-50): m = CN(c) reshaped (C, C) per sample;
//   W_b = tril(m, -1) + diag(exp(diag m))            [+ NN - I under contextflow]
//   z[b] = W_b x[b] per pixel;  ldj[b] = H W sum(diag m)   (the caller adds H W log|det NN| under contextflow -
//   the reference's own expression, not the log-det of W_b).
// row stride of the transposed per-sample matrix in LDS: C rounded up to 8 (16-byte rows for the float4 reads), plus 4
// when that is a multiple of 32 (the transposed writes would otherwise all land in one bank)
__host__ __device__ inline int conv1x1_ctx_cp(int C) {
    const int cp = (C + 7) & ~7;
    return (cp & 31) == 0 ? cp + 4 : cp;
}

// global -> LDS copy of n contiguous floats by the 256 threads of a workgroup, in batches of 8 independent loads per
// thread (16-byte loads when n and the source allow): a `dst[e] = src[e]` loop with a runtime bound is compiled to one
// dependent load -> store round trip per element, which makes these one-workgroup-per-sample kernels latency-bound.
__device__ __forceinline__ void stage_copy(float* __restrict__ dst, const float* __restrict__ src, int n, int tid) {
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int n4 = n >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int e0 = tid; e0 < n4; e0 += 8 * 256) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = s4[min(e0 + i * 256, n4 - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (e0 + i * 256 < n4) d4[e0 + i * 256] = v[i];
        }
    } else {
        for (int e0 = tid; e0 < n; e0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[min(e0 + i * 256, n - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (e0 + i * 256 < n) dst[e0 + i * 256] = v[i];
        }
    }
}

// One workgroup per sample: W_b^T (row stride CP = C rounded up to 8, zero padded) and x in LDS; a thread owns one
// pixel and 8 output channels at a time - per input channel one x read and two 16-byte (broadcast) reads of W_b^T.
__global__ __launch_bounds__(256) void k_conv1x1_ctx(const float* __restrict__ x, const float* __restrict__ m,
                                                     const float* __restrict__ Wm, float* __restrict__ z,
                                                     float* __restrict__ ldj, int C, int HW, int64_t xbs) {
    extern __shared__ __align__(16) float dyn[];
    const int CP = conv1x1_ctx_cp(C);
    float* Wt = dyn;                                  // Wt[i][o] = W_b[o][i], row stride CP
    float* xs = dyn + C * CP;                         // [C][HW]
    float* scr = xs + C * HW;                         // [4] (inside the dynamic block: no static LDS next to a 160 KiB request)
    float* ms = scr + 4;                              // [C*C] this sample's CN(c) output, staged before the transposed read
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* mb = m + (int64_t)b * C * C;
    const float* xb = x + (int64_t)b * xbs;
    stage_copy(ms, mb, C * C, tid);
    __syncthreads();
    float dsum = 0.f;
    // lanes walk i: ms read without bank conflicts, Wt written at stride CP; the shared NN matrix is fetched in batches of
    // independent loads (a load inside the element loop costs one full memory latency per element)
    for (int e0 = tid; e0 < CP * C; e0 += 8 * 256) {
        float wm[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wm[k] = Wm != nullptr ? Wm[min(e0 + k * 256, C * C - 1)] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + k * 256;
            if (e < CP * C) {
                const int o = e / C, i = e - o * C;
                float w = 0.f;
                if (o < C) {
                    const float v = ms[e];
                    w = o > i ? v : (o == i ? expf(v) : 0.f);
                    if (o == i) dsum += v;
                    if (Wm != nullptr) w += wm[k] - (o == i ? 1.f : 0.f);
                }
                Wt[i * CP + o] = w;
            }
        }
    }
    stage_copy(xs, xb, C * HW, tid);                  // xs is 16-byte aligned: C * CP is a multiple of 8
    dsum = cf_block_sum<4>(dsum, scr);                // also the barrier that publishes Wt / xs
    if (tid == 0) ldj[b] = dsum * (float)HW;
    float* zb = z + (int64_t)b * C * HW;
    const int nob = CP >> 3;
    for (int e = tid; e < nob * HW; e += 256) {
        const int ob = e / HW, p = e - ob * HW, o0 = ob * 8;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int i = 0; i < C; ++i) {
            const float xv = xs[i * HW + p];
            const float4 w0 = *reinterpret_cast<const float4*>(&Wt[i * CP + o0]);
            const float4 w1 = *reinterpret_cast<const float4*>(&Wt[i * CP + o0 + 4]);
            acc[0] = fmaf(w0.x, xv, acc[0]); acc[1] = fmaf(w0.y, xv, acc[1]);
            acc[2] = fmaf(w0.z, xv, acc[2]); acc[3] = fmaf(w0.w, xv, acc[3]);
            acc[4] = fmaf(w1.x, xv, acc[4]); acc[5] = fmaf(w1.y, xv, acc[5]);
            acc[6] = fmaf(w1.z, xv, acc[6]); acc[7] = fmaf(w1.w, xv, acc[7]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (o0 + k < C) zb[(int64_t)(o0 + k) * HW + p] = acc[k];
    }
}

// ActNorm with a context net (actnorm.py:40-60): m = CN(c) = [t_b | logs_b] per sample (+ the shared NN_t / NN_logs
// under contextflow); z = (x - t_b) exp(-logs_b); ldj[b] = sum_c logs_b (reference quirk: no H W factor).
__global__ __launch_bounds__(256) void k_actnorm_ctx(const float* __restrict__ x, const float* __restrict__ m,
                                                     const float* __restrict__ t, const float* __restrict__ logs,
                                                     float* __restrict__ z, float* __restrict__ ldj, int C, int HW,
                                                     int64_t xbs) {
    __shared__ float tb[128], sb[128];
    __shared__ float scr[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float lsum = 0.f;
    if (tid < C) {
        float tv = m[(int64_t)b * 2 * C + tid], lv = m[(int64_t)b * 2 * C + C + tid];
        if (t != nullptr) { tv += t[tid]; lv += logs[tid]; }
        tb[tid] = tv; sb[tid] = expf(-lv);
        lsum = lv;
    }
    lsum = cf_block_sum<4>(lsum, scr);
    if (tid == 0) ldj[b] = lsum;
    const float* xb = x + (int64_t)b * xbs;
    float* zb = z + (int64_t)b * C * HW;
    for (int e = tid; e < C * HW; e += 256) { const int c = e / HW; zb[e] = (xb[e] - tb[c]) * sb[c]; }
}

// Conv1x1 and ActNorm of a specialist step in ONE pass over the sample (evaluation): z = (W_b x - t_b) exp(-logs_b) with
// W_b as in k_conv1x1_ctx (m1 = Conv1x1.CN(c), (B, C*C)) and t_b / logs_b as in k_actnorm_ctx (m2 = ActNorm.CN(c'), (B, 2C)).
// SQ: x is the un-squeezed (C/4, 2H, 2W) sample and Squeeze((2,2)) (squeeze.py:10-11) is folded into the staging reads.
//   ldj[b] (+)= H W (sum diag m1 + lad[0]) + sum_c logs_b + cadd       lad: log|det NN| (device scalar) or NULL
// One workgroup per sample, as the two kernels it replaces: one read of x, one write of z instead of two of each (+ the
// Squeeze copy), no intermediate tensor.
template <bool SQ>
__global__ __launch_bounds__(256) void k_affine_ctx(const float* __restrict__ x, const float* __restrict__ m1,
                                                    const float* __restrict__ Wm, const float* __restrict__ m2,
                                                    const float* __restrict__ t, const float* __restrict__ logs,
                                                    const float* __restrict__ lad, float cadd, float* __restrict__ z,
                                                    float* __restrict__ ldj, int C, int H, int W, int64_t xbs, int accumulate) {
    extern __shared__ __align__(16) float dyn[];
    const int CP = conv1x1_ctx_cp(C), HW = H * W;
    float* Wt = dyn;                                  // Wt[i][o] = W_b[o][i], row stride CP
    float* xs = dyn + C * CP;                         // [C][HW]
    float* scr = xs + C * HW;                         // [4]
    float* tb = scr + 4;                              // [C]  t_b
    float* sb = tb + C;                               // [C]  exp(-logs_b)
    float* ms = sb + C;                               // [C*C] this sample's Conv1x1.CN(c), staged before the transposed read
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (int64_t)b * xbs;
    stage_copy(ms, m1 + (int64_t)b * C * C, C * C, tid);
    float dsum = 0.f;
    if (tid < C) {
        float tv = m2[(int64_t)b * 2 * C + tid], lv = m2[(int64_t)b * 2 * C + C + tid];
        if (t != nullptr) { tv += t[tid]; lv += logs[tid]; }
        tb[tid] = tv; sb[tid] = expf(-lv);
        dsum = lv;                                    // joins the block sum below with weight 1 (the diagonal gets H W)
    }
    if constexpr (!SQ) {
        stage_copy(xs, xb, C * HW, tid);              // xs is 16-byte aligned: C * CP is a multiple of 4
    } else {
        for (int e0 = tid; e0 < C * HW; e0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = min(e0 + k * 256, C * HW - 1), c = e / HW, p = e - c * HW, yy = p / W, xx = p - yy * W;
                v[k] = xb[(c >> 2) * 4 * HW + (2 * yy + ((c >> 1) & 1)) * 2 * W + 2 * xx + (c & 1)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (e0 + k * 256 < C * HW) xs[e0 + k * 256] = v[k];
        }
    }
    __syncthreads();
    const float hw = (float)HW;
    for (int e0 = tid; e0 < CP * C; e0 += 8 * 256) {
        float wm[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wm[k] = Wm != nullptr ? Wm[min(e0 + k * 256, C * C - 1)] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + k * 256;
            if (e < CP * C) {
                const int o = e / C, i = e - o * C;
                float w = 0.f;
                if (o < C) {
                    const float v = ms[e];
                    w = o > i ? v : (o == i ? expf(v) : 0.f);
                    if (o == i) dsum = fmaf(hw, v, dsum);
                    if (Wm != nullptr) w += wm[k] - (o == i ? 1.f : 0.f);
                }
                Wt[i * CP + o] = w;
            }
        }
    }
    dsum = cf_block_sum<4>(dsum, scr);                // also the barrier that publishes Wt
    if (tid == 0) {
        const float v = dsum + (lad != nullptr ? hw * lad[0] : 0.f) + cadd;
        ldj[b] = accumulate ? ldj[b] + v : v;
    }
    float* zb = z + (int64_t)b * C * HW;
    const int nob = CP >> 3;
    for (int e = tid; e < nob * HW; e += 256) {
        const int ob = e / HW, p = e - ob * HW, o0 = ob * 8;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int i = 0; i < C; ++i) {
            const float xv = xs[i * HW + p];
            const float4 w0 = *reinterpret_cast<const float4*>(&Wt[i * CP + o0]);
            const float4 w1 = *reinterpret_cast<const float4*>(&Wt[i * CP + o0 + 4]);
            acc[0] = fmaf(w0.x, xv, acc[0]); acc[1] = fmaf(w0.y, xv, acc[1]);
            acc[2] = fmaf(w0.z, xv, acc[2]); acc[3] = fmaf(w0.w, xv, acc[3]);
            acc[4] = fmaf(w1.x, xv, acc[4]); acc[5] = fmaf(w1.y, xv, acc[5]);
            acc[6] = fmaf(w1.z, xv, acc[6]); acc[7] = fmaf(w1.w, xv, acc[7]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (o0 + k < C) zb[(int64_t)(o0 + k) * HW + p] = (acc[k] - tb[o0 + k]) * sb[o0 + k];
    }
}

// The same on the matrix pipe for the three levels of the image flows, ONE WAVE per sample and no LDS: per sample the
// product is (C x C) (C x HW) = 64 v_mfma_f32_16x16x4_f32 whatever the level, and every operand is read from global memory
// exactly once, in the MFMA's own register layout:
//   A (W_b, 16 rows x 4 k per instruction): lane (n = lane & 15, kk = lane >> 4) loads the float4 m1[o = 16 rt + n][16 g + 4 kk ..]
//     and uses its 4 values for 4 successive k-steps - the k order of a product is free as long as A and B agree;
//   B (x): for those k-steps lane (n, kk) needs x[i = 16 g + 4 kk + e][pixel of column n].  With 64 pixels or more, column n of
//     column tile j IS pixel 4 n + j: one float4 of x serves the four column tiles, and the four results of a row come
//     back as one float4 store.  On 4x4 images (16 pixels) the column is the pixel: dword loads, 64-byte runs.
//   SQ: i = 4 q + 2 dy + dx with q = 4 g + kk, e = 2 dy + dx - the four k-steps of a lane are the four phases of ONE channel
//     of the un-squeezed tensor: rows 2 y, 2 y + 1, 8 (2) consecutive floats each.
// tril / exp(diag) / + NN - I are applied to the A fragments in registers; t_b, logs_b come as float4 per (row tile, kk).
// BLK: m1 holds only the 16 x 16 blocks on and below the diagonal of the per-sample matrix, block (rt, g <= rt) at
// (rt (rt + 1) / 2 + g) * 256, row-major inside (the caller permutes the rows of Conv1x1.CN accordingly): the blocks above the
// diagonal are never needed, and a fragment load is 1 KB of contiguous memory instead of 16 runs of 64 bytes.
template <int C, int W, bool SQ, bool BLK>
__global__ __launch_bounds__(256) void k_affine_ctx_wave(const float* __restrict__ x, const float* __restrict__ m1,
                                                         const float* __restrict__ Wm, const float* __restrict__ m2,
                                                         const float* __restrict__ t, const float* __restrict__ logs,
                                                         const float* __restrict__ lad, float cadd, float* __restrict__ z,
                                                         float* __restrict__ ldj, int B, int64_t xbs, int accumulate) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int HW = W * W, RT = C / 16, NG = C / 16, WIDE = HW >= 64, NT = WIDE ? HW / 64 : 1;
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;                                // a whole wave: the kernel has no barrier
    constexpr int NBLK = (C / 16) * (C / 16 + 1) / 2;
    const float* mb = m1 + (int64_t)b * (BLK ? NBLK * 256 : C * C);
    const float* xb = x + (int64_t)b * xbs;
    // ---- A fragments.  W_b is block lower triangular on the 16 x 16 grid of fragments: blocks above the diagonal hold NN alone
    // (never read from m1), blocks below m1 + NN, the diagonal blocks the element-wise tril / exp(diag) / - I.
    float4 a[RT][NG];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < NG; ++g)
            a[rt][g] = g <= rt ? *reinterpret_cast<const float4*>(BLK ? mb + (rt * (rt + 1) / 2 + g) * 256 + n * 16 + 4 * kk
                                                                        : mb + (16 * rt + n) * C + 16 * g + 4 * kk)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    float dsum = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int o = 16 * rt + n, i0 = 16 * g + 4 * kk;
            float v[4] = {a[rt][g].x, a[rt][g].y, a[rt][g].z, a[rt][g].w};
            if (g == rt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = i0 + e;
                    const float ex = expf(v[e]) - (Wm != nullptr ? 1.f : 0.f);
                    if (o == i) dsum += v[e];
                    v[e] = o > i ? v[e] : (o == i ? ex : 0.f);
                }
            }
            if (Wm != nullptr) {
                const float4 wv = *reinterpret_cast<const float4*>(Wm + o * C + i0);
                v[0] += wv.x; v[1] += wv.y; v[2] += wv.z; v[3] += wv.w;
            }
            a[rt][g] = make_float4(v[0], v[1], v[2], v[3]);
        }
    // ---- the product
    f32x4 acc[RT][NT][WIDE ? 4 : 1];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int j = 0; j < (WIDE ? 4 : 1); ++j) acc[rt][T][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float xv[4][WIDE ? 4 : 1];                 // [k-step e][column tile j]
            if constexpr (WIDE) {
                if constexpr (!SQ) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 v = *reinterpret_cast<const float4*>(xb + (16 * g + 4 * kk + e) * HW + 64 * T + 4 * n);
                        xv[e][0] = v.x; xv[e][1] = v.y; xv[e][2] = v.z; xv[e][3] = v.w;
                    }
                } else {
                    const int p0 = 64 * T + 4 * n, yy = p0 / W, xx = p0 - yy * W;       // 4 pixels of one image row (W >= 4)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const float* row = xb + (4 * g + kk) * 4 * HW + (2 * yy + dy) * 2 * W + 2 * xx;
                        const float4 u0 = *reinterpret_cast<const float4*>(row), u1 = *reinterpret_cast<const float4*>(row + 4);
                        xv[2 * dy][0] = u0.x; xv[2 * dy][1] = u0.z; xv[2 * dy][2] = u1.x; xv[2 * dy][3] = u1.z;
                        xv[2 * dy + 1][0] = u0.y; xv[2 * dy + 1][1] = u0.w; xv[2 * dy + 1][2] = u1.y; xv[2 * dy + 1][3] = u1.w;
                    }
                }
            } else {
                if constexpr (!SQ) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e][0] = xb[(16 * g + 4 * kk + e) * HW + n];
                } else {
                    const int yy = n / W, xx = n - yy * W;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const float2 u = *reinterpret_cast<const float2*>(xb + (4 * g + kk) * 4 * HW + (2 * yy + dy) * 2 * W + 2 * xx);
                        xv[2 * dy][0] = u.x; xv[2 * dy + 1][0] = u.y;
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float av[4] = {a[rt][g].x, a[rt][g].y, a[rt][g].z, a[rt][g].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < (WIDE ? 4 : 1); ++j)
                        acc[rt][T][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], xv[e][j], acc[rt][T][j], 0, 0, 0);
            }
        }
    // ---- ActNorm epilogue: rows o = 16 rt + 4 kk + r of this lane
    float* zb = z + (int64_t)b * C * HW;
    float lsum = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int o0 = 16 * rt + 4 * kk;
        float4 tv = *reinterpret_cast<const float4*>(m2 + (int64_t)b * 2 * C + o0);
        float4 lv = *reinterpret_cast<const float4*>(m2 + (int64_t)b * 2 * C + C + o0);
        if (t != nullptr) {
            const float4 t0 = *reinterpret_cast<const float4*>(t + o0), l0 = *reinterpret_cast<const float4*>(logs + o0);
            tv.x += t0.x; tv.y += t0.y; tv.z += t0.z; tv.w += t0.w;
            lv.x += l0.x; lv.y += l0.y; lv.z += l0.z; lv.w += l0.w;
        }
        if (n == 0) lsum += (lv.x + lv.y) + (lv.z + lv.w);
        const float tt[4] = {tv.x, tv.y, tv.z, tv.w};
        const float ss[4] = {expf(-lv.x), expf(-lv.y), expf(-lv.z), expf(-lv.w)};
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (WIDE) {
                    float4 o4;
                    o4.x = (acc[rt][T][0][r] - tt[r]) * ss[r]; o4.y = (acc[rt][T][1][r] - tt[r]) * ss[r];
                    o4.z = (acc[rt][T][2][r] - tt[r]) * ss[r]; o4.w = (acc[rt][T][3][r] - tt[r]) * ss[r];
                    *reinterpret_cast<float4*>(zb + (o0 + r) * HW + 64 * T + 4 * n) = o4;
                } else {
                    zb[(o0 + r) * HW + n] = (acc[rt][T][0][r] - tt[r]) * ss[r];
                }
            }
    }
    const float tot = cf_wave_sum(fmaf((float)HW, dsum, lsum));
    if (lane == 0) {
        const float v = tot + (lad != nullptr ? (float)HW * lad[0] : 0.f) + cadd;
        ldj[b] = accumulate ? ldj[b] + v : v;
    }
}

// h[b, c, p] = act(h[b, c, p] + bias[b, c])      (the CN(c) term of the coupling net, coupling.py:44-47)
__global__ __launch_bounds__(256) void k_add_sample_bias(float* __restrict__ h, const float* __restrict__ bias,
                                                         int HW, int64_t total, int relu) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const float v = h[e] + bias[e / HW];
        h[e] = relu ? fmaxf(v, 0.f) : v;
    }
}

// GaussianMixtureDistribution.log_prob with a context net (gaussian.py:142-158): per-sample additive shifts cm / cs
// (B, 2, M, K, D) of the component means and pre-softplus scales, constant over (h, w):
//   out[b, m] (+)= logsumexp_k [ logw[m,k] + sum_{d,p} ( -1/2 ((x - mu - cm)/sig)^2 - log sig - 1/2 log 2pi ) ],
//   sig = softplus(sG + cs).
// Work layout (forward and backward): a workgroup owns S samples (x in LDS), so every mu / sG value fetched from L2 is
// used S times; a wave owns whole channels - SEG = 2^k <= min(64, HW) lanes walk the pixels of one channel, 64/SEG
// channels side by side - so the per-channel shifts are loaded once per (component, channel) and no index division
// happens per term.  4 transcendentals per term (exp, log, rcp, log) on the hardware units: ~1e-7 relative each, far
// below the 1e-5 bits/dim tolerance after the sum over D*HW terms.
struct GmmCtxLanes {
    int SEG, CPW, G, sub, pl;
    __device__ __forceinline__ GmmCtxLanes(int D, int HW, int lane) {
        SEG = 64;
        while (SEG > HW) SEG >>= 1;                 // largest power of two <= min(64, HW)
        CPW = 64 / SEG;
        G = (D + CPW - 1) / CPW;                    // channel groups of the sample
        sub = lane / SEG;
        pl = lane - sub * SEG;
    }
};

__device__ __forceinline__ float softplus_fast(float s) { return s > 20.f ? s : __logf(1.0f + __expf(s)); }   // threshold 20, as torch

// log-joint lp[s][mk] of the S samples of this workgroup -> LDS (all threads must call; ends with a barrier)
// TAB: the scale shifts take few distinct values (embedding lookups, model.py:157,162): 1/sig and sum log sig come from
// tables indexed by the sample's key (cf_gmm_ctx_tables) - no transcendental left in the term loop.
struct GmmTab {
    const float* inv;      // (U, MK, N)  1 / softplus(sG + cs_u)
    const float* dsig;     // (U, MK, N)  softplus'(sG + cs_u)    (backward only)
    const float* lsum;     // (U, MK)     sum_e log softplus(sG + cs_u)
    const int* key;        // (B)         u of every sample
    const int* ckey;       // (B) or NULL: the mean shifts come from a table too - c is then (Um, MK, D) and ckey its row
};

template <int S, bool TAB>
__device__ __forceinline__ void gmm_ctx_logjoint(const float* __restrict__ xs, float* __restrict__ lpw, float* __restrict__ lp,
                                                 const float* __restrict__ mG, const float* __restrict__ sG,
                                                 const float* __restrict__ logw, const float* __restrict__ c, const int64_t (&co)[S],
                                                 const GmmTab& tb, const int64_t (&io)[S], int b0, int B, int MK,
                                                 int D, int HW, int lane, int wave) {
    const GmmCtxLanes L(D, HW, lane);
    const int N = D * HW;
    for (int mk = 0; mk < MK; ++mk) {
        const float* mu = mG + (int64_t)mk * N;
        const float* sg = sG + (int64_t)mk * N;
        float acc[S];
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = 0.f;
        for (int g = wave; g < L.G; g += 4) {
            const int d = g * L.CPW + L.sub;
            if (d < D) {
                float cm[S], cs[S];
#pragma unroll
                for (int s = 0; s < S; ++s) { cm[s] = c[co[s] + mk * D + d]; cs[s] = TAB ? 0.f : c[co[s] + (MK + mk) * D + d]; }
                for (int p = L.pl; p < HW; p += L.SEG) {
                    const int e = d * HW + p;
                    if constexpr (TAB) {
                        const float m = mu[e];
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const float r = (xs[s * N + e] - m - cm[s]) * tb.inv[io[s] + (int64_t)mk * N + e];
                            acc[s] = fmaf(-0.5f * r, r, acc[s]);
                        }
                    } else {
                        const float m = mu[e], sv = sg[e];
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const float sig = softplus_fast(sv + cs[s]);
                            const float r = (xs[s * N + e] - m - cm[s]) * __frcp_rn(sig);
                            acc[s] += -0.5f * r * r - __logf(sig);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float v = cf_wave_sum(acc[s]);
            if (lane == 0) lpw[(wave * S + s) * MK + mk] = v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S * MK; i += 256) {
        const int mk = i % MK;
        float v = ((lpw[i] + lpw[S * MK + i]) + (lpw[2 * S * MK + i] + lpw[3 * S * MK + i])) - 0.91893853320467274178f * (float)N +
                  logw[mk];
        if constexpr (TAB) v -= tb.lsum[(int64_t)tb.key[min(b0 + i / MK, B - 1)] * MK + mk];
        lp[i] = v;
    }
    __syncthreads();
}

// tables of GmmTab: one workgroup per (u, mk) row
__global__ __launch_bounds__(256) void k_gmm_ctx_tables(const float* __restrict__ sG, const float* __restrict__ cst,
                                                        float* __restrict__ inv, float* __restrict__ dsig,
                                                        float* __restrict__ lsum, int MK, int D, int HW) {
    __shared__ float part[4];
    const int u = blockIdx.x / MK, mk = blockIdx.x - u * MK, N = D * HW, tid = threadIdx.x;
    const float* sg = sG + (int64_t)mk * N;
    const float* cs = cst + ((int64_t)u * MK + mk) * D;
    const int64_t o = ((int64_t)u * MK + mk) * N;
    float acc = 0.f;
    for (int e = tid; e < N; e += 256) {
        const float t = sg[e] + cs[e / HW];
        const float sig = softplus_fast(t);
        inv[o + e] = __frcp_rn(sig);
        if (dsig) dsig[o + e] = t > 20.f ? 1.f : __frcp_rn(1.0f + __expf(-t));
        acc += __logf(sig);
    }
    acc = cf_wave_sum(acc);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) lsum[(int64_t)u * MK + mk] = (part[0] + part[1]) + (part[2] + part[3]);
}

// lp_out (optional, (B, M*K)): the per-component log-joints, kept by the training forward for cf_gmm_ctx_bwd
template <int S, bool TAB>
__global__ __launch_bounds__(256) void k_gmm_ctx(const float* __restrict__ x, const float* __restrict__ mG,
                                                 const float* __restrict__ sG, const float* __restrict__ logw,
                                                 const float* __restrict__ c, float* __restrict__ out,
                                                 float* __restrict__ lp_out, int B, int M, int K, int D, int HW,
                                                 int64_t xbs, int accumulate, GmmTab tb) {
    extern __shared__ __align__(16) float lds[];
    const int N = D * HW, MK = M * K;
    float* xs = lds;                       // [S][N]
    float* lpw = xs + S * N;               // [4][S][MK] per-wave partial sums
    float* lp = lpw + 4 * S * MK;          // [S][MK]
    const int b0 = blockIdx.x * S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t co[S], io[S];                  // offsets of the samples' shift rows in c (offsets, not pointers: see cf_step_common.h)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int b = min(b0 + s, B - 1);                       // ragged last workgroup: recompute the last sample
        co[s] = (TAB && tb.ckey) ? (int64_t)tb.ckey[b] * MK * D : (int64_t)b * 2 * MK * D;
        io[s] = TAB ? (int64_t)tb.key[b] * MK * N : 0;
        stage_copy(xs + s * N, x + (int64_t)b * xbs, N, tid);
    }
    __syncthreads();
    gmm_ctx_logjoint<S, TAB>(xs, lpw, lp, mG, sG, logw, c, co, tb, io, b0, B, MK, D, HW, lane, wave);
    for (int i = tid; i < S * M; i += 256) {
        const int s = i / M, m = i - s * M;
        if (b0 + s >= B) continue;
        const float* l = lp + s * MK + m * K;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, l[k]);
        float z = 0.f;
        for (int k = 0; k < K; ++k) z += expf(l[k] - mx);
        const float v = mx + logf(z);
        if (accumulate) out[(int64_t)(b0 + s) * M + m] += v;
        else out[(int64_t)(b0 + s) * M + m] = v;
    }
    if (lp_out)
        for (int i = tid; i < S * MK; i += 256)
            if (b0 + i / MK < B) lp_out[(int64_t)b0 * MK + i] = lp[i];
}

// ConditionalGaussianDistribution.sample (gaussian.py:263-270): c = [mean | log_scale] (B, 2D) from the context
// embedding; x = mean + exp(log_scale) eps; logp[b] = sum_d (-1/2 log 2pi - log_scale - eps^2 / 2).
__global__ __launch_bounds__(64) void k_cond_gauss_sample(const float* __restrict__ c, const float* __restrict__ eps,
                                                          float* __restrict__ x, float* __restrict__ logp, int D) {
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) {
        const float mean = c[(int64_t)b * 2 * D + d], ls = c[(int64_t)b * 2 * D + D + d], e = eps[(int64_t)b * D + d];
        const float xv = mean + expf(ls) * e;
        x[(int64_t)b * D + d] = xv;
        const float r = xv - mean;
        acc += -0.91893853320467274178f - ls - 0.5f * expf(-2.0f * ls) * r * r;
    }
    acc = cf_wave_sum(acc);
    if (threadIdx.x == 0) logp[b] = acc;
}

// Sigmoid activation layer with its log-det (activations.py:234-238, temperature 1):
// y = sigmoid(x), ldj[b] = sum_d (-softplus(-x) - softplus(x)).
__global__ __launch_bounds__(64) void k_sigmoid_ldj(const float* __restrict__ x, float* __restrict__ y,
                                                    float* __restrict__ ldj, int D) {
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) {
        const float v = x[(int64_t)b * D + d];
        y[(int64_t)b * D + d] = 1.0f / (1.0f + expf(-v));
        const float a = fabsf(v);
        acc -= a + 2.0f * log1pf(expf(-a));            // softplus(v) + softplus(-v) = |v| + 2 log(1 + e^-|v|)
    }
    acc = cf_wave_sum(acc);
    if (threadIdx.x == 0) ldj[b] = acc;
}

// ---- backward of the context branches (specialist training under contextflow: the CN nets and the priors' embedding
// tables are the trainable parameters; the generalist's own parameters are frozen, model.py / coupling.py:36) ------

// Conv1x1 with a context net, backward: gx[b] = W_b^T gz[b];  G = sum_p gz[b][:,p] x[b][:,p]^T;
// gm[b][o][i] = G[o][i] (o > i) | G[i][i] exp(m_ii) + H W gld[b] (o == i) | 0 (o < i)      (d/d CN(c) output)
// One workgroup per sample; W_b (row stride CP), x and gz (row stride HWP = odd: the (o, i) products below walk rows
// with one lane per row) in LDS.
__global__ __launch_bounds__(256) void k_conv1x1_ctx_bwd(const float* __restrict__ x, const float* __restrict__ m,
                                                         const float* __restrict__ Wm, const float* __restrict__ gz,
                                                         const float* __restrict__ gld, float* __restrict__ gx,
                                                         float* __restrict__ gm, int C, int HW, int64_t xbs, int64_t gzbs) {
    extern __shared__ __align__(16) float dyn[];
    const int CP = (C + 7) & ~7, HWP = HW | 1;
    float* Wb = dyn;                                  // Wb[o][i], row stride CP (zero padded)
    float* xs = dyn + C * CP;                         // [C][HWP]
    float* gs = xs + C * HWP;                         // [C][HWP]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* mb = m + (int64_t)b * C * C;
    const float* xb = x + (int64_t)b * xbs;
    const float* gb = gz + (int64_t)b * gzbs;
    for (int e0 = tid; e0 < C * HW; e0 += 4 * 256) {   // batches of independent loads (padded rows: scalar stores)
        float xv[4], gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int e = min(e0 + i * 256, C * HW - 1); xv[i] = xb[e]; gv[i] = gb[e]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i * 256;
            if (e < C * HW) { const int c = e / HW, p = e - c * HW; xs[c * HWP + p] = xv[i]; gs[c * HWP + p] = gv[i]; }
        }
    }
    for (int e0 = tid; e0 < C * CP; e0 += 8 * 256) {   // batches of independent loads (see k_conv1x1_ctx)
        float mv[8], wm[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = min(e0 + k * 256, C * CP - 1), o = e / CP, i = min(e - o * CP, C - 1);
            mv[k] = mb[o * C + i];
            wm[k] = Wm != nullptr ? Wm[o * C + i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + k * 256;
            if (e < C * CP) {
                const int o = e / CP, i = e - o * CP;
                float w = 0.f;
                if (i < C) {
                    w = o > i ? mv[k] : (o == i ? expf(mv[k]) : 0.f);
                    if (Wm != nullptr) w += wm[k] - (o == i ? 1.f : 0.f);
                }
                Wb[e] = w;
            }
        }
    }
    __syncthreads();
    float* gxb = gx + (int64_t)b * C * HW;
    const int nib = CP >> 3;
    for (int e = tid; e < nib * HW; e += 256) {       // gx = W_b^T gz: a pixel and 8 input channels per thread
        const int ib = e / HW, p = e - ib * HW, i0 = ib * 8;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int o = 0; o < C; ++o) {
            const float gv = gs[o * HWP + p];
            const float4 w0 = *reinterpret_cast<const float4*>(&Wb[o * CP + i0]);
            const float4 w1 = *reinterpret_cast<const float4*>(&Wb[o * CP + i0 + 4]);
            acc[0] = fmaf(w0.x, gv, acc[0]); acc[1] = fmaf(w0.y, gv, acc[1]);
            acc[2] = fmaf(w0.z, gv, acc[2]); acc[3] = fmaf(w0.w, gv, acc[3]);
            acc[4] = fmaf(w1.x, gv, acc[4]); acc[5] = fmaf(w1.y, gv, acc[5]);
            acc[6] = fmaf(w1.z, gv, acc[6]); acc[7] = fmaf(w1.w, gv, acc[7]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k < C) gxb[(int64_t)(i0 + k) * HW + p] = acc[k];
    }
    float* gmb = gm + (int64_t)b * C * C;
    const float gl = gld[b] * (float)HW;
    for (int e = tid; e < C * C; e += 256) {          // lanes walk i (rows of xs: odd stride), o is shared
        const int o = e / C, i = e - o * C;
        float g = 0.f;
        if (o >= i) {
            const float* go = gs + o * HWP;
            const float* xi = xs + i * HWP;
            float g0 = 0.f, g1 = 0.f;
            int p = 0;
            for (; p + 1 < HW; p += 2) { g0 = fmaf(go[p], xi[p], g0); g1 = fmaf(go[p + 1], xi[p + 1], g1); }
            if (p < HW) g0 = fmaf(go[p], xi[p], g0);
            g = g0 + g1;
            if (o == i) g = g * expf(mb[e]) + gl;
        }
        gmb[e] = g;
    }
}

// ActNorm with a context net, backward: gx = gz exp(-l_b); gm[b] = [ -exp(-l_b) sum_p gz | -sum_p gz z + gld[b] ]
__global__ __launch_bounds__(256) void k_actnorm_ctx_bwd(const float* __restrict__ x, const float* __restrict__ m,
                                                         const float* __restrict__ t, const float* __restrict__ logs,
                                                         const float* __restrict__ gz, const float* __restrict__ gld,
                                                         float* __restrict__ gx, float* __restrict__ gm, int C, int HW,
                                                         int64_t xbs, int64_t gzbs) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = x + (int64_t)b * xbs;
    const float* gb = gz + (int64_t)b * gzbs;
    float* gxb = gx + (int64_t)b * C * HW;
    for (int c = wave; c < C; c += 4) {               // a wave owns a channel at a time
        float tv = m[(int64_t)b * 2 * C + c], lv = m[(int64_t)b * 2 * C + C + c];
        if (t != nullptr) { tv += t[c]; lv += logs[c]; }
        const float s = expf(-lv);
        float s0 = 0.f, s1 = 0.f;
        for (int p = lane; p < HW; p += 64) {
            const float g = gb[(int64_t)c * HW + p], z = (xb[(int64_t)c * HW + p] - tv) * s;
            gxb[(int64_t)c * HW + p] = g * s;
            s0 += g; s1 = fmaf(g, z, s1);
        }
        s0 = cf_wave_sum(s0); s1 = cf_wave_sum(s1);
        if (lane == 0) {
            gm[(int64_t)b * 2 * C + c] = -s * s0;
            gm[(int64_t)b * 2 * C + C + c] = gld[b] - s1;
        }
    }
}

// out[b, c] = sum_p a[b, c, p]      (d/d of a per-sample bias: row sums of a gradient plane)
__global__ __launch_bounds__(256) void k_sample_channel_sums(const float* __restrict__ a, float* __restrict__ out, int HW,
                                                             int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int p = lane; p < HW; p += 64) s += a[row * HW + p];
    s = cf_wave_sum(s);
    if (lane == 0) out[row] = s;
}

// y = gy * (x > 0)   (ReLU backward of the CN nets)
__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ x, const float* __restrict__ gy,
                                                  float* __restrict__ out, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        out[e] = x[e] > 0.f ? gy[e] : 0.f;
}

// GMM with a context net, backward w.r.t. x and the per-sample shifts (gaussian.py:142-158; parameters frozen):
// r[mk] = g[b,m] softmax_k(lp)[mk];  with d = x - mu - cm, sig = softplus(s), s = sG + cs:
//   gx[e]      = sum_mk r * (-d / sig^2)
//   gc[0][mk][dch] = sum_hw r * d / sig^2
//   gc[1][mk][dch] = sum_hw r * (d^2 / sig^3 - 1 / sig) * sigmoid(s)
// Same work layout as k_gmm_ctx: S samples per workgroup, a wave owns channels.  With the channel fixed and the
// component loop inside, gx accumulates in registers (PIT pixels per lane and sample) and the (component, channel)
// sums are segmented shuffle reductions over the SEG lanes of the channel.  lp_in (optional): the log-joints kept by
// the forward; without it they are recomputed first.
template <int S, bool TAB>
__global__ __launch_bounds__(256) void k_gmm_ctx_bwd(const float* __restrict__ x, const float* __restrict__ mG,
                                                     const float* __restrict__ sG, const float* __restrict__ logw,
                                                     const float* __restrict__ c, const float* __restrict__ g,
                                                     const float* __restrict__ lp_in, float* __restrict__ gx,
                                                     float* __restrict__ gc, int B, int M, int K, int D, int HW,
                                                     int64_t xbs, GmmTab tb) {
    constexpr int PIT = 4;                 // pixels per lane and pass
    extern __shared__ __align__(16) float lds[];
    const int N = D * HW, MK = M * K;
    float* xs = lds;                       // [S][N]
    float* lpw = xs + S * N;               // [4][S][MK]
    float* lp = lpw + 4 * S * MK;          // [S][MK] log-joint, then responsibilities x upstream gradient
    const int b0 = blockIdx.x * S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t co[S], io[S];
    bool live[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int b = min(b0 + s, B - 1);
        live[s] = b0 + s < B;
        co[s] = (int64_t)b * 2 * MK * D;
        io[s] = TAB ? (int64_t)tb.key[b] * MK * N : 0;
        stage_copy(xs + s * N, x + (int64_t)b * xbs, N, tid);
    }
    __syncthreads();
    if (lp_in) {
        for (int i = tid; i < S * MK; i += 256) lp[i] = lp_in[(int64_t)min(b0 + i / MK, B - 1) * MK + i % MK];
        __syncthreads();
    } else {
        gmm_ctx_logjoint<S, TAB>(xs, lpw, lp, mG, sG, logw, c, co, tb, io, b0, B, MK, D, HW, lane, wave);
    }
    float* rr = lpw;                       // [S][MK] responsibilities (lpw is free now)
    for (int i = tid; i < S * M; i += 256) {
        const int s = i / M, m = i - s * M;
        const float* l = lp + s * MK + m * K;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, l[k]);
        float z = 0.f;
        for (int k = 0; k < K; ++k) z += expf(l[k] - mx);
        const float gm = g[(int64_t)min(b0 + s, B - 1) * M + m] / z;
        for (int k = 0; k < K; ++k) rr[s * MK + m * K + k] = expf(l[k] - mx) * gm;
    }
    __syncthreads();
    const GmmCtxLanes L(D, HW, lane);
    for (int gi = wave; gi < L.G; gi += 4) {
        const int d = gi * L.CPW + L.sub;
        const bool dok = d < D;
        for (int p0 = 0; p0 < HW; p0 += PIT * L.SEG) {          // one pass unless HW > 4 SEG
            float gxa[S][PIT];
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int it = 0; it < PIT; ++it) gxa[s][it] = 0.f;
            for (int mk = 0; mk < MK; ++mk) {
                const float* mu = mG + (int64_t)mk * N;
                const float* sg = sG + (int64_t)mk * N;
                float cm[S], cs[S], a0[S], a1[S], r[S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    r[s] = rr[s * MK + mk];
                    cm[s] = dok ? c[co[s] + mk * D + d] : 0.f;
                    cs[s] = dok ? c[co[s] + (MK + mk) * D + d] : 0.f;
                    a0[s] = 0.f;
                    a1[s] = 0.f;
                }
#pragma unroll
                for (int it = 0; it < PIT; ++it) {
                    const int p = p0 + it * L.SEG + L.pl;
                    if (dok && p < HW) {
                        const int e = d * HW + p;
                        const float m = mu[e];
                        float sv = 0.f;
                        if constexpr (!TAB) sv = sg[e];
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            float inv, dsig;
                            if constexpr (TAB) {
                                inv = tb.inv[io[s] + (int64_t)mk * N + e];
                                dsig = tb.dsig[io[s] + (int64_t)mk * N + e];
                            } else {
                                const float t = sv + cs[s];
                                inv = __frcp_rn(softplus_fast(t));
                                dsig = t > 20.f ? 1.f : __frcp_rn(1.0f + __expf(-t));           // softplus'
                            }
                            const float dd = xs[s * N + e] - m - cm[s];
                            const float q = dd * inv * inv;                                  // d / sig^2
                            gxa[s][it] = fmaf(-r[s], q, gxa[s][it]);
                            a0[s] += q;
                            a1[s] += (dd * q * inv - inv) * dsig;
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    for (int o = L.SEG >> 1; o > 0; o >>= 1) { a0[s] += __shfl_xor(a0[s], o, 64); a1[s] += __shfl_xor(a1[s], o, 64); }
                    if (L.pl == 0 && dok && live[s]) {
                        float* gcb = gc + co[s];
                        if (p0 == 0) { gcb[mk * D + d] = r[s] * a0[s]; gcb[(MK + mk) * D + d] = r[s] * a1[s]; }
                        else { gcb[mk * D + d] += r[s] * a0[s]; gcb[(MK + mk) * D + d] += r[s] * a1[s]; }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int it = 0; it < PIT; ++it) {
                    const int p = p0 + it * L.SEG + L.pl;
                    if (dok && p < HW && live[s]) gx[(int64_t)(b0 + s) * N + d * HW + p] = gxa[s][it];
                }
        }
    }
}

// h[b, c2, p] += x[b, c2 % C, p]   (the identity branch of MaskedResidualBlock2d: x repeated along channels,
// autoregressive/masked_conv_2d.py:93-98)
__global__ __launch_bounds__(256) void k_add_repeat(float* __restrict__ h, const float* __restrict__ x, int C2, int C,
                                                    int HW, int64_t total) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / ((int64_t)C2 * HW);
        const int r = (int)(e - b * (int64_t)C2 * HW), c2 = r / HW, p = r - c2 * HW;
        h[e] += x[(b * C + (c2 % C)) * HW + p];
    }
}

// Gradient w.r.t. the prior's OWN parameters for specialists trained without contextflow (gaussian.py:130-137: mG / sG / wG
// stay trainable), table form of the scale shifts:
//   gm[mk][e] = sum_b r[b][mk] d / sig^2,   gs[mk][e] = sum_b r[b][mk] (d^2 / sig^3 - 1 / sig) softplus'(s)
// with d = x - mu - cm, r = upstream gradient x responsibility.  A thread owns one (mk, e) and walks a slab of SB
// samples; the per-slab partials (nb, MK, N) are summed by the caller in slab order.
__global__ __launch_bounds__(256) void k_gmm_ctx_pgrad(const float* __restrict__ x, const float* __restrict__ mG,
                                                       const float* __restrict__ c, const float* __restrict__ r,
                                                       float* __restrict__ pgm, float* __restrict__ pgs, int B, int MK,
                                                       int D, int HW, int64_t xbs, int SB, GmmTab tb) {
    const int N = D * HW, e = blockIdx.x * 256 + threadIdx.x, mk = blockIdx.y;
    if (e >= N) return;
    const int d = e / HW, b0 = blockIdx.z * SB, b1 = min(B, b0 + SB);
    const float mu = mG[(int64_t)mk * N + e];
    float am = 0.f, as = 0.f;
    for (int b = b0; b < b1; ++b) {
        const int64_t o = ((int64_t)tb.key[b] * MK + mk) * N + e;
        const float rb = r[(int64_t)b * MK + mk];
        const float iv = tb.inv[o], ds = tb.dsig[o];
        const float dd = x[(int64_t)b * xbs + e] - mu - c[(int64_t)b * 2 * MK * D + mk * D + d];
        const float q = dd * iv * iv;
        am = fmaf(rb, q, am);
        as = fmaf(rb, (dd * q * iv - iv) * ds, as);
    }
    const int64_t po = ((int64_t)blockIdx.z * MK + mk) * N + e;
    pgm[po] = am;
    pgs[po] = as;
}

// samples per workgroup of the context-GMM kernels: as many as fit (x of S samples + the log-joint scratch in LDS)
static int gmm_ctx_group(int N, int MK, size_t* lds) {
    for (int S = 4; S >= 1; S >>= 1) {
        *lds = (size_t)(S * N + 5 * S * MK) * sizeof(float);
        if (*lds <= 64 * 1024) return S;
    }
    return *lds <= 160 * 1024 ? 1 : 0;
}

template <bool TAB>
static int gmm_ctx_fwd_launch(const char* who, const float* x, const float* mG, const float* sG, const float* logw, const float* c,
                              float* out, float* lp_out, int B, int M, int K, int D, int HW, int64_t xbs, int accumulate,
                              GmmTab tb, hipStream_t st) {
    size_t lds;
    const int S = gmm_ctx_group(D * HW, M * K, &lds);
    if (S == 0) { cf_set_error("%s: D*HW=%d needs %zu B of LDS", who, D * HW, lds); return CF_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_gmm_ctx<1, TAB>, 160 * 1024, raised, __func__)) return rc_;
    }
    const dim3 grid((B + S - 1) / S);
    if (S == 4) k_gmm_ctx<4, TAB><<<grid, dim3(256), lds, st>>>(x, mG, sG, logw, c, out, lp_out, B, M, K, D, HW, xbs, accumulate, tb);
    else if (S == 2) k_gmm_ctx<2, TAB><<<grid, dim3(256), lds, st>>>(x, mG, sG, logw, c, out, lp_out, B, M, K, D, HW, xbs, accumulate, tb);
    else k_gmm_ctx<1, TAB><<<grid, dim3(256), lds, st>>>(x, mG, sG, logw, c, out, lp_out, B, M, K, D, HW, xbs, accumulate, tb);
    return 0;
}

template <bool TAB>
static int gmm_ctx_bwd_launch(const char* who, const float* x, const float* mG, const float* sG, const float* logw, const float* c,
                              const float* g, const float* lp, float* gx, float* gc, int B, int M, int K, int D, int HW,
                              int64_t xbs, GmmTab tb, hipStream_t st) {
    size_t lds;
    const int S = gmm_ctx_group(D * HW, M * K, &lds);
    if (S == 0) { cf_set_error("%s: D*HW=%d needs %zu B of LDS", who, D * HW, lds); return CF_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_gmm_ctx_bwd<1, TAB>, 160 * 1024, raised, __func__)) return rc_;
    }
    const dim3 grid((B + S - 1) / S);
    if (S == 4) k_gmm_ctx_bwd<4, TAB><<<grid, dim3(256), lds, st>>>(x, mG, sG, logw, c, g, lp, gx, gc, B, M, K, D, HW, xbs, tb);
    else if (S == 2) k_gmm_ctx_bwd<2, TAB><<<grid, dim3(256), lds, st>>>(x, mG, sG, logw, c, g, lp, gx, gc, B, M, K, D, HW, xbs, tb);
    else k_gmm_ctx_bwd<1, TAB><<<grid, dim3(256), lds, st>>>(x, mG, sG, logw, c, g, lp, gx, gc, B, M, K, D, HW, xbs, tb);
    return 0;
}

}  // namespace

extern "C" {

int cf_ctx_encode(const int64_t* ctx, const float* u, const float* qbins, const int64_t* card, float* out, int B, int nctx,
                  int width, int onehot, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(ctx && u && out && B >= 0 && nctx > 0 && width > 0 && (!onehot || card) && (onehot == 2 || qbins));
    const int64_t total = (int64_t)B * width;
    k_ctx_encode<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, cf_s(stream)>>>(ctx, u, qbins, card, out, B, nctx,
                                                                                       width, onehot);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_conv1x1_ctx(const float* x, const float* m, const float* Wm, float* z, float* ldj, int B, int C, int HW,
                   int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && m && z && ldj && B >= 0 && C > 0 && HW > 0);
    const size_t lds = (size_t)(C * conv1x1_ctx_cp(C) + C * HW + 4 + C * C) * sizeof(float);
    if (lds > 160 * 1024) { cf_set_error("cf_conv1x1_ctx: C=%d, H*W=%d need %zu B of LDS", C, HW, lds); return CF_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_conv1x1_ctx, 160 * 1024, raised, __func__)) return rc_;
    }
    k_conv1x1_ctx<<<dim3(B), dim3(256), lds, cf_s(stream)>>>(x, m, Wm, z, ldj, C, HW, x_bstride);
    CF_LAUNCH_CHECK();
    return 0;
}

// number of floats per sample of the blocked form of m1 (cf_affine_ctx_fwd, m1_blocked != 0), 0 where it is not offered
int cf_affine_ctx_blocked_floats(int C, int H, int W) {
    const bool wave = H == W && ((C == 16 && W == 16) || (C == 32 && W == 8) || (C == 64 && W == 4));
    return wave ? (C / 16) * (C / 16 + 1) / 2 * 256 : 0;
}

int cf_affine_ctx_fwd(const float* x, const float* m1, const float* Wm, const float* m2, const float* t, const float* logs,
                      const float* lad, float cadd, float* z, float* ldj, int B, int C, int H, int W, int64_t x_bstride,
                      int in_squeeze, int accumulate, int m1_blocked, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(!m1_blocked || cf_affine_ctx_blocked_floats(C, H, W) > 0);
    CF_REQUIRE(x && m1 && m2 && z && ldj && B >= 0 && C > 0 && C <= 256 && H > 0 && W > 0 && ((t == nullptr) == (logs == nullptr)) &&
               x_bstride >= (int64_t)C * H * W && (!in_squeeze || C % 4 == 0));
    // the image flows' three levels: one wave per sample on the matrix pipe, operands straight from global memory
    const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(m1) | reinterpret_cast<uintptr_t>(m2) |
                      reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(Wm) | reinterpret_cast<uintptr_t>(t) |
                      reinterpret_cast<uintptr_t>(logs)) & 15) == 0 && x_bstride % 4 == 0;
    if (al && H == W && ((C == 16 && W == 16) || (C == 32 && W == 8) || (C == 64 && W == 4))) {
        const dim3 grid((B + 3) / 4), blk(256);
#define CF_AFF2(CC, WW, SQV, BLV) k_affine_ctx_wave<CC, WW, SQV, BLV><<<grid, blk, 0, cf_s(stream)>>>(x, m1, Wm, m2, t, logs, lad, cadd, z, ldj, B, x_bstride, accumulate)
#define CF_AFF(CC, WW) do { if (in_squeeze) { if (m1_blocked) CF_AFF2(CC, WW, true, true); else CF_AFF2(CC, WW, true, false); } \
                            else { if (m1_blocked) CF_AFF2(CC, WW, false, true); else CF_AFF2(CC, WW, false, false); } } while (0)
        if (C == 16) CF_AFF(16, 16); else if (C == 32) CF_AFF(32, 8); else CF_AFF(64, 4);
#undef CF_AFF
#undef CF_AFF2
        CF_LAUNCH_CHECK();
        return 0;
    }
    CF_REQUIRE(!m1_blocked);                             // (the blocked form exists for the wave kernel's shapes and needs their alignment)
    const size_t lds = (size_t)(C * conv1x1_ctx_cp(C) + C * H * W + 4 + 2 * C + C * C) * sizeof(float);
    if (lds > 160 * 1024) { cf_set_error("cf_affine_ctx_fwd: C=%d, H*W=%d need %zu B of LDS", C, H * W, lds); return CF_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> raised0{0}, raised1{0};
        if (int rc_ = in_squeeze ? cf_raise_dynamic_lds((const void*)k_affine_ctx<true>, 160 * 1024, raised1, __func__)
                                 : cf_raise_dynamic_lds((const void*)k_affine_ctx<false>, 160 * 1024, raised0, __func__)) return rc_;
    }
    if (in_squeeze) k_affine_ctx<true><<<dim3(B), dim3(256), lds, cf_s(stream)>>>(x, m1, Wm, m2, t, logs, lad, cadd, z, ldj, C, H, W, x_bstride, accumulate);
    else k_affine_ctx<false><<<dim3(B), dim3(256), lds, cf_s(stream)>>>(x, m1, Wm, m2, t, logs, lad, cadd, z, ldj, C, H, W, x_bstride, accumulate);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_actnorm_ctx(const float* x, const float* m, const float* t, const float* logs, float* z, float* ldj, int B, int C,
                   int HW, int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && m && z && ldj && B >= 0 && C > 0 && HW > 0 && ((t == nullptr) == (logs == nullptr)));
    if (C > 128) { cf_set_error("cf_actnorm_ctx: C=%d > 128 unsupported", C); return CF_ERR_UNSUPPORTED; }
    k_actnorm_ctx<<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, m, t, logs, z, ldj, C, HW, x_bstride);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_add_sample_bias(float* h, const float* bias, int B, int C, int HW, int relu, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(h && bias && B >= 0 && C > 0 && HW > 0);
    const int64_t total = (int64_t)B * C * HW;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    k_add_sample_bias<<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(h, bias, HW, total, relu);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_ctx_logprob(const float* x, const float* mG, const float* sG, const float* logw, const float* c, float* out,
                       float* lp_out, int B, int M, int K, int D, int HW, int64_t x_bstride, int accumulate,
                       cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && mG && sG && logw && c && out && B >= 0 && M > 0 && K > 0 && D > 0 && HW > 0);
    int rc = gmm_ctx_fwd_launch<false>("cf_gmm_ctx_logprob", x, mG, sG, logw, c, out, lp_out, B, M, K, D, HW, x_bstride, accumulate,
                                       GmmTab{nullptr, nullptr, nullptr, nullptr, nullptr}, cf_s(stream));
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_ctx_tables(const float* sG, const float* cs_tab, float* inv_sig, float* dsig, float* lsum, int U, int MK, int D,
                      int HW, cf_stream_t stream) {
    if (U == 0) return 0;
    CF_REQUIRE(sG && cs_tab && inv_sig && lsum && U > 0 && MK > 0 && D > 0 && HW > 0);
    k_gmm_ctx_tables<<<dim3(U * MK), dim3(256), 0, cf_s(stream)>>>(sG, cs_tab, inv_sig, dsig, lsum, MK, D, HW);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_ctx_logprob_tab(const float* x, const float* mG, const float* inv_sig, const float* lsum, const float* logw,
                           const float* c, const int* ckey, const int* key, float* out, float* lp_out, int B, int M, int K,
                           int D, int HW, int64_t x_bstride, int accumulate, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && mG && inv_sig && lsum && logw && c && key && out && B >= 0 && M > 0 && K > 0 && D > 0 && HW > 0);
    int rc = gmm_ctx_fwd_launch<true>("cf_gmm_ctx_logprob_tab", x, mG, nullptr, logw, c, out, lp_out, B, M, K, D, HW, x_bstride,
                                      accumulate, GmmTab{inv_sig, nullptr, lsum, key, ckey}, cf_s(stream));
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_cond_gauss_sample(const float* c, const float* eps, float* x, float* logp, int B, int D, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(c && eps && x && logp && B >= 0 && D > 0);
    k_cond_gauss_sample<<<dim3(B), dim3(64), 0, cf_s(stream)>>>(c, eps, x, logp, D);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_sigmoid_ldj(const float* x, float* y, float* ldj, int B, int D, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && y && ldj && B >= 0 && D > 0);
    k_sigmoid_ldj<<<dim3(B), dim3(64), 0, cf_s(stream)>>>(x, y, ldj, D);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_conv1x1_ctx_bwd(const float* x, const float* m, const float* Wm, const float* gz, const float* gld, float* gx,
                       float* gm, int B, int C, int HW, int64_t x_bstride, int64_t gz_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && m && gz && gld && gx && gm && B >= 0 && C > 0 && HW > 0);
    const size_t lds = (size_t)(C * ((C + 7) & ~7) + 2 * C * (HW | 1)) * sizeof(float);
    if (lds > 160 * 1024) { cf_set_error("cf_conv1x1_ctx_bwd: C=%d, H*W=%d need %zu B of LDS", C, HW, lds); return CF_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_conv1x1_ctx_bwd, 160 * 1024, raised, __func__)) return rc_;
    }
    k_conv1x1_ctx_bwd<<<dim3(B), dim3(256), lds, cf_s(stream)>>>(x, m, Wm, gz, gld, gx, gm, C, HW, x_bstride, gz_bstride);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_actnorm_ctx_bwd(const float* x, const float* m, const float* t, const float* logs, const float* gz, const float* gld,
                       float* gx, float* gm, int B, int C, int HW, int64_t x_bstride, int64_t gz_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && m && gz && gld && gx && gm && B >= 0 && C > 0 && HW > 0 && ((t == nullptr) == (logs == nullptr)));
    k_actnorm_ctx_bwd<<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, m, t, logs, gz, gld, gx, gm, C, HW, x_bstride, gz_bstride);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_sample_channel_sums(const float* a, float* out, int B, int C, int HW, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(a && out && B >= 0 && C > 0 && HW > 0);
    const int64_t rows = (int64_t)B * C;
    k_sample_channel_sums<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, cf_s(stream)>>>(a, out, HW, rows);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_relu_bwd(const float* x, const float* gy, float* out, int64_t n, cf_stream_t stream) {
    if (n == 0) return 0;
    CF_REQUIRE(x && gy && out && n >= 0);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    k_relu_bwd<<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, gy, out, n);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_ctx_bwd(const float* x, const float* mG, const float* sG, const float* logw, const float* c, const float* g,
                   const float* lp, float* gx, float* gc, int B, int M, int K, int D, int HW, int64_t x_bstride,
                   cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && mG && sG && logw && c && g && gx && gc && B >= 0 && M > 0 && K > 0 && D > 0 && HW > 0);
    int rc = gmm_ctx_bwd_launch<false>("cf_gmm_ctx_bwd", x, mG, sG, logw, c, g, lp, gx, gc, B, M, K, D, HW, x_bstride,
                                       GmmTab{nullptr, nullptr, nullptr, nullptr, nullptr}, cf_s(stream));
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_ctx_bwd_tab(const float* x, const float* mG, const float* inv_sig, const float* dsig, const float* lsum,
                       const float* logw, const float* c, const int* key, const float* g, const float* lp, float* gx,
                       float* gc, int B, int M, int K, int D, int HW, int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && mG && inv_sig && dsig && lsum && logw && c && key && g && gx && gc && B >= 0 && M > 0 && K > 0 && D > 0 && HW > 0);
    int rc = gmm_ctx_bwd_launch<true>("cf_gmm_ctx_bwd_tab", x, mG, nullptr, logw, c, g, lp, gx, gc, B, M, K, D, HW, x_bstride,
                                      GmmTab{inv_sig, dsig, lsum, key, nullptr}, cf_s(stream));
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_ctx_pgrad_tab(const float* x, const float* mG, const float* inv_sig, const float* dsig, const float* c,
                          const int* key, const float* r, float* pgm, float* pgs, int B, int M, int K, int D, int HW,
                          int64_t x_bstride, int slab, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && mG && inv_sig && dsig && c && key && r && pgm && pgs && B >= 0 && M > 0 && K > 0 && D > 0 && HW > 0 && slab > 0);
    const int N = D * HW, nb = (B + slab - 1) / slab;
    CF_REQUIRE(M * K <= 65535 && nb <= 65535);
    k_gmm_ctx_pgrad<<<dim3((N + 255) / 256, M * K, nb), dim3(256), 0, cf_s(stream)>>>(
        x, mG, c, r, pgm, pgs, B, M * K, D, HW, x_bstride, slab, GmmTab{inv_sig, dsig, nullptr, key, nullptr});
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_add_repeat(float* h, const float* x, int B, int C2, int C, int HW, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(h && x && B >= 0 && C2 > 0 && C > 0 && HW > 0);
    const int64_t total = (int64_t)B * C2 * HW;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    k_add_repeat<<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(h, x, C2, C, HW, total);
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
