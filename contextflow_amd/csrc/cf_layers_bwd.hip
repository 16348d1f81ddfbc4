// Backward kernels of the layer-by-layer path (TransCoupling's SimpleViT conditioner, generic ActNorm / affine map):
// what the training step (experiment_ad.py:207-213: loss.backward()) needs besides the fused conv-flow backward.
//
// All of them are small HBM- / latency-bound VALU kernels over token-major (rows, dim) or NCHW tensors; the dense
// contractions of the Linear layers' backward run on the repo's own MFMA kernels (gX = gY W: cf_linear with the
// transposed weight; gW = gY^T X: cf_linear_wgrad, cf_vit.hip - no library GEMM anywhere in the product:
// tests/test_host.py::test_no_library_gemm_in_the_product).  Parameter gradients that are reductions over rows are
// produced as per-workgroup partial sums in a fixed order (no float atomics: reproducible); the host sums the few
// hundred partial rows.
#include "cf_common.h"
#include <math.h>

namespace {

constexpr int kLnBlocks = 1024;     // partial-sum rows of cf_layernorm_bwd (4 workgroups per CU)
constexpr int kLnMaxPer = 16;       // features per lane: dim <= 256 (ATM: transformer width 152)
constexpr int kLnMaxDim = 16 * kLnMaxPer;

// LayerNorm backward (biased variance, eps; forward: simple_vit.py:33,50,74,104-106).  16 lanes per row, 16 rows per
// pass, grid-stride over row groups; a row's x and gy are read ONCE into registers (16 values per lane: dim <= 256).
// gx = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat));
// partial[blk][j] = sum_rows gy*xhat, partial[blk][dim + j] = sum_rows gy.
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ gy, float* __restrict__ gx,
                                                       float* __restrict__ part, int rows, int dim, float eps) {
    __shared__ float red[16][2 * kLnMaxDim];
    const int g = threadIdx.x & 15, rg = threadIdx.x >> 4;
    float aw[kLnMaxPer], ab[kLnMaxPer], wv[kLnMaxPer];
#pragma unroll
    for (int i = 0; i < kLnMaxPer; ++i) { aw[i] = 0.f; ab[i] = 0.f; wv[i] = (g + 16 * i < dim) ? w[g + 16 * i] : 0.f; }
    const float inv = 1.0f / (float)dim;
    for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < rows; r0 += (int64_t)gridDim.x * 16) {
        const int64_t row = r0 + rg;
        const bool ok = row < rows;
        const float* xr = x + (ok ? row : 0) * dim;
        const float* gr = gy + (ok ? row : 0) * dim;
        float xv[kLnMaxPer], gv[kLnMaxPer];
#pragma unroll
        for (int i = 0; i < kLnMaxPer; ++i) {
            const int j = g + 16 * i;
            const bool in = j < dim;
            xv[i] = in ? xr[j] : 0.f;
            gv[i] = (in && ok) ? gr[j] : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < kLnMaxPer; ++i) s += xv[i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * inv;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < kLnMaxPer; ++i) {
            const float d = (g + 16 * i < dim) ? xv[i] - mean : 0.f;
            xv[i] = d;
            v = fmaf(d, d, v);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const float rstd = 1.0f / sqrtf(v * inv + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < kLnMaxPer; ++i) {
            xv[i] *= rstd;                                               // xhat
            const float gw = gv[i] * wv[i];
            s1 += gw;
            s2 = fmaf(gw, xv[i], s2);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        s1 *= inv; s2 *= inv;
#pragma unroll
        for (int i = 0; i < kLnMaxPer; ++i) {
            const int j = g + 16 * i;
            if (ok && j < dim) gx[row * dim + j] = rstd * (gv[i] * wv[i] - s1 - xv[i] * s2);
            aw[i] = fmaf(gv[i], xv[i], aw[i]);                           // gv = 0 for rows past the end
            ab[i] += gv[i];
        }
    }
#pragma unroll
    for (int i = 0; i < kLnMaxPer; ++i) {
        const int j = g + 16 * i;
        if (j < dim) { red[rg][j] = aw[i]; red[rg][kLnMaxDim + j] = ab[i]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * dim; e += 256) {
        const int j = e < dim ? e : kLnMaxDim + (e - dim);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][j];
        part[(int64_t)blockIdx.x * 2 * dim + e] = t;
    }
}

// single-head attention backward on N tokens per sample (forward: k_attention, simple_vit.py:56-68).
// qkv rows [q | k | v], go = d/d out; gqkv rows [dq | dk | dv].  One workgroup per sample; the softmax is
// recomputed.  dS = P o (dP - rowsum(P o dP)), dP = go V^T, dV = P^T go, dQ = scale dS K, dK = scale dS^T Q.
// LDS rows have odd strides (3 dh + 1, dh + 1, N + 1): the products walk rows / columns with one lane each.
__global__ __launch_bounds__(256) void k_attention_bwd(const float* __restrict__ qkv, const float* __restrict__ go,
                                                       float* __restrict__ gqkv, int N, int dh, float scale) {
    extern __shared__ __align__(16) float lds[];
    const int RS = 3 * dh + 1, GS = dh + 1, NS = N | 1, nt = blockDim.x;
    float* s_qkv = lds;                     // [N][RS]
    float* s_go = s_qkv + N * RS;           // [N][GS]
    float* P = s_go + N * GS;               // [N][NS]
    float* dS = P + N * NS;                 // [N][NS]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* src = qkv + (int64_t)b * N * 3 * dh;
    const float* gsrc = go + (int64_t)b * N * dh;
    // staging in batches of 8 independent 16-byte loads per thread (dh is a multiple of 4, rows are 16-byte aligned)
    auto stage = [&](const float* g, float* l, int width, int stride) {
        const int q4 = width / 4, total = N * q4;
        const float4* g4 = reinterpret_cast<const float4*>(g);
        for (int e0 = tid; e0 < total; e0 += 8 * nt) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = g4[min(e0 + i * nt, total - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * nt;
                if (e < total) {
                    const int r = e / q4, c = 4 * (e - r * q4);
                    float* d = l + r * stride + c;
                    d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
                }
            }
        }
    };
    stage(src, s_qkv, 3 * dh, RS);
    stage(gsrc, s_go, dh, GS);
    __syncthreads();
    // P = q k^T and dP = go v^T: a thread owns key j and 2 query rows - one k / v read feeds 2 FMAs each
    const int NQ = (N + 1) >> 1;
    for (int e = tid; e < NQ * N; e += nt) {
        const int iq = e / N, j = e - iq * N, i0 = iq * 2, i1 = min(i0 + 1, N - 1);
        const float* k = s_qkv + j * RS + dh;
        const float* v = s_qkv + j * RS + 2 * dh;
        const float* q0 = s_qkv + i0 * RS;
        const float* q1 = s_qkv + i1 * RS;
        const float* g0 = s_go + i0 * GS;
        const float* g1 = s_go + i1 * GS;
        float s0 = 0.f, s1 = 0.f, d0 = 0.f, d1 = 0.f;
        for (int c = 0; c < dh; ++c) {
            const float kv = k[c], vv = v[c];
            s0 = fmaf(q0[c], kv, s0); s1 = fmaf(q1[c], kv, s1);
            d0 = fmaf(g0[c], vv, d0); d1 = fmaf(g1[c], vv, d1);
        }
        P[i0 * NS + j] = s0 * scale;
        dS[i0 * NS + j] = d0;               // dP for now
        if (i0 + 1 < N) { P[i1 * NS + j] = s1 * scale; dS[i1 * NS + j] = d1; }
    }
    __syncthreads();
    for (int i = tid; i < N; i += nt) {     // row softmax, then dS = P o (dP - sum_j P dP)
        float* Pi = P + i * NS;
        float* Di = dS + i * NS;
        float m = -INFINITY;
        for (int j = 0; j < N; ++j) m = fmaxf(m, Pi[j]);
        float z = 0.f;
        for (int j = 0; j < N; ++j) { const float ev = expf(Pi[j] - m); Pi[j] = ev; z += ev; }
        const float rz = 1.0f / z;
        float dot = 0.f;
        for (int j = 0; j < N; ++j) { Pi[j] *= rz; dot = fmaf(Pi[j], Di[j], dot); }
        for (int j = 0; j < N; ++j) Di[j] = Pi[j] * (Di[j] - dot);
    }
    __syncthreads();
    float* dst = gqkv + (int64_t)b * N * 3 * dh;
    for (int e = tid; e < N * dh; e += nt) {
        const int i = e / dh, c = e - i * dh;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < N; ++j) {
            dq = fmaf(dS[i * NS + j], s_qkv[j * RS + dh + c], dq);          // dS[i][j] K[j][c]
            dk = fmaf(dS[j * NS + i], s_qkv[j * RS + c], dk);               // dS[j][i] Q[j][c]
            dv = fmaf(P[j * NS + i], s_go[j * GS + c], dv);                 // P[j][i] go[j][c]
        }
        dst[i * 3 * dh + c] = dq * scale;
        dst[i * 3 * dh + dh + c] = dk * scale;
        dst[i * 3 * dh + 2 * dh + c] = dv;
    }
}

// exact (erf) GELU, nn.GELU() default (simple_vit.py:36): y = x Phi(x); dy/dx = Phi(x) + x phi(x)
template <bool BWD>
__global__ __launch_bounds__(256) void k_gelu(const float* __restrict__ x, const float* __restrict__ gy,
                                              float* __restrict__ out, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float v = x[e];
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
        if (BWD) out[e] = gy[e] * (cdf + v * 0.39894228040143267794f * expf(-0.5f * v * v));
        else out[e] = v * cdf;
    }
}

// Backward of the affine coupling map (coupling.py:52-66): z = [x0 | x1 s + t], ldj = sum log_s,
// log_s = 2 tanh(raw / 2), h = [t | raw].  gx = [gz0 | gz1 s]  (the conditioner's own gradient w.r.t. x0 is added by
// the caller), gh = [gz1 | (gz1 x1 s + gld_b) (1 - (log_s/2)^2)].
__global__ __launch_bounds__(256) void k_coupling_apply_bwd(const float* __restrict__ x, const float* __restrict__ h,
                                                            const float* __restrict__ gz, const float* __restrict__ gld,
                                                            float* __restrict__ gx, float* __restrict__ gh, int C,
                                                            int HW, int64_t total_half, int64_t xbs, int64_t gzbs) {
    const int half = C / 2;
    const int64_t per = (int64_t)half * HW;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total_half; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / per, r = e - b * per;
        const float x1 = x[b * xbs + per + r];
        const float g0 = gz[b * gzbs + r], g1 = gz[b * gzbs + per + r];
        const float raw = h[b * 2 * per + per + r];
        const float th = tanhf(0.5f * raw), ls = 2.0f * th, s = expf(ls);
        gx[b * 2 * per + r] = g0;
        gx[b * 2 * per + per + r] = g1 * s;
        gh[b * 2 * per + r] = g1;
        gh[b * 2 * per + per + r] = (g1 * x1 * s + gld[b]) * (1.0f - th * th);
    }
}

// out[c] = sum_{b,p} a[b][c][p], out[C + c] = sum_{b,p} a * b2   (per-channel reductions of the ActNorm backward).
// Two passes with a fixed summation order: (channel, batch slice) workgroups write partial sums, one workgroup adds the
// slices of each channel in order.  (Round 1 ran ONE workgroup per channel: 26 workgroups on 256 CUs, 600 us per call on
// the SMAP training step.)
constexpr int kChanSlices = 128;
__global__ __launch_bounds__(256) void k_channel_sums(const float* __restrict__ a, const float* __restrict__ b2,
                                                      float* __restrict__ part, int B, int C, int HW, int64_t abs_,
                                                      int64_t bbs) {
    const int c = blockIdx.x, sl = blockIdx.y, nsl = gridDim.y;
    const int b_lo = (int)((int64_t)B * sl / nsl), b_hi = (int)((int64_t)B * (sl + 1) / nsl);
    const int64_t n = (int64_t)(b_hi - b_lo) * HW;
    float s0 = 0.f, s1 = 0.f;
    for (int64_t e = threadIdx.x; e < n; e += 256) {
        const int64_t b = b_lo + e / HW;
        const int p = (int)(e % HW);
        const float av = a[b * abs_ + (int64_t)c * HW + p];
        s0 += av;
        if (b2) s1 = fmaf(av, b2[b * bbs + (int64_t)c * HW + p], s1);
    }
    __shared__ float scr[4];
    s0 = cf_block_sum<4>(s0, scr);
    s1 = cf_block_sum<4>(s1, scr);
    if (threadIdx.x == 0) { part[((int64_t)sl * C + c) * 2] = s0; part[((int64_t)sl * C + c) * 2 + 1] = s1; }
}
__global__ __launch_bounds__(256) void k_channel_sums_finish(const float* __restrict__ part, float* __restrict__ out, int C, int nsl) {
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        const int c = e % C, j = e / C;
        float s = 0.f;
        for (int sl = 0; sl < nsl; ++sl) s += part[((int64_t)sl * C + c) * 2 + j];
        out[e] = s;
    }
}

// Parameter gradients of the Conv1x1 + ActNorm pair of a fused flow step from the gradient of the folded matrix /
// bias the step-backward produced (W' = diag(s) Wm, b' = -t s, s = exp(-logs); autograd.py step_backward):
//   gNN[o][i]  = s[o] gWp[o][i] + G H W Winv[i][o]          (+ d(H W log|det Wm|)/dWm, G = sum_b gld[b])
//   gt[o]      = -s[o] gbp[o]
//   glogs[o]   = -sum_i gWp[o][i] s[o] Wm[o][i] + gbp[o] t[o] s[o] + G      (reference quirk: ldj = +sum logs)
// One workgroup (C <= 128): replaces ~15 parameter-sized torch kernels per flow step of a training step.
// blockIdx.x = flow step of a batch (cf_step_param_grads_batch)
constexpr int kParamBatch = 8;
struct StepParamBatch {
    const float *gWp[kParamBatch], *gbp[kParamBatch], *Wm[kParamBatch], *t[kParamBatch], *logs[kParamBatch], *winv[kParamBatch];
    float *gNN[kParamBatch], *gt[kParamBatch], *glogs[kParamBatch];
};
__global__ __launch_bounds__(256) void k_step_param_grads(const StepParamBatch pb, const float* __restrict__ Gsum, float hw, int C) {
    const int bi = blockIdx.x;
    const float* __restrict__ gWp = pb.gWp[bi]; const float* __restrict__ gbp = pb.gbp[bi]; const float* __restrict__ Wm = pb.Wm[bi];
    const float* __restrict__ t = pb.t[bi]; const float* __restrict__ logs = pb.logs[bi]; const float* __restrict__ winv = pb.winv[bi];
    float* __restrict__ gNN = pb.gNN[bi]; float* __restrict__ gt = pb.gt[bi]; float* __restrict__ glogs = pb.glogs[bi];
    const float G = Gsum[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < C * C; e += 256) {
        const int o = e / C, i = e - o * C;
        gNN[e] = expf(-logs[o]) * gWp[e] + G * hw * winv[i * C + o];
    }
    for (int o = wave; o < C; o += 4) {                   // a wave per output channel
        const float so = expf(-logs[o]);
        float acc = 0.f;
        for (int i = lane; i < C; i += 64) acc = fmaf(gWp[o * C + i], Wm[o * C + i], acc);
        acc = cf_wave_sum(acc);
        if (lane == 0) {
            gt[o] = -so * gbp[o];
            glogs[o] = -so * acc + gbp[o] * t[o] * so + G;
        }
    }
}

}  // namespace

extern "C" {

int cf_layernorm_bwd_parts(void) { return kLnBlocks; }

int cf_layernorm_bwd(const float* x, const float* w, const float* gy, float* gx, float* partial, int rows, int dim,
                     float eps, cf_stream_t stream) {
    CF_REQUIRE(x && w && gy && gx && partial && rows >= 0 && dim > 0);
    if (dim > 16 * kLnMaxPer) { cf_set_error("cf_layernorm_bwd: dim=%d > %d unsupported", dim, 16 * kLnMaxPer); return CF_ERR_UNSUPPORTED; }
    k_layernorm_bwd<<<dim3(kLnBlocks), dim3(256), 0, cf_s(stream)>>>(x, w, gy, gx, partial, rows, dim, eps);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_attention_bwd(const float* qkv, const float* go, float* gqkv, int B, int N, int dh, float scale, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(qkv && go && gqkv && B >= 0 && N > 0 && dh > 0 && dh % 4 == 0 &&
               ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(go)) & 15) == 0);
    const size_t lds = (size_t)(N * (3 * dh + 1) + N * (dh + 1) + 2 * N * (N | 1)) * sizeof(float);
    if (lds > 64 * 1024) { cf_set_error("cf_attention_bwd: N=%d dh=%d needs %zu B of LDS", N, dh, lds); return CF_ERR_UNSUPPORTED; }
    k_attention_bwd<<<dim3(B), dim3(N >= 16 ? 256 : 64), lds, cf_s(stream)>>>(qkv, go, gqkv, N, dh, scale);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gelu(const float* x, const float* gy, float* out, int64_t n, int backward, cf_stream_t stream) {
    if (n == 0) return 0;
    CF_REQUIRE(x && out && n >= 0 && (!backward || gy));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (backward) k_gelu<true><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, gy, out, n);
    else k_gelu<false><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, nullptr, out, n);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_coupling_apply_bwd(const float* x, const float* h, const float* gz, const float* gld, float* gx, float* gh, int B,
                          int C, int HW, int64_t x_bstride, int64_t gz_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && h && gz && gld && gx && gh && B >= 0 && C > 0 && C % 2 == 0 && HW > 0);
    const int64_t total = (int64_t)B * (C / 2) * HW;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    k_coupling_apply_bwd<<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, h, gz, gld, gx, gh, C, HW, total,
                                                                                x_bstride, gz_bstride);
    CF_LAUNCH_CHECK();
    return 0;
}

static int channel_slices(int B) { return B < kChanSlices ? (B > 0 ? B : 1) : kChanSlices; }

int64_t cf_channel_sums_ws_bytes(int B, int C) { return (int64_t)channel_slices(B) * C * 2 * 4; }

int cf_channel_sums(const float* a, const float* b2, float* out, void* ws, int B, int C, int HW, int64_t a_bstride,
                    int64_t b_bstride, cf_stream_t stream) {
    CF_REQUIRE(a && out && ws && B >= 0 && C > 0 && HW > 0);
    const int nsl = channel_slices(B);
    k_channel_sums<<<dim3(C, nsl), dim3(256), 0, cf_s(stream)>>>(a, b2, (float*)ws, B, C, HW, a_bstride, b_bstride);
    k_channel_sums_finish<<<dim3(1), dim3(256), 0, cf_s(stream)>>>((const float*)ws, out, C, nsl);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_step_param_grads_batch(int n, const float* const* gWp, const float* const* gbp, const float* const* Wm, const float* const* t,
                              const float* const* logs, const float* const* winv, const float* gld_sum, int HW, float* const* gNN,
                              float* const* gt, float* const* glogs, int C, cf_stream_t stream);

int cf_step_param_grads(const float* gWp, const float* gbp, const float* Wm, const float* t, const float* logs,
                        const float* winv, const float* gld_sum, int HW, float* gNN, float* gt, float* glogs, int C,
                        cf_stream_t stream) {
    CF_REQUIRE(gWp && gbp && Wm && t && logs && winv && gld_sum && gNN && gt && glogs && C > 0 && HW > 0);
    return cf_step_param_grads_batch(1, &gWp, &gbp, &Wm, &t, &logs, &winv, gld_sum, HW, &gNN, &gt, &glogs, C, stream);
}

// the parameter chains of n flow steps of one width in one launch (workgroup = step); gld_sum is shared by the steps
int cf_step_param_grads_batch(int n, const float* const* gWp, const float* const* gbp, const float* const* Wm, const float* const* t,
                              const float* const* logs, const float* const* winv, const float* gld_sum, int HW, float* const* gNN,
                              float* const* gt, float* const* glogs, int C, cf_stream_t stream) {
    CF_REQUIRE(n >= 0 && gWp && gbp && Wm && t && logs && winv && gld_sum && gNN && gt && glogs && C > 0 && HW > 0);
    for (int i0 = 0; i0 < n; i0 += kParamBatch) {
        const int m = n - i0 < kParamBatch ? n - i0 : kParamBatch;
        StepParamBatch pb{};
        for (int i = 0; i < m; ++i) {
            const int j = i0 + i;
            CF_REQUIRE(gWp[j] && gbp[j] && Wm[j] && t[j] && logs[j] && winv[j] && gNN[j] && gt[j] && glogs[j]);
            pb.gWp[i] = gWp[j]; pb.gbp[i] = gbp[j]; pb.Wm[i] = Wm[j]; pb.t[i] = t[j]; pb.logs[i] = logs[j]; pb.winv[i] = winv[j];
            pb.gNN[i] = gNN[j]; pb.gt[i] = gt[j]; pb.glogs[i] = glogs[j];
        }
        k_step_param_grads<<<dim3(m), dim3(256), 0, cf_s(stream)>>>(pb, gld_sum, (float)HW, C);
        CF_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
