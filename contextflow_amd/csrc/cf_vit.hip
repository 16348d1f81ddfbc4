// SimpleViT conditioner of TransCoupling (contextflow/layers/simple_vit.py:18-127) on gfx950.
//
// The dense contractions (patch embedding, qkv, out-projection, MLP) run on the exact-fp32 matrix
// cores (v_mfma_f32_32x32x2_f32): rows = samples x tokens, K, N <= a few hundred.  One workgroup
// owns 128 rows x (<= 96) output features; X and the W slice go through LDS in K-chunks of 32 with
// an odd row stride, so both MFMA operand reads (lane -> row, lane>>5 -> k) are bank-conflict free.
// LayerNorm, the tiny (tokens x tokens) attention and the patch index maps are VALU kernels.
#include "cf_common.h"
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LIN_ROWS = 128;     // rows per workgroup (4 waves x 32)
constexpr int LIN_COLS = 96;      // output features per workgroup (3 MFMA column tiles)
constexpr int LIN_KC = 32;        // K chunk staged in LDS per pass
constexpr int LIN_KP = LIN_KC + 1;   // odd row stride: conflict-free operand reads (lane -> row)

// y[r, n] = act(sum_k x[r,k] W[n,k] + bias[n]) + res[r,n]
// ACT: 0 none, 1 exact GELU (erf), 2 ReLU             simple_vit.py:32-38,52-53; coupling.py:37 (CN nets)
// K is walked in chunks of 32 through a 29 KiB LDS stage (any K; 4-5 workgroups per CU overlap each other's staging
// and MFMA phases); the next chunk's global loads are issued into registers before the MFMAs of the current one.
template <int ACT>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ x, const float* __restrict__ Wt,
                                                const float* __restrict__ bias, const float* __restrict__ res,
                                                float* __restrict__ y, int rows, int K, int N) {
    __shared__ float xs[LIN_ROWS * LIN_KP];
    __shared__ float ws[LIN_COLS * LIN_KP];
    constexpr int NXI = LIN_ROWS * LIN_KC / 256, NWI = LIN_COLS * LIN_KC / 256;   // staged elements per thread: 16 + 12
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * LIN_ROWS, n0 = blockIdx.y * LIN_COLS;
    const int sk = tid & 31, sr = tid >> 5;              // staging: column k of the chunk, first row (rows sr + 8 i)
    float xr[NXI], wr[NWI];
    auto fetch = [&](int k0) {
        const int k = k0 + sk;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int r = r0 + sr + 8 * i;
            xr[i] = (r < rows && k < K) ? x[(int64_t)r * K + k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            const int n = n0 + sr + 8 * i;
            wr[i] = (n < N && k < K) ? Wt[(int64_t)n * K + k] : 0.f;
        }
    };
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    const float* xa = xs + (wave * 32 + li) * LIN_KP + lk;     // A[i = row][k]
    const float* wb = ws + li * LIN_KP + lk;                    // B[k][j = feature]
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += LIN_KC) {
#pragma unroll
        for (int i = 0; i < NXI; ++i) xs[(sr + 8 * i) * LIN_KP + sk] = xr[i];
#pragma unroll
        for (int i = 0; i < NWI; ++i) ws[(sr + 8 * i) * LIN_KP + sk] = wr[i];
        __syncthreads();
        if (k0 + LIN_KC < K) fetch(k0 + LIN_KC);
#pragma unroll
        for (int kk = 0; kk < LIN_KC; kk += 2) {
            const float a = xa[kk];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float b = wb[t * 32 * LIN_KP + kk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D[i][j]: lane holds column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = n0 + t * 32 + li;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row >= rows) continue;
            float v = acc[t][r] + bv;
            if (ACT == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
            if (ACT == 2) v = fmaxf(v, 0.f);
            if (res) v += res[(int64_t)row * N + n];
            y[(int64_t)row * N + n] = v;
        }
    }
}

// LayerNorm over the last dim (biased variance, eps) + optional positional embedding add:
// y[r, :] = LN(x[r, :]) * w + b (+ pe[r % ntok, :]).  16 lanes per row.   simple_vit.py:33,50,74,104-106,122
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ b, const float* __restrict__ pe,
                                                   float* __restrict__ y, int rows, int dim, int ntok, float eps) {
    const int g = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = row < rows;
    const float* xr = x + (ok ? row : 0) * dim;
    float s = 0.f;
    for (int j = g; j < dim; j += 16) s += xr[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)dim;
    float v = 0.f;
    for (int j = g; j < dim; j += 16) { const float d = xr[j] - mean; v = fmaf(d, d, v); }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float rstd = 1.0f / sqrtf(v / (float)dim + eps);
    if (!ok) return;
    const float* per = pe ? pe + (row % ntok) * dim : nullptr;
    for (int j = g; j < dim; j += 16) {
        float o = (xr[j] - mean) * rstd * w[j] + b[j];
        if (per) o += per[j];
        y[row * dim + j] = o;
    }
}

// single-head attention on N tokens per sample: out = softmax(q k^T * scale) v.   simple_vit.py:56-68
// qkv rows are [q | k | v] of width 3*dh; one workgroup per sample.  LDS rows have the odd stride 3 dh + 1: the q k^T
// products walk the k rows with one lane per row.
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, float* __restrict__ out, int N, int dh,
                                                   float scale) {
    extern __shared__ __align__(16) float lds[];
    const int RS = 3 * dh + 1, nt = blockDim.x;
    float* s_qkv = lds;                   // [N][RS]
    float* dots = lds + N * RS;           // [N][N]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* src = qkv + (int64_t)b * N * 3 * dh;
    for (int e = tid; e < N * 3 * dh; e += nt) s_qkv[(e / (3 * dh)) * RS + e % (3 * dh)] = src[e];
    __syncthreads();
    for (int e = tid; e < N * N; e += nt) {
        const int i = e / N, j = e - i * N;
        const float* q = s_qkv + i * RS;
        const float* k = s_qkv + j * RS + dh;
        float a0 = 0.f, a1 = 0.f;
        for (int d = 0; d < dh; d += 2) { a0 = fmaf(q[d], k[d], a0); a1 = fmaf(q[d + 1], k[d + 1], a1); }
        dots[e] = (a0 + a1) * scale;
    }
    __syncthreads();
    for (int i = tid; i < N; i += nt) {
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, dots[i * N + j]);
        float sum = 0.f;
        for (int j = 0; j < N; ++j) { const float e = expf(dots[i * N + j] - mx); dots[i * N + j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < N; ++j) dots[i * N + j] *= inv;
    }
    __syncthreads();
    for (int e = tid; e < N * dh; e += nt) {
        const int i = e / dh, d = e - i * dh;
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc = fmaf(dots[i * N + j], s_qkv[j * RS + 2 * dh + d], acc);
        out[((int64_t)b * N + i) * dh + d] = acc;
    }
}

// INV=false: tok[b, h*gw+w, (i1*p2+i2)*C + c] = x[b, c, h*p1+i1, w*p2+i2]     simple_vit.py:101
// INV=true : x[b, c, h*p1+i1, w*p2+i2] = tok[...]                              simple_vit.py:115
template <bool INV>
__global__ __launch_bounds__(256) void k_patch(const float* __restrict__ src, float* __restrict__ dst, int C, int H,
                                               int W, int p1, int p2, int64_t img_bs, int64_t total) {
    const int gw = W / p2;
    const int64_t per = (int64_t)C * H * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / per;
        int r = (int)(e - b * per);                          // token-major index inside the sample
        const int c = r % C; r /= C;
        const int i2 = r % p2; r /= p2;
        const int i1 = r % p1; r /= p1;
        const int w = r % gw; const int h = r / gw;
        const int64_t img = b * img_bs + ((int64_t)c * H + (h * p1 + i1)) * W + (w * p2 + i2);
        if (!INV) dst[e] = src[img];
        else dst[img] = src[e];
    }
}

}  // namespace

extern "C" {

int cf_linear(const float* x, const float* Wt, const float* bias, const float* res, float* y, int rows, int K, int N,
              int act, cf_stream_t stream) {
    if (rows == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && Wt && y && rows >= 0 && K > 0 && N > 0 && act >= 0 && act <= 2);
    dim3 grid((rows + LIN_ROWS - 1) / LIN_ROWS, (N + LIN_COLS - 1) / LIN_COLS);
    if (act == 0) k_linear<0><<<grid, dim3(256), 0, cf_s(stream)>>>(x, Wt, bias, res, y, rows, K, N);
    else if (act == 1) k_linear<1><<<grid, dim3(256), 0, cf_s(stream)>>>(x, Wt, bias, res, y, rows, K, N);
    else k_linear<2><<<grid, dim3(256), 0, cf_s(stream)>>>(x, Wt, bias, res, y, rows, K, N);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_layernorm(const float* x, const float* w, const float* b, const float* pos, float* y, int rows, int dim,
                 int ntok, float eps, cf_stream_t stream) {
    if (rows == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && w && b && y && rows >= 0 && dim > 0 && (pos == nullptr || ntok > 0));
    if (rows == 0) return 0;
    k_layernorm<<<dim3((rows + 15) / 16), dim3(256), 0, cf_s(stream)>>>(x, w, b, pos, y, rows, dim, ntok > 0 ? ntok : 1, eps);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_attention(const float* qkv, float* out, int B, int N, int dh, float scale, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(qkv && out && B >= 0 && N > 0 && dh > 0);
    CF_REQUIRE(dh % 2 == 0);
    const size_t lds = (size_t)(N * (3 * dh + 1) + N * N) * sizeof(float);
    if (lds > 64 * 1024) { cf_set_error("cf_attention: N=%d dh=%d needs %zu B of LDS", N, dh, lds); return CF_ERR_UNSUPPORTED; }
    k_attention<<<dim3(B), dim3(N >= 16 ? 256 : 64), lds, cf_s(stream)>>>(qkv, out, N, dh, scale);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_patchify(const float* src, float* dst, int B, int C, int H, int W, int p1, int p2, int64_t img_bstride,
                int inverse, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(src && dst && B >= 0 && C > 0 && H > 0 && W > 0 && p1 > 0 && p2 > 0 && H % p1 == 0 && W % p2 == 0);
    const int64_t total = (int64_t)B * C * H * W;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (inverse) k_patch<true><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(src, dst, C, H, W, p1, p2, img_bstride, total);
    else k_patch<false><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(src, dst, C, H, W, p1, p2, img_bstride, total);
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
