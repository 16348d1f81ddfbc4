// HBM-bound elementwise / per-sample-reduction kernels of the flow path (gfx950).
//
// Every kernel streams fp32 with 16-byte accesses per lane when shapes allow (VEC=4) and reduces
// per-sample log-det terms with 64-lane wave shuffles + one LDS hop per block: one block owns one
// sample, so log-dets are written with plain stores — no float atomics, bitwise reproducible.
#include "cf_common.h"
#include <math.h>

namespace {

constexpr float kLog2Pi = 1.8378770664093453f;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int V> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<4> { using T = float4; };

template <int V> __device__ __forceinline__ void vload(const float* p, float (&r)[V]) {
    if constexpr (V == 4) { float4 t = *reinterpret_cast<const float4*>(p); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
    else r[0] = *p;
}
template <int V> __device__ __forceinline__ void vstore(float* p, const float (&r)[V]) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    else *p = r[0];
}

// ---------------------------------------------------------------------------------------------
// flat elementwise ops
// ---------------------------------------------------------------------------------------------
enum { OP_ADD2 = 0, OP_AFFINE_FWD, OP_AFFINE_INV, OP_SIGMOID, OP_FLOOR };

template <int OP, int V>
__global__ __launch_bounds__(256) void k_flat(const float* __restrict__ x, const float* __restrict__ x2,
                                              float* __restrict__ y, int64_t n_items, float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
        float r[V], s[V];
        vload<V>(x + i * V, r);
        if constexpr (OP == OP_ADD2) vload<V>(x2 + i * V, s);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if constexpr (OP == OP_ADD2) r[j] = r[j] + s[j];
            else if constexpr (OP == OP_AFFINE_FWD) r[j] = r[j] / b + a;        // normalize.py:32
            else if constexpr (OP == OP_AFFINE_INV) r[j] = (r[j] - a) * b;      // normalize.py:40
            else if constexpr (OP == OP_SIGMOID) r[j] = 1.0f / (1.0f + expf(-r[j]));
            else if constexpr (OP == OP_FLOOR) r[j] = floorf(r[j]);
        }
        vstore<V>(y + i * V, r);
    }
}

template <int OP>
int launch_flat(const float* x, const float* x2, float* y, int64_t n, float a, float b, hipStream_t s) {
    if (n == 0) return 0;
    const bool vec = (n % 4 == 0) && aligned16(x) && aligned16(y) && (x2 == nullptr || aligned16(x2));
    const int64_t items = vec ? n / 4 : n;
    int64_t blocks = (items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (vec) k_flat<OP, 4><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, x2, y, items, a, b);
    else k_flat<OP, 1><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, x2, y, items, a, b);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// per-sample kernels: one block per sample
// ---------------------------------------------------------------------------------------------
// MODE 0: logit          y = log x - log(1-x), ldj = sum(-log x - log(1-x))
// MODE 1: preprocess     v = ((x+u)/s1+t1)/s2+t2 then logit(v); ldj = c0 + sum(...)
// MODE 2: std-normal nll out = 0.5*sum x^2 + 0.5*N*log(2pi)   (no y)
template <int MODE, int V, int NT>
__global__ __launch_bounds__(NT) void k_sample(const float* __restrict__ x, const float* __restrict__ u,
                                               float* __restrict__ y, float* __restrict__ ldj, int n_items,
                                               int64_t x_bstride, int64_t y_bstride,
                                               float t1, float s1, float t2, float s2, float c0) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    const float* xb = x + (int64_t)b * x_bstride;
    const float* ub = (MODE == 1) ? u + (int64_t)b * x_bstride : nullptr;
    float* yb = (MODE == 2) ? nullptr : y + (int64_t)b * y_bstride;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_items; i += NT) {
        float r[V], s[V];
        vload<V>(xb + (int64_t)i * V, r);
        if constexpr (MODE == 1) vload<V>(ub + (int64_t)i * V, s);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if constexpr (MODE == 2) {
                acc = fmaf(r[j], r[j], acc);
            } else {
                float v = r[j];
                if constexpr (MODE == 1) v = ((v + s[j]) / s1 + t1) / s2 + t2;
                const float l0 = logf(v), l1 = logf(1.0f - v);
                r[j] = l0 - l1;
                acc += -l0 - l1;
            }
        }
        if constexpr (MODE != 2) vstore<V>(yb + (int64_t)i * V, r);
    }
    acc = cf_block_sum<NT / 64>(acc, red);
    if (threadIdx.x == 0) {
        if constexpr (MODE == 2) ldj[b] = 0.5f * acc + c0;
        else ldj[b] = acc + c0;
    }
}

template <int MODE>
int launch_sample(const float* x, const float* u, float* y, float* ldj, int B, int N, int64_t xbs, int64_t ybs,
                  float t1, float s1, float t2, float s2, float c0, hipStream_t s) {
    if (B == 0) return 0;
    const bool vec = (N % 4 == 0) && (xbs % 4 == 0) && (ybs % 4 == 0) && aligned16(x) &&
                     (y == nullptr || aligned16(y)) && (u == nullptr || aligned16(u));
    const int items = vec ? N / 4 : N;
    const bool big = items >= 256;
#define CF_GO(V, NT) k_sample<MODE, V, NT><<<dim3(B), dim3(NT), 0, s>>>(x, u, y, ldj, items, xbs, ybs, t1, s1, t2, s2, c0)
    if (vec) { if (big) CF_GO(4, 256); else CF_GO(4, 64); }
    else     { if (big) CF_GO(1, 256); else CF_GO(1, 64); }
#undef CF_GO
    return 0;
}

// ---------------------------------------------------------------------------------------------
// pre-processing with the noise drawn INSIDE the kernel (no separate RNG kernels, no 4-byte/element noise tensors
// through HBM): Philox4x32-10 keyed by `seed`, counter = (float4 index, kind, offset) - one call gives the four
// uniforms of a float4 of pixels.  The augment channel (augment.py:14-18, gaussian.py:50-72) is filled with
// Box-Muller normals from the same generator and contributes -log q(eps) = sum(eps^2/2 + log(2 pi)/2).
// state[0] = offset: either a per-call value the caller draws from its own generator (FlowSequential: torch's CUDA
// generator, graph-safe), or a self-advancing word (`advance` = 1: k_rng_advance on the same stream).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }   // [0, 1), 24 bits

template <int NT>
__global__ __launch_bounds__(NT) void k_preprocess_rng(const float* __restrict__ x, float* __restrict__ y,
                                                       float* __restrict__ ldj, const unsigned long long* __restrict__ state,
                                                       unsigned long long seed, int n4, int aug4, int64_t x_bstride,
                                                       int64_t y_bstride, float t1, float s1, float t2, float s2, float c0) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    const unsigned long long off = state[0];
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    const float* xb = x + (int64_t)b * x_bstride;
    float* yb = y + (int64_t)b * y_bstride;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n4; i += NT) {
        float r[4];
        unsigned rn[4];
        vload<4>(xb + (int64_t)i * 4, r);
        const unsigned long long idx = (unsigned long long)b * (unsigned)(n4 + aug4) + (unsigned)i;
        philox4x32_10((unsigned)idx, (unsigned)(idx >> 32), (unsigned)off, (unsigned)(off >> 32), k0, k1, rn);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = ((r[j] + u01(rn[j])) / s1 + t1) / s2 + t2;
            const float l0 = logf(v), l1 = logf(1.0f - v);
            r[j] = l0 - l1;
            acc += -l0 - l1;
        }
        vstore<4>(yb + (int64_t)i * 4, r);
    }
    for (int i = threadIdx.x; i < aug4; i += NT) {            // augment channel: 4 standard normals per Philox call
        unsigned rn[4];
        float e[4];
        const unsigned long long idx = (unsigned long long)b * (unsigned)(n4 + aug4) + (unsigned)(n4 + i);
        philox4x32_10((unsigned)idx, (unsigned)(idx >> 32), (unsigned)off, (unsigned)(off >> 32), k0, k1, rn);
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const float rad = sqrtf(-2.0f * logf(1.0f - u01(rn[j]))), ang = 6.28318530717958647692f * u01(rn[j + 1]);
            e[j] = rad * cosf(ang); e[j + 1] = rad * sinf(ang);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += 0.5f * e[j] * e[j] + 0.5f * kLog2Pi;
        vstore<4>(yb + (int64_t)(n4 + i) * 4, e);
    }
    acc = cf_block_sum<NT / 64>(acc, red);
    if (threadIdx.x == 0) ldj[b] = acc + c0;
}

__global__ void k_rng_advance(unsigned long long* state) { state[0] += 1; }

// ---------------------------------------------------------------------------------------------
// squeeze: out[b, c*p1*p2 + i1*p2 + i2, h, w] = in[b, c, h*p1+i1, w*p2+i2]     squeeze.py:10-14
// thread per element of the SQUEEZED tensor (coalesced on that side)
// ---------------------------------------------------------------------------------------------
template <bool INV>
__global__ __launch_bounds__(256) void k_squeeze(const float* __restrict__ x, float* __restrict__ y, int B, int C,
                                                 int H, int W, int p1, int p2, int64_t xbs, int64_t ybs) {
    const int h2 = H / p1, w2 = W / p2;
    const int64_t per = (int64_t)C * H * W;
    const int64_t total = per * B;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / per);
        int r = (int)(i - (int64_t)b * per);                 // index inside the squeezed sample
        const int w = r % w2; r /= w2;
        const int h = r % h2; r /= h2;
        const int i2 = r % p2; r /= p2;
        const int i1 = r % p1; const int c = r / p1;
        const int64_t sq = i - (int64_t)b * per;              // (c p1 p2) h w offset
        const int64_t un = ((int64_t)c * H + (h * p1 + i1)) * W + (w * p2 + i2);
        if (!INV) y[(int64_t)b * ybs + sq] = x[(int64_t)b * xbs + un];
        else      y[(int64_t)b * ybs + un] = x[(int64_t)b * xbs + sq];
    }
}

// ---------------------------------------------------------------------------------------------
// ActNorm
// ---------------------------------------------------------------------------------------------
constexpr int kStatSplits = 64;

// partial sums in fp64: ws[(c*S + s)*2 + {0,1}] = {sum x, sum x^2} over this split's samples
__global__ __launch_bounds__(256) void k_actnorm_partial(const float* __restrict__ x, double* __restrict__ ws,
                                                         int B, int C, int HW, int64_t xbs) {
    __shared__ double red[2][4];
    const int c = blockIdx.x, s = blockIdx.y;
    double a0 = 0.0, a1 = 0.0;
    for (int b = s; b < B; b += kStatSplits) {
        const float* p = x + (int64_t)b * xbs + (int64_t)c * HW;
        for (int i = threadIdx.x; i < HW; i += 256) { const double v = p[i]; a0 += v; a1 += v * v; }
    }
    a0 = cf_wave_sum_d(a0); a1 = cf_wave_sum_d(a1);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = a0; red[1][w] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[((int64_t)c * kStatSplits + s) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ws[((int64_t)c * kStatSplits + s) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// per-channel totals of the split partials: sums[c] = sum x, sums[C + c] = sum x^2 (fp64).  Kept as their own tensor so
// that data-parallel ranks can all-reduce them (2C doubles per layer) and initialise from the GLOBAL batch statistics
__global__ __launch_bounds__(64) void k_actnorm_reduce(const double* __restrict__ ws, double* __restrict__ sums, int C) {
    const int c = blockIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (int s = threadIdx.x; s < kStatSplits; s += 64) {
        a0 += ws[((int64_t)c * kStatSplits + s) * 2 + 0];
        a1 += ws[((int64_t)c * kStatSplits + s) * 2 + 1];
    }
    a0 = cf_wave_sum_d(a0); a1 = cf_wave_sum_d(a1);
    if (threadIdx.x == 0) { sums[c] = a0; sums[C + c] = a1; }
}

// actnorm.py:31-33: mean, log(unbiased std + 1e-8); n = number of elements per channel behind the sums.  `n_dev`
// (optional) overrides n from device memory: the all-reduced element count of a sharded batch
__global__ __launch_bounds__(256) void k_actnorm_from_sums(const double* __restrict__ sums, const double* __restrict__ n_dev,
                                                           double n, float* __restrict__ t, float* __restrict__ logs, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (n_dev != nullptr) n = n_dev[0];
    const double a0 = sums[c], a1 = sums[C + c];
    const double mean = a0 / n;
    double var = (a1 - a0 * mean) / (n - 1.0);
    if (var < 0.0) var = 0.0;
    t[c] = (float)mean;
    logs[c] = (float)log(sqrt(var) + 1e-8);
}

template <bool INV>
__global__ __launch_bounds__(256) void k_actnorm(const float* __restrict__ x, const float* __restrict__ t,
                                                 const float* __restrict__ logs, float* __restrict__ z,
                                                 float* __restrict__ ldj_scalar, int C, int HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        const float tc = t[c], lc = logs[c];
        z[i] = INV ? x[i] * expf(lc) + tc : (x[i] - tc) * expf(-lc);       // actnorm.py:59 / :78
    }
    if (!INV && ldj_scalar != nullptr && blockIdx.x == 0 && threadIdx.x < 64) {
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 64) s += logs[c];
        s = cf_wave_sum(s);
        if (threadIdx.x == 0) ldj_scalar[0] = s;                          // actnorm.py:58 (quirk: +sum logs)
    }
}

// ---------------------------------------------------------------------------------------------
// affine coupling map from the net output                                       coupling.py:52-73
// ---------------------------------------------------------------------------------------------
template <bool INV, int NT>
__global__ __launch_bounds__(NT) void k_coupling_apply(const float* __restrict__ x, const float* __restrict__ h,
                                                       float* __restrict__ z, float* __restrict__ ldj, int half_n) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    const float* xb = x + (int64_t)b * 2 * half_n;
    const float* hb = h + (int64_t)b * 2 * half_n;
    float* zb = z + (int64_t)b * 2 * half_n;
    float acc = 0.f;
    for (int i = threadIdx.x; i < half_n; i += NT) {
        zb[i] = xb[i];
        const float tt = hb[i];
        const float ls = 2.0f * tanhf(hb[half_n + i] * 0.5f);
        const float x1 = xb[half_n + i];
        zb[half_n + i] = INV ? (x1 - tt) / expf(ls) : x1 * expf(ls) + tt;
        acc += ls;
    }
    if (!INV) {
        acc = cf_block_sum<NT / 64>(acc, red);
        if (threadIdx.x == 0) ldj[b] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// log-det bookkeeping                                                      flowsequential.py:18-27
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_logdet_combine(const float* __restrict__ ldM, const float* __restrict__ ld1,
                                                        float* __restrict__ out, int M, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        out[i] = ldM[i] + ld1[i / M];
}

// acc[0] += sum_b logsumexp_m logp[b,m]: fp64 block partials, one fp64 atomic per block (a reported
// scalar, not a parity output)
__global__ __launch_bounds__(256) void k_nll_sum(const float* __restrict__ logp, double* __restrict__ acc, int B, int M) {
    __shared__ double red[4];
    double a = 0.0;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        const float* p = logp + (int64_t)b * M;
        float mx = p[0];
        for (int m = 1; m < M; ++m) mx = fmaxf(mx, p[m]);
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += expf(p[m] - mx);
        a += (double)(mx + logf(s));
    }
    a = cf_wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}

// ---- elementwise flow activations with their log-det (activations.py:34-118, 213-245) ---------------------------
// mode 0 Identity, 1 LeakyRelu(a), 2 SmoothLeakyRelu(a): a x + (1 - a) softplus(x), 3 SmoothTanh(a, b): tanh(a x) + b x,
// 4 Sigmoid(temperature a, eps b), 5 LearnableLeakyRelu: slope sigmoid(*slope_logit) + 0.5 read on the device.
// One workgroup per row of D elements: y, and ldj[row] = sum log|f'(x)| (forward only).  The inverses of the smooth
// activations are the reference's Newton iteration (100 steps from x0 = y, derivative clamped at 1e-2).
__device__ __forceinline__ float act_f(int mode, float x, float a, float b) {
    switch (mode) {
        case 1: case 5: return x < 0.f ? a * x : x;
        case 2: return a * x + (1.f - a) * (fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))));     // logsumexp(0, x)
        case 3: return tanhf(a * x) + b * x;
        case 4: return 1.0f / (1.0f + expf(-a * x));
        default: return x;
    }
}
__device__ __forceinline__ float act_df(int mode, float x, float a, float b) {
    switch (mode) {
        case 1: case 5: return x < 0.f ? a : 1.f;
        case 2: return a + (1.f - a) / (1.f + expf(-x));
        case 3: { const float c = coshf(a * x); return b + a / (c * c); }
        default: return 1.f;
    }
}

__global__ __launch_bounds__(256) void k_activation(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ ldj,
                                                    int D, int mode, float a, float b, const float* __restrict__ slope_logit,
                                                    int inverse) {
    __shared__ float scr[4];
    const int64_t row = blockIdx.x;
    if (mode == 5) a = 1.0f / (1.0f + expf(-slope_logit[0])) + 0.5f;
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float v = x[row * D + d];
        float out;
        if (!inverse) {
            out = act_f(mode, v, a, b);
            if (mode == 4) {                              // log T - softplus(-T x) - softplus(T x)
                const float t = fabsf(a * v);
                acc += logf(a) - (t + 2.0f * log1pf(expf(-t)));
            } else if (mode != 0) {
                acc += logf(fabsf(act_df(mode, v, a, b)));
            }
        } else if (mode == 1 || mode == 5) {
            out = v < 0.f ? v / a : v;
        } else if (mode == 2 || mode == 3) {
            float xi = v;
            for (int it = 0; it < 100; ++it) xi -= (act_f(mode, xi, a, b) - v) / fmaxf(act_df(mode, xi, a, b), 1e-2f);
            out = xi;
        } else if (mode == 4) {
            const float z = fminf(fmaxf(v, b), 1.0f - b);
            out = (logf(z) - log1pf(-z)) / a;
        } else {
            out = v;
        }
        y[row * D + d] = out;
    }
    if (!inverse && ldj != nullptr) {
        acc = cf_block_sum<4>(acc, scr);
        if (threadIdx.x == 0) ldj[row] = acc;
    }
}

// The tail of `inverse` / `sample` in one pass: [Augment.reverse] -> LogitTransform.reverse -> Normalization.reverse x 2 ->
// Dequantization.reverse = x[b, i] = floor(((sigmoid(z[b, i]) - t2) * s2 - t1) * s1) for the first n_keep elements of every sample
// (z_bstride > n_keep drops the augmented channels).  The same operations in the same order as the four k_flat launches + the copy
// of the channel slice it replaces - subtraction and multiplication kept apart (no fma contraction), so the results are bitwise equal.
template <int V>
__global__ __launch_bounds__(256) void k_postprocess_inv(const float* __restrict__ z, float* __restrict__ x, int64_t n_items, int per_sample,
                                                         int64_t zbs, float t2, float s2, float t1, float s1) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_sample;
        const int j = (int)(i - b * per_sample);
        float r[V];
        vload<V>(z + b * zbs + (int64_t)j * V, r);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = 1.0f / (1.0f + expf(-r[e]));               // OP_SIGMOID
            v = __fmul_rn(__fsub_rn(v, t2), s2);                 // OP_AFFINE_INV, Normalization 2
            v = __fmul_rn(__fsub_rn(v, t1), s1);                 // OP_AFFINE_INV, Normalization 1
            r[e] = floorf(v);                                    // OP_FLOOR
        }
        vstore<V>(x + i * V, r);
    }
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int cf_dequant_fwd(const float* x, const float* u, float* y, int64_t n, cf_stream_t stream) {
    if (n == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && u && y && n >= 0);
    launch_flat<OP_ADD2>(x, u, y, n, 0.f, 0.f, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_affine(const float* x, float* y, int64_t n, float translation, float scale, int inverse, cf_stream_t stream) {
    if (n == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && y && n >= 0);
    if (inverse) launch_flat<OP_AFFINE_INV>(x, nullptr, y, n, translation, scale, cf_s(stream));
    else launch_flat<OP_AFFINE_FWD>(x, nullptr, y, n, translation, scale, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_sigmoid(const float* x, float* y, int64_t n, cf_stream_t stream) {
    if (n == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && y && n >= 0);
    launch_flat<OP_SIGMOID>(x, nullptr, y, n, 0.f, 0.f, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_postprocess_inv(const float* z, float* x, int B, int n_keep, int64_t z_bstride, float t2, float s2, float t1, float s1,
                       cf_stream_t stream) {
    if (B == 0 || n_keep == 0) return 0;
    CF_REQUIRE(z && x && B > 0 && n_keep > 0 && z_bstride >= n_keep);
    const bool vec = n_keep % 4 == 0 && z_bstride % 4 == 0 && aligned16(z) && aligned16(x);
    const int per = vec ? n_keep / 4 : n_keep;
    const int64_t items = (int64_t)B * per;
    int64_t blocks = (items + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (vec) k_postprocess_inv<4><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(z, x, items, per, z_bstride, t2, s2, t1, s1);
    else k_postprocess_inv<1><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(z, x, items, per, z_bstride, t2, s2, t1, s1);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_floor(const float* x, float* y, int64_t n, cf_stream_t stream) {
    if (n == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && y && n >= 0);
    launch_flat<OP_FLOOR>(x, nullptr, y, n, 0.f, 0.f, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_logit_fwd(const float* x, float* y, float* ldj, int B, int N, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && y && ldj && B >= 0 && N > 0);
    launch_sample<0>(x, nullptr, y, ldj, B, N, N, N, 0, 1, 0, 1, 0.f, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_preprocess_fwd(const float* x, const float* u, float* y, float* ldj, int B, int N, int64_t y_bstride,
                      float t1, float s1, float t2, float s2, float ldj_const, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && u && y && ldj && B >= 0 && N > 0 && y_bstride >= N);
    launch_sample<1>(x, u, y, ldj, B, N, N, y_bstride, t1, s1, t2, s2, ldj_const, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_preprocess_rng_fwd(const float* x, float* y, float* ldj, uint64_t* rng_state, uint64_t seed, int B, int N,
                          int aug_n, int64_t y_bstride, float t1, float s1, float t2, float s2, float ldj_const,
                          int advance, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && y && ldj && rng_state && B >= 0 && N > 0 && aug_n >= 0 && y_bstride >= N + aug_n);
    CF_REQUIRE(N % 4 == 0 && aug_n % 4 == 0 && y_bstride % 4 == 0 && aligned16(x) && aligned16(y));
    k_preprocess_rng<256><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, y, ldj, (const unsigned long long*)rng_state,
                                                                  (unsigned long long)seed, N / 4, aug_n / 4, N, y_bstride,
                                                                  t1, s1, t2, s2, ldj_const);
    if (advance) k_rng_advance<<<dim3(1), dim3(1), 0, cf_s(stream)>>>((unsigned long long*)rng_state);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_std_normal_nll(const float* eps, float* out, int B, int N, int64_t eps_bstride, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(eps && out && B >= 0 && N > 0 && eps_bstride >= N);
    launch_sample<2>(eps, nullptr, nullptr, out, B, N, eps_bstride, 0, 0, 1, 0, 1, 0.5f * N * kLog2Pi, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_squeeze(const float* x, float* y, int B, int C, int H, int W, int p1, int p2, int64_t x_bstride,
               int64_t y_bstride, int inverse, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && y && B >= 0 && C > 0 && H > 0 && W > 0 && p1 > 0 && p2 > 0 && H % p1 == 0 && W % p2 == 0);
    const int64_t total = (int64_t)B * C * H * W;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (inverse) k_squeeze<true><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, y, B, C, H, W, p1, p2, x_bstride, y_bstride);
    else k_squeeze<false><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, y, B, C, H, W, p1, p2, x_bstride, y_bstride);
    CF_LAUNCH_CHECK();
    return 0;
}

int64_t cf_actnorm_stats_ws_bytes(int C) { return (int64_t)C * (kStatSplits + 1) * 2 * sizeof(double); }

int cf_actnorm_sums(const float* x, double* sums, void* ws, int B, int C, int HW, int64_t x_bstride, cf_stream_t stream) {
    CF_REQUIRE(x && sums && ws && B > 0 && C > 0 && HW > 0);
    k_actnorm_partial<<<dim3(C, kStatSplits), dim3(256), 0, cf_s(stream)>>>(x, (double*)ws, B, C, HW, x_bstride);
    k_actnorm_reduce<<<dim3(C), dim3(64), 0, cf_s(stream)>>>((const double*)ws, sums, C);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_actnorm_from_sums(const double* sums, const double* count_dev, double count, float* t, float* logs, int C,
                         cf_stream_t stream) {
    CF_REQUIRE(sums && t && logs && C > 0 && (count_dev || count > 1.0));
    k_actnorm_from_sums<<<dim3((C + 255) / 256), dim3(256), 0, cf_s(stream)>>>(sums, count_dev, count, t, logs, C);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_actnorm_stats(const float* x, float* t, float* logs, void* ws, int B, int C, int HW, int64_t x_bstride,
                     cf_stream_t stream) {
    CF_REQUIRE(x && t && logs && ws && B > 0 && C > 0 && HW > 0 && (int64_t)B * HW > 1);
    double* sums = (double*)ws + (int64_t)C * kStatSplits * 2;
    int rc = cf_actnorm_sums(x, sums, ws, B, C, HW, x_bstride, stream);
    if (rc) return rc;
    return cf_actnorm_from_sums(sums, nullptr, (double)B * HW, t, logs, C, stream);
}

int cf_actnorm(const float* x, const float* t, const float* logs, float* z, float* ldj_scalar, int B, int C, int HW,
               int inverse, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && t && logs && z && B >= 0 && C > 0 && HW > 0);
    const int64_t total = (int64_t)B * C * HW;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    if (inverse) k_actnorm<true><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, t, logs, z, ldj_scalar, C, HW, total);
    else k_actnorm<false><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, t, logs, z, ldj_scalar, C, HW, total);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_coupling_apply(const float* x, const float* h, float* z, float* ldj, int B, int C, int HW, int inverse,
                      cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && h && z && B >= 0 && C > 0 && C % 2 == 0 && HW > 0 && (inverse || ldj));
    if (B == 0) return 0;
    const int half_n = (C / 2) * HW;
    if (half_n >= 256) {
        if (inverse) k_coupling_apply<true, 256><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, h, z, ldj, half_n);
        else k_coupling_apply<false, 256><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, h, z, ldj, half_n);
    } else {
        if (inverse) k_coupling_apply<true, 64><<<dim3(B), dim3(64), 0, cf_s(stream)>>>(x, h, z, ldj, half_n);
        else k_coupling_apply<false, 64><<<dim3(B), dim3(64), 0, cf_s(stream)>>>(x, h, z, ldj, half_n);
    }
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_logdet_combine(const float* ldM, const float* ld1, float* out, int B, int M, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(ldM && ld1 && out && B >= 0 && M > 0);
    const int64_t total = (int64_t)B * M;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    k_logdet_combine<<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(ldM, ld1, out, M, total);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_nll_sum(const float* logp, double* acc, int B, int M, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(logp && acc && B >= 0 && M > 0);
    if (B == 0) return 0;
    int blocks = (B + 255) / 256;
    if (blocks > 256) blocks = 256;
    k_nll_sum<<<dim3(blocks), dim3(256), 0, cf_s(stream)>>>(logp, acc, B, M);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_activation(const float* x, float* y, float* ldj, int64_t rows, int D, int mode, float a, float b,
                  const float* slope_logit, int inverse, cf_stream_t stream) {
    if (rows == 0) return 0;
    CF_REQUIRE(x && y && rows >= 0 && rows <= 0x7fffffff && D > 0 && mode >= 0 && mode <= 5 && (inverse || ldj) &&
               (mode != 5 || slope_logit));
    k_activation<<<dim3((unsigned)rows), dim3(256), 0, cf_s(stream)>>>(x, y, ldj, D, mode, a, b, slope_logit, inverse);
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
