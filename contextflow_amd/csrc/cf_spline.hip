// SplineActivation: elementwise monotone rational-quadratic spline with linear tails, forward / inverse and
// log|det J|  (contextflow/layers/activations.py:120-211, layers/splines/rational_quadratic.py:21-176).
//
// HBM-bound elementwise op.  The knot tables (cumulative widths / heights, knot derivatives: 3*(K+1) floats per
// parameter set) are built once per call by k_spline_tables — one thread per parameter set, i.e. one thread in
// total for shared weights, C*H*W threads for `individual_weights` — and read through L1/L2 by the main kernel,
// which does the bin search as a K-step compare-and-count (K = 5), then the closed-form RQ map.  One block per
// sample reduces the log-det with wave shuffles.
#include "cf_common.h"
#include <math.h>

namespace {

constexpr int kMaxBins = 32;
constexpr float kMinW = 1e-3f, kMinH = 1e-3f, kMinD = 1e-3f;

__device__ __forceinline__ float softplus1(float v) { return v > 20.f ? v : log1pf(expf(v)); }

// table[p] = [cumwidths(K+1) | cumheights(K+1) | derivatives(K+1)]      rational_quadratic.py:36-48,98-118
__global__ __launch_bounds__(64) void k_spline_tables(const float* __restrict__ uw, const float* __restrict__ uh,
                                                      const float* __restrict__ ud, float* __restrict__ table, int P, int K,
                                                      float bound) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= P) return;
    const float* w = uw + (int64_t)p * K;
    const float* h = uh + (int64_t)p * K;
    const float* d = ud + (int64_t)p * (K - 1);
    float* t = table + (int64_t)p * 3 * (K + 1);
    for (int pass = 0; pass < 2; ++pass) {
        const float* u = pass ? h : w;
        const float mn = pass ? kMinH : kMinW;
        float* c = t + pass * (K + 1);
        float mx = u[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, u[k]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(u[k] - mx);
        float cum = 0.f;
        c[0] = -bound;
        for (int k = 0; k < K; ++k) {
            cum += mn + (1.f - mn * K) * (expf(u[k] - mx) / sum);
            c[k + 1] = 2.f * bound * cum - bound;
        }
        c[K] = bound;
    }
    const float cst = logf(expf(1.f - kMinD) - 1.f);       // boundary derivatives = 1 exactly (linear tails)
    float* dv = t + 2 * (K + 1);
    for (int k = 0; k <= K; ++k) dv[k] = kMinD + softplus1(((k == 0 || k == K) ? 0.f : d[k - 1]) + cst);
}

template <bool INV, int NT>
__global__ __launch_bounds__(NT) void k_spline(const float* __restrict__ x, const float* __restrict__ table,
                                               float* __restrict__ y, float* __restrict__ ldj, int N, int P, int K, float bound) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    const float* xb = x + (int64_t)b * N;
    float* yb = y + (int64_t)b * N;
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += NT) {
        const float v = xb[i];
        float out = v, lad = 0.f;
        if (v >= -bound && v <= bound) {
            const float* t = table + (int64_t)(P == 1 ? 0 : i) * 3 * (K + 1);
            const float* cw = t;
            const float* ch = t + (K + 1);
            const float* dv = t + 2 * (K + 1);
            const float* loc = INV ? ch : cw;
            int idx = -1;                                     // searchsorted: count(knots <= v) - 1, last knot + 1e-6
            for (int k = 0; k <= K; ++k) idx += (v >= (k == K ? loc[k] + 1e-6f : loc[k])) ? 1 : 0;
            idx = min(max(idx, 0), K - 1);
            const float w0 = cw[idx], h0 = ch[idx];
            const float w = cw[idx + 1] - w0, h = ch[idx + 1] - h0;
            const float d0 = dv[idx], d1 = dv[idx + 1];
            const float delta = h / w;
            const float s2 = d0 + d1 - 2.f * delta;
            float th;
            if (INV) {
                const float u = v - h0;
                const float a = u * s2 + h * (delta - d0);
                const float bq = h * d0 - u * s2;
                const float c = -delta * u;
                th = (2.f * c) / (-bq - sqrtf(bq * bq - 4.f * a * c));
                out = th * w + w0;
            } else {
                th = (v - w0) / w;
                out = h0 + h * (delta * th * th + d0 * th * (1.f - th)) / (delta + s2 * th * (1.f - th));
            }
            const float om = 1.f - th;
            const float den = delta + s2 * th * om;
            lad = logf(delta * delta * (d1 * th * th + 2.f * delta * th * om + d0 * om * om)) - 2.f * logf(den);
        }
        yb[i] = out;
        acc += lad;
    }
    if (!INV) {
        acc = cf_block_sum<NT / 64>(acc, red);
        if (threadIdx.x == 0) ldj[b] = acc;
    }
}

}  // namespace

extern "C" {

int64_t cf_spline_table_floats(int P, int K) { return (int64_t)P * 3 * (K + 1); }

int cf_spline_prepare(const float* uw, const float* uh, const float* ud, float* table, int P, int K, float tail_bound,
                      cf_stream_t stream) {
    CF_REQUIRE(uw && uh && ud && table && P > 0 && K >= 2 && K <= kMaxBins && tail_bound > 0.f);
    k_spline_tables<<<dim3((P + 63) / 64), dim3(64), 0, cf_s(stream)>>>(uw, uh, ud, table, P, K, tail_bound);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_spline(const float* x, const float* table, float* y, float* ldj, int B, int N, int P, int K, float tail_bound,
              int inverse, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && table && y && (inverse || ldj) && N > 0 && (P == 1 || P == N) && K >= 2 && K <= kMaxBins);
    if (N >= 1024) {
        if (inverse) k_spline<true, 256><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, table, y, ldj, N, P, K, tail_bound);
        else k_spline<false, 256><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, table, y, ldj, N, P, K, tail_bound);
    } else {
        if (inverse) k_spline<true, 64><<<dim3(B), dim3(64), 0, cf_s(stream)>>>(x, table, y, ldj, N, P, K, tail_bound);
        else k_spline<false, 64><<<dim3(B), dim3(64), 0, cf_s(stream)>>>(x, table, y, ldj, N, P, K, tail_bound);
    }
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
