// Generic k x k stride-1 convolution with reflect padding (coupling.py:26-29), fp32 direct form.
//
// This is the shape-agnostic path (any channel count, (3,1) time-series kernels, odd images): one
// thread per output element, weights and activations served from L1/L2.  The benchmark shapes use
// the fused fp32-MFMA step kernel in cf_step.hip instead.
#include "cf_common.h"

namespace {

__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

template <bool RELU>
__global__ __launch_bounds__(256) void k_conv2d_reflect(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int Cin, int Cout, int H, int W, int kh, int kw, int ph, int pw,
                                                        int64_t xbs, int64_t total) {
    const int HW = H * W;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        int64_t r = g;
        const int px = (int)(r % W); r /= W;
        const int py = (int)(r % H); r /= H;
        const int co = (int)(r % Cout);
        const int64_t b = r / Cout;
        const float* xb = x + b * xbs;
        const float* wc = w + (int64_t)co * Cin * kh * kw;
        float acc = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < Cin; ++ci) {
            const float* xc = xb + (int64_t)ci * HW;
            const float* wk = wc + ci * kh * kw;
            for (int ky = 0; ky < kh; ++ky) {
                const int yy = reflect(py + ky - ph, H);
                for (int kx = 0; kx < kw; ++kx) {
                    const int xx = reflect(px + kx - pw, W);
                    acc = fmaf(wk[ky * kw + kx], xc[yy * W + xx], acc);
                }
            }
        }
        y[g] = RELU ? fmaxf(acc, 0.f) : acc;
    }
}

}  // namespace

extern "C" int cf_conv2d_reflect(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                                 int H, int W, int kh, int kw, int ph, int pw, int relu, int64_t x_bstride,
                                 cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && w && y && B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && kh > 0 && kw > 0);
    CF_REQUIRE(ph >= 0 && pw >= 0 && ph < H && pw < W && kh == 2 * ph + 1 && kw == 2 * pw + 1);
    const int64_t total = (int64_t)B * Cout * H * W;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (relu) k_conv2d_reflect<true><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, w, bias, y, Cin, Cout, H, W, kh, kw, ph, pw, x_bstride, total);
    else k_conv2d_reflect<false><<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(x, w, bias, y, Cin, Cout, H, W, kh, kw, ph, pw, x_bstride, total);
    CF_LAUNCH_CHECK();
    return 0;
}
