// Generic k x k stride-1 convolution (coupling.py:26-29 with any channel count / image size / (3,1) time-series
// kernels; the masked convolutions of MaskedCoupling, ar.py; every first, ActNorm-initialising call of the image flows)
// as an implicit GEMM on the exact-fp32 matrix cores.
//
//   y[b, co, o] = bias[co] + sum_{tap, ci} w[co, ci, tap] * x[b, ci, src(o, tap)]
//   rows = output channels (A operand: weights), columns = output pixels of the whole batch (B operand: gathered x),
//   K = taps x input channels, walked tap by tap in chunks of 16 channels.
// A workgroup = 4 waves x 32 pixel columns x RT row tiles of 32 output channels.  Per chunk the weight slice
// W[rows][16 channels] of the current tap goes through LDS (stored [k][row]: the A fragment reads are 32 consecutive
// floats), double buffered, fetched while the previous chunk's MFMAs run; the B operand is gathered straight from
// global memory (consecutive lanes = consecutive pixels: coalesced except at the reflected borders; the 9 taps of a
// pixel hit L1), one chunk ahead as well.  Padding: 'reflect' (padding_mode of the coupling nets) or zeros with an
// output larger than the input (the transposed convolution of the backward: cf_conv2d_zero).
// The benchmark shapes run the fused step kernels (cf_step.hip) instead; this is the shape-agnostic path.
#include "cf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

constexpr int CK = 16;            // input channels per chunk (8 k-steps)

template <int RT, bool REFLECT, bool RELU>
__global__ __launch_bounds__(256) void k_conv_mfma(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y, int Cin, int Cout,
                                                   int Hi, int Wi, int Ho, int Wo, int kh, int kw, int ph, int pw,
                                                   int64_t xbs, int64_t N) {
    constexpr int ROWS = 32 * RT;
    __shared__ float wl[2][CK * ROWS];                 // [buffer][k][row]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int HWi = Hi * Wi, HWo = Ho * Wo, T = kh * kw;
    const int co0 = blockIdx.y * ROWS;
    const int64_t n = ((int64_t)blockIdx.x * 4 + wave) * 32 + li;           // output pixel of the whole batch
    const bool live = n < N;
    const int64_t nn = live ? n : N - 1;
    const int b = (int)(nn / HWo), o = (int)(nn - (int64_t)b * HWo), oy = o / Wo, ox = o - oy * Wo;
    const float* xb = x + (int64_t)b * xbs;
    const int nchunk = (Cin + CK - 1) / CK, nsteps = T * nchunk;

    // staging of W[co0 + row][ci0 .. ci0 + 15][tap] -> wl[k][row]: thread t owns row t % ROWS and the 16 * ROWS / 256 k's
    // t / ROWS, + 256 / ROWS, ...
    constexpr int WPT = CK * ROWS / 256;               // weight values per thread and chunk (2 or 4)
    const int srow = tid % ROWS, sk0 = tid / ROWS;
    float wreg[WPT], breg[CK / 2];
    auto fetch = [&](int step) {                       // operands of chunk `step` -> registers
        const int tap = step / nchunk, ci0 = (step - tap * nchunk) * CK;
        const int co = co0 + srow;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int ci = ci0 + sk0 + i * (256 / ROWS);
            wreg[i] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * T + tap] : 0.f;
        }
        const int ky = tap / kw, kx = tap - ky * kw;
        int sy = oy + ky - ph, sx = ox + kx - pw;
        bool ok = true;
        if (REFLECT) { sy = reflect(sy, Hi); sx = reflect(sx, Wi); }
        else { ok = sy >= 0 && sy < Hi && sx >= 0 && sx < Wi; sy = min(max(sy, 0), Hi - 1); sx = min(max(sx, 0), Wi - 1); }
        const float* xs = xb + sy * Wi + sx;
#pragma unroll
        for (int s = 0; s < CK / 2; ++s) {
            const int ci = min(ci0 + 2 * s + lk, Cin - 1);                   // padded channels: zero weights, finite data
            const float v = xs[(int64_t)ci * HWi];
            breg[s] = ok ? v : 0.f;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) wl[buf][(sk0 + i * (256 / ROWS)) * ROWS + srow] = wreg[i];
    };

    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;

    fetch(0);
    stage(0);
    float bcur[CK / 2];
#pragma unroll
    for (int s = 0; s < CK / 2; ++s) bcur[s] = breg[s];
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) fetch(step + 1);        // next chunk's weights and pixels in flight during the MFMAs
#pragma unroll
        for (int s = 0; s < CK / 2; ++s) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[buf][(2 * s + lk) * ROWS + rt * 32 + li], bcur[s], acc[rt], 0, 0, 0);
        }
        if (step + 1 < nsteps) {
            stage(buf ^ 1);                            // the other buffer: last read two steps ago, behind a barrier
#pragma unroll
            for (int s = 0; s < CK / 2; ++s) bcur[s] = breg[s];
        }
        __syncthreads();
    }
    if (!live) return;
    float* yb = y + (int64_t)b * Cout * HWo + o;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (co < Cout) {
                float v = acc[rt][r] + (bias ? bias[co] : 0.f);
                if (RELU) v = fmaxf(v, 0.f);
                yb[(int64_t)co * HWo] = v;
            }
        }
}

template <bool REFLECT>
int launch_conv(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Hi, int Wi, int Ho,
                int Wo, int kh, int kw, int ph, int pw, int relu, int64_t xbs, hipStream_t s) {
    const int64_t N = (int64_t)B * Ho * Wo;
    const unsigned gx = (unsigned)((N + 127) / 128);
#define CF_CONV(RT, RL) k_conv_mfma<RT, REFLECT, RL><<<dim3(gx, (Cout + 32 * RT - 1) / (32 * RT)), dim3(256), 0, s>>>( \
        x, w, bias, y, Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, ph, pw, xbs, N)
    if (Cout <= 32) { if (relu) CF_CONV(1, true); else CF_CONV(1, false); }
    else            { if (relu) CF_CONV(2, true); else CF_CONV(2, false); }
#undef CF_CONV
    return 0;
}

// adjoint of reflect padding: gx[b, c, p] = sum of gpad over the padded positions whose source pixel is p
__global__ __launch_bounds__(256) void k_reflect_pad_adjoint(const float* __restrict__ gpad, float* __restrict__ gx, int H,
                                                             int W, int ph, int pw, int64_t total) {
    const int Hp = H + 2 * ph, Wp = W + 2 * pw;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int xq = (int)(e % W), yq = (int)((e / W) % H);
        const int64_t bc = e / ((int64_t)W * H);
        const float* g = gpad + bc * Hp * Wp;
        // padded rows mapping to yq: yq + ph itself, the top mirror ph - yq (1 <= yq <= ph), the bottom mirror
        int ys[3], nys = 0, xs[3], nxs = 0;
        ys[nys++] = yq + ph;
        if (yq >= 1 && yq <= ph) ys[nys++] = ph - yq;
        if (yq <= H - 2 && yq >= H - 1 - ph) ys[nys++] = ph + 2 * (H - 1) - yq;
        xs[nxs++] = xq + pw;
        if (xq >= 1 && xq <= pw) xs[nxs++] = pw - xq;
        if (xq <= W - 2 && xq >= W - 1 - pw) xs[nxs++] = pw + 2 * (W - 1) - xq;
        float s = 0.f;
        for (int i = 0; i < nys; ++i)
            for (int j = 0; j < nxs; ++j) s += g[ys[i] * Wp + xs[j]];
        gx[e] = s;
    }
}

}  // namespace

extern "C" {

int cf_conv2d_reflect(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                      int H, int W, int kh, int kw, int ph, int pw, int relu, int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && w && y && B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && kh > 0 && kw > 0);
    CF_REQUIRE(ph >= 0 && pw >= 0 && ph < H && pw < W && kh == 2 * ph + 1 && kw == 2 * pw + 1);
    launch_conv<true>(x, w, bias, y, B, Cin, Cout, H, W, H, W, kh, kw, ph, pw, relu, x_bstride, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_conv2d_zero(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Hi, int Wi,
                   int kh, int kw, int ph, int pw, int relu, int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && w && y && B >= 0 && Cin > 0 && Cout > 0 && Hi > 0 && Wi > 0 && kh > 0 && kw > 0 && ph >= 0 && pw >= 0);
    const int Ho = Hi + 2 * ph - kh + 1, Wo = Wi + 2 * pw - kw + 1;
    CF_REQUIRE(Ho > 0 && Wo > 0);
    launch_conv<false>(x, w, bias, y, B, Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, ph, pw, relu, x_bstride, cf_s(stream));
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_reflect_pad_adjoint(const float* gpad, float* gx, int BC, int H, int W, int ph, int pw, cf_stream_t stream) {
    if (BC == 0) return 0;
    CF_REQUIRE(gpad && gx && BC > 0 && H > 0 && W > 0 && ph >= 0 && pw >= 0 && ph < H && pw < W);
    const int64_t total = (int64_t)BC * H * W;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    k_reflect_pad_adjoint<<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(gpad, gx, H, W, ph, pw, total);
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
