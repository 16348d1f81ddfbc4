// Invertible 1x1 convolution (per-pixel C x C mat-vec) and the tiny dense linear algebra it needs
// (log|det W| and W^-1), gfx950.
//
// cf_conv1x1_fwd is the generic, shape-agnostic kernel (any C <= 128, any H*W): HBM-bound in the
// ideal, in practice LDS-broadcast-bound; the fused MFMA step kernel (cf_step.hip) replaces it on
// the benchmark shapes.  W^T lives in LDS (padded to 8 outputs), one thread owns one pixel and
// produces 8 outputs at a time from two broadcast ds_read_b128 per input channel.
#include "cf_common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void k_conv1x1(const float* __restrict__ x, const float* __restrict__ Wm,
                                                 const float* __restrict__ bias, float* __restrict__ z,
                                                 int C, int Cp, int HW, int64_t npix, int64_t xbs, int64_t zbs) {
    extern __shared__ __align__(16) float Wt[];          // [C][Cp]  Wt[i][o] = Wm[o][i], zero padded
    for (int e = threadIdx.x; e < C * Cp; e += 256) {
        const int i = e / Cp, o = e - i * Cp;
        Wt[e] = (o < C) ? Wm[o * C + i] : 0.f;
    }
    __syncthreads();
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= npix) return;
    const int64_t b = g / HW;
    const int p = (int)(g - b * HW);
    const float* xp = x + b * xbs + p;
    float* zp = z + b * zbs + p;
    for (int o0 = 0; o0 < Cp; o0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (bias != nullptr && o0 + j < C) ? bias[o0 + j] : 0.f;
        for (int i = 0; i < C; ++i) {
            const float xv = xp[(int64_t)i * HW];
            const float4 w0 = *reinterpret_cast<const float4*>(&Wt[i * Cp + o0]);
            const float4 w1 = *reinterpret_cast<const float4*>(&Wt[i * Cp + o0 + 4]);
            acc[0] = fmaf(w0.x, xv, acc[0]); acc[1] = fmaf(w0.y, xv, acc[1]);
            acc[2] = fmaf(w0.z, xv, acc[2]); acc[3] = fmaf(w0.w, xv, acc[3]);
            acc[4] = fmaf(w1.x, xv, acc[4]); acc[5] = fmaf(w1.y, xv, acc[5]);
            acc[6] = fmaf(w1.z, xv, acc[6]); acc[7] = fmaf(w1.w, xv, acc[7]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (o0 + j < C) zp[(int64_t)(o0 + j) * HW] = acc[j];
    }
}

// In-place Gauss-Jordan with partial pivoting in fp64: log|det| = sum log|pivot|, inverse optional.
// One block; C <= 64 (32 KiB of LDS).
constexpr int kMaxLU = 64;

__global__ __launch_bounds__(256) void k_slogdet_inverse(const float* __restrict__ Wm, int C,
                                                         float* __restrict__ logabsdet, float* __restrict__ inv) {
    __shared__ double A[kMaxLU * kMaxLU];
    __shared__ double colk[kMaxLU];
    __shared__ int perm[kMaxLU];
    __shared__ int prow;
    __shared__ double lsum;
    const int tid = threadIdx.x;
    for (int e = tid; e < C * C; e += 256) A[e] = (double)Wm[e];
    if (tid == 0) lsum = 0.0;
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        if (tid == 0) {
            int r = k; double best = fabs(A[k * C + k]);
            for (int i = k + 1; i < C; ++i) { const double v = fabs(A[i * C + k]); if (v > best) { best = v; r = i; } }
            prow = r; perm[k] = r;
            lsum += log(best);
        }
        __syncthreads();
        const int r = prow;
        if (r != k) {
            for (int j = tid; j < C; j += 256) { const double tmp = A[k * C + j]; A[k * C + j] = A[r * C + j]; A[r * C + j] = tmp; }
        }
        __syncthreads();
        const double piv = A[k * C + k];
        if (tid < C) colk[tid] = A[tid * C + k];             // column k before it is overwritten
        __syncthreads();
        // row k: A[k][k] = 1, then scale by 1/piv
        for (int j = tid; j < C; j += 256) A[k * C + j] = ((j == k) ? 1.0 : A[k * C + j]) / piv;
        __syncthreads();
        // other rows: A[i][k] = 0, then A[i][:] -= f_i * A[k][:]
        for (int e = tid; e < C * C; e += 256) {
            const int i = e / C, j = e - i * C;
            if (i == k) continue;
            const double f = colk[i];
            const double cur = (j == k) ? 0.0 : A[e];
            A[e] = cur - f * A[k * C + j];
        }
        __syncthreads();
    }
    if (tid == 0) logabsdet[0] = (float)lsum;
    if (inv != nullptr) {
        for (int k = C - 1; k >= 0; --k) {                   // undo the row swaps as column swaps
            const int r = perm[k];
            if (r != k) {
                for (int i = tid; i < C; i += 256) { const double tmp = A[i * C + k]; A[i * C + k] = A[i * C + r]; A[i * C + r] = tmp; }
            }
            __syncthreads();
        }
        for (int e = tid; e < C * C; e += 256) inv[e] = (float)A[e];
    }
}

// log|det W| only (the per-call hot path of Conv1x1.forward): Gaussian elimination with IMPLICIT partial
// pivoting in fp64.  The matrix sits column-major in LDS (lane = row: conflict-free), rows are never
// swapped: the pivot row index r is wave-uniform, so its elements are read as LDS broadcasts.  Every
// wave finds the pivot redundantly (6-shuffle argmax, no cross-wave traffic); the 4 waves split the
// columns still to be updated; one barrier per step.  log|det| = sum log|pivot| (sign irrelevant).
__global__ __launch_bounds__(256) void k_slogdet_lds(const float* __restrict__ Wm, int C, float* __restrict__ logabsdet) {
    __shared__ double A[kMaxLU * kMaxLU];                       // A[j*64 + i] = W[i][j]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < C * C; e += 256) { const int i = e / C, j = e - i * C; A[j * kMaxLU + i] = (double)Wm[e]; }
    bool used = lane >= C;
    double lsum = 0.0;
    for (int k = 0; k < C; ++k) {
        __syncthreads();                                        // column k is final, earlier updates visible
        const double ak = A[k * kMaxLU + lane];
        double best = used ? -1.0 : fabs(ak);
        int bi = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        const int r = __builtin_amdgcn_readfirstlane(bi);       // pivot row, identical in every lane and wave
        lsum += log(best);
        const double piv = A[k * kMaxLU + r];
        const double f = (used || lane == r) ? 0.0 : ak / piv;
        for (int j = k + 1 + w; j < C; j += 4) A[j * kMaxLU + lane] -= f * A[j * kMaxLU + r];
        if (lane == r) used = true;
    }
    if (tid == 0) logabsdet[0] = (float)lsum;
}

}  // namespace

extern "C" {

int cf_conv1x1_fwd(const float* x, const float* Wm, const float* bias, float* z, int B, int C, int HW,
                   int64_t x_bstride, int64_t z_bstride, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && Wm && z && B >= 0 && C > 0 && C <= 128 && HW > 0);
    const int64_t npix = (int64_t)B * HW;
    if (npix == 0) return 0;
    const int Cp = (C + 7) / 8 * 8;
    const size_t lds = (size_t)C * Cp * sizeof(float);
    k_conv1x1<<<dim3((unsigned)((npix + 255) / 256)), dim3(256), lds, cf_s(stream)>>>(x, Wm, bias, z, C, Cp, HW, npix,
                                                                                      x_bstride, z_bstride);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_slogdet_inverse(const float* Wm, int C, float* logabsdet, float* inv, cf_stream_t stream) {
    CF_REQUIRE(Wm && logabsdet && C > 0);
    if (C > kMaxLU) { cf_set_error("cf_slogdet_inverse: C=%d > %d unsupported", C, kMaxLU); return CF_ERR_UNSUPPORTED; }
    if (inv == nullptr) k_slogdet_lds<<<dim3(1), dim3(256), 0, cf_s(stream)>>>(Wm, C, logabsdet);
    else k_slogdet_inverse<<<dim3(1), dim3(256), 0, cf_s(stream)>>>(Wm, C, logabsdet, inv);
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
