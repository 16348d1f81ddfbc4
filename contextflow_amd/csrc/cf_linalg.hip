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

// log|det W| and (optionally) W^-1 of one C x C matrix, C <= 64: the per-call parameter work of Conv1x1
// (reference conv1x1.py:21-31: torch.slogdet / torch.inverse on every forward / reverse call).
//
// A latency problem, not a throughput one (64 dependent pivot steps), so the matrix lives in REGISTERS:
// lane = row, wave w owns the columns j = 4 jj + w (NJ = CMAX/4 doubles per lane, statically indexed: the
// pivot loop is unrolled by template recursion).  Gaussian elimination in fp64 with IMPLICIT partial
// pivoting - rows are never swapped, the pivot row index r is wave-uniform:
//   * column k is broadcast to the 4 waves through a double-buffered 64-double LDS line (1 barrier / step);
//   * every wave finds the pivot redundantly: 6 xor-shuffles of one 32-bit key (fp32 magnitude, low 6 bits =
//     63 - lane so that ties go to the lowest row; a near-maximal pivot is as stable as the maximal one);
//   * pivot-row operands come from v_readlane (SGPR operands of the fp64 FMAs), no LDS traffic.
// INV: Gauss-Jordan on [A | I] (rows that already served as pivots keep being eliminated), the pivot-row
// scaling is deferred to the store: W^-1[k][:] = E[r_k][:] / pivot_k.  log|det| = sum_k log|pivot_k|.
constexpr int kMaxLU = 64;

__device__ __forceinline__ double readlane_f64(double v, int r) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), r);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), r);
    return __hiloint2double(hi, lo);
}

template <int NJ> struct LuState {
    double a[NJ];      // A[lane][4 jj + w]
    double e[NJ];      // E[lane][4 jj + w]  (INV only)
    double piv;        // pivot of the step this row served in
    int col;           // that step
    bool used;
};

template <int CMAX, bool INV, int K>
__device__ __forceinline__ void lu_steps(LuState<CMAX / 4>& s, double* colbuf, int C, int lane, int w) {
    if constexpr (K < CMAX) {
        constexpr int NJ = CMAX / 4;
        if (K < C) {                                              // uniform
            double* line = colbuf + (K & 1) * kMaxLU;
            if (w == (K & 3)) line[lane] = s.a[K >> 2];
            __syncthreads();
            const double ak = line[lane];
            unsigned key = s.used ? 0u : ((__float_as_uint((float)fabs(ak)) & ~63u) | (unsigned)(63 - lane));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned ok = (unsigned)__shfl_xor((int)key, o, 64); key = ok > key ? ok : key; }
            const int r = __builtin_amdgcn_readfirstlane(63 - (int)(key & 63u));
            const double piv = readlane_f64(ak, r);
            const bool self = lane == r;
            const double f = (self || (!INV && s.used)) ? 0.0 : ak / piv;
#pragma unroll
            for (int jj = K >> 2; jj < NJ; ++jj) s.a[jj] = fma(-f, readlane_f64(s.a[jj], r), s.a[jj]);
            if constexpr (INV) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) s.e[jj] = fma(-f, readlane_f64(s.e[jj], r), s.e[jj]);
            }
            if (self) { s.used = true; s.piv = ak; s.col = K; }
        }
        lu_steps<CMAX, INV, K + 1>(s, colbuf, C, lane, w);
    }
}

// One workgroup per matrix; blockIdx.y picks the matrix of a batch (cf_slogdet_inverse_batch: the Conv1x1 weights of all flow
// steps of a resolution level factorise side by side - at a batch of 256 the twelve serial 15-68 us launches of a cifar10
// training step were 0.4 ms of its 2.5).
constexpr int kMaxBatch = 16;
struct SlogdetBatch {
    const float* Wm[kMaxBatch];
    float* lad[kMaxBatch];
    float* inv[kMaxBatch];
};
template <int CMAX, bool INV>
__global__ __launch_bounds__(256) void k_slogdet(const SlogdetBatch bt, int C) {
    const float* __restrict__ Wm = bt.Wm[blockIdx.y];
    float* __restrict__ logabsdet = bt.lad[blockIdx.y];
    float* __restrict__ inv = bt.inv[blockIdx.y];
    constexpr int NJ = CMAX / 4;
    __shared__ double colbuf[2 * kMaxLU];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    LuState<NJ> s;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = 4 * jj + w;
        s.a[jj] = (lane < C && j < C) ? (double)Wm[lane * C + j] : 0.0;
        s.e[jj] = (lane == j) ? 1.0 : 0.0;
    }
    s.used = lane >= C; s.piv = 1.0; s.col = 0;
    lu_steps<CMAX, INV, 0>(s, colbuf, C, lane, w);
    if (w == 0) {
        const double l = cf_wave_sum_d(lane < C ? log(fabs(s.piv)) : 0.0);
        if (lane == 0) logabsdet[0] = (float)l;
    }
    if constexpr (INV) {
        if (lane < C) {
            const double rp = 1.0 / s.piv;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = 4 * jj + w;
                if (j < C) inv[s.col * C + j] = (float)(s.e[jj] * rp);
            }
        }
    }
}

// 64 < C <= 128, log|det| only (the ATM topology's 72 / 76-channel Conv1x1, model.py:149-151): the same scheme with
// TWO rows per lane (row = lane + 64 h); the pivot row's slot h is wave-uniform.
struct LuState2 {
    double a[2][32];   // A[lane + 64 h][4 jj + w]
    double piv[2];
    bool used[2];
};

template <int K>
__device__ __forceinline__ void lu_steps2(LuState2& s, double* colbuf, int C, int lane, int w) {
    if constexpr (K < 128) {
        if (K < C) {                                              // uniform
            double* line = colbuf + (K & 1) * 128;
            if (w == (K & 3)) { line[lane] = s.a[0][K >> 2]; line[64 + lane] = s.a[1][K >> 2]; }
            __syncthreads();
            const double ak0 = line[lane], ak1 = line[64 + lane];
            const unsigned k0 = s.used[0] ? 0u : ((__float_as_uint((float)fabs(ak0)) & ~127u) | (unsigned)(127 - lane));
            const unsigned k1 = s.used[1] ? 0u : ((__float_as_uint((float)fabs(ak1)) & ~127u) | (unsigned)(63 - lane));
            unsigned key = k0 > k1 ? k0 : k1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned ok = (unsigned)__shfl_xor((int)key, o, 64); key = ok > key ? ok : key; }
            const int r = __builtin_amdgcn_readfirstlane(127 - (int)(key & 127u));   // pivot row in [0, 128)
            const int rl = r & 63;
            const bool hi = r >= 64;                                                    // uniform
            const double piv = readlane_f64(hi ? ak1 : ak0, rl);
            const bool self0 = !hi && lane == rl, self1 = hi && lane == rl;
            const double f0 = (self0 || s.used[0]) ? 0.0 : ak0 / piv;
            const double f1 = (self1 || s.used[1]) ? 0.0 : ak1 / piv;
#pragma unroll
            for (int jj = K >> 2; jj < 32; ++jj) {
                const double pr = readlane_f64(hi ? s.a[1][jj] : s.a[0][jj], rl);
                s.a[0][jj] = fma(-f0, pr, s.a[0][jj]);
                s.a[1][jj] = fma(-f1, pr, s.a[1][jj]);
            }
            if (self0) { s.used[0] = true; s.piv[0] = ak0; }
            if (self1) { s.used[1] = true; s.piv[1] = ak1; }
        }
        lu_steps2<K + 1>(s, colbuf, C, lane, w);
    }
}

__global__ __launch_bounds__(256) void k_slogdet128(const float* __restrict__ Wm, int C, float* __restrict__ logabsdet) {
    __shared__ double colbuf[2 * 128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    LuState2 s;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = lane + 64 * h;
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) {
            const int j = 4 * jj + w;
            s.a[h][jj] = (row < C && j < C) ? (double)Wm[row * C + j] : 0.0;
        }
        s.used[h] = row >= C; s.piv[h] = 1.0;
    }
    lu_steps2<0>(s, colbuf, C, lane, w);
    if (w == 0) {
        const double l = cf_wave_sum_d((lane < C ? log(fabs(s.piv[0])) : 0.0) + (lane + 64 < C ? log(fabs(s.piv[1])) : 0.0));
        if (lane == 0) logabsdet[0] = (float)l;
    }
}

template <int CMAX>
void launch_slogdet(const SlogdetBatch& bt, int n, int C, bool with_inverse, hipStream_t st) {
    if (!with_inverse) k_slogdet<CMAX, false><<<dim3(1, n), dim3(256), 0, st>>>(bt, C);
    else k_slogdet<CMAX, true><<<dim3(1, n), dim3(256), 0, st>>>(bt, C);
}

// Inverse for 64 < C <= 128 (ATM: 76-channel Conv1x1, training only: d log|det W| / dW = W^-T) and log|det| + inverse
// for 128 < C <= 192 (the FC layers of the ATM context-encoder flows: width 144 / 152, model.py:52-66).  In-place
// Gauss-Jordan with partial pivoting on an fp32 copy in LDS (row stride C + 1), one workgroup; the column swaps that
// undo the row pivoting run at the end; log|det| = sum log|pivot| accumulated in fp64.  Conv1x1 weights start orthogonal
// and stay well conditioned.
constexpr int kMaxLdsLU = 192;
__global__ __launch_bounds__(256) void k_inverse_lds(const float* __restrict__ Wm, int C, float* __restrict__ inv,
                                                     float* __restrict__ logabsdet) {
    extern __shared__ float A[];                   // [C][C+1]
    __shared__ int piv[kMaxLdsLU];
    __shared__ float colk[kMaxLdsLU];
    double lad = 0.0;                              // thread 0
    __shared__ float rv[4];
    __shared__ int ri[4];
    const int S = C + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < C * C; e += 256) A[(e / C) * S + (e % C)] = Wm[e];
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        // pivot search over rows k .. C-1 of column k
        float best = -1.f;
        int bi = k;
        for (int r = k + tid; r < C; r += 256) {
            const float v = fabsf(A[r * S + k]);
            if (v > best) { best = v; bi = r; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float b = rv[0];
            int p = ri[0];
            for (int w = 1; w < 4; ++w)
                if (rv[w] > b || (rv[w] == b && ri[w] < p)) { b = rv[w]; p = ri[w]; }
            piv[k] = p;
        }
        __syncthreads();
        const int p = piv[k];
        if (p != k)
            for (int c = tid; c < C; c += 256) { const float t = A[k * S + c]; A[k * S + c] = A[p * S + c]; A[p * S + c] = t; }
        __syncthreads();
        const float pk = A[k * S + k];
        const float d = 1.0f / pk;
        if (tid == 0) lad += log(fabs((double)pk));
        __syncthreads();
        for (int c = tid; c < C; c += 256) A[k * S + c] = (c == k ? 1.0f : A[k * S + c]) * d;
        for (int r = tid; r < C; r += 256) colk[r] = A[r * S + k];
        __syncthreads();
        for (int e = tid; e < C * C; e += 256) {
            const int r = e / C, c = e - r * C;
            if (r != k) A[r * S + c] = (c == k ? 0.f : A[r * S + c]) - colk[r] * A[k * S + c];
        }
        __syncthreads();
    }
    for (int k = C - 1; k >= 0; --k) {
        const int p = piv[k];
        if (p != k)
            for (int r = tid; r < C; r += 256) { const float t = A[r * S + k]; A[r * S + k] = A[r * S + p]; A[r * S + p] = t; }
        __syncthreads();
    }
    if (inv != nullptr)
        for (int e = tid; e < C * C; e += 256) inv[e] = A[(e / C) * S + (e % C)];
    if (logabsdet != nullptr && tid == 0) logabsdet[0] = (float)lad;
}

}  // namespace

extern "C" {

int cf_conv1x1_fwd(const float* x, const float* Wm, const float* bias, float* z, int B, int C, int HW,
                   int64_t x_bstride, int64_t z_bstride, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && Wm && z && B >= 0 && C > 0 && C <= 192 && HW > 0);
    const int64_t npix = (int64_t)B * HW;
    if (npix == 0) return 0;
    const int Cp = (C + 7) / 8 * 8;
    const size_t lds = (size_t)C * Cp * sizeof(float);
    if (lds > 64 * 1024) {                      // 128 < C <= 192: the FC layers of the ATM context-encoder flows
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_conv1x1, 160 * 1024, raised, __func__)) return rc_;
    }
    k_conv1x1<<<dim3((unsigned)((npix + 255) / 256)), dim3(256), lds, cf_s(stream)>>>(x, Wm, bias, z, C, Cp, HW, npix,
                                                                                      x_bstride, z_bstride);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_slogdet_inverse(const float* Wm, int C, float* logabsdet, float* inv, cf_stream_t stream) {
    CF_REQUIRE(Wm && logabsdet && C > 0);
    if (C > kMaxLU) {
        if (C > kMaxLdsLU) { cf_set_error("cf_slogdet_inverse: C=%d unsupported (up to %d)", C, kMaxLdsLU); return CF_ERR_UNSUPPORTED; }
        if (C <= 128) k_slogdet128<<<dim3(1), dim3(256), 0, cf_s(stream)>>>(Wm, C, logabsdet);
        if (inv != nullptr || C > 128) {
            const size_t lds = (size_t)C * (C + 1) * sizeof(float);
            if (lds > 64 * 1024) {
                static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_inverse_lds, 152 * 1024, raised, __func__)) return rc_;
            }
            k_inverse_lds<<<dim3(1), dim3(256), lds, cf_s(stream)>>>(Wm, C, inv, C > 128 ? logabsdet : nullptr);
        }
        CF_LAUNCH_CHECK();
        return 0;
    }
    const float* wm1[1] = {Wm};
    float* lad1[1] = {logabsdet};
    float* inv1[1] = {inv};
    return cf_slogdet_inverse_batch(1, wm1, C, lad1, inv ? inv1 : nullptr, stream);
}

int cf_slogdet_inverse_batch(int n, const float* const* Wm, int C, float* const* logabsdet, float* const* inv, cf_stream_t stream) {
    CF_REQUIRE(n >= 0 && Wm && logabsdet && C > 0);
    if (C > kMaxLU) {                            // wide matrices: one at a time
        for (int i = 0; i < n; ++i)
            if (int rc = cf_slogdet_inverse(Wm[i], C, logabsdet[i], inv ? inv[i] : nullptr, stream)) return rc;
        return 0;
    }
    for (int i0 = 0; i0 < n; i0 += kMaxBatch) {
        const int m = n - i0 < kMaxBatch ? n - i0 : kMaxBatch;
        SlogdetBatch bt{};
        for (int i = 0; i < m; ++i) {
            CF_REQUIRE(Wm[i0 + i] && logabsdet[i0 + i] && (!inv || inv[i0 + i]));
            bt.Wm[i] = Wm[i0 + i]; bt.lad[i] = logabsdet[i0 + i]; bt.inv[i] = inv ? inv[i0 + i] : nullptr;
        }
        if (C <= 8) launch_slogdet<8>(bt, m, C, inv != nullptr, cf_s(stream));
        else if (C <= 16) launch_slogdet<16>(bt, m, C, inv != nullptr, cf_s(stream));
        else if (C <= 32) launch_slogdet<32>(bt, m, C, inv != nullptr, cf_s(stream));
        else launch_slogdet<64>(bt, m, C, inv != nullptr, cf_s(stream));
        CF_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
