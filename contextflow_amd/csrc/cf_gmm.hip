// GaussianMixtureDistribution.log_prob (layers/distributions/gaussian.py:138-161) without the
// (B, M, K, D) broadcast the reference materialises.
//
//   out[b,m] = logsumexp_k( cst[m,k] - 1/2 * sum_d ((x[b,d] - mu[mk,d]) / sigma[mk,d])^2 )
//
// Work shape: a "GEMM with a square in the inner product" — (B samples) x (M*K components) x D.
// VALU-bound (2 FMA per term) when operands are reused from registers, so the kernel is register
// tiled: a 256-thread workgroup owns 128 samples x 80 components, each thread an 8 x 5 micro-tile;
// x and the (a, nm) = (1/sigma, -mu) rows stream through LDS in 32-wide d-chunks with
// 36-float row stride (every ds_read_b128 lane group hits distinct 4-bank groups), the next
// chunk's global loads are issued before the current chunk's FMAs.  D can be split over
// blockIdx.z (partials in a caller workspace + a small finishing kernel) to fill 256 CUs when
// B/128 is small.
#include "cf_common.h"
#include <math.h>

namespace {

constexpr float kLog2Pi = 1.8378770664093453f;
constexpr int TB = 128;        // samples per workgroup
constexpr int DC = 32;         // d-chunk
constexpr int LD = DC + 4;     // LDS row stride (floats)
constexpr int SPT = 8;         // samples per thread (strided by 16)

__device__ __forceinline__ float softplus_ref(float v) {      // torch softplus, beta=1, threshold=20
    return v > 20.f ? v : log1pf(expf(v));
}

// one block per component row mk
__global__ __launch_bounds__(256) void k_gmm_prepare(const float* __restrict__ mG, const float* __restrict__ sG,
                                                     const float* __restrict__ wG, float* __restrict__ a,
                                                     float* __restrict__ nm, float* __restrict__ cst, int K, int D) {
    __shared__ float red[4];
    const int mk = blockIdx.x;
    const float* mu = mG + (int64_t)mk * D;
    const float* sg = sG + (int64_t)mk * D;
    float acc = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float s = softplus_ref(sg[d]);
        const float inv = 1.0f / s;
        a[(int64_t)mk * D + d] = inv;
        nm[(int64_t)mk * D + d] = -mu[d];
        acc += logf(s);
    }
    acc = cf_block_sum<4>(acc, red);
    if (threadIdx.x == 0) {
        const int m = mk / K;
        const float* wr = wG + m * K;
        float mx = wr[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, wr[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(wr[k] - mx);
        const float logw = wG[mk] - mx - logf(s);                       // log_softmax(wG[m])[k]
        cst[mk] = logw - acc - 0.5f * (float)D * kLog2Pi;
    }
}

// guarded 4-float load for ragged tails: elements at d >= dend read as 0 (a = nm = 0 makes the term vanish)
template <bool VEC>
__device__ __forceinline__ float4 ld4_tail(const float* row, int d, int dend) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
        if (d < dend) v = *reinterpret_cast<const float4*>(row + d);
    } else {
        if (d + 0 < dend) v.x = row[d + 0];
        if (d + 1 < dend) v.y = row[d + 1];
        if (d + 2 < dend) v.z = row[d + 2];
        if (d + 3 < dend) v.w = row[d + 3];
    }
    return v;
}

// MKT = components per thread; the workgroup covers MKB = 16*MKT components starting at blockIdx.y*MKB.
// Partial sums q[b, mk] over d in [dlo, dhi) go to qout (workspace, [nsplit][B][MKtot]) when SPLIT,
// otherwise the logsumexp epilogue runs here.
// dlo, dhi: this workgroup's range of d; qb: its slot of partial sums [B][MKtot] (SPLIT)
// KEYED: the samples of this workgroup are rows[0 .. nvalid) (indices into x and qb) instead of b0 .. b0 + TB
template <int MKT, bool VEC, bool SPLIT, bool KEYED = false>
__device__ __forceinline__ void gmm_logprob_body(const float* __restrict__ x, const float* __restrict__ a,
                                                 const float* __restrict__ nm, const float* __restrict__ cst,
                                                 float* __restrict__ out, float* __restrict__ qb,
                                                 int B, int MKtot, int K, int D, int dlo, int dhi,
                                                 int64_t xbs, int accumulate, int Mtot,
                                                 const int* __restrict__ rows = nullptr, int nvalid = 0) {
    static_assert(!KEYED || SPLIT, "the keyed form writes partial sums");
    constexpr int MKB = 16 * MKT;
    constexpr int XITEMS = TB * (DC / 4) / 256;                    // float4 per thread for the x chunk (4)
    constexpr int PITEMS = (MKB * (DC / 4) + 255) / 256;           // float4 per thread per parameter array
    constexpr int LDS_MAIN = (TB + 2 * MKB) * LD;
    constexpr int LDS_EPI = TB * (MKB + 1);
    constexpr int LDS_FLOATS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
    __shared__ __align__(16) float lds[LDS_FLOATS];
    float* xs = lds;                   // [TB][LD]
    float* as_ = lds + TB * LD;        // [MKB][LD]
    float* bs = as_ + MKB * LD;        // [MKB][LD]

    const int tid = threadIdx.x;
    const int tm = tid & 15, ts = tid >> 4;
    const int b0 = blockIdx.x * TB;
    const int mk0 = blockIdx.y * MKB;

    // packed along d: element .x sums the even, .y the odd feature columns - both v_pk_fma_f32 operands of a term are then
    // natural register pairs (x[s][d, d+1], a[mk][d, d+1]) with no broadcast.  (Written with scalars, hipcc packs across
    // samples / components instead and spends 140 v_mov per 160 v_pk_fma on building the broadcast pairs.)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[SPT][MKT];
#pragma unroll
    for (int i = 0; i < SPT; ++i)
#pragma unroll
        for (int j = 0; j < MKT; ++j) acc2[i][j] = f32x2{0.f, 0.f};

    float4 px[XITEMS], pa[PITEMS], pb[PITEMS];
    // rows past B / past the last component are clamped to a valid row: their results are never written
    auto gload = [&](int d0) {
        const bool full = VEC && (d0 + DC <= dhi);                 // wave-uniform: unguarded 16-B loads
#pragma unroll
        for (int q = 0; q < XITEMS; ++q) {
            const int e = q * 256 + tid, row = e >> 3, c4 = (e & 7) * 4;
            const float* rp = x + (int64_t)(KEYED ? rows[min(row, nvalid - 1)] : min(b0 + row, B - 1)) * xbs;
#ifndef CF_GMM_NT
#define CF_GMM_NT 1                  // x is read once per component block: non-temporal, the parameter rows every workgroup re-reads stay cached
#endif
            if (CF_GMM_NT && full) {
                typedef float f32x4nt __attribute__((ext_vector_type(4)));
                const f32x4nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(rp + d0 + c4));
                px[q] = make_float4(v[0], v[1], v[2], v[3]);
            } else
            px[q] = full ? *reinterpret_cast<const float4*>(rp + d0 + c4) : ld4_tail<VEC>(rp, d0 + c4, dhi);
        }
#pragma unroll
        for (int q = 0; q < PITEMS; ++q) {
            const int e = q * 256 + tid, row = e >> 3, c4 = (e & 7) * 4;
            const int64_t off = (int64_t)min(mk0 + row, MKtot - 1) * D;
            pa[q] = full ? *reinterpret_cast<const float4*>(a + off + d0 + c4) : ld4_tail<VEC>(a + off, d0 + c4, dhi);
            pb[q] = full ? *reinterpret_cast<const float4*>(nm + off + d0 + c4) : ld4_tail<VEC>(nm + off, d0 + c4, dhi);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < XITEMS; ++q) {
            const int e = q * 256 + tid, row = e >> 3, c4 = (e & 7) * 4;
            *reinterpret_cast<float4*>(&xs[row * LD + c4]) = px[q];
        }
#pragma unroll
        for (int q = 0; q < PITEMS; ++q) {
            const int e = q * 256 + tid, row = e >> 3, c4 = (e & 7) * 4;
            if (row < MKB) {
                *reinterpret_cast<float4*>(&as_[row * LD + c4]) = pa[q];
                *reinterpret_cast<float4*>(&bs[row * LD + c4]) = pb[q];
            }
        }
    };

    gload(dlo);
    for (int d0 = dlo; d0 < dhi; d0 += DC) {
        __syncthreads();                      // previous chunk fully consumed
        lstore();
        __syncthreads();
        if (d0 + DC < dhi) gload(d0 + DC);    // prefetch the next chunk behind the FMAs
#pragma unroll 2
        for (int dd = 0; dd < DC; dd += 2) {
            // 8-byte LDS reads: same bytes per LDS cycle as 16-byte ones, half the operand registers
            f32x2 xv[SPT], av[MKT], bv[MKT];
#pragma unroll
            for (int i = 0; i < SPT; ++i) xv[i] = *reinterpret_cast<const f32x2*>(&xs[(ts + 16 * i) * LD + dd]);
#pragma unroll
            for (int j = 0; j < MKT; ++j) {
                av[j] = *reinterpret_cast<const f32x2*>(&as_[(tm + 16 * j) * LD + dd]);
                bv[j] = *reinterpret_cast<const f32x2*>(&bs[(tm + 16 * j) * LD + dd]);
            }
            // the reference's form (x - mu) / sigma (gaussian.py:142-161): the difference of two close values is exact,
            // whereas fma(x, 1/sigma, -mu/sigma) carries the rounding of mu/sigma - an absolute error of ulp(mu/sigma) on a
            // term of size O(1) once a mixture is fitted with small sigma (SMAP "extreme" fixtures: 1.7e-5 bits/dim).  One
            // more packed VALU per term than the fma form.  Written stage by stage over the MKT components of a sample: left
            // to itself hipcc chains add -> mul -> fma of ONE term through one temporary, with a wait state between every
            // pair of dependent packed instructions (118 s_nop per 240 v_pk in the loop, two waves per SIMD to cover them):
            // 1.70 -> 1.47 ms per 262 144 samples.  (The 2-way LDS bank conflict of the parameter reads - lanes tm and tm + 8
            // lie 9 x 32 banks apart - was removed by rotating the column order per lane half and measured: no gain.)
#pragma unroll
            for (int i = 0; i < SPT; ++i) {
                f32x2 t[MKT];
#pragma unroll
                for (int j = 0; j < MKT; ++j) t[j] = xv[i] + bv[j];
#pragma unroll
                for (int j = 0; j < MKT; ++j) t[j] = t[j] * av[j];
#pragma unroll
                for (int j = 0; j < MKT; ++j) acc2[i][j] = __builtin_elementwise_fma(t[j], t[j], acc2[i][j]);
            }
        }
    }
    float acc[SPT][MKT];
#pragma unroll
    for (int i = 0; i < SPT; ++i)
#pragma unroll
        for (int j = 0; j < MKT; ++j) acc[i][j] = acc2[i][j].x + acc2[i][j].y;

    if (SPLIT) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const bool live = KEYED ? ts + 16 * i < nvalid : b0 + ts + 16 * i < B;
            const int b = KEYED ? (live ? rows[ts + 16 * i] : 0) : b0 + ts + 16 * i;
#pragma unroll
            for (int j = 0; j < MKT; ++j) {
                const int mk = mk0 + tm + 16 * j;
                if (live && mk < MKtot) qb[(int64_t)b * MKtot + mk] = acc[i][j];
            }
        }
        return;
    }

    // epilogue: q -> LDS, then logsumexp over the K components of every (sample, mixture)
    __syncthreads();
    float* qs = lds;                                  // [TB][MKB+1]
#pragma unroll
    for (int i = 0; i < SPT; ++i)
#pragma unroll
        for (int j = 0; j < MKT; ++j) qs[(ts + 16 * i) * (MKB + 1) + tm + 16 * j] = acc[i][j];
    __syncthreads();
    const int mloc = MKB / K;                         // mixtures fully inside this component block
    for (int e = tid; e < TB * mloc; e += 256) {
        const int s = e / mloc, ml = e - s * mloc;
        const int b = b0 + s, mk = mk0 + ml * K;
        if (b >= B || mk >= MKtot) continue;
        const float* qrow = &qs[s * (MKB + 1) + ml * K];
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, cst[mk + k] - 0.5f * qrow[k]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(cst[mk + k] - 0.5f * qrow[k] - mx);
        const float r = mx + logf(sum);
        float* o = out + (int64_t)b * Mtot + (mk / K);
        *o = accumulate ? *o + r : r;
    }
}

template <int MKT, bool VEC, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_gmm_logprob(const float* __restrict__ x, const float* __restrict__ a,
                                                     const float* __restrict__ nm, const float* __restrict__ cst,
                                                     float* __restrict__ out, float* __restrict__ qout,
                                                     int B, int MKtot, int K, int D, int dsplit,
                                                     int64_t xbs, int accumulate, int Mtot) {
    const int dlo = blockIdx.z * dsplit;
    gmm_logprob_body<MKT, VEC, SPLIT>(x, a, nm, cst, out, qout + (int64_t)blockIdx.z * B * MKtot, B, MKtot, K, D, dlo,
                                      min(D, dlo + dsplit), xbs, accumulate, Mtot);
}

// Small batches (the launch, not the arithmetic, is what a mixture costs there - 10.5 + 5.8 us per level at a batch of
// 256): the mixtures of ALL levels of a flow (the Split priors + the final prior) in one launch of the D-split form.
// blockIdx.z runs over the d-slices of level 0, then level 1, ...; one finishing kernel sums each level's partials,
// takes its logsumexp, adds the levels in order and the per-sample log-det (cf_logdet_combine folded in).
constexpr int GMM_MAX_LEVELS = 4;
struct GmmLevels {
    const float* x[GMM_MAX_LEVELS]; const float* a[GMM_MAX_LEVELS]; const float* nm[GMM_MAX_LEVELS];
    const float* cst[GMM_MAX_LEVELS];
    int64_t xbs[GMM_MAX_LEVELS];
    int D[GMM_MAX_LEVELS], dsplit[GMM_MAX_LEVELS], z0[GMM_MAX_LEVELS + 1];
    int n;
};

template <int MKT>
__global__ __launch_bounds__(256, 2) void k_gmm_logprob_levels(GmmLevels L, float* __restrict__ qout, int B, int MKtot, int K,
                                                            int Mtot) {
    int l = 0;
    while (l + 1 < L.n && (int)blockIdx.z >= L.z0[l + 1]) ++l;      // uniform
    const int dlo = ((int)blockIdx.z - L.z0[l]) * L.dsplit[l];
    gmm_logprob_body<MKT, true, true>(L.x[l], L.a[l], L.nm[l], nullptr, nullptr, qout + (int64_t)blockIdx.z * B * MKtot, B,
                                      MKtot, K, L.D[l], dlo, min(L.D[l], dlo + L.dsplit[l]), L.xbs[l], 0, Mtot);
}

// one thread per (b, component) as in k_gmm_finish; out[b, m] = (ldM[b, m] +) sum_levels logsumexp_k(...) (+ ld1[b])
__global__ __launch_bounds__(256) void k_gmm_finish_levels(const float* __restrict__ q, GmmLevels L,
                                                           const float* __restrict__ ldM, const float* __restrict__ ld1,
                                                           float* __restrict__ out, int B, int M, int K) {
    __shared__ float l[256];
    const int MK = M * K, spb = blockDim.x / MK;
    const int sl = threadIdx.x / MK, mk = threadIdx.x - sl * MK;
    const int b = blockIdx.x * spb + sl;
    const bool live = b < B && sl < spb;
    const bool head = live && mk % K == 0;
    const int64_t e = (int64_t)b * M + mk / K;
    float r = 0.f;
    bool first = true;
    if (head && ldM) { r = ldM[e]; first = false; }
    for (int lev = 0; lev < L.n; ++lev) {
        float s = 0.f;
        if (live) {
            const float* qp = q + (int64_t)b * MK + mk;
            const int64_t zs = (int64_t)B * MK;
            int z = L.z0[lev];
            const int zend = L.z0[lev + 1];
            for (; z + 8 <= zend; z += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = qp[(z + j) * zs];
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            for (; z < zend; ++z) s += qp[z * zs];
        }
        __syncthreads();
        l[threadIdx.x] = live ? L.cst[lev][mk] - 0.5f * s : 0.f;
        __syncthreads();
        if (head) {
            const float* lp = l + threadIdx.x;
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, lp[k]);
            float sum = 0.f;
            for (int k = 0; k < K; ++k) sum += expf(lp[k] - mx);
            const float v = mx + logf(sum);
            r = first ? v : r + v;
            first = false;
        }
    }
    if (head) out[e] = ld1 ? r + ld1[b] : r;
}

// Context-shifted mixtures whose shifts are embedding lookups (model.py:157,162: CatEmbeddings + EyeSampling): a sample's
// mixture is one of Um x Us parameter sets - means mu + cm[km], scales softplus(sG + cs[ks]) - chosen by its context.
// With the samples bucketed by (km, ks) a workgroup's 128 samples share their parameter rows, i.e. the register-tiled
// kernel above applies with its x rows gathered through an index list and its tables chosen per tile (the per-sample
// kernel of cf_context.hip reads one 1/sigma per term per sample from L2: 7x slower at saturating batches).
// tiles (T, 4) int32: [ks, km, first position in `order`, number of samples (0: unused tile)].
template <int MKT>
__global__ __launch_bounds__(256, 2) void k_gmm_logprob_keyed(const float* __restrict__ x, const float* __restrict__ a_tab,
                                                           const float* __restrict__ nm_tab, const int* __restrict__ tiles,
                                                           const int* __restrict__ order, float* __restrict__ qout, int B,
                                                           int MKtot, int K, int D, int dsplit, int64_t xbs, int Mtot) {
    const int4 tl = *reinterpret_cast<const int4*>(tiles + 4 * blockIdx.x);
    if (tl.w <= 0) return;                                           // uniform
    const int dlo = blockIdx.z * dsplit;
    const int64_t tab = (int64_t)MKtot * D;
    gmm_logprob_body<MKT, true, true, true>(x, a_tab + tl.x * tab, nm_tab + tl.y * tab, nullptr, nullptr,
                                            qout + (int64_t)blockIdx.z * B * MKtot, B, MKtot, K, D, dlo, min(D, dlo + dsplit),
                                            xbs, 0, Mtot, order + tl.z, tl.w);
}

// k_gmm_finish with the constant row chosen by the sample's scale key
__global__ __launch_bounds__(256) void k_gmm_finish_keyed(const float* __restrict__ q, const float* __restrict__ cst_tab,
                                                          const int* __restrict__ key, float* __restrict__ out, int B, int M,
                                                          int K, int nsplit, int accumulate) {
    __shared__ float l[256];
    const int MK = M * K, spb = blockDim.x / MK;
    const int sl = threadIdx.x / MK, mk = threadIdx.x - sl * MK;
    const int b = blockIdx.x * spb + sl;
    const bool live = b < B && sl < spb;
    float s = 0.f;
    if (live) {
        const float* qp = q + (int64_t)b * MK + mk;
        const int64_t zs = (int64_t)B * MK;
        for (int z = 0; z < nsplit; ++z) s += qp[z * zs];
    }
    l[threadIdx.x] = live ? cst_tab[(int64_t)key[b] * MK + mk] - 0.5f * s : 0.f;
    __syncthreads();
    if (live && mk % K == 0) {
        const float* lp = l + threadIdx.x;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, lp[k]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(lp[k] - mx);
        const float r = mx + logf(sum);
        const int64_t e = (int64_t)b * M + mk / K;
        out[e] = accumulate ? out[e] + r : r;
    }
}

// finishing kernel for the D-split form: sum partials, logsumexp.  One thread per (b, component): the partial sums of
// one sample are read as runs of MK consecutive floats; the K log-joints of a mixture meet in LDS.  A block holds
// SPB = blockDim / MK whole samples (small batches are latency-bound: the one-thread-per-(b, m) form with its
// 2 K nsplit strided loads per thread took longer than the main kernel).
__global__ __launch_bounds__(256) void k_gmm_finish(const float* __restrict__ q, const float* __restrict__ cst,
                                                    float* __restrict__ out, int B, int M, int K, int nsplit,
                                                    int accumulate) {
    __shared__ float l[256];
    const int MK = M * K, spb = blockDim.x / MK;
    const int sl = threadIdx.x / MK, mk = threadIdx.x - sl * MK;
    const int b = blockIdx.x * spb + sl;
    const bool live = b < B && sl < spb;
    float s = 0.f;
    if (live) {
        const float* qp = q + (int64_t)b * MK + mk;
        const int64_t zs = (int64_t)B * MK;
        int z = 0;
        for (; z + 8 <= nsplit; z += 8) {              // 8 independent loads in flight (the partials sit 4 B MK bytes apart)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = qp[(z + j) * zs];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; z < nsplit; ++z) s += qp[z * zs];
    }
    l[threadIdx.x] = live ? cst[mk] - 0.5f * s : 0.f;
    __syncthreads();
    if (live && mk % K == 0) {
        const float* lp = l + threadIdx.x;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, lp[k]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(lp[k] - mx);
        const float r = mx + logf(sum);
        const int64_t e = (int64_t)b * M + mk / K;
        out[e] = accumulate ? out[e] + r : r;
    }
}

// same for mixtures with more than 256 components in all: one thread per (b, m)
__global__ __launch_bounds__(256) void k_gmm_finish_bm(const float* __restrict__ q, const float* __restrict__ cst,
                                                       float* __restrict__ out, int B, int M, int K, int nsplit,
                                                       int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * M) return;
    const int b = (int)(e / M), m = (int)(e - (int64_t)b * M);
    const int MK = M * K;
    float lj[16];
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += q[((int64_t)z * B + b) * MK + m * K + k];
        lj[k] = cst[m * K + k] - 0.5f * s;
        mx = fmaxf(mx, lj[k]);
    }
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += expf(lj[k] - mx);
    const float r = mx + logf(sum);
    out[e] = accumulate ? out[e] + r : r;
}

// finishing kernel of the backward: sum the D-split partials, r[b, mk] = softmax_k(cst - q/2)[mk] * g[b, m]
// (responsibilities times the upstream gradient).  One thread per (b, m, k) as in k_gmm_finish - the partial sums are
// nsplit loads per thread, 8 in flight, instead of K nsplit serial ones (small batches: 53 -> 8 us at B = 256); the K
// log-joints of a mixture meet in LDS and every thread normalises its own component.
__global__ __launch_bounds__(256) void k_gmm_resp_finish_mk(const float* __restrict__ q, const float* __restrict__ cst,
                                                            const float* __restrict__ g, float* __restrict__ r, int B, int M,
                                                            int K, int nsplit) {
    __shared__ float l[256];
    const int MK = M * K, spb = blockDim.x / MK;
    const int sl = threadIdx.x / MK, mk = threadIdx.x - sl * MK;
    const int b = blockIdx.x * spb + sl;
    const bool live = b < B && sl < spb;
    float s = 0.f;
    if (live) {
        const float* qp = q + (int64_t)b * MK + mk;
        const int64_t zs = (int64_t)B * MK;
        int z = 0;
        for (; z + 8 <= nsplit; z += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = qp[(z + j) * zs];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; z < nsplit; ++z) s += qp[z * zs];
    }
    const float lj = live ? cst[mk] - 0.5f * s : 0.f;
    l[threadIdx.x] = lj;
    __syncthreads();
    if (live) {
        const int m = mk / K;
        const float* lp = l + threadIdx.x - (mk - m * K);
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, lp[k]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(lp[k] - mx);
        r[(int64_t)b * MK + mk] = expf(lj - mx) * (g[(int64_t)b * M + m] / sum);
    }
}

// same for mixtures with more than 256 components in all: one thread per (b, m)
__global__ __launch_bounds__(256) void k_gmm_resp_finish(const float* __restrict__ q, const float* __restrict__ cst,
                                                         const float* __restrict__ g, float* __restrict__ r, int B, int M,
                                                         int K, int nsplit) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * M) return;
    const int b = (int)(e / M), m = (int)(e - (int64_t)b * M);
    const int MK = M * K;
    float l[16];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < K) {
            float s = 0.f;
            for (int z = 0; z < nsplit; ++z) s += q[((int64_t)z * B + b) * MK + m * K + k];
            l[k] = cst[m * K + k] - 0.5f * s;
            mx = fmaxf(mx, l[k]);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < K) { l[k] = expf(l[k] - mx); sum += l[k]; }
    const float sc = g[e] / sum;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < K) r[(int64_t)b * MK + m * K + k] = l[k] * sc;
}

// ---- elementwise pieces of the mixture backward (autograd.py gmm_backward), one launch each instead of a chain of
// parameter-sized torch kernels ------------------------------------------------------------------------------------
// A2 = a a, AB = a a nm (nm = -mu): right-hand sides of the two (B x MK) x (MK x D) products of d/dx
// TR: outputs transposed, (D, MK) - the Wt operand of cf_linear for G = r A  (coalesced writes, strided L2-resident reads)
template <bool TR>
__global__ __launch_bounds__(256) void k_gmm_bwd_coeffs(const float* __restrict__ a, const float* __restrict__ nm,
                                                        float* __restrict__ A2, float* __restrict__ AB, int64_t n, int MK, int D) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t src = TR ? (e % MK) * D + e / MK : e;
        const float av = a[src];
        A2[e] = av * av;
        AB[e] = av * av * nm[src];
    }
}
// gx[b, d] = -(x[b, d] G1[b, d] + G2[b, d]),  G1 = r A2, G2 = r AB
__global__ __launch_bounds__(256) void k_gmm_bwd_gx(const float* __restrict__ x, const float* __restrict__ G1,
                                                    const float* __restrict__ G2, float* __restrict__ gx, int D, int64_t xbs,
                                                    int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / D;
        gx[e] = -fmaf(x[b * xbs + (e - b * D)], G1[e], G2[e]);
    }
}
// parameter gradients from the batch sums S0 (MK) = sum_b r, S1 = r^T x, S2 = r^T x^2 (MK x D):
//   t = a (x + nm), nm = -mu;  g_mu = a sum_b r t;  g_sG = a (sum_b r t^2 - S0) sigmoid(sG)      (softplus' = sigmoid)
// ... and, when gw is given, the mixture-weight gradient  g_wG[m][k] = S0[mk] - gcol[m] softmax_k(wG[m])[k]   (the log-weights
// enter through log softmax; gcol[m] = sum_b g[b][m]): M K <= a few hundred values, the first workgroup's job
__global__ __launch_bounds__(256) void k_gmm_bwd_params(const float* __restrict__ a, const float* __restrict__ nm,
                                                        const float* __restrict__ sG, const float* __restrict__ S0,
                                                        const float* __restrict__ S1, const float* __restrict__ S2,
                                                        float* __restrict__ gmu, float* __restrict__ gsig, int D, int64_t n,
                                                        const float* __restrict__ wG = nullptr, const float* __restrict__ gcol = nullptr,
                                                        float* __restrict__ gw = nullptr, int M = 0, int K = 0) {
    if (gw != nullptr && blockIdx.x == 0) {
        for (int mk = threadIdx.x; mk < M * K; mk += 256) {
            const int m = mk / K;
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, wG[m * K + k]);
            float sum = 0.f;
            for (int k = 0; k < K; ++k) sum += expf(wG[m * K + k] - mx);
            gw[mk] = S0[mk] - gcol[m] * (expf(wG[mk] - mx) / sum);
        }
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float av = a[e], bv = nm[e], s0 = S0[e / D], s1 = S1[e], s2 = S2[e];
        const float rt = av * (s1 + bv * s0);
        const float rt2 = av * av * (s2 + 2.0f * bv * s1 + bv * bv * s0);
        gmu[e] = av * rt;
        gsig[e] = av * (rt2 - s0) / (1.0f + expf(-sG[e]));
    }
}

// prior sampling (gaussian.py:163-169): x[n,:] = mG[row_n,:] + softplus(sG[row_n,:]) * eps[n,:], row_n = m*K + k_n
__global__ __launch_bounds__(256) void k_gmm_sample(const float* __restrict__ mG, const float* __restrict__ sG,
                                                    const int64_t* __restrict__ rows, const float* __restrict__ eps,
                                                    float* __restrict__ out, int D, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / D;
        const int d = (int)(i - n * D);
        const int64_t r = rows[n] * D + d;
        out[i] = fmaf(softplus_ref(sG[r]), eps[i], mG[r]);
    }
}

// ... 16 bytes per lane, the sample index by a 32-bit division per FOUR elements (the scalar form above spends a 64-bit division per
// element: 142 us for 16 384 x 2 048 values, 0.95 TB/s)
__global__ __launch_bounds__(256) void k_gmm_sample4(const float* __restrict__ mG, const float* __restrict__ sG,
                                                     const int64_t* __restrict__ rows, const float* __restrict__ eps,
                                                     float* __restrict__ out, int D4, int total4) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const int n = i / D4, d4 = i - n * D4;
        const int64_t r = rows[n] * D4 + d4;
        const float4 m = reinterpret_cast<const float4*>(mG)[r], sg = reinterpret_cast<const float4*>(sG)[r];
        const float4 e = reinterpret_cast<const float4*>(eps)[i];
        float4 o;
        o.x = fmaf(softplus_ref(sg.x), e.x, m.x); o.y = fmaf(softplus_ref(sg.y), e.y, m.y);
        o.z = fmaf(softplus_ref(sg.z), e.z, m.z); o.w = fmaf(softplus_ref(sg.w), e.w, m.w);
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// D-split heuristic: enough workgroups to cover the chip (~2 per CU), chunks stay multiples of DC
// ---- parameter sums of the mixture backward: S0[mk] = sum_b r[b][mk], S1[mk][d] = sum_b r[b][mk] x[b][d],
// S2[mk][d] = sum_b r[b][mk] x[b][d]^2 - one (MK x B)(B x D) product with two right-hand sides.  MK = 80 rows are five
// 16-row tiles of v_mfma_f32_16x16x4_f32 exactly (no padding rows); a WAVE owns 32 columns of d and one slice of the batch:
// 20 accumulator tiles (5 row tiles x 2 column tiles x {x, x^2}) in 80 registers, per k-step (4 samples) 5 A and 2 B
// dword loads straight from global memory in the MFMA layout for 20 MFMAs - no LDS, no barrier.  Partials per batch slice
// are summed in slice order by k_gmm_sums_reduce (no float atomics).  (cf_linear_wgrad, built for K, N <= a few hundred,
// took 13 launches over the 1536 columns and re-read r in each: 1.45 ms per training step for the three priors.)
constexpr int GS_RT = 5;             // row tiles: MK = 80
template <int RT>
__global__ __launch_bounds__(256) void k_gmm_sums(const float* __restrict__ x, const float* __restrict__ r,
                                                  float* __restrict__ part, int B, int D, int64_t xbs, int nsp, int spb) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int MK = 16 * RT;
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);           // (column block of 32, batch slice)
    const int nct = D / 32;
    if (unit >= nct * nsp) return;                                   // whole wave
    const int cb = unit % nct, sp = unit / nct;
    const int b0 = sp * spb, b1 = min(B, b0 + spb);
    f32x4 acc[RT][2][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[rt][c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s0[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) s0[rt] = 0.f;
    // uniform slice bases + 32-bit lane offsets that advance by a constant per k-step (4 samples): the loop's address
    // arithmetic is two v_add - every VALU instruction here is time taken from the matrix pipe
    const float* xs = x + (int64_t)b0 * xbs + 32 * cb;
    const float* rsl = r + (int64_t)b0 * MK;
    const int nb = b1 - b0, xstep = 4 * (int)xbs;
    int xo = kk * (int)xbs + n, ro = kk * MK + n;
    auto load = [&](float (&av)[RT], float (&bv)[2]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) av[rt] = rsl[ro + 16 * rt];
#pragma unroll
        for (int c = 0; c < 2; ++c) bv[c] = xs[xo + 16 * c];
        xo += xstep; ro += 4 * MK;
    };
    auto mma = [&](const float (&av)[RT], const float (&bv)[2]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            s0[rt] += av[rt];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[rt][c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv[c], acc[rt][c][0], 0, 0, 0);
                acc[rt][c][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv[c] * bv[c], acc[rt][c][1], 0, 0, 0);
            }
        }
    };
    float a0[RT], x0[2], a1[RT], x1[2];
    const int full = nb / 8;                                         // trips of two whole k-steps
    if (full > 0) {
        load(a0, x0);
        for (int it = 0; it < full; ++it) {
            load(a1, x1);
            mma(a0, x0);
            if (it + 1 < full) load(a0, x0);
            mma(a1, x1);
        }
    }
    for (int bb = 8 * full; bb < nb; bb += 4) {                      // ragged tail: samples past the slice contribute r = 0
        const bool ok = bb + kk < nb;
        const int rr = ok ? ro : n, xx = ok ? xo : n;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { const float v = rsl[rr + 16 * rt]; a0[rt] = ok ? v : 0.f; }
#pragma unroll
        for (int c = 0; c < 2; ++c) x0[c] = xs[xx + 16 * c];
        xo += xstep; ro += 4 * MK;
        mma(a0, x0);
    }
    // partial layout per slice: [S1 (MK, D) | S2 (MK, D) | S0 (MK)]
    float* ps = part + (int64_t)sp * (2 * (int64_t)MK * D + MK);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ps[(int64_t)q * MK * D + (int64_t)(16 * rt + 4 * kk + j) * D + 32 * cb + 16 * c + n] = acc[rt][c][q][j];
    if (cb == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float v = s0[rt];                                        // lanes (n, kk): sum over the 4 kk groups
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kk == 0) ps[2 * (int64_t)MK * D + 16 * rt + n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_gmm_sums_reduce(const float* __restrict__ part, float* __restrict__ S0,
                                                         float* __restrict__ S1, float* __restrict__ S2, int MK, int D, int nsp) {
    const int64_t per = 2 * (int64_t)MK * D + MK;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    float v = 0.f;
    for (int sp = 0; sp < nsp; ++sp) v += part[sp * per + e];
    if (e < (int64_t)MK * D) S1[e] = v;
    else if (e < 2 * (int64_t)MK * D) S2[e - (int64_t)MK * D] = v;
    else S0[e - 2 * (int64_t)MK * D] = v;
}

// batch slices: about two waves per SIMD over the chip, at least 64 samples each, a multiple of 8
void gmm_sums_split(int B, int D, int& nsp, int& spb) {
    const int nct = D / 32;
    int want = (2048 + nct - 1) / nct;
    if (want < 1) want = 1;
    spb = (B + want - 1) / want;
    if (spb < 64) spb = 64;
    spb = (spb + 7) / 8 * 8;
    nsp = (B + spb - 1) / spb;
}

int choose_nsplit(int B, int MK, int D) {
    const int mkb = MK <= 16 ? 16 : 80;
    const int64_t base = (int64_t)((B + TB - 1) / TB) * ((MK + mkb - 1) / mkb);
    int ns = 1;
    while (base * ns < 512 && ns < 64 && D / (ns * 2) >= DC) ns *= 2;
    return ns;
}

}  // namespace

extern "C" {

int cf_gmm_prepare(const float* mG, const float* sG, const float* wG, float* a, float* nm, float* cst, int M, int K,
                   int D, cf_stream_t stream) {
    CF_REQUIRE(mG && sG && wG && a && nm && cst && M > 0 && K > 0 && D > 0);
    k_gmm_prepare<<<dim3(M * K), dim3(256), 0, cf_s(stream)>>>(mG, sG, wG, a, nm, cst, K, D);
    CF_LAUNCH_CHECK();
    return 0;
}

// q[b, mk] = sum_d ((x[b,d] + nm[mk,d]) * a[mk,d])^2 — the quadratic forms alone (used by the backward pass to rebuild
// the component responsibilities)
int cf_gmm_quad(const float* x, const float* a, const float* nm, float* q, int B, int M, int K, int D, int64_t x_bstride,
                cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && a && nm && q && M > 0 && K > 0 && K <= 16 && D > 0 && x_bstride >= D);
    const int MK = M * K;
    const bool small = MK <= 16;
    const bool vec = (D % 4 == 0) && (x_bstride % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(a) & 15) == 0) && ((reinterpret_cast<uintptr_t>(nm) & 15) == 0);
    const int mkb = small ? 16 : 80;
    dim3 grid((B + TB - 1) / TB, (MK + mkb - 1) / mkb, 1);
#define CF_GO(MKT, V) k_gmm_logprob<MKT, V, true><<<grid, dim3(256), 0, cf_s(stream)>>>(x, a, nm, nullptr, nullptr, q, B, MK, K, D, D, x_bstride, 0, M)
    if (small) { if (vec) CF_GO(1, true); else CF_GO(1, false); }
    else       { if (vec) CF_GO(5, true); else CF_GO(5, false); }
#undef CF_GO
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_sample(const float* mG, const float* sG, const int64_t* rows, const float* eps, float* out, int N, int D,
                  cf_stream_t stream) {
    if (N == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(mG && sG && rows && eps && out && N >= 0 && D > 0);
    const int64_t total = (int64_t)N * D;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const bool vec = D % 4 == 0 && total / 4 < (1ll << 31) &&
                     ((reinterpret_cast<uintptr_t>(mG) | reinterpret_cast<uintptr_t>(sG) | reinterpret_cast<uintptr_t>(eps) |
                       reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (vec) {
        int64_t b4 = (total / 4 + 255) / 256;
        if (b4 > 16384) b4 = 16384;
        k_gmm_sample4<<<dim3((unsigned)b4), dim3(256), 0, cf_s(stream)>>>(mG, sG, rows, eps, out, D / 4, (int)(total / 4));
    } else
        k_gmm_sample<<<dim3((unsigned)blocks), dim3(256), 0, cf_s(stream)>>>(mG, sG, rows, eps, out, D, total);
    CF_LAUNCH_CHECK();
    return 0;
}

int64_t cf_gmm_ws_bytes(int B, int M, int K, int D) {
    const int ns = choose_nsplit(B, M * K, D);
    return ns > 1 ? (int64_t)ns * B * M * K * sizeof(float) : 0;
}

int cf_gmm_logprob(const float* x, const float* a, const float* nm, const float* cst, float* out, void* ws,
                   int B, int M, int K, int D, int64_t x_bstride, int accumulate, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && a && nm && cst && out && B >= 0 && M > 0 && K > 0 && K <= 16 && D > 0 && x_bstride >= D);
    if (B == 0) return 0;
    const int MK = M * K;
    const bool small = MK <= 16;
    CF_REQUIRE(small ? (16 % K == 0 || M == 1) : (80 % K == 0));
    int ns = ws ? choose_nsplit(B, MK, D) : 1;
    int dsplit = D;
    if (ns > 1) { dsplit = ((D + ns - 1) / ns + DC - 1) / DC * DC; ns = (D + dsplit - 1) / dsplit; }
    const bool vec = (D % 4 == 0) && (x_bstride % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(a) & 15) == 0) && ((reinterpret_cast<uintptr_t>(nm) & 15) == 0) &&
                     (dsplit % 4 == 0);
    const int mkb = small ? 16 : 80;
    dim3 grid((B + TB - 1) / TB, (MK + mkb - 1) / mkb, ns);
    float* q = (float*)ws;
#define CF_GO(MKT, V, S) k_gmm_logprob<MKT, V, S><<<grid, dim3(256), 0, cf_s(stream)>>>(x, a, nm, cst, out, q, B, MK, K, D, dsplit, x_bstride, accumulate, M)
    if (ns > 1) {
        if (small) { if (vec) CF_GO(1, true, true); else CF_GO(1, false, true); }
        else       { if (vec) CF_GO(5, true, true); else CF_GO(5, false, true); }
        const int spb = 256 / MK;
        if (spb >= 1) {
            k_gmm_finish<<<dim3((unsigned)((B + spb - 1) / spb)), dim3(spb * MK), 0, cf_s(stream)>>>(q, cst, out, B, M, K, ns, accumulate);
        } else {
            const int64_t n = (int64_t)B * M;
            k_gmm_finish_bm<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cf_s(stream)>>>(q, cst, out, B, M, K, ns, accumulate);
        }
    } else {
        if (small) { if (vec) CF_GO(1, true, false); else CF_GO(1, false, false); }
        else       { if (vec) CF_GO(5, true, false); else CF_GO(5, false, false); }
    }
#undef CF_GO
    CF_LAUNCH_CHECK();
    return 0;
}

// all mixtures of a flow in two launches (small batches).  Level l: x[l] (B rows of D[l] floats, row stride x_bstride[l]),
// its prepared tables a[l], nm[l] (M*K, D[l]) and cst[l] (M*K).  out[b, m] = (ldM[b, m] +) sum_l log p_l(x_l[b] | m) (+ ld1[b]);
// ldM and ld1 may be null.  The levels are summed in order, each with the D split cf_gmm_logprob would choose: the result
// equals the chain of cf_gmm_logprob(..., accumulate) calls + cf_logdet_combine bit for bit.
// Requires n <= 4, M*K <= 256, D[l] % 4 == 0, x_bstride[l] % 4 == 0, 16-byte aligned x[l] / a[l] / nm[l].
int64_t cf_gmm_levels_ws_bytes(int n, const int* D, int B, int M, int K) {
    if (n < 1 || n > GMM_MAX_LEVELS || !D) return -1;
    int64_t z = 0;
    for (int l = 0; l < n; ++l) {
        int ns = choose_nsplit(B, M * K, D[l]);
        if (ns > 1) { const int ds = ((D[l] + ns - 1) / ns + DC - 1) / DC * DC; ns = (D[l] + ds - 1) / ds; }
        z += ns;
    }
    return z * B * M * K * (int64_t)sizeof(float);
}

int cf_gmm_logprob_levels(int n, const float* const* x, const float* const* a, const float* const* nm,
                          const float* const* cst, const int* D, const int64_t* x_bstride, const float* ldM,
                          const float* ld1, float* out, void* ws, int B, int M, int K, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(n >= 1 && n <= GMM_MAX_LEVELS && x && a && nm && cst && D && x_bstride && out && ws && B >= 0 && M > 0 &&
               K > 0 && K <= 16 && M * K <= 256);
    const int MK = M * K;
    const bool small = MK <= 16;
    CF_REQUIRE(small ? (16 % K == 0 || M == 1) : (80 % K == 0));
    GmmLevels L;
    L.n = n;
    int z = 0;
    for (int l = 0; l < GMM_MAX_LEVELS; ++l) {
        const int s = l < n ? l : n - 1;
        CF_REQUIRE(x[s] && a[s] && nm[s] && cst[s] && D[s] > 0 && D[s] % 4 == 0 && x_bstride[s] >= D[s] && x_bstride[s] % 4 == 0);
        CF_REQUIRE(((reinterpret_cast<uintptr_t>(x[s]) | reinterpret_cast<uintptr_t>(a[s]) | reinterpret_cast<uintptr_t>(nm[s])) & 15) == 0);
        L.x[l] = x[s]; L.a[l] = a[s]; L.nm[l] = nm[s]; L.cst[l] = cst[s]; L.xbs[l] = x_bstride[s]; L.D[l] = D[s];
        int ns = choose_nsplit(B, MK, D[s]), ds = D[s];
        if (ns > 1) { ds = ((D[s] + ns - 1) / ns + DC - 1) / DC * DC; ns = (D[s] + ds - 1) / ds; }
        L.dsplit[l] = ds;
        L.z0[l] = z;
        if (l < n) z += ns;
    }
    L.z0[GMM_MAX_LEVELS] = z;
    for (int l = n; l < GMM_MAX_LEVELS; ++l) L.z0[l] = z;
    const int mkb = small ? 16 : 80;
    dim3 grid((B + TB - 1) / TB, (MK + mkb - 1) / mkb, z);
    float* q = (float*)ws;
    if (small) k_gmm_logprob_levels<1><<<grid, dim3(256), 0, cf_s(stream)>>>(L, q, B, MK, K, M);
    else k_gmm_logprob_levels<5><<<grid, dim3(256), 0, cf_s(stream)>>>(L, q, B, MK, K, M);
    const int spb = 256 / MK;
    k_gmm_finish_levels<<<dim3((unsigned)((B + spb - 1) / spb)), dim3(spb * MK), 0, cf_s(stream)>>>(q, L, ldM, ld1, out, B, M, K);
    CF_LAUNCH_CHECK();
    return 0;
}

// Mixtures with keyed parameter sets (the embedding-lookup context nets of the specialist flows).  x (B rows of D floats -
// D counts every feature, channels x pixels), a_tab (Us, M*K, D) = 1/sigma per scale key, nm_tab (Um, M*K, D) = -(mu + shift)
// per mean key, cst_tab (Us, M*K), key_s (B): every sample's scale key.  order (B): the sample indices grouped by
// (scale key, mean key); tiles (T, 4) int32 rows [ks, km, first position in order, count <= 128 (0 = unused tile)].
// out[b, m] (+)= logsumexp_k(cst_tab[key_s[b]] - 1/2 sum_d ((x + nm) a)^2).  M*K <= 256 and a multiple of 80 / K as for
// cf_gmm_logprob with M*K > 16; D % 4 == 0, x_bstride % 4 == 0, 16-byte aligned x and tables.
static int keyed_nsplit(int T, int MK, int D) {
    const int64_t base = (int64_t)T * ((MK + 79) / 80);
    int ns = 1;
    while (base * ns < 512 && ns < 64 && D / (ns * 2) >= DC) ns *= 2;
    return ns;
}

int64_t cf_gmm_keyed_ws_bytes(int T, int B, int M, int K, int D) {
    return (int64_t)keyed_nsplit(T, M * K, D) * B * M * K * (int64_t)sizeof(float);
}

int cf_gmm_logprob_keyed(const float* x, const float* a_tab, const float* nm_tab, const float* cst_tab, const int* key_s,
                         const int* tiles, const int* order, float* out, void* ws, int T, int B, int M, int K, int D,
                         int64_t x_bstride, int accumulate, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && a_tab && nm_tab && cst_tab && key_s && tiles && order && out && ws && T > 0 && B > 0 && M > 0 && K > 0 &&
               K <= 16 && M * K > 16 && M * K <= 256 && 80 % K == 0 && D > 0 && D % 4 == 0 && x_bstride >= D && x_bstride % 4 == 0);
    CF_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(a_tab) | reinterpret_cast<uintptr_t>(nm_tab) |
                 reinterpret_cast<uintptr_t>(tiles)) & 15) == 0);
    const int MK = M * K;
    int ns = keyed_nsplit(T, MK, D), dsplit = D;
    if (ns > 1) { dsplit = ((D + ns - 1) / ns + DC - 1) / DC * DC; ns = (D + dsplit - 1) / dsplit; }
    float* q = (float*)ws;
    k_gmm_logprob_keyed<5><<<dim3(T, (MK + 79) / 80, ns), dim3(256), 0, cf_s(stream)>>>(x, a_tab, nm_tab, tiles, order, q, B, MK,
                                                                                       K, D, dsplit, x_bstride, M);
    const int spb = 256 / MK;
    k_gmm_finish_keyed<<<dim3((unsigned)((B + spb - 1) / spb)), dim3(spb * MK), 0, cf_s(stream)>>>(q, cst_tab, key_s, out, B, M, K,
                                                                                                 ns, accumulate);
    CF_LAUNCH_CHECK();
    return 0;
}

// backward of the mixture prior, first half: r (B, M*K) = responsibilities x upstream gradient g (B, M), D split over
// blockIdx.z when the batch alone does not fill the chip.  ws: cf_gmm_resp_ws_bytes(...) bytes.
int64_t cf_gmm_resp_ws_bytes(int B, int M, int K, int D) {
    const int ns = choose_nsplit(B, M * K, D);
    return (int64_t)ns * B * M * K * sizeof(float);
}

int cf_gmm_resp(const float* x, const float* a, const float* nm, const float* cst, const float* g, float* r, void* ws, int B,
                int M, int K, int D, int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && a && nm && cst && g && r && ws && M > 0 && K > 0 && K <= 16 && D > 0 && x_bstride >= D);
    const int MK = M * K;
    const bool small = MK <= 16;
    int ns = choose_nsplit(B, MK, D);
    int dsplit = D;
    if (ns > 1) { dsplit = ((D + ns - 1) / ns + DC - 1) / DC * DC; ns = (D + dsplit - 1) / dsplit; }
    const bool vec = (D % 4 == 0) && (x_bstride % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(a) & 15) == 0) && ((reinterpret_cast<uintptr_t>(nm) & 15) == 0) &&
                     (dsplit % 4 == 0);
    const int mkb = small ? 16 : 80;
    dim3 grid((B + TB - 1) / TB, (MK + mkb - 1) / mkb, ns);
    float* q = (float*)ws;
#define CF_GO(MKT, V) k_gmm_logprob<MKT, V, true><<<grid, dim3(256), 0, cf_s(stream)>>>(x, a, nm, nullptr, nullptr, q, B, MK, K, D, dsplit, x_bstride, 0, M)
    if (small) { if (vec) CF_GO(1, true); else CF_GO(1, false); }
    else       { if (vec) CF_GO(5, true); else CF_GO(5, false); }
#undef CF_GO
    if (MK <= 256) {
        const int spb = 256 / MK;
        k_gmm_resp_finish_mk<<<dim3((unsigned)((B + spb - 1) / spb)), dim3(256), 0, cf_s(stream)>>>(q, cst, g, r, B, M, K, ns);
    } else {
        const int64_t n = (int64_t)B * M;
        k_gmm_resp_finish<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cf_s(stream)>>>(q, cst, g, r, B, M, K, ns);
    }
    CF_LAUNCH_CHECK();
    return 0;
}

static unsigned gmm_ew_blocks(int64_t n) { const int64_t b = (n + 255) / 256; return (unsigned)(b > 8192 ? 8192 : (b > 0 ? b : 1)); }

int cf_gmm_bwd_coeffs(const float* a, const float* nm, float* A2, float* AB, int MK, int D, int transposed,
                      cf_stream_t stream) {
    CF_REQUIRE(a && nm && A2 && AB && MK > 0 && D > 0);
    const int64_t n = (int64_t)MK * D;
    if (transposed) k_gmm_bwd_coeffs<true><<<dim3(gmm_ew_blocks(n)), dim3(256), 0, cf_s(stream)>>>(a, nm, A2, AB, n, MK, D);
    else k_gmm_bwd_coeffs<false><<<dim3(gmm_ew_blocks(n)), dim3(256), 0, cf_s(stream)>>>(a, nm, A2, AB, n, MK, D);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_bwd_gx(const float* x, const float* G1, const float* G2, float* gx, int B, int D, int64_t x_bstride,
                  cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && G1 && G2 && gx && D > 0 && x_bstride >= D);
    const int64_t n = (int64_t)B * D;
    k_gmm_bwd_gx<<<dim3(gmm_ew_blocks(n)), dim3(256), 0, cf_s(stream)>>>(x, G1, G2, gx, D, x_bstride, n);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_gmm_bwd_params(const float* a, const float* nm, const float* sG, const float* S0, const float* S1, const float* S2,
                      float* gmu, float* gsig, int MK, int D, cf_stream_t stream) {
    CF_REQUIRE(a && nm && sG && S0 && S1 && S2 && gmu && gsig && MK > 0 && D > 0);
    const int64_t n = (int64_t)MK * D;
    k_gmm_bwd_params<<<dim3(gmm_ew_blocks(n)), dim3(256), 0, cf_s(stream)>>>(a, nm, sG, S0, S1, S2, gmu, gsig, D, n);
    CF_LAUNCH_CHECK();
    return 0;
}

// S0 (MK), S1, S2 (MK, D) of the mixture backward from the responsibilities r (B, MK) and x (B rows of D floats, row stride
// x_bstride): see k_gmm_sums.  MK = 80, D % 32 == 0 (cf_gmm_bwd_sums_supported); ws: cf_gmm_bwd_sums_ws_bytes(B, MK, D).
int cf_gmm_bwd_sums_supported(int MK, int D) { return MK == 16 * GS_RT && D > 0 && D % 32 == 0; }

int64_t cf_gmm_bwd_sums_ws_bytes(int B, int MK, int D) {
    if (!cf_gmm_bwd_sums_supported(MK, D) || B <= 0) return 0;
    int nsp, spb;
    gmm_sums_split(B, D, nsp, spb);
    return (int64_t)nsp * (2 * (int64_t)MK * D + MK) * (int64_t)sizeof(float);
}

int cf_gmm_bwd_sums(const float* x, const float* r, float* S0, float* S1, float* S2, void* ws, int B, int MK, int D,
                    int64_t x_bstride, cf_stream_t stream) {
    CF_REQUIRE(x && r && S0 && S1 && S2 && ws && B > 0 && cf_gmm_bwd_sums_supported(MK, D) && x_bstride >= D && x_bstride < (1 << 20));
    int nsp, spb;
    gmm_sums_split(B, D, nsp, spb);
    const int units = (D / 32) * nsp;
    k_gmm_sums<GS_RT><<<dim3((units + 3) / 4), dim3(256), 0, cf_s(stream)>>>(x, r, (float*)ws, B, D, x_bstride, nsp, spb);
    const int64_t per = 2 * (int64_t)MK * D + MK;
    k_gmm_sums_reduce<<<dim3((unsigned)((per + 255) / 256)), dim3(256), 0, cf_s(stream)>>>((const float*)ws, S0, S1, S2, MK, D, nsp);
    CF_LAUNCH_CHECK();
    return 0;
}

// the same + the mixture-weight gradient gw (M, K) = S0 - gcol softmax(wG) in the same launch (gcol (M) = column sums of
// the upstream gradient)
int cf_gmm_bwd_params_w(const float* a, const float* nm, const float* sG, const float* S0, const float* S1, const float* S2,
                        const float* wG, const float* gcol, float* gmu, float* gsig, float* gw, int M, int K, int D,
                        cf_stream_t stream) {
    CF_REQUIRE(a && nm && sG && S0 && S1 && S2 && wG && gcol && gmu && gsig && gw && M > 0 && K > 0 && D > 0);
    const int64_t n = (int64_t)M * K * D;
    k_gmm_bwd_params<<<dim3(gmm_ew_blocks(n)), dim3(256), 0, cf_s(stream)>>>(a, nm, sG, S0, S1, S2, gmu, gsig, D, n, wG, gcol, gw, M, K);
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
