// Weight gradients of the coupling net (and of the folded Conv1x1/ActNorm matrix) as one MFMA kernel family:
//
//     gw[t][m][n] = sum_{b, p}  A[b][m][p] * Bm[b][n][src_t(p)]          t = tap (1 or 9), src_t = reflect-shifted pixel
//
// i.e. a split-K GEMM whose K axis is (sample, pixel).  A = upstream gradient plane (g_h2, g_h, g_h1, g_y),
// Bm = forward activation plane (h1, h2, y0, x), both (B, rows, H*W) as written by cf_flow_step_bwd.
//
// MI355X design: the contraction index (pixel) must sit on the MFMA k axis, so both operands are staged in LDS
// TRANSPOSED — T[pixel][channel] with an odd row stride: the global reads stay coalesced along pixels, the LDS
// writes (lanes = consecutive pixels, stride odd) and the operand reads (lanes = consecutive channels) are both
// bank-conflict free.  The 3x3 taps are nine B-operand reads of the same staged tile through an LDS index table
// (reflect padding resolved once per workgroup).  A workgroup owns one 32-row tile of A and ALL columns / taps and
// keeps its <= 9 accumulator tiles per wave in registers over its whole K range; it writes its partial once with
// plain coalesced stores and a small second kernel sums the partials in a fixed order (no float atomics: the
// outputs are tiny and shared by every workgroup, and the result stays bitwise reproducible).  Output layout
// [t][m][n]; the caller permutes to the reference's [m][n][kh][kw].
#include "cf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// NT = 32-column tiles of Bm (1, 2 or 4); the 4 waves split (column tile) x (K quarter): KW = 4 / NT
template <int H, int W, int TAPS, int NT>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ A, const float* __restrict__ Bm,
                                               float* __restrict__ gw, float* __restrict__ gbias, int B, int MR, int NR) {
    constexpr int HW = H * W;
    constexpr int KC = HW >= 64 ? HW : 64;            // pixels per chunk (whole samples)
    constexpr int SPC = KC / HW;                      // samples per chunk
    constexpr int KW = 4 / NT;
    constexpr int SA = 33, SB = NT * 32 + 1;          // odd LDS row strides
    constexpr int IA = 32 * KC / 256, IB = NT * 32 * KC / 256;    // staged elements per thread
    extern __shared__ __align__(16) float lds[];
    float* TA = lds;                                  // [KC][SA]   A tile, transposed
    float* TB = lds + KC * SA;                        // [KC][SB]   B tile, transposed
    int* tab = reinterpret_cast<int*>(TB + KC * SB);  // [TAPS][KC] source pixel of every tap
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int nt = wave % NT, kq = wave / NT;
    const int m0 = blockIdx.x * 32;

    for (int e = tid; e < TAPS * KC; e += 256) {
        const int t = e / KC, pix = e - t * KC, p = pix % HW;
        int yy = p / W, xx = p % W;
        if (TAPS == 9) {
            yy += t / 3 - 1; xx += t % 3 - 1;
            yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
            xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
        }
        tab[e] = (pix - p) + yy * W + xx;
    }
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;                                 // row sum of A (= the bias gradient), lanes of column tile 0

    // global -> registers (lanes along pixels: coalesced); one chunk ahead of the MFMAs
    float ra[IA], rb[IB];
    auto gload = [&](int c) {
        const int s0 = c * SPC;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int e = i * 256 + tid, ch = e / KC, pix = e - ch * KC, b = s0 + pix / HW, m = m0 + ch;
            ra[i] = (m < MR && b < B) ? A[((int64_t)b * MR + m) * HW + pix % HW] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int e = i * 256 + tid, ch = e / KC, pix = e - ch * KC, b = s0 + pix / HW;
            rb[i] = (ch < NR && b < B) ? Bm[((int64_t)b * NR + ch) * HW + pix % HW] : 0.f;
        }
    };
    const int nchunks = (B + SPC - 1) / SPC;
    int c = blockIdx.y;
    if (c < nchunks) gload(c);
    for (; c < nchunks; c += gridDim.y) {
        __syncthreads();                              // previous chunk consumed (and tab written)
        // registers -> LDS, transposed (lanes = consecutive pixels, odd stride: conflict-free)
#pragma unroll
        for (int i = 0; i < IA; ++i) { const int e = i * 256 + tid, ch = e / KC, pix = e - ch * KC; TA[pix * SA + ch] = ra[i]; }
#pragma unroll
        for (int i = 0; i < IB; ++i) { const int e = i * 256 + tid, ch = e / KC, pix = e - ch * KC; TB[pix * SB + ch] = rb[i]; }
        __syncthreads();
        if (c + (int)gridDim.y < nchunks) gload(c + gridDim.y);       // next chunk in flight behind the MFMAs
        // K loop: k-step s covers pixels 2s, 2s+1; this wave takes the steps s = kq (mod KW)
#pragma unroll 2
        for (int s = kq; s < KC / 2; s += KW) {
            const int pix = 2 * s + lk;
            const float a = TA[pix * SA + li];                       // A[i = m][k = pixel]
            bsum += a;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const float b = TB[tab[t * KC + pix] * SB + nt * 32 + li];   // B[k = pixel][j = n], tap-shifted
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // combine the KW K-quarters of this workgroup in LDS, in a fixed order (deterministic), into the kq == 0 waves
    if (KW > 1) {
        float* R = lds;                               // [NT][TAPS][16][64] (+ [NT][64] for the bias sums)
        for (int k = 1; k < KW; ++k) {
            __syncthreads();
            if (kq == k) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) R[((nt * TAPS + t) * 16 + r) * 64 + lane] = acc[t][r];
                R[NT * TAPS * 1024 + nt * 64 + lane] = bsum;
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += R[((nt * TAPS + t) * 16 + r) * 64 + lane];
                bsum += R[NT * TAPS * 1024 + nt * 64 + lane];
            }
        }
    }
    // flush: D[i = m][j = n]: lane holds column n = nt*32 + li, rows (r&3) + 8*(r>>2) + 4*lk.  Every K split writes
    // its own partial with plain coalesced stores; k_wgrad_reduce sums them in a fixed order (atomics into the tiny,
    // shared output would serialise at the memory side — and would not be reproducible).
    if (kq != 0) return;
    const int n = nt * 32 + li;
    float* pw = gw + (int64_t)blockIdx.y * TAPS * MR * NR;
    if (n < NR) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < MR) pw[((int64_t)t * MR + m) * NR + n] = acc[t][r];
            }
    }
    if (nt == 0) {                                    // every A element is seen once by the waves of column tile 0
        bsum += __shfl_xor(bsum, 32, 64);
        if (lk == 0 && m0 + li < MR) gbias[(int64_t)blockIdx.y * MR + m0 + li] = bsum;
    }
}

// out[e] = sum_s part[s][e] in a fixed order: a wave covers 64 consecutive outputs, the 4 waves of a block split S
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ out, int n, int S) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int i = w;
        for (; i + 12 < S; i += 16) {
            s0 += part[(int64_t)i * n + e]; s1 += part[(int64_t)(i + 4) * n + e];
            s2 += part[(int64_t)(i + 8) * n + e]; s3 += part[(int64_t)(i + 12) * n + e];
        }
        for (; i < S; i += 4) s0 += part[(int64_t)i * n + e];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && e < n) out[e] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

inline int wgrad_splits(int B, int MR, int HW) {
    const int KC = HW >= 64 ? HW : 64, SPC = KC / HW;
    const int mtiles = (MR + 31) / 32, nchunks = (B + SPC - 1) / SPC;
    int splits = 512 / mtiles;                        // ~2 workgroups per CU in flight
    if (splits > nchunks) splits = nchunks;
    return splits < 1 ? 1 : splits;
}

template <int H, int W, int TAPS, int NT>
int launch_wgrad(const float* A, const float* Bm, float* gw, float* gbias, float* ws, int B, int MR, int NR, hipStream_t s) {
    constexpr int HW = H * W, KC = HW >= 64 ? HW : 64, KW = 4 / NT;
    constexpr size_t lds_main = (size_t)(KC * 33 + KC * (NT * 32 + 1) + TAPS * KC) * 4;
    constexpr size_t lds_comb = KW > 1 ? (size_t)(NT * TAPS * 1024 + NT * 64) * 4 : 0;
    constexpr size_t lds = lds_main > lds_comb ? lds_main : lds_comb;
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute((const void*)k_wgrad<H, W, TAPS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { cf_set_error("cf_wgrad: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return (int)e; }
            raised = true;
        }
    }
    const int mtiles = (MR + 31) / 32;
    const int splits = wgrad_splits(B, MR, HW);
    const int S = splits, nw = TAPS * MR * NR;
    float* pw = ws;                                   // [S][TAPS][MR][NR]
    float* pb = ws + (int64_t)S * nw;                 // [S][MR]
    k_wgrad<H, W, TAPS, NT><<<dim3(mtiles, splits), dim3(256), lds, s>>>(A, Bm, pw, pb, B, MR, NR);
    k_wgrad_reduce<<<dim3((nw + 63) / 64), dim3(256), 0, s>>>(pw, gw, nw, S);
    if (gbias) k_wgrad_reduce<<<dim3((MR + 63) / 64), dim3(256), 0, s>>>(pb, gbias, MR, S);
    return 0;
}

template <int H, int W, int TAPS>
int dispatch_nt(const float* A, const float* Bm, float* gw, float* gbias, float* ws, int B, int MR, int NR, hipStream_t s) {
    if (NR <= 32) return launch_wgrad<H, W, TAPS, 1>(A, Bm, gw, gbias, ws, B, MR, NR, s);
    if (NR <= 64) return launch_wgrad<H, W, TAPS, 2>(A, Bm, gw, gbias, ws, B, MR, NR, s);
    return launch_wgrad<H, W, TAPS, 4>(A, Bm, gw, gbias, ws, B, MR, NR, s);
}

}  // namespace

extern "C" {

// workspace for the split-K partials: [splits][taps*MR*NR + MR] floats
int64_t cf_wgrad_ws_bytes(int B, int MR, int NR, int H, int W, int taps) {
    const int S = wgrad_splits(B, MR, H * W);
    return (int64_t)S * ((int64_t)taps * MR * NR + MR) * 4;
}

int cf_wgrad(const float* A, const float* Bm, float* gw, float* gbias, void* ws, int B, int MR, int NR, int H, int W,
             int taps, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(A && Bm && gw && ws && MR > 0 && NR > 0 && NR <= 128 && (taps == 1 || taps == 9));
    int rc;
    hipStream_t s = cf_s(stream);
    float* w = (float*)ws;
#define CF_W(HH, WW) rc = taps == 9 ? dispatch_nt<HH, WW, 9>(A, Bm, gw, gbias, w, B, MR, NR, s) : dispatch_nt<HH, WW, 1>(A, Bm, gw, gbias, w, B, MR, NR, s)
    if (H == 16 && W == 16) CF_W(16, 16);
    else if (H == 8 && W == 8) CF_W(8, 8);
    else if (H == 4 && W == 4) CF_W(4, 4);
    else { cf_set_error("cf_wgrad: image %dx%d unsupported", H, W); return CF_ERR_UNSUPPORTED; }
#undef CF_W
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
