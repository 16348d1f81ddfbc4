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
// bank-conflict free.  The 3x3 taps are nine B-operand reads of the same staged tile; the reflected source
// rows are scalar arithmetic (the k-step index is wave-uniform), the three source columns three VALU ops.  A workgroup owns one 32-row tile of A and ALL columns / taps and
// keeps its <= 9 accumulator tiles per wave in registers over its whole K range; it writes its partial once with
// plain coalesced stores and a small second kernel sums the partials in a fixed order (no float atomics: the
// outputs are tiny and shared by every workgroup, and the result stays bitwise reproducible).  Output layout
// [t][m][n]; the caller permutes to the reference's [m][n][kh][kw].
// The 3x3 runs in the Winograd form F(3x3, 2x2) (further down: 16 MFMAs per 8 pixels instead of 36) unless
// CONTEXTFLOW_DIRECT_CONV=1; the direct 3x3 path stays as the second implementation the tests compare it with.
#include "cf_common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Operands of one k-step (pixels 2s, 2s+1) for all taps.  s is wave-uniform, so the sample base and the (reflected)
// source rows of the 3 vertical taps are scalar; only the 3 horizontal sources depend on the lane half.
template <int TAPS> struct WgOps { float a; float b[TAPS]; };

// LDS float offsets of the operands of k-step s: ad[0] = A, ad[1 + t] = B of tap t
template <int H, int W, int TAPS, int SA, int SB>
__device__ __forceinline__ void wg_addr(int (&ad)[TAPS + 1], int offTB, int s, int li, int lk, int ncol) {
    constexpr int HW = H * W;
    const int pix0 = 2 * s;                                   // scalar; W >= 4 is even: pix0 and pix0 + 1 share a row
    ad[0] = (pix0 + lk) * SA + li;
    if constexpr (TAPS == 1) {
        ad[1] = offTB + (pix0 + lk) * SB + ncol;
    } else {
        const int p = pix0 & (HW - 1), base = pix0 - p, y = p / W, x = (p & (W - 1)) + lk;
        int rowoff[3], coloff[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            int yy = y + d - 1; yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
            int xx = x + d - 1; xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
            rowoff[d] = offTB + (base + yy * W) * SB;         // scalar
            coloff[d] = xx * SB + ncol;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) ad[1 + t] = rowoff[t / 3] + coloff[t % 3];
    }
}

template <int TAPS>
__device__ __forceinline__ void wg_load(WgOps<TAPS>& o, const float* __restrict__ lds, const int (&ad)[TAPS + 1]) {
    o.a = lds[ad[0]];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) o.b[t] = lds[ad[1 + t]];
}

template <int TAPS>
__device__ __forceinline__ void wg_mma(f32x16 (&acc)[TAPS], const WgOps<TAPS>& o, float& bsum) {
    bsum += o.a;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a, o.b[t], acc[t], 0, 0, 0);
}

// ---- 3x3 path ---------------------------------------------------------------------------------------------
// Measured on gfx950 (tools/micro/mfma_issue.hip): VALU instructions never overlap the MFMAs of their SIMD (the
// first one after an MFMA costs ~16 cycles, each further one ~4.5; SALU and LDS instructions are free), so the K
// loop is built to need almost none:
//   * a wave owns whole image rows: the source rows of the three vertical taps are scalar (reflect on the SALU), the
//     W/2 k-steps of a row are unrolled, so every operand is `row register + immediate`: 5 v_add per row;
//   * the horizontal reflect costs nothing: at the left border the dx=-1 operand (columns 1,0) is the dx=0 operand
//     with the two k-halves swapped, and swapping the halves of B equals swapping those of A - one extra A read
//     (`as`) instead of lane selects; same at the right border for dx=+1;
//   * operand reads of k-step j+1 are issued as one batch before the MFMAs of step j (one VALU<->MFMA switch a step).
struct WgRow { int a, as, b[3]; };                    // LDS byte offsets of a row's operands, lane part included
struct WgOps9 { float a, as, b[9]; };

__device__ __forceinline__ float wg_ldf(const char* ldsb, int off) { return *reinterpret_cast<const float*>(ldsb + off); }

template <int H, int W, int SA, int SB>
__device__ __forceinline__ void wg_row_addr(WgRow& r, int g, int offTBb, int laneA, int laneAs, int laneB) {
    const int smp = g / H, y = g - smp * H, base = smp * (H * W);     // scalar
    r.a = (base + y * W) * (SA * 4) + laneA;
    r.as = (base + y * W) * (SA * 4) + laneAs;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int yy = y + d - 1; yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
        r.b[d] = offTBb + (base + yy * W) * (SB * 4) + laneB;
    }
}

template <int W, int XS> constexpr bool wg_swapped(int dx) { return (2 * XS == 0 && dx == -1) || (2 * XS == W - 2 && dx == 1); }

template <int W, int SA, int SB, int XS>
__device__ __forceinline__ void wg_row_load(WgOps9& o, const char* ldsb, const WgRow& r) {
    constexpr int x0 = 2 * XS;
    o.a = wg_ldf(ldsb, r.a + x0 * SA * 4);
    if constexpr (x0 == 0 || x0 == W - 2) o.as = wg_ldf(ldsb, r.as + x0 * SA * 4);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
            if (!wg_swapped<W, XS>(dx)) o.b[dy * 3 + dx + 1] = wg_ldf(ldsb, r.b[dy] + (x0 + dx) * SB * 4);
}

template <int W, int XS>
__device__ __forceinline__ void wg_row_mma(f32x16 (&acc)[9], const WgOps9& o, float& bsum) {
    bsum += o.a;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int t = dy * 3 + dx + 1;
            if (wg_swapped<W, XS>(dx)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.as, o.b[dy * 3 + 1], acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a, o.b[t], acc[t], 0, 0, 0);
        }
}

// the W/2 k-steps of one row; operands alternate between o0 / o1 (W/2 is even), the last step prefetches the next row
template <int W, int SA, int SB, int XS>
__device__ __forceinline__ void wg_row_steps(f32x16 (&acc)[9], WgOps9& o0, WgOps9& o1, const char* ldsb, const WgRow& cur,
                                             const WgRow& nxt, float& bsum) {
    if constexpr (XS < W / 2) {
        WgOps9& use = (XS & 1) ? o1 : o0;
        WgOps9& fill = (XS & 1) ? o0 : o1;
        if constexpr (XS + 1 < W / 2) wg_row_load<W, SA, SB, XS + 1>(fill, ldsb, cur);
        else wg_row_load<W, SA, SB, 0>(fill, ldsb, nxt);
        __builtin_amdgcn_sched_barrier(0);
        wg_row_mma<W, XS>(acc, use, bsum);
        __builtin_amdgcn_sched_barrier(0);
        wg_row_steps<W, SA, SB, XS + 1>(acc, o0, o1, ldsb, cur, nxt, bsum);
    }
}

// ---- 3x3 path, Winograd form F(3x3, 2x2) -----------------------------------------------------------------------
// The weight gradient of a 3x3 convolution is itself a convolution with 3x3 outputs (the taps) and the 2x2 tiles of the
// upstream gradient as filter:  gw = A^T [ sum_{sample, tile} (G dy G^T) . (B^T d B) ] A   with dy the 2x2 tile of A, d the
// 4x4 reflect-padded patch of Bm around it.  The sum over tiles and samples runs in the Winograd domain: 16 positions,
// each a rank-update of a 32x32 accumulator tile, i.e. 16 MFMAs per 2 tiles (= 8 pixels) where the direct form issues
// 36.  The k axis of the MFMA is the TILE index: a lane (channel li, tile lk) reads its own 2x2 / 4x4 raw values from the
// transposed LDS tiles (conflict-free as before), transforms them in registers (12 + 32 additions; G' = 2G is used
// unscaled, the factors 1/2 are folded into the output transform) and feeds the 16 results straight into the MFMAs as
// operands - no LDS round trip for the transformed data.  The 16 accumulator tiles (256 registers per lane) live in the
// AGPR half of the register file: one wave per SIMD, which is all the MFMA pipe needs here (16 independent
// accumulators back to back; the raw reads of the next step are issued in front of them).
// Reflect: the patch column -1 of the leftmost tile is column 1 and the tile of the other lane half needs column
// x0 - 1 = 1 as well - BOTH halves read column 1, so the border is an address without the lane-half term (same at the
// right border with column W - 2): no selects.  Rows are scalar as in the direct form.
// Bias gradient: position (1, 1) of G' dy G'^T is the plain sum of the tile.
struct WwRow { int a[2], b[4], c[4]; };               // LDS byte offsets: rows of A; rows of Bm with / without the lane-half term
struct WwRaw { float a[4], b[16]; };

template <int H, int W, int SA, int SB>
__device__ __forceinline__ void ww_row_addr(WwRow& r, int g, int offTBb, int laneA, int laneB, int laneC) {
    constexpr int TR = H / 2;                          // tile rows per sample
    const int smp = g / TR, ty = g - smp * TR, base = smp * (H * W);      // scalar
#pragma unroll
    for (int i = 0; i < 2; ++i) r.a[i] = (base + (2 * ty + i) * W) * (SA * 4) + laneA;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int yy = 2 * ty - 1 + i; yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
        const int rb = offTBb + (base + yy * W) * (SB * 4);
        r.b[i] = rb + laneB;
        r.c[i] = rb + laneC;
    }
}

// raw operands of k-step XS of a tile row: tiles 2 XS + lk, i.e. x0 = 4 XS + 2 lk (the 2 lk is in laneA / laneB)
template <int W, int SA, int SB, int XS>
__device__ __forceinline__ void ww_load(WwRaw& o, const char* ldsb, const WwRow& r) {
    constexpr int x0 = 4 * XS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) o.a[i * 2 + j] = wg_ldf(ldsb, r.a[i] + (x0 + j) * SA * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (XS == 0 && j == 0) o.b[i * 4 + j] = wg_ldf(ldsb, r.c[i] + 1 * SB * 4);
            else if (XS == W / 4 - 1 && j == 3) o.b[i * 4 + j] = wg_ldf(ldsb, r.c[i] + (W - 2) * SB * 4);
            else o.b[i * 4 + j] = wg_ldf(ldsb, r.b[i] + (x0 + j - 1) * SB * 4);
        }
}

__device__ __forceinline__ void ww_transform(const WwRaw& o, float (&at)[16], float (&bt)[16]) {
    {   // G' dy G'^T, G' = [[1,0],[1,1],[1,-1],[0,1]]
        const float p[4] = {o.a[0], o.a[0] + o.a[2], o.a[0] - o.a[2], o.a[2]};
        const float q[4] = {o.a[1], o.a[1] + o.a[3], o.a[1] - o.a[3], o.a[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) { at[4 * i] = p[i]; at[4 * i + 1] = p[i] + q[i]; at[4 * i + 2] = p[i] - q[i]; at[4 * i + 3] = q[i]; }
    }
    float t[16];                                       // B^T d
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[j] = o.b[j] - o.b[8 + j]; t[4 + j] = o.b[4 + j] + o.b[8 + j];
        t[8 + j] = o.b[8 + j] - o.b[4 + j]; t[12 + j] = o.b[4 + j] - o.b[12 + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                      // (B^T d) B
        bt[4 * i] = t[4 * i] - t[4 * i + 2]; bt[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
        bt[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1]; bt[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
    }
}

// the W/4 k-steps of one tile row; the raw reads of the next step (the last: of the next row) fly behind the MFMAs
template <int W, int SA, int SB, int XS>
__device__ __forceinline__ void ww_row_steps(f32x16 (&acc)[16], WwRaw& raw, const char* ldsb, const WwRow& cur, const WwRow& nxt,
                                             float& bsum) {
    if constexpr (XS < W / 4) {
        float at[16], bt[16];
        ww_transform(raw, at, bt);
        bsum += at[5];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (XS + 1 < W / 4) ww_load<W, SA, SB, XS + 1>(raw, ldsb, cur);
        else ww_load<W, SA, SB, 0>(raw, ldsb, nxt);
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[p], bt[p], acc[p], 0, 0, 0);
        // with one wave per SIMD every instruction costs an issue slot of the wave: the LDS reads go into the shadow of
        // the MFMAs (an MFMA occupies the pipe for 64 cycles; LDS / scalar instructions issue next to it)
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // 2 DS reads
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
        __builtin_amdgcn_sched_barrier(0);
        ww_row_steps<W, SA, SB, XS + 1>(acc, raw, ldsb, cur, nxt, bsum);
    }
}

// output transform of one accumulator element: gw(3x3) = A^T (s M s) A, A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]], s = (1, 1/2, 1/2, 1)
__device__ __forceinline__ void ww_output(const float (&m)[16], float (&o)[9]) {
    float r[12];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float p = 0.5f * (m[4 * x + 1] + m[4 * x + 2]), q = 0.5f * (m[4 * x + 1] - m[4 * x + 2]);
        r[3 * x] = m[4 * x] + p; r[3 * x + 1] = q; r[3 * x + 2] = p - m[4 * x + 3];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float p = 0.5f * (r[3 + c] + r[6 + c]), q = 0.5f * (r[3 + c] - r[6 + c]);
        o[c] = r[c] + p; o[3 + c] = q; o[6 + c] = p - r[9 + c];
    }
}

// NT = 32-column tiles of Bm (1, 2 or 4); the 4 waves split (column tile) x (K quarter): KW = 4 / NT
// blockIdx.z = flow step of a batch (cf_step_wgrads_batch: the steps of one resolution level in one launch per product)
constexpr int kWgBatch = 8;
struct WgBatch { const float* A[kWgBatch]; const float* Bm[kWgBatch]; float* part[kWgBatch]; };

template <int H, int W, int TAPS, int NT, bool WINO = false>
__global__ __launch_bounds__(256) void k_wgrad(const WgBatch wb, int B, int MR, int NR, int64_t bsB, int sqB) {
    const float* __restrict__ A = wb.A[blockIdx.z]; const float* __restrict__ Bm = wb.Bm[blockIdx.z];
    float* __restrict__ part = wb.part[blockIdx.z];
    constexpr int HW = H * W;
    constexpr int KC = HW >= 64 ? HW : 64;            // pixels per chunk (whole samples)
    constexpr int SPC = KC / HW;                      // samples per chunk
    constexpr int KW = 4 / NT;
    constexpr int SA = 33, SB = NT * 32 + 1;          // odd LDS row strides
    constexpr int IA = 32 * KC / 256, IB = NT * 32 * KC / 256;    // staged elements per thread
    extern __shared__ __align__(16) float lds[];
    float* TA = lds;                                  // [KC][SA]   A tile, transposed
    float* TB = lds + KC * SA;                        // [KC][SB]   B tile, transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int nt = wave % NT, kq = __builtin_amdgcn_readfirstlane(wave / NT);
    const int m0 = blockIdx.x * 32;

    static_assert(!WINO || TAPS == 9, "the Winograd form is the 3x3's");
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x16 wacc[WINO ? 16 : 1];                       // Winograd-domain accumulators (16 positions)
#pragma unroll
    for (int t = 0; t < (WINO ? 16 : 1); ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[t][r] = 0.f;
    float bsum = 0.f;                                 // row sum of A (= the bias gradient), lanes of column tile 0

    // global -> registers (lanes along pixels: coalesced); one chunk ahead of the MFMAs.  Element i of thread tid is
    // channel i*(256/KC) + tid/KC (wave-uniform), pixel tid % KC: a scalar row base + a lane offset that does not
    // depend on the chunk - no vector address arithmetic per load.  Rows / columns past MR / NR are clamped (their
    // products land in accumulator rows / columns that are never stored); samples past B are clamped and the A
    // operand zeroed.
    constexpr int CPI = 256 / KC;                     // channels per staging iteration (1 or 4)
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int chw = CPI == 1 ? 0 : wv / (KC / 64);    // channel offset of this wave inside an iteration
    const int pixl = tid % KC, sj = pixl / HW, pl = pixl % HW;
    const bool fullA = m0 + 32 <= MR, fullB = NT * 32 <= NR;
    float ra[IA], rb[IB];
    auto gload = [&](int c) {
        const int s0 = c * SPC;
        int sjc = sj;
        if constexpr (SPC > 1) sjc = min(sj, B - 1 - s0);
        const unsigned offA = sjc * MR * HW + pl, offB = sjc * (unsigned)bsB + pl;
        if (fullA) {                                  // whole 32-row tile: one vector base, immediate row offsets
            const float* pa = A + ((int64_t)s0 * MR + m0 + chw) * HW + offA;
#pragma unroll
            for (int i = 0; i < IA; ++i) ra[i] = pa[i * CPI * HW];
        } else {
            int m0v = m0, mrv = MR;
            asm volatile("" : "+s"(m0v), "+s"(mrv));  // keep the row bases out of LICM (live SGPRs spill to lanes)
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const float* rowp = A + ((int64_t)s0 * MR + min(m0v + i * CPI + chw, mrv - 1)) * HW;
                ra[i] = rowp[offA];
            }
        }
        bool doneB = false;
        if constexpr (TAPS == 1 && !WINO) {
            if (sqB) {               // Bm is the tensor BEFORE Squeeze((2,2)) (squeeze.py:10-11), (NR/4, 2H, 2W) per sample with batch
                //                      stride bsB: channel c = 4 q + 2 dy + dx at (y, x) is element (q, 2 y + dy, 2 x + dx)
                const int yy = pl / W, xx = pl - yy * W;
                const float* pb = Bm + (int64_t)s0 * bsB + sjc * (unsigned)bsB + 4 * yy * W + 2 * xx;
                int nrv = NR;
                asm volatile("" : "+s"(nrv));
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const int c = min(i * CPI + chw, nrv - 1);
                    rb[i] = pb[(c >> 2) * 4 * HW + ((c >> 1) & 1) * 2 * W + (c & 1)];
                }
                doneB = true;
            }
        }
        if (doneB) {
        } else if (fullB) {
            const float* pb = Bm + (int64_t)s0 * bsB + chw * HW + offB;
#pragma unroll
            for (int i = 0; i < IB; ++i) rb[i] = pb[i * CPI * HW];
        } else {
            int nrv = NR;
            asm volatile("" : "+s"(nrv));
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const float* rowp = Bm + (int64_t)s0 * bsB + min(i * CPI + chw, nrv - 1) * HW;
                rb[i] = rowp[offB];
            }
        }
        if constexpr (SPC > 1) {
            const float valid = (sj <= B - 1 - s0) ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < IA; ++i) ra[i] *= valid;
        }
    };
    // Software pipeline over this workgroup's chunks: the loads of chunk k+1 are issued, then the MFMAs of chunk k
    // (operands in LDS) run in front of them, then chunk k+1 goes registers -> LDS.  One gload / one MFMA site.
    const int nchunks = (B + SPC - 1) / SPC;
    bool have = false;                                // LDS holds a chunk
    for (int cn = blockIdx.y;; cn += gridDim.y) {
        const bool more = cn < nchunks;
        if (more) gload(cn);
        if (have) {
        constexpr int offTB = KC * SA;
        const int ncol = nt * 32 + li;
        if constexpr (WINO) {
            // this wave takes the TILE rows g = kq (mod KW) of the chunk's SPC samples
            constexpr int TROWS = SPC * H / 2;
            static_assert(TROWS % KW == 0 && W % 4 == 0, "tile rows split evenly; two tiles per k-step");
            const char* ldsb = reinterpret_cast<const char*>(lds);
            const int laneA = (2 * lk * SA + li) * 4, laneB = (2 * lk * SB + ncol) * 4, laneC = ncol * 4;
            WwRaw raw;
            WwRow cur, nxt;
            ww_row_addr<H, W, SA, SB>(cur, kq, offTB * 4, laneA, laneB, laneC);
            ww_load<W, SA, SB, 0>(raw, ldsb, cur);
#pragma unroll 1
            for (int g = kq; g < TROWS; g += KW) {
                const int gn = g + KW < TROWS ? g + KW : kq;          // scalar select, no branch
                ww_row_addr<H, W, SA, SB>(nxt, gn, offTB * 4, laneA, laneB, laneC);
                ww_row_steps<W, SA, SB, 0>(wacc, raw, ldsb, cur, nxt, bsum);
                cur = nxt;
            }
        } else if constexpr (TAPS == 9) {
            // this wave takes the image rows g = kq (mod KW) of the chunk's SPC samples
            constexpr int ROWS = SPC * H;
            static_assert(ROWS % KW == 0 && (W / 2) % 2 == 0, "rows split evenly; even number of k-steps per row");
            const char* ldsb = reinterpret_cast<const char*>(lds);
            const int laneA = (lk * SA + li) * 4, laneAs = ((1 - lk) * SA + li) * 4, laneB = (lk * SB + ncol) * 4;
            WgOps9 o0, o1;
            WgRow cur, nxt;
            wg_row_addr<H, W, SA, SB>(cur, kq, offTB * 4, laneA, laneAs, laneB);
            wg_row_load<W, SA, SB, 0>(o0, ldsb, cur);
#pragma unroll 1
            for (int g = kq; g < ROWS; g += KW) {
                const int gn = g + KW < ROWS ? g + KW : kq;           // scalar select, no branch (see below)
                wg_row_addr<H, W, SA, SB>(nxt, gn, offTB * 4, laneA, laneAs, laneB);
                wg_row_steps<W, SA, SB, 0>(acc, o0, o1, ldsb, cur, nxt, bsum);
                cur = nxt;
            }
        } else {
            // 1x1: bound by the staging.  k-step s covers pixels 2s, 2s+1; this wave takes s = kq (mod KW); two plain
            // stages, no branch inside the loop (a join would force lgkmcnt(0)).
            constexpr int NS = KC / 2 / KW;
            static_assert(NS % 2 == 0, "even number of k-steps per wave");
            WgOps<TAPS> o0, o1;
            int ad[TAPS + 1];
            wg_addr<H, W, TAPS, SA, SB>(ad, offTB, kq, li, lk, ncol);
            wg_load<TAPS>(o0, lds, ad);
            wg_addr<H, W, TAPS, SA, SB>(ad, offTB, kq + KW, li, lk, ncol);
#pragma unroll 1
            for (int i = 0; i < NS - 2; i += 2) {
                wg_load<TAPS>(o1, lds, ad);
                wg_addr<H, W, TAPS, SA, SB>(ad, offTB, kq + (i + 2) * KW, li, lk, ncol);
                __builtin_amdgcn_sched_barrier(0);
                wg_mma<TAPS>(acc, o0, bsum);
                __builtin_amdgcn_sched_barrier(0);
                wg_load<TAPS>(o0, lds, ad);
                wg_addr<H, W, TAPS, SA, SB>(ad, offTB, kq + (i + 3) * KW, li, lk, ncol);
                __builtin_amdgcn_sched_barrier(0);
                wg_mma<TAPS>(acc, o1, bsum);
                __builtin_amdgcn_sched_barrier(0);
            }
            wg_load<TAPS>(o1, lds, ad);
            __builtin_amdgcn_sched_barrier(0);
            wg_mma<TAPS>(acc, o0, bsum);
            wg_mma<TAPS>(acc, o1, bsum);
        }
        }
        if (!more) break;
        __syncthreads();                              // previous chunk consumed
        // registers -> LDS, transposed (lanes = consecutive pixels, odd stride: conflict-free)
#pragma unroll
        for (int i = 0; i < IA; ++i) TA[pixl * SA + chw + i * CPI] = ra[i];
#pragma unroll
        for (int i = 0; i < IB; ++i) TB[pixl * SB + chw + i * CPI] = rb[i];
        __syncthreads();
        have = true;
    }
    if constexpr (WINO) {                              // back to the 9 taps, once
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float m[16], o[9];
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] = wacc[p][r];
            ww_output(m, o);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t][r] = o[t];
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
    float* pw = part + (int64_t)blockIdx.y * (TAPS * MR * NR + MR);        // this split's partial: [TAPS][MR][NR] | [MR]
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
        if (lk == 0 && m0 + li < MR) pw[TAPS * MR * NR + m0 + li] = bsum;
    }
}

// ---- 1x1 weight gradients without LDS ---------------------------------------------------------------------------------
// gw[m][n] = sum_{b, p} A[b][m][p] Bm[b][n][p] with 8..128 rows on either side: a skinny product whose cost is reading the
// two planes once (C = 16 on 16x16: 48 rows x 16.8 MB = 100 us at HBM speed; the staged 32x32x2 kernel above took 171 us,
// loading 32-row tiles for 16 / 8 rows and going through LDS transposes).  Here a WAVE owns all RT x CT 16x16 output
// tiles and a contiguous range of (sample, 16-pixel group) units, and reads both operands from global memory directly in
// the layout of v_mfma_f32_16x16x4_f32: lane (n = lane & 15, kk = lane >> 4) loads the float4 A[b][16 rt + n][16 g + 4 kk ..]
// and the float4 Bm[b][16 ct + n][16 g + 4 kk ..] - its four values are the operands of four successive k-steps whose k
// index (the lane group kk) stands for pixel 16 g + 4 kk + e on BOTH sides; the order in which a sum visits its pixels is
// free.  No LDS, no barrier; the row sums of A (bias gradient) are four adds per load.  Partials per wave in the layout of
// k_wgrad ([MR][NR] | [MR]), summed in order by the same reduce kernels.
// SQB: Bm is the tensor BEFORE Squeeze((2,2)) - channel c = 4 q + 2 dy + dx at (y, x) is element (q, 2 y + dy, 2 x + dx):
// the 4 pixels of a lane are 8 consecutive floats of row 2 y + dy, of which it keeps those of its dx.
__device__ __forceinline__ float wg_f4e(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

template <int RT, int CT, bool SQB, int NF>
__global__ __launch_bounds__(256) void k_wgrad1x1(const WgBatch wb, int MR, int NR, int HW, int W, int64_t bsB,
                                                  int Q, int qper, int nsplit) {
    const float* __restrict__ A = wb.A[blockIdx.z]; const float* __restrict__ Bm = wb.Bm[blockIdx.z];
    float* __restrict__ part = wb.part[blockIdx.z];
    // NF = float4 per lane, row and unit: a unit is 16 NF pixels, of which lane group kk takes the 4 NF consecutive ones from
    // 4 NF kk - with NF = 2 the four lane groups of a row read 128 contiguous bytes (whole cache lines; 16x16 / 8x8 images)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, n = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sp = blockIdx.x * 4 + wave;              // wave-uniform: the unit index and everything derived from it (sample,
    //                                                    group, base pointers) live on the SALU
    const int q0 = min(Q, sp * qper), q1 = min(Q, q0 + qper);       // (a wave past the end has an empty range: it still joins the barriers)
    const int gps = HW / (16 * NF);                    // units per sample
    f32x4 acc[RT][CT];
    float bs[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        bs[rt] = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int arow[RT], brow[CT];                            // row offsets inside a sample (clamped: results of rows past MR / NR are not stored)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) arow[rt] = min(16 * rt + n, MR - 1) * HW + 4 * NF * kk;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = min(16 * ct + n, NR - 1);
        brow[ct] = SQB ? (c >> 2) * 4 * HW + ((c >> 1) & 1) * 2 * W + (c & 1) : c * HW + 4 * NF * kk;
    }
    auto load = [&](int q, float4 (&av)[RT][NF], float4 (&bv)[CT][NF]) {
        const int b = q / gps, g = q - b * gps;
        const float* ab = A + (int64_t)b * MR * HW + 16 * NF * g;
        const float* bb = Bm + (int64_t)b * bsB;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int f = 0; f < NF; ++f) av[rt][f] = *reinterpret_cast<const float4*>(ab + arow[rt] + 4 * f);
        if constexpr (!SQB) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int f = 0; f < NF; ++f) bv[ct][f] = *reinterpret_cast<const float4*>(bb + 16 * NF * g + brow[ct] + 4 * f);
        } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int p0 = 16 * NF * g + 4 * NF * kk + 4 * f, yy = p0 / W, xx = p0 - yy * W;     // 4 pixels of one image row (W >= 4)
                const float* src = bb + 4 * yy * W + 2 * xx;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float* r = src + (brow[ct] & ~1);
                    const float4 u0 = *reinterpret_cast<const float4*>(r), u1 = *reinterpret_cast<const float4*>(r + 4);
                    bv[ct][f] = (brow[ct] & 1) ? make_float4(u0.y, u0.w, u1.y, u1.w) : make_float4(u0.x, u0.z, u1.x, u1.z);
                }
            }
        }
    };
    auto mma = [&](const float4 (&av)[RT][NF], const float4 (&bv)[CT][NF]) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) bs[rt] += (av[rt][f].x + av[rt][f].y) + (av[rt][f].z + av[rt][f].w);
#pragma unroll
            for (int e = 0; e < 4; ++e)              // k-step outermost: successive MFMAs go to different accumulators
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg_f4e(av[rt][f], e), wg_f4e(bv[ct][f], e), acc[rt][ct], 0, 0, 0);
        }
    };
    // two units in flight behind the one being multiplied
    float4 a0[RT][NF], b0[CT][NF], a1[RT][NF], b1[CT][NF], a2[RT][NF], b2[CT][NF];
    int q = q0;
    if (q < q1) load(q, a0, b0);
    if (q + 1 < q1) load(q + 1, a1, b1);
    for (; q + 2 < q1; q += 3) {
        load(q + 2, a2, b2);
        mma(a0, b0);
        if (q + 3 < q1) load(q + 3, a0, b0);
        mma(a1, b1);
        if (q + 4 < q1) load(q + 4, a1, b1);
        mma(a2, b2);
    }
    if (q < q1) mma(a0, b0);
    if (q + 1 < q1) mma(a1, b1);
    // the four waves of the workgroup meet in LDS (waves 3, 2, 1 in that order onto wave 0): one partial per WORKGROUP
    extern __shared__ __align__(16) float xch[];       // [RT * CT * 4 + RT][64]
#pragma unroll 1
    for (int src = 3; src >= 1; --src) {
        if (wave == src) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int j = 0; j < 4; ++j) xch[((rt * CT + ct) * 4 + j) * 64 + lane] = acc[rt][ct][j];
                xch[(RT * CT * 4 + rt) * 64 + lane] = bs[rt];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[rt][ct][j] += xch[((rt * CT + ct) * 4 + j) * 64 + lane];
                bs[rt] += xch[(RT * CT * 4 + rt) * 64 + lane];
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
    float* pw = part + (int64_t)blockIdx.x * ((int64_t)MR * NR + MR);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = 16 * rt + 4 * kk + j, c = 16 * ct + n;
                if (m < MR && c < NR) pw[m * NR + c] = acc[rt][ct][j];
            }
        float v = bs[rt];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (kk == 0 && 16 * rt + n < MR) pw[MR * NR + 16 * rt + n] = v;
    }
}

// out[e] = sum_s part[s][e] in a fixed order: a wave covers 64 consecutive outputs, the 4 waves of a block split S.
// The first n0 outputs go to out0 (weights), the rest to out1 (bias; may be null).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ out0,
                                                      float* __restrict__ out1, int n0, int n, int S) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int i = w;
        for (; i + 60 < S; i += 64) {                 // the same sums in the same order as the loop below, 16 loads in flight
            float v[16];                               // (one partial per WAVE of k_wgrad1x1: a thread walks up to 512 of them)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = part[(int64_t)(i + 4 * j) * n + e];
#pragma unroll
            for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
        }
        for (; i + 12 < S; i += 16) {
            s0 += part[(int64_t)i * n + e]; s1 += part[(int64_t)(i + 4) * n + e];
            s2 += part[(int64_t)(i + 8) * n + e]; s3 += part[(int64_t)(i + 12) * n + e];
        }
        for (; i < S; i += 4) s0 += part[(int64_t)i * n + e];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && e < n) {
        const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (e < n0) out0[e] = v;
        else if (out1 != nullptr) out1[e - n0] = v;
    }
}

// the same sum for up to four problems in one launch (blockIdx.y = problem): the four weight gradients of a flow step
struct WgReduce4 { const float* part[4]; float* out0[4]; float* out1[4]; int n0[4], n[4], S[4], taps[4]; };   // taps > 1: out0 as [m][n][tap]
struct WgReduce4B { WgReduce4 d[kWgBatch]; };         // blockIdx.z = flow step
__global__ __launch_bounds__(256) void k_wgrad_reduce4(const WgReduce4B db) {
    __shared__ float red[4][64];
    const WgReduce4& d = db.d[blockIdx.z];
    const int q = blockIdx.y;
    const float* __restrict__ part = d.part[q];
    const int n = d.n[q], n0 = d.n0[q], S = d.S[q];
    if ((int)blockIdx.x * 64 >= n) return;            // uniform per block
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int i = w;
        for (; i + 60 < S; i += 64) {                 // the same sums in the same order as the loop below, 16 loads in flight
            float v[16];                               // (one partial per WAVE of k_wgrad1x1: a thread walks up to 512 of them)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = part[(int64_t)(i + 4 * j) * n + e];
#pragma unroll
            for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
        }
        for (; i + 12 < S; i += 16) {
            s0 += part[(int64_t)i * n + e]; s1 += part[(int64_t)(i + 4) * n + e];
            s2 += part[(int64_t)(i + 8) * n + e]; s3 += part[(int64_t)(i + 12) * n + e];
        }
        for (; i < S; i += 4) s0 += part[(int64_t)i * n + e];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && e < n) {
        const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int tp = d.taps[q], mn = n0 / tp;       // partials are [tap][m][n]; the 3x3 leaves as the reference's [m][n][kh][kw]
        if (e < n0) d.out0[q][tp > 1 ? (e % mn) * tp + e / mn : e] = v;
        else if (d.out1[q] != nullptr) d.out1[q][e - n0] = v;
    }
}

// cf_step_wgrads: the k_wgrad launches leave their partials in place and report their split count; one reduce follows
static thread_local bool g_wgrad_defer = false;
static thread_local int g_wgrad_last_S = 0;
// cf_step_wgrads_batch: the launch covers g_wgrad_nb steps whose operands are g_wgrad_batch's (the pointer arguments of the
// launchers are step 0's then)
static thread_local const WgBatch* g_wgrad_batch = nullptr;
static thread_local int g_wgrad_nb = 1;
inline WgBatch wg_batch(const float* A, const float* Bm, float* part) {
    if (g_wgrad_batch) return *g_wgrad_batch;
    WgBatch wb{};
    wb.A[0] = A; wb.Bm[0] = Bm; wb.part[0] = part;
    return wb;
}

// wgs = workgroups to aim for: 512 (two per CU in flight) for the direct forms; the Winograd form runs one workgroup per
// CU (512 registers per lane), so 256 - one round, one epilogue per CU
inline int wgrad_splits(int B, int MR, int HW, int wgs = 512) {
    const int KC = HW >= 64 ? HW : 64, SPC = KC / HW;
    const int mtiles = (MR + 31) / 32, nchunks = (B + SPC - 1) / SPC;
    int splits = wgs / mtiles;
    // small batches: at least CF_WGRAD_MIN_CHUNKS chunks per workgroup - a split costs a whole epilogue and a row of the reduce
    // (batch of 256 at 4x4: 64 splits of ONE chunk each; with 16, the captured cifar10 training step 1.51 -> 1.43 ms)
#ifndef CF_WGRAD_MIN_CHUNKS
#define CF_WGRAD_MIN_CHUNKS 4
#endif
    if (splits > nchunks / CF_WGRAD_MIN_CHUNKS) splits = nchunks / CF_WGRAD_MIN_CHUNKS;
    if (splits > nchunks) splits = nchunks;
    return splits < 1 ? 1 : splits;
}

// k_wgrad1x1: one partial per wave; about two waves per SIMD, at least 8 units each.  A unit is 32 pixels on 16x16 / 8x8
// images (NF = 2), 16 on 4x4
inline int wgrad1x1_nf(int HW) { return HW >= 64 ? 2 : 1; }
// nsplit = number of PARTIALS = workgroups of four waves; waves: about 4 per SIMD for the small tiles (<= 128 registers), 2 otherwise
inline void wgrad1x1_split(int B, int MR, int NR, int HW, int& Q, int& qper, int& nsplit) {
    const int rt = (MR + 15) / 16, ct = (NR + 15) / 16;
    const int waves = rt * ct <= 2 ? 4096 : 2048;
    Q = B * (HW / (16 * wgrad1x1_nf(HW)));
    qper = (Q + waves - 1) / waves;                    // (small batches: one unit per wave - the units are what fills the chip)
#ifndef CF_W1_MIN_UNITS
#define CF_W1_MIN_UNITS 1
#endif
    if (qper < CF_W1_MIN_UNITS) qper = CF_W1_MIN_UNITS;
    nsplit = ((Q + qper - 1) / qper + 3) / 4;
}
inline bool wgrad1x1_ok(int MR, int NR, int HW) {
    const int rt = (MR + 15) / 16, ct = (NR + 15) / 16;
    return HW % 16 == 0 && rt <= 8 && ct <= 8 && rt * ct <= (wgrad1x1_nf(HW) == 2 ? 8 : 32);
}

template <int RT, int CT, int NF>
int launch_wgrad1x1(const float* A, const float* Bm, float* ws, int MR, int NR, int HW, int W, int64_t bsB, int sqB, int Q,
                    int qper, int nsplit, hipStream_t s) {
    const dim3 grid(nsplit, 1, g_wgrad_nb), blk(256);
    const size_t lds = (size_t)(RT * CT * 4 + RT) * 64 * sizeof(float);          // <= 33 KB
    const WgBatch wb = wg_batch(A, Bm, ws);
    if (sqB) k_wgrad1x1<RT, CT, true, NF><<<grid, blk, lds, s>>>(wb, MR, NR, HW, W, bsB, Q, qper, nsplit);
    else k_wgrad1x1<RT, CT, false, NF><<<grid, blk, lds, s>>>(wb, MR, NR, HW, W, bsB, Q, qper, nsplit);
    return 0;
}

template <int H, int W, int TAPS, int NT, bool WINO>
int launch_wgrad(const float* A, const float* Bm, float* gw, float* gbias, float* ws, int B, int MR, int NR, hipStream_t s,
                 int64_t bsB, int sqB) {
    constexpr int HW = H * W, KC = HW >= 64 ? HW : 64, KW = 4 / NT;
    constexpr size_t lds_main = (size_t)(KC * 33 + KC * (NT * 32 + 1)) * 4;
    constexpr size_t lds_comb = KW > 1 ? (size_t)(NT * TAPS * 1024 + NT * 64) * 4 : 0;
    constexpr size_t lds = lds_main > lds_comb ? lds_main : lds_comb;
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_wgrad<H, W, TAPS, NT, WINO>, 160 * 1024, raised, __func__)) return rc_;
    }
    const int mtiles = (MR + 31) / 32;
    const int splits = wgrad_splits(B, MR, HW, WINO ? 256 : 512);
    const int S = splits, nw = TAPS * MR * NR;
    // partials: [S][TAPS*MR*NR + MR] (weights | bias of one split contiguous: ONE reduce launch)
    k_wgrad<H, W, TAPS, NT, WINO><<<dim3(mtiles, splits, g_wgrad_nb), dim3(256), lds, s>>>(wg_batch(A, Bm, ws), B, MR, NR, bsB, sqB);
    g_wgrad_last_S = S;
    if (!g_wgrad_defer) k_wgrad_reduce<<<dim3((nw + MR + 63) / 64), dim3(256), 0, s>>>(ws, gw, gbias, nw, nw + MR, S);
    return 0;
}

// the 3x3 takes the Winograd form unless CONTEXTFLOW_DIRECT_CONV=1 (same switch as the step kernels)
static thread_local int g_wgrad_form = -1;            // test hook (cf_wgrad_form): 0 direct, 1 Winograd, -1 environment
static bool wgrad_direct_only() {
    static const bool v = [] { const char* e = getenv("CONTEXTFLOW_DIRECT_CONV"); return e && e[0] == '1'; }();
    return g_wgrad_form < 0 ? v : g_wgrad_form == 0;
}

template <int H, int W, int TAPS, int NT>
int launch_form(const float* A, const float* Bm, float* gw, float* gbias, float* ws, int B, int MR, int NR, hipStream_t s,
                int64_t bsB, int sqB) {
    // (16x16 with 128 columns would stage 160 values per thread next to the 256 accumulators: it keeps the direct form)
    if constexpr (TAPS == 9 && !(H * W == 256 && NT == 4)) {
        if (!wgrad_direct_only()) return launch_wgrad<H, W, TAPS, NT, true>(A, Bm, gw, gbias, ws, B, MR, NR, s, bsB, sqB);
    }
    return launch_wgrad<H, W, TAPS, NT, false>(A, Bm, gw, gbias, ws, B, MR, NR, s, bsB, sqB);
}

template <int H, int W, int TAPS>
int dispatch_nt(const float* A, const float* Bm, float* gw, float* gbias, float* ws, int B, int MR, int NR, hipStream_t s,
                int64_t bsB, int sqB) {
    if (NR <= 32) return launch_form<H, W, TAPS, 1>(A, Bm, gw, gbias, ws, B, MR, NR, s, bsB, sqB);
    if (NR <= 64) return launch_form<H, W, TAPS, 2>(A, Bm, gw, gbias, ws, B, MR, NR, s, bsB, sqB);
    return launch_form<H, W, TAPS, 4>(A, Bm, gw, gbias, ws, B, MR, NR, s, bsB, sqB);
}

}  // namespace

extern "C" {

// workspace for the split-K partials: [splits][taps*MR*NR + MR] floats (sized for the larger split count of the two forms)
int64_t cf_wgrad_ws_bytes(int B, int MR, int NR, int H, int W, int taps) {
    int S = wgrad_splits(B, MR, H * W);
    if (taps == 1 && wgrad1x1_ok(MR, NR, H * W)) {        // k_wgrad1x1: one partial per wave
        int Q, qper, nsplit;
        wgrad1x1_split(B, MR, NR, H * W, Q, qper, nsplit);
        if (nsplit > S) S = nsplit;
    }
    return (int64_t)S * ((int64_t)taps * MR * NR + MR) * 4;
}

// bsB: batch stride of Bm in floats; sqB (taps == 1 only): Bm is the tensor before Squeeze((2,2)), read through the index map
static int wgrad_impl(const float* A, const float* Bm, float* gw, float* gbias, void* ws, int B, int MR, int NR, int H, int W,
                      int taps, int64_t bsB, int sqB, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(A && Bm && gw && ws && MR > 0 && NR > 0 && NR <= 128 && (taps == 1 || taps == 9));
    CF_REQUIRE(bsB >= (int64_t)NR * H * W && bsB < (1 << 24) && (!sqB || (taps == 1 && NR % 4 == 0)));
    int rc;
    hipStream_t s = cf_s(stream);
    float* w = (float*)ws;
    if (taps == 1 && wgrad1x1_ok(MR, NR, H * W) && (H == W) && (H == 16 || H == 8 || H == 4) &&
        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bm)) & 15) == 0 && bsB % 4 == 0) {
        // the skinny 1x1 gradients: operands straight from global memory in MFMA layout (k_wgrad1x1)
        int Q, qper, nsplit;
        wgrad1x1_split(B, MR, NR, H * W, Q, qper, nsplit);
        const int rt = (MR + 15) / 16, ct = (NR + 15) / 16;
        rc = -1;
#define CF_W1(R, Cc, F) if (rt == R && ct == Cc && nf == F) rc = launch_wgrad1x1<R, Cc, F>(A, Bm, w, MR, NR, H * W, W, bsB, sqB, Q, qper, nsplit, s)
        const int nf = wgrad1x1_nf(H * W);
        CF_W1(1, 1, 2); CF_W1(1, 2, 2); CF_W1(2, 1, 2); CF_W1(2, 2, 2); CF_W1(2, 4, 2); CF_W1(4, 1, 2); CF_W1(4, 2, 2);
        CF_W1(1, 1, 1); CF_W1(1, 2, 1); CF_W1(2, 1, 1); CF_W1(2, 2, 1); CF_W1(2, 4, 1); CF_W1(4, 1, 1); CF_W1(4, 2, 1); CF_W1(4, 4, 1);
        CF_W1(4, 8, 1); CF_W1(8, 2, 1);
#undef CF_W1
        if (rc == 0) {
            g_wgrad_last_S = nsplit;
            const int nw = MR * NR;
            if (!g_wgrad_defer) k_wgrad_reduce<<<dim3((nw + MR + 63) / 64), dim3(256), 0, s>>>(w, gw, gbias, nw, nw + MR, nsplit);
            CF_LAUNCH_CHECK();
            return 0;
        }
    }
#define CF_W(HH, WW) rc = taps == 9 ? dispatch_nt<HH, WW, 9>(A, Bm, gw, gbias, w, B, MR, NR, s, bsB, sqB) : dispatch_nt<HH, WW, 1>(A, Bm, gw, gbias, w, B, MR, NR, s, bsB, sqB)
    if (H == 16 && W == 16) CF_W(16, 16);
    else if (H == 8 && W == 8) CF_W(8, 8);
    else if (H == 4 && W == 4) CF_W(4, 4);
    else { cf_set_error("cf_wgrad: image %dx%d unsupported", H, W); return CF_ERR_UNSUPPORTED; }
#undef CF_W
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_wgrad(const float* A, const float* Bm, float* gw, float* gbias, void* ws, int B, int MR, int NR, int H, int W,
             int taps, cf_stream_t stream) {
    return wgrad_impl(A, Bm, gw, gbias, ws, B, MR, NR, H, W, taps, (int64_t)NR * H * W, 0, stream);
}

// The four weight gradients of one flow step - NN.4 (s_gh x t_h2), NN.2 (3x3: s_gh2 x t_h1), NN.0 (s_gh1 x t_y0), the
// folded Conv1x1 / ActNorm matrix (s_gy x xs) - as four k_wgrad launches and ONE reduce launch (small batches: the step's
// backward is a chain of launches of a few microseconds each).  Same results as four cf_wgrad calls, bit for bit; gw2
// leaves in the reference's layout (2C, 2C, 3, 3) - no permute launch behind it.
int64_t cf_step_wgrads_ws_bytes(int B, int C, int H, int W) {
    const int HID = 2 * C, HALF = C / 2;
    return cf_wgrad_ws_bytes(B, C, HID, H, W, 1) + cf_wgrad_ws_bytes(B, HID, HID, H, W, 9) +
           cf_wgrad_ws_bytes(B, HID, HALF, H, W, 1) + cf_wgrad_ws_bytes(B, C, C, H, W, 1);
}

// n steps of one shape: per product ONE launch over all steps (blockIdx.z) - the Conv1x1 product once per (stride, layout) of
// the step inputs: the first step of a level reads the tensor in front of its Squeeze - and ONE reduce launch.  Each step's
// workspace is laid out as cf_step_wgrads' (cf_step_wgrads_ws_bytes); results bit for bit those of n cf_step_wgrads calls.
static int step_wgrads_impl(int n, const float* const* s_gh, const float* const* s_gh2, const float* const* s_gh1,
                            const float* const* s_gy, const float* const* t_h2, const float* const* t_h1, const float* const* t_y0,
                            const float* const* xs, float* const* gw3, float* const* gb3, float* const* gw2, float* const* gb2,
                            float* const* gw1, float* const* gb1, float* const* gwp, float* const* gbp, void* const* ws, int B, int C,
                            int H, int W, const int64_t* xs_bstride, const int* xs_unsqueezed, cf_stream_t stream) {
    if (B == 0 || n == 0) return 0;
    CF_REQUIRE(n > 0 && n <= kWgBatch && C >= 2 && C % 2 == 0 && 2 * C <= 128);
    const int HID = 2 * C, HALF = C / 2;
    const float* const* As[4] = {s_gh, s_gh2, s_gh1, s_gy};
    const float* const* Bs[4] = {t_h2, t_h1, t_y0, xs};
    float* const* gws[4] = {gw3, gw2, gw1, gwp};
    float* const* gbs[4] = {gb3, gb2, gb1, gbp};
    const int MRs[4] = {C, HID, HID, C}, NRs[4] = {HID, HID, HALF, C}, tps[4] = {1, 9, 1, 1};
    for (int i = 0; i < n; ++i)
        for (int q = 0; q < 4; ++q) CF_REQUIRE(As[q][i] && Bs[q][i] && gws[q][i] && gbs[q][i] && ws[i]);
    WgReduce4B db{};
    int64_t woff = 0;
    int nmax = 0;
    for (int q = 0; q < 4; ++q) {
        // steps that share the operand layout of this product go into one launch
        bool done[kWgBatch] = {};
        for (int i0 = 0; i0 < n; ++i0) {
            if (done[i0]) continue;
            WgBatch wb{};
            int idx[kWgBatch], m = 0;
            for (int i = i0; i < n; ++i) {
                if (done[i] || (q == 3 && (xs_bstride[i] != xs_bstride[i0] || (xs_unsqueezed[i] != 0) != (xs_unsqueezed[i0] != 0)))) continue;
                // (alignment decides the kernel inside wgrad_impl: only steps whose operands agree with the leader's share a launch)
                if ((((reinterpret_cast<uintptr_t>(As[q][i]) | reinterpret_cast<uintptr_t>(Bs[q][i])) & 15) == 0) !=
                    (((reinterpret_cast<uintptr_t>(As[q][i0]) | reinterpret_cast<uintptr_t>(Bs[q][i0])) & 15) == 0)) continue;
                done[i] = true; idx[m] = i;
                wb.A[m] = As[q][i]; wb.Bm[m] = Bs[q][i]; wb.part[m] = (float*)((char*)ws[i] + woff);
                ++m;
            }
            g_wgrad_defer = true; g_wgrad_batch = &wb; g_wgrad_nb = m;
            const int rc = wgrad_impl(wb.A[0], wb.Bm[0], gws[q][idx[0]], gbs[q][idx[0]], wb.part[0], B, MRs[q], NRs[q], H, W, tps[q],
                                      q == 3 ? xs_bstride[i0] : (int64_t)NRs[q] * H * W, q == 3 && xs_unsqueezed[i0] != 0, stream);
            g_wgrad_defer = false; g_wgrad_batch = nullptr; g_wgrad_nb = 1;
            if (rc) return rc;
            const int nw = tps[q] * MRs[q] * NRs[q];
            for (int j = 0; j < m; ++j) {
                WgReduce4& d = db.d[idx[j]];
                d.part[q] = wb.part[j]; d.out0[q] = gws[q][idx[j]]; d.out1[q] = gbs[q][idx[j]];
                d.n0[q] = nw; d.n[q] = nw + MRs[q]; d.S[q] = g_wgrad_last_S; d.taps[q] = tps[q];
            }
            nmax = nw + MRs[q] > nmax ? nw + MRs[q] : nmax;
        }
        woff += cf_wgrad_ws_bytes(B, MRs[q], NRs[q], H, W, tps[q]);
    }
    k_wgrad_reduce4<<<dim3((nmax + 63) / 64, 4, n), dim3(256), 0, cf_s(stream)>>>(db);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_step_wgrads(const float* s_gh, const float* s_gh2, const float* s_gh1, const float* s_gy, const float* t_h2,
                   const float* t_h1, const float* t_y0, const float* xs, float* gw3, float* gb3, float* gw2, float* gb2,
                   float* gw1, float* gb1, float* gwp, float* gbp, void* ws, int B, int C, int H, int W, int64_t xs_bstride,
                   int xs_unsqueezed, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(s_gh && s_gh2 && s_gh1 && s_gy && t_h2 && t_h1 && t_y0 && xs && gw3 && gb3 && gw2 && gb2 && gw1 && gb1 && gwp && gbp && ws);
    return step_wgrads_impl(1, &s_gh, &s_gh2, &s_gh1, &s_gy, &t_h2, &t_h1, &t_y0, &xs, &gw3, &gb3, &gw2, &gb2, &gw1, &gb1, &gwp, &gbp,
                            &ws, B, C, H, W, &xs_bstride, &xs_unsqueezed, stream);
}

int cf_step_wgrads_batch(int n, const float* const* s_gh, const float* const* s_gh2, const float* const* s_gh1,
                         const float* const* s_gy, const float* const* t_h2, const float* const* t_h1, const float* const* t_y0,
                         const float* const* xs, float* const* gw3, float* const* gb3, float* const* gw2, float* const* gb2,
                         float* const* gw1, float* const* gb1, float* const* gwp, float* const* gbp, void* const* ws, int B, int C,
                         int H, int W, const int64_t* xs_bstride, const int* xs_unsqueezed, cf_stream_t stream) {
    CF_REQUIRE(n >= 0 && s_gh && s_gh2 && s_gh1 && s_gy && t_h2 && t_h1 && t_y0 && xs && gw3 && gb3 && gw2 && gb2 && gw1 && gb1 && gwp &&
               gbp && ws && xs_bstride && xs_unsqueezed);
    for (int i0 = 0; i0 < n; i0 += kWgBatch) {
        const int m = n - i0 < kWgBatch ? n - i0 : kWgBatch;
        if (int rc = step_wgrads_impl(m, s_gh + i0, s_gh2 + i0, s_gh1 + i0, s_gy + i0, t_h2 + i0, t_h1 + i0, t_y0 + i0, xs + i0, gw3 + i0,
                                      gb3 + i0, gw2 + i0, gb2 + i0, gw1 + i0, gb1 + i0, gwp + i0, gbp + i0, ws + i0, B, C, H, W,
                                      xs_bstride + i0, xs_unsqueezed + i0, stream)) return rc;
    }
    return 0;
}

// multiply-adds per sample the matrix pipe executes for the four weight gradients of a step (bench.py): direct 40 C^2 HW;
// the Winograd form F(3x3, 2x2) of the 3x3 (launch_form: every shape but 16x16 with 128 columns) runs 16 instead of 36 C^2
int64_t cf_step_wgrads_macs(int B, int C, int H, int W) {
    (void)B;
    const bool wino = !wgrad_direct_only() && !(H * W == 256 && 2 * C > 64);
    return (wino ? 20ll : 40ll) * C * C * H * W;
}

// test hook (not part of the public header): cf_wgrad with the form of the 3x3 chosen by the caller (0 direct, 1 Winograd)
int cf_wgrad_form(const float* A, const float* Bm, float* gw, float* gbias, void* ws, int B, int MR, int NR, int H, int W,
                  int taps, int form, cf_stream_t stream) {
    g_wgrad_form = form ? 1 : 0;
    const int rc = cf_wgrad(A, Bm, gw, gbias, ws, B, MR, NR, H, W, taps, stream);
    g_wgrad_form = -1;
    return rc;
}

}  // extern "C"
