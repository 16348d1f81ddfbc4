// SimpleViT conditioner of TransCoupling (contextflow/layers/simple_vit.py:18-127) on gfx950.
//
// The dense contractions (patch embedding, qkv, out-projection, MLP) run on the exact-fp32 matrix
// cores (v_mfma_f32_32x32x2_f32): rows = samples x tokens, K, N <= a few hundred.  One workgroup
// owns 128 rows x (<= 96) output features; X and the W slice go through LDS in K-chunks of 16 with
// an odd row stride, so both MFMA operand reads (lane -> row, lane>>5 -> k) are bank-conflict free.
// LayerNorm, the tiny (tokens x tokens) attention and the patch index maps are VALU kernels.
#include "cf_common.h"
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LIN_ROWS = 128;     // rows per workgroup (4 waves x 32)
constexpr int LIN_KC = 16;        // K chunk staged in LDS per pass
constexpr int LIN_KP = LIN_KC + 1;   // odd row stride: conflict-free operand reads (lane -> row)

// y[r, n] = act(sum_k x[r,k] W[n,k] + bias[n]) + res[r,n]
// ACT: 0 none, 1 exact GELU (erf), 2 ReLU             simple_vit.py:32-38,52-53; coupling.py:37 (CN nets)
// K is walked in chunks of 16 through a 15 KiB LDS stage (any K; 5 workgroups per CU overlap each other's staging
// and MFMA phases - 32- and 64-wide chunks measured slower); the next chunk's global loads are issued into registers
// before the MFMAs of the current one.
// WT: the weight is given as (K, N) row-major - element (n, k) = Wt[k N + n] - i.e. the nn.Linear weight of the layer whose
// BACKWARD this is (gx = gy W): staged with lanes along n (coalesced), same LDS image, same summation order.
template <int ACT, int NTL, bool VEC, bool WT = false>
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ x, const float* __restrict__ Wt,
                                                const float* __restrict__ bias, const float* __restrict__ res,
                                                float* __restrict__ y, int rows, int K, int N) {
    __shared__ float xs[LIN_ROWS * LIN_KP];
    constexpr int LIN_COLS = 32 * NTL;                 // output features per workgroup (NTL MFMA column tiles)
    __shared__ float ws[LIN_COLS * LIN_KP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * LIN_ROWS, n0 = blockIdx.y * LIN_COLS;
    // staging.  VEC (K % 4 == 0): a thread fetches 16 bytes = 4 consecutive k of a row (rows sr + RSTEP i); otherwise one
    // float per load.  Branch-free: clamped 32-bit offsets from the workgroup's base pointers (out-of-range positions
    // are replaced by zeros when the values go to LDS), all loads of a chunk in flight together.
    constexpr int EPL = VEC ? 4 : 1;                     // floats per load
    constexpr int RSTEP = 256 * EPL / LIN_KC;            // rows covered by one load of the 256 threads
    constexpr int NXI = LIN_ROWS / RSTEP, NWI = (LIN_COLS + RSTEP - 1) / RSTEP;
    const int sk = (tid % (LIN_KC / EPL)) * EPL, sr = tid / (LIN_KC / EPL);
    const float* __restrict__ xb = x + (int64_t)r0 * K;
    const float* __restrict__ wbp = WT ? Wt + n0 : Wt + (int64_t)n0 * K;
    constexpr int NWT = LIN_COLS * LIN_KC / 256;         // WT: elements per thread (n fastest)
    static_assert(!WT || (LIN_COLS * LIN_KC) % 256 == 0, "WT staging covers the tile exactly");
    float wrt[WT ? NWT : 1];
    const int rmax = rows - 1 - r0, nmax = N - 1 - n0;
    int xo[NXI], wo[NWI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) xo[i] = min(sr + RSTEP * i, rmax) * K;
#pragma unroll
    for (int i = 0; i < NWI; ++i) wo[i] = min(sr + RSTEP * i, nmax) * K;
    float xr[NXI][EPL], wr[NWI][EPL];
    auto fetch = [&](int k0) {
        const int kc = min(k0 + sk, K - EPL);
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            if constexpr (VEC) { const float4 v = *reinterpret_cast<const float4*>(xb + xo[i] + kc); xr[i][0] = v.x; xr[i][1] = v.y; xr[i][2] = v.z; xr[i][3] = v.w; }
            else xr[i][0] = xb[xo[i] + kc];
        }
        if constexpr (WT) {
#pragma unroll
            for (int i = 0; i < NWT; ++i) {
                const int e = tid + 256 * i, nl = e % LIN_COLS, kl = e / LIN_COLS;
                wrt[i] = wbp[(int64_t)min(k0 + kl, K - 1) * N + min(nl, nmax)];
            }
        } else {
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            if constexpr (VEC) { const float4 v = *reinterpret_cast<const float4*>(wbp + wo[i] + kc); wr[i][0] = v.x; wr[i][1] = v.y; wr[i][2] = v.z; wr[i][3] = v.w; }
            else wr[i][0] = wbp[wo[i] + kc];
        }
        }
    };
    f32x16 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    const float* xa = xs + (wave * 32 + li) * LIN_KP + lk;     // A[i = row][k]
    const float* wb = ws + li * LIN_KP + lk;                    // B[k][j = feature]
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += LIN_KC) {
        const bool kin = k0 + sk < K;                     // masks are applied here, after the loads have landed
#pragma unroll
        for (int i = 0; i < NXI; ++i)
#pragma unroll
            for (int j = 0; j < EPL; ++j) xs[(sr + RSTEP * i) * LIN_KP + sk + j] = (kin && sr + RSTEP * i <= rmax) ? xr[i][j] : 0.f;
        if constexpr (WT) {
#pragma unroll
            for (int i = 0; i < NWT; ++i) {
                const int e = tid + 256 * i, nl = e % LIN_COLS, kl = e / LIN_COLS;
                ws[nl * LIN_KP + kl] = (k0 + kl < K && nl <= nmax) ? wrt[i] : 0.f;
            }
        } else {
#pragma unroll
        for (int i = 0; i < NWI; ++i)
#pragma unroll
            for (int j = 0; j < EPL; ++j)
                if (sr + RSTEP * i < LIN_COLS) ws[(sr + RSTEP * i) * LIN_KP + sk + j] = (kin && sr + RSTEP * i <= nmax) ? wr[i][j] : 0.f;
        }
        __syncthreads();
        if (k0 + LIN_KC < K) fetch(k0 + LIN_KC);
#pragma unroll
        for (int kk = 0; kk < LIN_KC; kk += 2) {
            const float a = xa[kk];
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const float b = wb[t * 32 * LIN_KP + kk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D[i][j]: lane holds column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
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

// ---- grouped Linears over one batch (the CN nets of a specialist flow, layers/specialist.py) ---------------------------------------
// A specialist flow runs five small Linears and three context encodings per flow step, all functions of the context alone: 36
// encodings + 60 Linears of 5 - 25 us in the cifar10 flow (profiles/r3_spec_fwd_b32768_kstats_final.txt).  Here blockIdx.z walks up to
// kLinGroup problems of one launch: y_g = act_g(x_g W_g^T + b_g), the same rows and K for every problem, N / activation per problem
// (workgroups beyond a problem's N leave at once).  ENC: x_g is not read but FORMED while it is staged - the uniform
// dequantisation of the integer context (k_ctx_encode's arithmetic: (code + u_g) / qbins_g, one-hot code or the context itself), so
// the encoder outputs never exist in memory.  Tile, staging and summation order are k_linear's.
constexpr int kLinGroup = 48;
struct LinGroup {
    const float* x[kLinGroup];          // (rows, K) activations; ENC: the encoder's uniforms u (rows, K)
    const float* q[kLinGroup];          // ENC: qbins (K)
    float* c[kLinGroup];                // ENC: where the code itself goes too (rows, K), or null (training keeps it for the backward)
    const float* W[kLinGroup];          // (N, K)
    const float* b[kLinGroup];          // (N) or null
    float* y[kLinGroup];                // (rows, N)
    int N[kLinGroup];
    int act[kLinGroup];                 // 0 none, 2 ReLU
};
template <bool VEC, bool ENC>
__global__ __launch_bounds__(256) void k_linear_group(const LinGroup pg, const int64_t* __restrict__ ctx, const int64_t* __restrict__ card,
                                                      int nctx, int onehot, int rows, int K) {
    constexpr int NTL = 3, LIN_COLS = 32 * NTL;
    const int g = blockIdx.z, N = pg.N[g];
    const int r0 = blockIdx.x * LIN_ROWS, n0 = blockIdx.y * LIN_COLS;
    if (n0 >= N) return;
    __shared__ float xs[LIN_ROWS * LIN_KP];
    __shared__ float ws[LIN_COLS * LIN_KP];
    const float* __restrict__ x = pg.x[g];
    const float* __restrict__ qb = pg.q[g];
    float* __restrict__ cout = pg.c[g];
    const float* __restrict__ Wt = pg.W[g];
    const float* __restrict__ bias = pg.b[g];
    float* __restrict__ y = pg.y[g];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int EPL = VEC ? 4 : 1;
    constexpr int RSTEP = 256 * EPL / LIN_KC;
    constexpr int NXI = LIN_ROWS / RSTEP, NWI = (LIN_COLS + RSTEP - 1) / RSTEP;
    const int sk = (tid % (LIN_KC / EPL)) * EPL, sr = tid / (LIN_KC / EPL);
    const float* __restrict__ xb = x + (int64_t)r0 * K;
    const float* __restrict__ wbp = Wt + (int64_t)n0 * K;
    const int rmax = rows - 1 - r0, nmax = N - 1 - n0;
    int xo[NXI], wo[NWI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) xo[i] = min(sr + RSTEP * i, rmax) * K;
#pragma unroll
    for (int i = 0; i < NWI; ++i) wo[i] = min(sr + RSTEP * i, nmax) * K;
    float xr[NXI][EPL], wr[NWI][EPL];
    auto fetch = [&](int k0) {
        const int kc = min(k0 + sk, K - EPL);
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            if constexpr (VEC) { const float4 v = *reinterpret_cast<const float4*>(xb + xo[i] + kc); xr[i][0] = v.x; xr[i][1] = v.y; xr[i][2] = v.z; xr[i][3] = v.w; }
            else xr[i][0] = xb[xo[i] + kc];
            if constexpr (ENC) {          // (code + u) / qbins, dequantize.py:55-64 - the arithmetic of k_ctx_encode
                const int64_t* __restrict__ crow = ctx + (int64_t)(r0 + min(sr + RSTEP * i, rmax)) * nctx;
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int k = kc + j;
                    float code;
                    if (onehot) {
                        int v = 0, off = 0;
                        while (v < nctx - 1 && k >= off + (int)card[v]) { off += (int)card[v]; ++v; }
                        code = (crow[v] == (int64_t)(k - off)) ? 1.f : 0.f;
                    } else {
                        code = (float)crow[k];
                    }
                    xr[i][j] = (code + xr[i][j]) / qb[k];
                }
                // the code itself, for callers that need it (first column block only; clamped rows / k positions repeat a neighbour's
                // store with the same values)
                if (cout != nullptr && n0 == 0) {
                    float* cp = cout + (int64_t)(r0 + min(sr + RSTEP * i, rmax)) * K + kc;
#pragma unroll
                    for (int j = 0; j < EPL; ++j) cp[j] = xr[i][j];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            if constexpr (VEC) { const float4 v = *reinterpret_cast<const float4*>(wbp + wo[i] + kc); wr[i][0] = v.x; wr[i][1] = v.y; wr[i][2] = v.z; wr[i][3] = v.w; }
            else wr[i][0] = wbp[wo[i] + kc];
        }
    };
    f32x16 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    const float* xa = xs + (wave * 32 + li) * LIN_KP + lk;
    const float* wb = ws + li * LIN_KP + lk;
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += LIN_KC) {
        const bool kin = k0 + sk < K;
#pragma unroll
        for (int i = 0; i < NXI; ++i)
#pragma unroll
            for (int j = 0; j < EPL; ++j) xs[(sr + RSTEP * i) * LIN_KP + sk + j] = (kin && sr + RSTEP * i <= rmax) ? xr[i][j] : 0.f;
#pragma unroll
        for (int i = 0; i < NWI; ++i)
#pragma unroll
            for (int j = 0; j < EPL; ++j)
                if (sr + RSTEP * i < LIN_COLS) ws[(sr + RSTEP * i) * LIN_KP + sk + j] = (kin && sr + RSTEP * i <= nmax) ? wr[i][j] : 0.f;
        __syncthreads();
        if (k0 + LIN_KC < K) fetch(k0 + LIN_KC);
#pragma unroll
        for (int kk = 0; kk < LIN_KC; kk += 2) {
            const float a = xa[kk];
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const float b = wb[t * 32 * LIN_KP + kk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const bool relu = pg.act[g] == 2;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int n = n0 + t * 32 + li;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row >= rows) continue;
            float v = acc[t][r] + bv;
            if (relu) v = fmaxf(v, 0.f);
            y[(int64_t)row * N + n] = v;
        }
    }
}

// Weight / bias gradient of a Linear over row-major activations (backward of k_linear; simple_vit.py's nn.Linear):
//   gW[n][k] = sum_r gy[r][n] x[r][k],   gb[n] = sum_r gy[r][n]        (rows = samples x tokens: 1e5 .. 1e6)
// A split-K GEMM whose reduction dimension is the ROW index: both MFMA operands come straight from row-major LDS tiles
// with lanes along the feature dimension (A[i = n][k = r] = gy[r][n], B[k = r][j = kcol] = x[r][kcol]) - no transposes.
// The bias gradient rides along as column K of x (ones).  The NT x KT output tiles (32 x 32) are dealt round-robin
// to the 4 waves (<= WG_TPW each, a block of 4 WG_TPW tiles per blockIdx.y); blockIdx.x walks 32-row chunks with a
// grid stride and leaves ONE partial per workgroup; k_linear_wgrad_reduce sums them in a fixed order.
constexpr int WG_RC = 32;         // rows per staged chunk
constexpr int WG_TPW = 8;         // output tiles per wave
constexpr int WG_MAXF = 12;       // 32-feature column blocks of gy and [x | 1] together
constexpr int WG_LS = WG_MAXF * 32;   // LDS row stride (floats): 48 KiB per workgroup

// TPW = tiles per wave of this launch (1..WG_TPW): slots past the last tile recompute tile 0 and are not stored - a
// branch around an MFMA would put a full LDS wait in front of every one of them.
// ldx / ldy: row strides of x / gy (a launch may cover a column block of wider matrices); XSQ: the B operand is x^2
// (second-moment sums of the mixture backward).
template <int TPW, bool XSQ>
__device__ __forceinline__ void linear_wgrad_body(const float* __restrict__ x, const float* __restrict__ gy,
                                                  float* __restrict__ part, int rows, int Ktot, int N, int NT, int kbs, int ybase,
                                                  int64_t ldx, int64_t ldy, int64_t zstride, int bx, int by, int bz, int gdx) {
    // bz = column block of a wide x (kbs columns each; small batches run all blocks in one launch)
    const int K = min(kbs, Ktot - bz * kbs), KT = (K + 1 + 31) / 32;
    x += (int64_t)bz * kbs;
    part += (int64_t)bz * zstride;
    // one LDS row = the 32-feature blocks [gy (NT) | x, 1, 0.. (KT)] of one activation row at the FIXED stride WG_LS: the
    // operand reads of the MFMA loop are then base register + immediate offset (no address arithmetic between MFMAs)
    __shared__ float lds[WG_RC * WG_LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int NS = NT * 32, ntiles = NT * KT, t0 = (ybase + by) * 4 * WG_TPW;
    int aoff[TPW], boff[TPW];
    bool live[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = t0 + wave + 4 * i;
        live[i] = t < ntiles;
        const int tt = live[i] ? t : 0;
        aoff[i] = lk * WG_LS + (tt / KT) * 32 + li;
        boff[i] = lk * WG_LS + NS + (tt % KT) * 32 + li;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // staging: a thread owns the 32-feature column c32 of rows (tid>>5) + 8 i.  Two workgroups per CU overlap each
    // other's staging and MFMA phases.
    const int nchunks = (rows + WG_RC - 1) / WG_RC;
    const int c32 = tid & 31, rs = tid >> 5;
    // 16-byte staging when every row of both operands is a whole number of aligned float4 (the planes of the transformer
    // step backward: 52 / 64 / 192 floats per row): a thread takes 4 consecutive features of ONE row per 32-feature block -
    // a quarter of the load instructions, the same LDS image.  (wave-uniform; x^2 sums keep the scalar form)
    const bool vec4 = !XSQ && (N & 3) == 0 && (K & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
    for (int c = bx; c < nchunks; c += gdx) {
        const int r0 = c * WG_RC;
        if (vec4) {
            const int row = tid >> 3, c4 = (tid & 7) * 4;
            const bool ok = r0 + row < rows;
            const int64_t rc = ok ? r0 + row : rows - 1;
            float4 st4[WG_MAXF];
#pragma unroll
            for (int cb = 0; cb < WG_MAXF; ++cb) {
                const bool isg = cb < NT;                                         // uniform
                const int col = cb * 32 + c4 - (isg ? 0 : NS);
                const int lim = isg ? N : K;
                st4[cb] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cb < NT + KT && col < lim) st4[cb] = *reinterpret_cast<const float4*>((isg ? gy + rc * ldy : x + rc * ldx) + col);
            }
#pragma unroll
            for (int cb = 0; cb < WG_MAXF; ++cb) {
                const bool isg = cb < NT;
                const int col = cb * 32 + c4 - (isg ? 0 : NS);
                float4 v = st4[cb];
                if (isg) { if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f); }        // rows past the end contribute nothing
                else if (col == K) v.x = 1.f;                                     // the bias column (K is a multiple of 4)
                if (cb < NT + KT) *reinterpret_cast<float4*>(&lds[row * WG_LS + cb * 32 + c4]) = v;
            }
        } else {
        // branch-free batch: all 4 x WG_MAXF loads of a thread are in flight together (clamped addresses; the values of
        // padding positions are replaced afterwards), then the LDS writes
        float stg[WG_RC / 8][WG_MAXF];
#pragma unroll
        for (int i = 0; i < WG_RC / 8; ++i) {
            const int r = r0 + rs + 8 * i;
            const int64_t rc = r < rows ? r : rows - 1;
#pragma unroll
            for (int cb = 0; cb < WG_MAXF; ++cb) {
                const bool isg = cb < NT;                                         // uniform
                const int col = cb * 32 + c32 - (isg ? 0 : NS);                   // n, or k
                const float* src = isg ? gy + rc * ldy + min(col, N - 1) : x + rc * ldx + min(col, K - 1);
                stg[i][cb] = cb < NT + KT ? *src : 0.f;                           // (uniform) blocks past [gy | x, 1] are never read
                if (XSQ && !isg) stg[i][cb] *= stg[i][cb];
            }
        }
#pragma unroll
        for (int i = 0; i < WG_RC / 8; ++i) {
            const int r = rs + 8 * i;
            const bool ok = r0 + r < rows;
#pragma unroll
            for (int cb = 0; cb < WG_MAXF; ++cb) {
                const bool isg = cb < NT;
                const int col = cb * 32 + c32 - (isg ? 0 : NS);
                const float v = isg ? ((ok && col < N) ? stg[i][cb] : 0.f)        // rows past the end contribute nothing
                                    : (col < K ? stg[i][cb] : (col == K ? 1.f : 0.f));
                if (cb < NT + KT) lds[r * WG_LS + cb * 32 + c32] = v;
            }
        }
        }
        __syncthreads();
        // operands of row pair rr + 2 are requested before the MFMAs of row pair rr (two register sets)
        float oa[2][TPW], ob[2][TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) { oa[0][i] = lds[aoff[i]]; ob[0][i] = lds[boff[i]]; }
#pragma unroll
        for (int rr = 0; rr < WG_RC; rr += 2) {
            const int cur = (rr >> 1) & 1;
            if (rr + 2 < WG_RC) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) { oa[cur ^ 1][i] = lds[aoff[i] + (rr + 2) * WG_LS]; ob[cur ^ 1][i] = lds[boff[i] + (rr + 2) * WG_LS]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[cur][i], ob[cur][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // partial of this workgroup: full 32 x 32 tiles, [tile][i = n][j = kcol]
    float* pw = part + (int64_t)bx * ntiles * 1024;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        if (!live[i]) continue;
        float* pt = pw + (int64_t)(t0 + wave + 4 * i) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + li] = acc[i][r];
    }
}

template <int TPW, bool XSQ = false>
__global__ __launch_bounds__(256) void k_linear_wgrad(const float* __restrict__ x, const float* __restrict__ gy,
                                                      float* __restrict__ part, int rows, int Ktot, int N, int NT, int kbs, int ybase,
                                                      int64_t ldx, int64_t ldy, int64_t zstride) {
    linear_wgrad_body<TPW, XSQ>(x, gy, part, rows, Ktot, N, NT, kbs, ybase, ldx, ldy, zstride, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// 64 output elements per workgroup, the G partials split over the 4 waves (every 4th partial each), then summed across
// the waves in a fixed order
__device__ __forceinline__ void linear_wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ gW,
                                                         float* __restrict__ gb, int Ktot, int N, int kbs, int NT, int G,
                                                         int64_t ldw, int64_t zstride, int bx, int z) {
    __shared__ float red[4][64];
    // z: column block of x, as in k_linear_wgrad
    const int K = min(kbs, Ktot - z * kbs), KT = (K + 1 + 31) / 32, ntiles = NT * KT;
    if (bx * 64 >= ntiles * 1024) return;    // uniform: the last block may have fewer tiles
    gW += (int64_t)z * kbs;
    if (z != 0) gb = nullptr;                             // the bias column rides with the first block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = bx * 64 + lane;                 // element of the padded tile storage
    const float* p = part + (int64_t)z * zstride + e;
    float s0 = 0.f, s1 = 0.f;
    int g = wave;
    for (; g + 4 < G; g += 8) { s0 += p[(int64_t)g * ntiles * 1024]; s1 += p[(int64_t)(g + 4) * ntiles * 1024]; }
    if (g < G) s0 += p[(int64_t)g * ntiles * 1024];
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0) {
        const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int t = e >> 10, n = (t / KT) * 32 + ((e >> 5) & 31), k = (t % KT) * 32 + (e & 31);
        if (n < N) {
            if (k < K) gW[(int64_t)n * ldw + k] = v;
            else if (k == K && gb) gb[n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_linear_wgrad_reduce(const float* __restrict__ part, float* __restrict__ gW,
                                                             float* __restrict__ gb, int Ktot, int N, int kbs, int NT, int G,
                                                             int64_t ldw, int64_t zstride) {
    linear_wgrad_reduce_body(part, gW, gb, Ktot, N, kbs, NT, G, ldw, zstride, blockIdx.x, blockIdx.y);
}
// LayerNorm over the last dim (biased variance, eps) + optional positional embedding add:
// y[r, :] = LN(x[r, :]) * w + b (+ pe[r % ntok, :]).  16 lanes per row.   simple_vit.py:33,50,74,104-106,122
// CACHED (dim <= 256): the row is read once into registers (16 values per lane, all loads in flight together).
template <bool CACHED>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ b, const float* __restrict__ pe,
                                                   float* __restrict__ y, int rows, int dim, int ntok, float eps) {
    const int g = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = row < rows;
    const float* xr = x + (ok ? row : 0) * dim;
    float xv[CACHED ? 16 : 1];
    float s = 0.f;
    if constexpr (CACHED) {
#pragma unroll
        for (int i = 0; i < 16; ++i) xv[i] = (g + 16 * i < dim) ? xr[g + 16 * i] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += xv[i];
    } else {
        for (int j = g; j < dim; j += 16) s += xr[j];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)dim;
    float v = 0.f;
    if constexpr (CACHED) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = (g + 16 * i < dim) ? xv[i] - mean : 0.f; xv[i] = d; v = fmaf(d, d, v); }
    } else {
        for (int j = g; j < dim; j += 16) { const float d = xr[j] - mean; v = fmaf(d, d, v); }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const float rstd = 1.0f / sqrtf(v / (float)dim + eps);
    if (!ok) return;
    const float* per = pe ? pe + (row % ntok) * dim : nullptr;
    if constexpr (CACHED) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = g + 16 * i;
            if (j < dim) {
                float o = xv[i] * rstd * w[j] + b[j];
                if (per) o += per[j];
                y[row * dim + j] = o;
            }
        }
    } else {
        for (int j = g; j < dim; j += 16) {
            float o = (xr[j] - mean) * rstd * w[j] + b[j];
            if (per) o += per[j];
            y[row * dim + j] = o;
        }
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
    // staging in batches of 8 independent 16-byte loads per thread (one dependent load -> store round trip per element
    // was the whole cost of this kernel); 3 dh is a multiple of 4 and the rows of qkv are 16-byte aligned
    {
        const int q4 = 3 * dh / 4, total = N * q4;                    // float4 per row, per sample
        const float4* src4 = reinterpret_cast<const float4*>(src);
        for (int e0 = tid; e0 < total; e0 += 8 * nt) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src4[min(e0 + i * nt, total - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * nt;
                if (e < total) {
                    const int r = e / q4, c = 4 * (e - r * q4);
                    float* d = s_qkv + r * RS + c;
                    d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
                }
            }
        }
    }
    __syncthreads();
    // q k^T: a thread owns key j and 4 query rows - one k read feeds 4 FMAs (the q reads are wave broadcasts)
    const int NQ = (N + 3) >> 2;
    for (int e = tid; e < NQ * N; e += nt) {
        const int iq = e / N, j = e - iq * N, i0 = iq * 4;
        const float* k = s_qkv + j * RS + dh;
        const float* q0 = s_qkv + i0 * RS;
        const float* q1 = s_qkv + min(i0 + 1, N - 1) * RS;
        const float* q2 = s_qkv + min(i0 + 2, N - 1) * RS;
        const float* q3 = s_qkv + min(i0 + 3, N - 1) * RS;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int d = 0; d < dh; ++d) {
            const float kv = k[d];
            a0 = fmaf(q0[d], kv, a0); a1 = fmaf(q1[d], kv, a1); a2 = fmaf(q2[d], kv, a2); a3 = fmaf(q3[d], kv, a3);
        }
        dots[i0 * N + j] = a0 * scale;
        if (i0 + 1 < N) dots[(i0 + 1) * N + j] = a1 * scale;
        if (i0 + 2 < N) dots[(i0 + 2) * N + j] = a2 * scale;
        if (i0 + 3 < N) dots[(i0 + 3) * N + j] = a3 * scale;
    }
    __syncthreads();
    // row softmax: 8 lanes per row (columns j = l, l + 8, ...), max / sum through xor shuffles inside the group
    for (int i = tid >> 3; i < N; i += nt >> 3) {
        const int l = tid & 7;
        float* row = dots + i * N;
        float mx = -INFINITY;
        for (int j = l; j < N; j += 8) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int j = l; j < N; j += 8) { const float e = expf(row[j] - mx); row[j] = e; sum += e; }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = 1.0f / sum;
        for (int j = l; j < N; j += 8) row[j] *= inv;
    }
    __syncthreads();
    // P v: a thread owns feature d of 4 output rows - one v read feeds 4 FMAs
    for (int e = tid; e < NQ * dh; e += nt) {
        const int iq = e / dh, d = e - iq * dh, i0 = iq * 4;
        const float* p0 = dots + i0 * N;
        const float* p1 = dots + min(i0 + 1, N - 1) * N;
        const float* p2 = dots + min(i0 + 2, N - 1) * N;
        const float* p3 = dots + min(i0 + 3, N - 1) * N;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int j = 0; j < N; ++j) {
            const float vv = s_qkv[j * RS + 2 * dh + d];
            a0 = fmaf(p0[j], vv, a0); a1 = fmaf(p1[j], vv, a1); a2 = fmaf(p2[j], vv, a2); a3 = fmaf(p3[j], vv, a3);
        }
        float* o = out + ((int64_t)b * N + i0) * dh + d;
        o[0] = a0;
        if (i0 + 1 < N) o[dh] = a1;
        if (i0 + 2 < N) o[2 * dh] = a2;
        if (i0 + 3 < N) o[3 * dh] = a3;
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
    // column tiles per workgroup: 3 (96 features).  5 / 6 tiles (N = 152 / 192 in one pass, no padded MFMA work) were
    // measured: faster at 65 K rows, slower at 147 K rows and in the ATM forward end to end - fewer, longer workgroups
    // quantise worse over the 256 CUs (a sweep over sizes)
    constexpr int ntl = 3;
    const int nt = (N + 31) / 32;
    dim3 grid((rows + LIN_ROWS - 1) / LIN_ROWS, (nt + ntl - 1) / ntl);
    hipStream_t st = cf_s(stream);
    const bool vec = K % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(Wt)) & 15) == 0;
#define CF_LIN(A, T, V) k_linear<A, T, V><<<grid, dim3(256), 0, st>>>(x, Wt, bias, res, y, rows, K, N)
#define CF_LIN_A(T, V) (act == 0 ? CF_LIN(0, T, V) : (act == 1 ? CF_LIN(1, T, V) : CF_LIN(2, T, V)))
    if (vec) CF_LIN_A(3, true);
    else CF_LIN_A(3, false);
#undef CF_LIN_A
#undef CF_LIN
    CF_LAUNCH_CHECK();
    return 0;
}

// n Linears over the same rows in one launch (k_linear_group): y_g = act_g(x_g W_g^T + b_g), x_g (rows, K), W_g (N_g, K), b_g (N_g) or
// null, act_g 0 | 2 (ReLU).  With ctx != null the inputs are FORMED from the integer context (rows, nctx) instead of read:
// x_g[r, k] = (code(ctx[r], k) + u_g[r, k]) / qbins_g[k] - x[] then holds the uniforms u_g and q[] the encoders' qbins; onehot != 0:
// code = the concatenated one-hot code with cardinalities card (device int64, nctx entries), else the context itself (K == nctx);
// c_out (or null; entries may be null): the codes x_g themselves are written there as well (rows, K).
int cf_linear_group(int n, const float* const* x, const float* const* q, const float* const* W, const float* const* b, float* const* y,
                    float* const* c_out, const int* N, const int* act, const int64_t* ctx, const int64_t* card, int nctx, int onehot,
                    int rows, int K, cf_stream_t stream) {
    if (n == 0 || rows == 0) return 0;
    CF_REQUIRE(n > 0 && x && W && b && y && N && act && rows > 0 && K > 0 && (!ctx || (q && nctx > 0 && (onehot ? card != nullptr : K == nctx))));
    hipStream_t st = cf_s(stream);
    for (int i0 = 0; i0 < n; i0 += kLinGroup) {
        const int m = n - i0 < kLinGroup ? n - i0 : kLinGroup;
        LinGroup pg{};
        int nmax = 0;
        bool vec = K % 4 == 0;
        for (int i = 0; i < m; ++i) {
            const int j = i0 + i;
            CF_REQUIRE(x[j] && W[j] && y[j] && N[j] > 0 && (act[j] == 0 || act[j] == 2) && (!ctx || q[j]));
            pg.x[i] = x[j]; pg.q[i] = ctx ? q[j] : nullptr; pg.c[i] = (ctx && c_out) ? c_out[j] : nullptr; pg.W[i] = W[j]; pg.b[i] = b[j]; pg.y[i] = y[j]; pg.N[i] = N[j]; pg.act[i] = act[j];
            nmax = N[j] > nmax ? N[j] : nmax;
            vec = vec && ((reinterpret_cast<uintptr_t>(x[j]) | reinterpret_cast<uintptr_t>(W[j])) & 15) == 0;
        }
        dim3 grid((rows + LIN_ROWS - 1) / LIN_ROWS, (nmax + 95) / 96, m);
        if (ctx) {
            if (vec) k_linear_group<true, true><<<grid, dim3(256), 0, st>>>(pg, ctx, card, nctx, onehot, rows, K);
            else k_linear_group<false, true><<<grid, dim3(256), 0, st>>>(pg, ctx, card, nctx, onehot, rows, K);
        } else {
            if (vec) k_linear_group<true, false><<<grid, dim3(256), 0, st>>>(pg, nullptr, nullptr, 0, 0, rows, K);
            else k_linear_group<false, false><<<grid, dim3(256), 0, st>>>(pg, nullptr, nullptr, 0, 0, rows, K);
        }
        CF_LAUNCH_CHECK();
    }
    return 0;
}

// gx = gy W for an nn.Linear weight W (N_out, K_in) as stored: y[r, n] = sum_k x[r, k] W[k, n] - the data gradient of
// cf_linear without a transposed copy of the weight (K = the layer's N_out, N = its K_in)
int cf_linear_tn(const float* x, const float* W, float* y, int rows, int K, int N, cf_stream_t stream) {
    if (rows == 0) return 0;
    CF_REQUIRE(x && W && y && rows >= 0 && K > 0 && N > 0);
    constexpr int ntl = 3;
    const int nt = (N + 31) / 32;
    dim3 grid((rows + LIN_ROWS - 1) / LIN_ROWS, (nt + ntl - 1) / ntl);
    const bool vec = K % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) k_linear<0, 3, true, true><<<grid, dim3(256), 0, cf_s(stream)>>>(x, W, nullptr, nullptr, y, rows, K, N);
    else k_linear<0, 3, false, true><<<grid, dim3(256), 0, cf_s(stream)>>>(x, W, nullptr, nullptr, y, rows, K, N);
    CF_LAUNCH_CHECK();
    return 0;
}

static int linear_wgrad_groups(int rows) {
    const int chunks = (rows + WG_RC - 1) / WG_RC;
    return chunks < 512 ? (chunks > 0 ? chunks : 1) : 512;      // two workgroups per CU
}

// Column blocking of wide problems: the LDS stage holds WG_MAXF 32-feature blocks of [gy | x, 1]; gy goes in blocks of at
// most 4 (128 output rows), x in blocks of what is left.
static void wgrad_blocks(int K, int N, int& nbs, int& kbs) {
    nbs = N < 128 ? N : 128;
    const int NTc = (nbs + 31) / 32;
    kbs = (WG_MAXF - NTc) * 32 - 1;
    if (kbs > K) kbs = K;
}

// column blocks of x that share one launch (blockIdx.z): all of them while the grid stays small (small batches, where a
// launch costs more than its kernel), else one launch pair per block (the partials of one block at a time)
static int wgrad_zblocks(int G, int K, int kbs) {
    const int nkb = (K + kbs - 1) / kbs;
    return G * nkb <= 512 ? nkb : 1;
}

int64_t cf_linear_wgrad_ws_bytes(int rows, int K, int N) {
    int nbs, kbs;
    wgrad_blocks(K, N, nbs, kbs);
    const int NT = (nbs + 31) / 32, KT = (kbs + 1 + 31) / 32, G = linear_wgrad_groups(rows);
    return (int64_t)G * NT * KT * 1024 * sizeof(float) * wgrad_zblocks(G, K, kbs);
}

static int linear_wgrad_any(const float* x, const float* gy, float* gW, float* gb, void* ws, int rows, int K, int N,
                            cf_stream_t stream, bool xsq) {
    CF_REQUIRE(x && gy && gW && ws && rows >= 0 && K > 0 && N > 0);
    int nbs, kbs;
    wgrad_blocks(K, N, nbs, kbs);
    const int G = linear_wgrad_groups(rows);
    const int ZB = wgrad_zblocks(G, K, kbs);              // column blocks per launch: all, or 1
    float* part = (float*)ws;
    hipStream_t st = cf_s(stream);
    for (int n0 = 0; n0 < N; n0 += nbs)
        for (int k0 = 0; k0 < K; k0 += kbs * ZB) {
            const int Nc = N - n0 < nbs ? N - n0 : nbs, Krem = K - k0, Kc = Krem < kbs ? Krem : kbs;
            const int nz = ZB > 1 ? (Krem + kbs - 1) / kbs : 1;
            const int Kl = ZB > 1 ? Krem : Kc;            // columns this launch covers
            const int NT = (Nc + 31) / 32, KT = (Kc + 1 + 31) / 32, ntiles = NT * KT;       // of the widest (first) block
            const int64_t zs = (int64_t)G * ntiles * 1024;
            const float* xp = x + k0;
            const float* gp = gy + n0;
            // full blocks of 4 WG_TPW tiles, then the remainder with exactly as many tile slots per wave as it needs
            const int full = ntiles / (4 * WG_TPW), rem = ntiles - full * 4 * WG_TPW;
            if (full > 0) {
                if (xsq) k_linear_wgrad<WG_TPW, true><<<dim3(G, full, nz), dim3(256), 0, st>>>(xp, gp, part, rows, Kl, Nc, NT, kbs, 0, K, N, zs);
                else k_linear_wgrad<WG_TPW, false><<<dim3(G, full, nz), dim3(256), 0, st>>>(xp, gp, part, rows, Kl, Nc, NT, kbs, 0, K, N, zs);
            }
#define CF_WG_TAIL(T) do { if (xsq) k_linear_wgrad<T, true><<<dim3(G, 1, nz), dim3(256), 0, st>>>(xp, gp, part, rows, Kl, Nc, NT, kbs, full, K, N, zs); \
                           else k_linear_wgrad<T, false><<<dim3(G, 1, nz), dim3(256), 0, st>>>(xp, gp, part, rows, Kl, Nc, NT, kbs, full, K, N, zs); } while (0)
            switch ((rem + 3) / 4) {
                case 0: break;
                case 1: CF_WG_TAIL(1); break;
                case 2: CF_WG_TAIL(2); break;
                case 3: CF_WG_TAIL(3); break;
                case 4: CF_WG_TAIL(4); break;
                case 5: CF_WG_TAIL(5); break;
                case 6: CF_WG_TAIL(6); break;
                case 7: CF_WG_TAIL(7); break;
                default: CF_WG_TAIL(8); break;
            }
#undef CF_WG_TAIL
            // the bias gradient (column sums of gy) rides along with the first x block only
            k_linear_wgrad_reduce<<<dim3(ntiles * 16, nz), dim3(256), 0, st>>>(part, gW + (int64_t)n0 * K + k0, (gb && k0 == 0) ? gb + n0 : nullptr,
                                                                             Kl, Nc, kbs, NT, G, K, zs);
        }
    CF_LAUNCH_CHECK();
    return 0;
}

// (a GROUP of weight gradients in one launch pair - cf_linear_wgrad_group - lives in cf_rowgemm.hip)
int cf_linear_wgrad(const float* x, const float* gy, float* gW, float* gb, void* ws, int rows, int K, int N,
                    cf_stream_t stream) {
    return linear_wgrad_any(x, gy, gW, gb, ws, rows, K, N, stream, false);
}

int cf_linear_wgrad_x2(const float* x, const float* gy, float* gW, void* ws, int rows, int K, int N, cf_stream_t stream) {
    return linear_wgrad_any(x, gy, gW, nullptr, ws, rows, K, N, stream, true);
}

int cf_layernorm(const float* x, const float* w, const float* b, const float* pos, float* y, int rows, int dim,
                 int ntok, float eps, cf_stream_t stream) {
    if (rows == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && w && b && y && rows >= 0 && dim > 0 && (pos == nullptr || ntok > 0));
    if (rows == 0) return 0;
    if (dim <= 256) k_layernorm<true><<<dim3((rows + 15) / 16), dim3(256), 0, cf_s(stream)>>>(x, w, b, pos, y, rows, dim, ntok > 0 ? ntok : 1, eps);
    else k_layernorm<false><<<dim3((rows + 15) / 16), dim3(256), 0, cf_s(stream)>>>(x, w, b, pos, y, rows, dim, ntok > 0 ? ntok : 1, eps);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_attention(const float* qkv, float* out, int B, int N, int dh, float scale, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(qkv && out && B >= 0 && N > 0 && dh > 0);
    CF_REQUIRE(dh % 4 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0);
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
