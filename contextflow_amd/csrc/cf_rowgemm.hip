// A GROUP of Linear weight gradients in one launch pair - cf_linear_wgrad_group:
//
//     gW_i (N_i, K_i) = gy_i^T x_i ,   gb_i (N_i) = column sums of gy_i          x_i (rows_i, K_i), gy_i (rows_i, N_i) dense
//
// The transformer flow step leaves 26 such operand pairs per step (cf_vit_step_bwd: K, N in {26, 52, 64, 192}, rows =
// tokens of the batch): skinny products whose cost is reading the planes once (1.9 GB per step at 32 768 samples) plus
// 30 GFLOP.  Round 3's first form staged 32-row tiles through LDS with 52 -> 64 / 26 -> 32 padding and two barriers per 32
// rows: 1.14 ms per step, 1.7 TB/s.  This form has no LDS and no barrier: a WAVE owns up to 6 x 4 output tiles of
// v_mfma_f32_16x16x4_f32 (16 features of gy x 16 features of x) and a contiguous range of rows, and reads both operands
// from global memory directly in the MFMA's layout - lane (m = lane & 15, kk = lane >> 4) loads gy[r + kk][16 rt + m] and
// x[r + kk][16 ct + m] for the k-step of rows r .. r + 3: 64-byte runs of four consecutive rows.  The operands of the next
// k-step are requested before the MFMAs of the current one.  Partials per (member, tile group, row range), summed in range
// order by a second kernel (no float atomics).
#include "cf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RG_MAX = 32;           // members per group
constexpr int RG_RT = 6, RG_CT = 4;  // output tiles per wave: 96 x 64 (96 accumulator registers)

struct RgMember {
    const float* x; const float* gy; float* part; float* gW; float* gb;
    int rows, K, N;
    int rtg, ctg;                    // tile groups along N (of RG_RT tiles) / along K (of RG_CT tiles)
    int S, rps;                      // row ranges and rows per range (a multiple of 8)
    int unit0;                       // first wave unit of this member
};
struct RgGroup { RgMember m[RG_MAX]; int n, units; };

// NRT x NCT tiles of one unit.  Columns past N / K are clamped (their products land in accumulator rows / columns that are
// never stored); rows past the range contribute gy = 0.
template <int NRT, int NCT>
__device__ __forceinline__ void rowgemm_body(const RgMember& d, int rg, int cg, int s, int lane) {
    const int m = lane & 15, kk = lane >> 4;
    const int N = d.N, K = d.K;
    const int r0 = s * d.rps, r1 = min(d.rows, r0 + d.rps);
    int ac[NRT], bc[NCT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) ac[rt] = min(16 * (RG_RT * rg + rt) + m, N - 1);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bc[ct] = min(16 * (RG_CT * cg + ct) + m, K - 1);
    f32x4 acc[NRT][NCT];
    float bs[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        bs[rt] = 0.f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // uniform range base + 32-bit lane offsets advancing by a constant per k-step
    const float* gyb = d.gy + (int64_t)r0 * N;
    const float* xb = d.x + (int64_t)r0 * K;
    int go = kk * N, xo = kk * K;
    const int nr = r1 - r0;
    auto load = [&](float (&av)[NRT], float (&bv)[NCT]) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) av[rt] = gyb[go + ac[rt]];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) bv[ct] = xb[xo + bc[ct]];
        go += 4 * N; xo += 4 * K;
    };
    auto mma = [&](const float (&av)[NRT], const float (&bv)[NCT]) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            bs[rt] += av[rt];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv[ct], acc[rt][ct], 0, 0, 0);
        }
    };
    float a0[NRT], b0[NCT], a1[NRT], b1[NCT];
    const int full = nr / 8;                           // trips of two whole k-steps
    if (full > 0) {
        load(a0, b0);
        for (int it = 0; it < full; ++it) {
            load(a1, b1);
            mma(a0, b0);
            if (it + 1 < full) load(a0, b0);
            mma(a1, b1);
        }
    }
    for (int rr = 8 * full; rr < nr; rr += 4) {        // ragged tail
        const bool ok = rr + kk < nr;
        const int g2 = ok ? go : 0, x2 = ok ? xo : 0;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) { const float v = gyb[g2 + ac[rt]]; a0[rt] = ok ? v : 0.f; }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) b0[ct] = xb[x2 + bc[ct]];
        go += 4 * N; xo += 4 * K;
        mma(a0, b0);
    }
    // partial of this range: [N][K] | [N]
    float* pw = d.part + (int64_t)s * ((int64_t)N * K + N);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = 16 * (RG_RT * rg + rt) + 4 * kk + j, k = 16 * (RG_CT * cg + ct) + m;
                if (n < N && k < K) pw[(int64_t)n * K + k] = acc[rt][ct][j];
            }
        if (cg == 0) {
            float v = bs[rt];                           // lane (m, kk): column m of gy over the rows of lane group kk
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = 16 * (RG_RT * rg + rt) + m;
            if (kk == 0 && n < N) pw[(int64_t)N * K + n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_rowgemm_group(const RgGroup grp) {
    const int u = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (u >= grp.units) return;                        // whole wave; the kernel has no barrier
    int i = 0;
    while (i + 1 < grp.n && u >= grp.m[i + 1].unit0) ++i;          // scalar
    const RgMember& d = grp.m[i];
    const int local = u - d.unit0, s = local % d.S, tg = local / d.S, rg = tg % d.rtg, cg = tg / d.rtg;
    const int nrt = min(RG_RT, (d.N + 15) / 16 - RG_RT * rg), nct = min(RG_CT, (d.K + 15) / 16 - RG_CT * cg);
    const int lane = threadIdx.x & 63;
    // the tile counts of a unit are wave-uniform: one of a few fully unrolled bodies
#define RG_CASE(R, C) if (nrt == R && nct == C) { rowgemm_body<R, C>(d, rg, cg, s, lane); return; }
    RG_CASE(6, 4) RG_CASE(4, 4) RG_CASE(4, 2) RG_CASE(2, 2) RG_CASE(6, 2) RG_CASE(2, 4)
    RG_CASE(1, 1) RG_CASE(1, 2) RG_CASE(2, 1) RG_CASE(1, 4) RG_CASE(4, 1) RG_CASE(6, 1) RG_CASE(3, 4) RG_CASE(4, 3)
    RG_CASE(3, 2) RG_CASE(2, 3) RG_CASE(3, 3) RG_CASE(5, 4) RG_CASE(6, 3) RG_CASE(5, 2) RG_CASE(3, 1) RG_CASE(1, 3)
    RG_CASE(5, 1) RG_CASE(5, 3)
#undef RG_CASE
}

// gW, gb = sum over the row ranges, in range order; one thread per output element, 16 loads in flight
__global__ __launch_bounds__(256) void k_rowgemm_reduce(const RgGroup grp) {
    const RgMember& d = grp.m[blockIdx.y];
    const int64_t per = (int64_t)d.N * d.K + d.N;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    const float* p = d.part + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 16 <= d.S; s += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = p[(int64_t)(s + j) * per];
#pragma unroll
        for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    }
    for (; s < d.S; ++s) s0 += p[(int64_t)s * per];
    const float v = (s0 + s1) + (s2 + s3);
    if (e < (int64_t)d.N * d.K) d.gW[e] = v;
    else if (d.gb != nullptr) d.gb[e - (int64_t)d.N * d.K] = v;
}

// row ranges of a member: about 4096 waves for the whole group, at least 64 rows each, a multiple of 8
void rowgemm_plan(const int* rows, const int* K, const int* N, int n, int i, int& rtg, int& ctg, int& S, int& rps) {
    int tgs = 0;
    for (int j = 0; j < n; ++j) tgs += (((N[j] + 15) / 16 + RG_RT - 1) / RG_RT) * (((K[j] + 15) / 16 + RG_CT - 1) / RG_CT);
    rtg = ((N[i] + 15) / 16 + RG_RT - 1) / RG_RT;
    ctg = ((K[i] + 15) / 16 + RG_CT - 1) / RG_CT;
    int want = (4096 + tgs - 1) / tgs;
    if (want < 1) want = 1;
    rps = (rows[i] + want - 1) / want;
    if (rps < 64) rps = 64;
    rps = (rps + 7) / 8 * 8;
    S = rows[i] > 0 ? (rows[i] + rps - 1) / rps : 1;
}

}  // namespace

extern "C" {

int64_t cf_linear_wgrad_group_ws_bytes(const int* rows, const int* K, const int* N, int n) {
    if (!rows || !K || !N || n < 1 || n > RG_MAX) return -1;
    int64_t f = 0;
    for (int i = 0; i < n; ++i) {
        int rtg, ctg, S, rps;
        rowgemm_plan(rows, K, N, n, i, rtg, ctg, S, rps);
        f += (int64_t)S * ((int64_t)N[i] * K[i] + N[i]);
    }
    return f * (int64_t)sizeof(float);
}

int cf_linear_wgrad_group(const float* const* x, const float* const* gy, float* const* gW, float* const* gb, const int* rows,
                          const int* K, const int* N, int n, void* ws, cf_stream_t stream) {
    CF_REQUIRE(x && gy && gW && gb && rows && K && N && ws && n >= 1 && n <= RG_MAX);
    RgGroup grp;
    float* part = (float*)ws;
    int units = 0;
    int64_t permax = 1;
    for (int i = 0; i < n; ++i) {
        CF_REQUIRE(x[i] && gy[i] && gW[i] && rows[i] > 0 && K[i] > 0 && N[i] > 0 && (int64_t)rows[i] * (K[i] > N[i] ? K[i] : N[i]) < (1ll << 40));
        RgMember& m = grp.m[i];
        m.x = x[i]; m.gy = gy[i]; m.part = part; m.gW = gW[i]; m.gb = gb[i];
        m.rows = rows[i]; m.K = K[i]; m.N = N[i];
        rowgemm_plan(rows, K, N, n, i, m.rtg, m.ctg, m.S, m.rps);
        CF_REQUIRE((int64_t)m.rps * (K[i] > N[i] ? K[i] : N[i]) < (1ll << 30));          // 32-bit offsets inside a row range
        m.unit0 = units;
        units += m.rtg * m.ctg * m.S;
        const int64_t per = (int64_t)N[i] * K[i] + N[i];
        part += (int64_t)m.S * per;
        permax = per > permax ? per : permax;
    }
    for (int i = n; i < RG_MAX; ++i) grp.m[i] = grp.m[0];
    grp.n = n; grp.units = units;
    hipStream_t st = cf_s(stream);
    k_rowgemm_group<<<dim3((units + 3) / 4), dim3(256), 0, st>>>(grp);
    k_rowgemm_reduce<<<dim3((unsigned)((permax + 255) / 256), n), dim3(256), 0, st>>>(grp);
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
