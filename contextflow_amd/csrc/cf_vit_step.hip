// One transformer-coupling flow step — Conv1x1 -> ActNorm -> TransCoupling (patchify, SimpleViT, un-patchify, affine
// map, log-det) — as ONE gfx950 kernel for the time-series topologies (H x 1 windows, patch (2,1), 4 tokens per sample:
// SMAP).  Reference: contextflow/model.py:129-147 (the per-step triple), layers/conv1x1.py:52-57, layers/actnorm.py:
// 53-60, layers/coupling.py:100-159, layers/simple_vit.py:18-127.
//
// Design (MI355X).  Everything between the input x and the output z stays in REGISTERS:
//  * every Linear runs on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) with the weights as the A operand (row =
//    output feature) and the activations as the B operand (column = token).  A result tile has the TOKEN on the lane and
//    the FEATURES in the 16 accumulator registers - and a following product that sums over those features can take the
//    accumulator registers AS its B operand, with no lane movement and no LDS: k-step s of the next Linear is simply
//    "register s", i.e. lane half 0 supplies feature row(s, 0) and lane half 1 feature row(s, 1) of its own token; the
//    packed weights are stored in that k order (k_vit_step_pack).  The residual stream, LayerNorm outputs, attention
//    operands and the MLP hidden layer never leave the register file;
//  * round 4 - the kernel was bound by its MFMA count (with every vector instruction removed it ran 14 % faster, not 40 %),
//    so the count went down, 2 144 -> 1 326 per wave:
//      - a single head of 64 on a width of 52 makes q.k and the value / output pair factor through the 52-wide stream:
//        q_i.k_j = u_i^T (Wq^T Wk) u_j and to_out(sum_j p_ij Wv u_j) = (Wout Wv) sum_j p_ij u_j.  The two 52 x 52 matrices
//        are formed once per parameter version in fp64 (k_vit_step_pack) and an attention block is TWO 52 x 52 products
//        instead of 52 -> 192 and 64 -> 52 (18 720 -> 10 816 multiply-adds per token and layer);
//      - every LayerNorm that feeds a Linear leaves its affine part in that Linear's packed weights and bias (fp64), and
//        with u_j = g (.) n_j + b the score terms that do not depend on the key j drop out of the softmax exactly:
//        s_ij = (A1 n_i + c1) . n_j;
//      - the 2C features of a tile pair sit on the physical rows {0 .. 24, 28} of each 32-row tile: both lane halves hold
//        C/2 = 13 valid registers per tile, so a product over the stream has 26 k-steps (it had 28: rows 28 / 29 of one
//        half were padding) and no loop depends on the lane half;
//  * LayerNorm statistics are lane-local over the registers + one v_permlane32_swap between the two lane halves that share
//    a token (no LDS round trip), GELU is elementwise on the accumulators (erf by Abramowitz-Stegun 7.1.26 on the
//    hardware exp / rcp: the resulting GELU is as close to the exact one as torch's, 1.0e-7 rms over [-8, 8]);
//  * a sample's 4 tokens sit in 4 consecutive lanes: the scores and the probability-weighted sum use DPP quad
//    permutations, the softmax over 4 scores is exact (no online rescaling);
//  * Conv1x1 + ActNorm run as a first product with rows (position-in-patch, channel): its two result tiles ARE the
//    patchified conditioner input x0 (tile 0) and the half that gets transformed, x1 (tile 1).  The model's 2C features
//    are laid out so that t (tile 0) and raw log-scale (tile 1) of one (position, channel) share lane AND register index
//    with x1: the affine epilogue is lane-local;
//  * a wave owns 32 token columns = 8 samples; nothing crosses a wave: no LDS, no barrier.  One launch = one flow step:
//    HBM traffic = read x + write z.
// Geometry covered: C even, C <= 32 (SMAP: 26), H = 8, W = 1, patch (2, 1), dim = 2C, 1 head x 64, any depth.  Other
// transformer couplings keep the LDS-plane kernel (cf_vit_fused.hip) or the layer-by-layer kernels (cf_vit.hip).
#include "cf_common.h"
#include "cf_vit_fuse.h"
#include <math.h>

extern "C" int cf_slogdet_inverse(const float* W, int C, float* logabsdet, float* Winv, cf_stream_t stream);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// accumulator register r of lane half lk <-> row of a 32-row result tile (v_mfma_f32_32x32x2_f32)
__host__ __device__ constexpr int trow(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
__host__ __device__ constexpr int reg_of_row(int q) { return (q & 3) + 4 * (q >> 3); }
__host__ __device__ constexpr int ngrp(int ks) { return (ks + 3) / 4; }

// ---- compile-time geometry ------------------------------------------------------------------------------------
template <int C_> struct VS {
    static constexpr int C = C_, CIN = C / 2, HW = 8, NTOK = 4, DIM = 2 * C, HEAD = 64;
    static constexpr int KPT = C / 2;                 // valid registers per tile and lane: rows trow(r, lk), r < KPT, BOTH halves
    static constexpr int KS_RES = 2 * KPT;            // k-steps when the B operand is the two residual tiles
    static constexpr int KS_IN = KPT;                 // ... tile 0 only (patch features)
    static constexpr int KS0 = C;                     // Conv1x1: one k-step per input channel (lane half = position in patch)
    // workspace (floats).  "vec" = 64 floats in register order [lk][tile][r]; "mat" = A fragments of a 2-tile product
    static constexpr int MAT_RES = ngrp(KS_RES) * 2 * 256;
    static constexpr int OFF_B0 = 4, OFF_A0 = OFF_B0 + 64;
    static constexpr int OFF_WE = OFF_A0 + ngrp(KS0) * 2 * 256, OFF_BE = OFF_WE + ngrp(KS_IN) * 2 * 256;
    static constexpr int OFF_LN1 = OFF_BE + 64;                                // [w | b + pos_0 | .. | b + pos_3]
    static constexpr int OFF_LAYER = OFF_LN1 + 64 + NTOK * 64;
    // per layer: s = (A1 n + c1) . n, x += A2 (sum_j p_j n_j) + c2, x += W2 gelu(W1 n' + b1) + b2
    static constexpr int L_A1 = 0, L_C1 = L_A1 + MAT_RES, L_A2 = L_C1 + 64, L_C2 = L_A2 + MAT_RES;
    static constexpr int L_W1 = L_C2 + 64, L_B1 = L_W1 + MAT_RES, L_W2 = L_B1 + 64, L_B2 = L_W2 + MAT_RES;
    static constexpr int L_STRIDE = L_B2 + 64;
    static_assert(C % 2 == 0 && C >= 4 && C <= 32, "C even, <= 32");
};
template <class V> __host__ __device__ constexpr int off_lno(int depth) { return V::OFF_LAYER + depth * V::L_STRIDE; }
template <class V> __host__ __device__ constexpr int off_fuse(int depth) { return off_lno<V>(depth) + 128; }      // k_vit_fuse scratch
template <class V> __host__ __device__ constexpr int ws_floats(int depth) { return off_fuse<V>(depth) + depth * VitFuse<V::DIM, V::HEAD>::LAYER_FLOATS; }

// row q (0..31) of a tile -> logical index j in [0, C) (the valid rows in increasing order), or -1 for padding.  A row is
// valid when its register index is below KPT; for C = 26 these are the rows 0 .. 24 and 28.
template <class V> __host__ __device__ constexpr int logical_of_row(int q) {
    if (reg_of_row(q) >= V::KPT) return -1;
    int j = 0;
    for (int p = 0; p < q; ++p) j += reg_of_row(p) < V::KPT ? 1 : 0;
    return j;
}
// physical row p (0..63: tile = p >> 5) of the residual layout -> model feature f = ii * C + ch, or -1 for padding.
// tile 0 holds channels ch < C/2 (t after the last LayerNorm), tile 1 channels ch >= C/2 (raw log-scale), both at logical
// index ii * C/2 + c: the same (lane, register) in the two tiles belongs to one (position ii, channel c).
template <class V> __host__ __device__ constexpr int feat_of_phys(int p) {
    const int j = logical_of_row<V>(p & 31);
    if (j < 0) return -1;
    return (j / V::CIN) * V::C + (j % V::CIN) + V::CIN * (p >> 5);
}
// k-step s of a product whose B operand is the residual tile pair -> physical row supplied by lane half lk
template <class V> __host__ __device__ constexpr int phys_of_kstep(int s, int lk) { return trow(s % V::KPT, lk) + 32 * (s / V::KPT); }

// ---- packing (cf_vit_step_prepare) ----------------------------------------------------------------------------------
// fragment element ((g * RT + rt) * 64 + lane) * 4 + e = A[row = rt * 32 + (lane & 31)][k-step 4 g + e, lane half]
template <class V>
__global__ __launch_bounds__(256) void k_vit_step_pack(const float* __restrict__ Wm, const float* __restrict__ t,
                                                       const float* __restrict__ logs, const float* __restrict__ flat,
                                                       const float* __restrict__ pos, float* __restrict__ ws, int depth) {
    constexpr int C = V::C, CIN = V::CIN, DIM = V::DIM, HEAD = V::HEAD;
    using F = VitFuse<DIM, HEAD>;
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    // register-order vector of a per-feature quantity val(f), f = model feature of the residual layout
    auto vec = [&](float* dst, auto val) {
        for (int i = gtid; i < 64; i += gsz) {
            const int lk = i >> 5, rt = (i >> 4) & 1, r = i & 15;
            const int f = feat_of_phys<V>(trow(r, lk) + 32 * rt);
            dst[i] = f >= 0 ? (float)val(f) : 0.f;
        }
    };
    // A fragments of a 2-tile product: physical output rows -> model rows through rowmap; k-steps through kmap; val(row, k)
    auto frags = [&](float* dst, int nks, auto rowmap, auto kmap, auto val) {
        const int n = ngrp(nks) * 2 * 256;
        for (int i = gtid; i < n; i += gsz) {
            const int e = i & 3, lane = (i >> 2) & 63, q = i >> 8, rt = q % 2, g = q / 2;
            const int s = 4 * g + e;
            const int row = rowmap(rt * 32 + (lane & 31));
            const int k = s < nks ? kmap(s, lane >> 5) : -1;
            dst[i] = (row >= 0 && k >= 0) ? (float)val(row, k) : 0.f;
        }
    };
    auto res_row = [](int p) { return feat_of_phys<V>(p); };
    auto res_k = [](int s, int lk) { return feat_of_phys<V>(phys_of_kstep<V>(s, lk)); };
    auto in_k = [](int s, int lk) { return logical_of_row<V>(trow(s, lk)); };                   // patch feature = logical tile-0 row

    if (gtid == 0) {       // ldj_const = H*W*log|det Wm| + sum_c logs  (ws[1] holds log|det| from cf_slogdet_inverse)
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += logs[c];
        ws[0] = (float)V::HW * ws[1] + s;                                   // conv1x1.py:53 + actnorm.py:58
    }
    // Conv1x1 + ActNorm as a product over (position-in-patch, channel): rows = physical rows, k-step s = input channel s,
    // lane half = position ii'; block diagonal in the position
    for (int i = gtid; i < 64; i += gsz) {
        const int lk = i >> 5, rt = (i >> 4) & 1, r = i & 15, j = logical_of_row<V>(trow(r, lk));
        const int ch = j >= 0 ? (j % CIN) + CIN * rt : -1;
        ws[V::OFF_B0 + i] = ch >= 0 ? (float)(-(double)t[ch] * exp(-(double)logs[ch])) : 0.f;        // (x - t) e^{-logs}
    }
    for (int i = gtid; i < ngrp(V::KS0) * 2 * 256; i += gsz) {
        const int e = i & 3, lane = (i >> 2) & 63, qq = i >> 8, rt = qq % 2, g = qq / 2, s = 4 * g + e;
        const int j = logical_of_row<V>(lane & 31), ii = j >= 0 ? j / CIN : -1, ch = j >= 0 ? (j % CIN) + CIN * rt : -1;
        ws[V::OFF_A0 + i] = (ch >= 0 && s < C && ii == (lane >> 5)) ? (float)(exp(-(double)logs[ch]) * (double)Wm[ch * C + s]) : 0.f;
    }
    const float* p = flat;       // order of TransCoupling._flat_params(): see cf_vit_fused.hip
    {   // to_patch_embedding: LN(pd = C) -> Linear(C, DIM) -> LN(DIM); the first LayerNorm's affine part folded into the Linear
        const float *g0 = p, *b0 = p + C, *We = p + 2 * C, *be = We + DIM * C, *g1 = be + DIM, *b1 = g1 + DIM;
        frags(ws + V::OFF_WE, V::KS_IN, res_row, in_k, [&](int f, int k) { return (double)We[f * C + k] * (double)g0[k]; });
        vec(ws + V::OFF_BE, [&](int f) { double a = be[f]; for (int k = 0; k < C; ++k) a += (double)We[f * C + k] * (double)b0[k]; return a; });
        vec(ws + V::OFF_LN1, [&](int f) { return g1[f]; });
        for (int n = 0; n < V::NTOK; ++n) vec(ws + V::OFF_LN1 + 64 * (1 + n), [&](int f) { return b1[f] + pos[n * DIM + f]; });
        p = b1 + DIM;
    }
    for (int l = 0; l < depth; ++l) {
        float* w = ws + V::OFF_LAYER + l * V::L_STRIDE;
        const float* Wo = p + 2 * DIM + 3 * HEAD * DIM;
        const float *gf = Wo + DIM * HEAD, *bf = gf + DIM, *W1 = bf + DIM, *b1 = W1 + DIM * DIM, *W2 = b1 + DIM, *b2 = W2 + DIM * DIM;
        // attention through the fused matrices (cf_vit_fuse.h: k_vit_fuse has formed them in fp64):  s_ij = (A1 n_i + c1) . n_j,
        // x += A2 (sum_j p_ij n_j) + c2
        const float* fz = ws + off_fuse<V>(depth) + l * F::LAYER_FLOATS;
        frags(w + V::L_A1, V::KS_RES, res_row, res_k, [&](int a, int b) { return fz[F::M1 + a * DIM + b]; });
        vec(w + V::L_C1, [&](int a) { return fz[F::C1 + a]; });
        frags(w + V::L_A2, V::KS_RES, res_row, res_k, [&](int f, int b) { return fz[F::M2 + f * DIM + b]; });
        vec(w + V::L_C2, [&](int f) { return fz[F::C2 + f]; });
        // FeedForward (simple_vit.py:30-40): LayerNorm affine folded into the first Linear
        frags(w + V::L_W1, V::KS_RES, res_row, res_k, [&](int f, int b) { return (double)W1[f * DIM + b] * (double)gf[b]; });
        vec(w + V::L_B1, [&](int f) { double a = b1[f]; for (int k = 0; k < DIM; ++k) a += (double)W1[f * DIM + k] * (double)bf[k]; return a; });
        frags(w + V::L_W2, V::KS_RES, res_row, res_k, [&](int f, int b) { return (double)W2[f * DIM + b]; });
        vec(w + V::L_B2, [&](int f) { return b2[f]; });
        p = b2 + DIM;
    }
    vec(ws + off_lno<V>(depth), [&](int f) { return p[f]; });
    vec(ws + off_lno<V>(depth) + 64, [&](int f) { return p[DIM + f]; });
}

// ---- device helpers ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ rsrc_t make_rsrc(const float* ws, int floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, floats * 4, 0x00020000);
}
__device__ __forceinline__ float4 frag(rsrc_t rs, int lane, int foff) {
#ifdef CF_ABL_VS_NOFRAG
    return make_float4(1e-3f * lane, 2e-3f, 3e-3f * foff, 1e-3f);
#endif
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, foff * 4, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// register-order vector (64 floats [lk][tile][r]) -> this lane's two tiles
__device__ __forceinline__ void load_vec(f32x16 (&v)[2], const float* __restrict__ base, int lk) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = *reinterpret_cast<const float4*>(base + lk * 32 + rt * 16 + 4 * q);
            v[rt][4 * q + 0] = x.x; v[rt][4 * q + 1] = x.y; v[rt][4 * q + 2] = x.z; v[rt][4 * q + 3] = x.w;
        }
}

// acc[rt] += sum over k-steps A(ws frags at foff) * B, B operand = registers: k-step s -> bop(s).
// Fragments of group g+1 are requested before the MFMAs of group g (pinned: hipcc otherwise sinks the loads).
// ZERO: the accumulators start from zero - passed to the first MFMA as the inline constant instead of 16 v_mov per tile.
template <int RT, int NKS, bool ZERO = false, class BOP>
__device__ __forceinline__ void gemm_regs(f32x16 (&acc)[RT], rsrc_t rs, int lane, int foff, BOP bop) {
    constexpr int NG = (NKS + 3) / 4;
    // fragment groups in flight ahead of the MFMAs (a group = 4 k-steps = 8 MFMAs of 64 cycles).  Measured at 524 288 samples:
    // 1 group ahead 3 380 us, 2 groups 3 545, 3 groups 3 700 - more requests in flight only raise the L2 traffic's share of the
    // power budget (the shader clock falls from 2.39 GHz without the loads to 2.26 with them: tools/dev/vit_clock.py)
#ifndef CF_VS_AHEAD
#define CF_VS_AHEAD 1
#endif
    constexpr int AH = CF_VS_AHEAD, NR = AH + 1;
    float4 a[NR][RT];
#pragma unroll
    for (int g = 0; g < AH && g < NG; ++g)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[g][rt] = frag(rs, lane, foff + (g * RT + rt) * 256);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + AH < NG) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[(g + AH) % NR][rt] = frag(rs, lane, foff + ((g + AH) * RT + rt) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g + e < NKS) {
                const float b = bop(4 * g + e);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (ZERO && g == 0 && e == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a[g % NR][rt], e), b, zero, 0, 0, 0);
                    } else {
                        acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a[g % NR][rt], e), b, acc[rt], 0, 0, 0);
                    }
                }
            }
    }
}

// sum of a per-lane partial over the two lane halves that share a token (v_permlane32_swap: no LDS round trip); both
// halves receive lo + hi in that order, i.e. the same bits
__device__ __forceinline__ float half_sum(float s) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Packed fp32 (v_pk_add / v_pk_mul / v_pk_fma_f32: two elements per instruction).  The fp32 matrix pipe shares its lanes with
// the vector ALU (tools/micro/mfma_issue.hip: no overlap), so every vector instruction of this kernel is time taken from the
// MFMA stream: the elementwise stages work on the aligned register pairs (2i, 2i + 1) of a tile - registers 0 .. 11 of the
// KPT = 13 valid ones - and treat register 12 on its own.  (Left to itself hipcc pairs registers of different tiles and pays
// for it in v_mov: 103 of them per transformer layer.)
#define CF_PAIR(v, i) (f32x2{(v)[2 * (i)], (v)[2 * (i) + 1]})
__device__ __forceinline__ void set_pair(f32x16& v, int i, f32x2 p) { v[2 * i] = p.x; v[2 * i + 1] = p.y; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }

// the token's features minus their mean, times 1 / sqrt(biased variance + 1e-5) (torch.nn.LayerNorm without its affine part:
// that sits in the packed weights of the Linear behind it, or is applied by layernorm_affine).  NT tiles of KPT valid
// registers per lane; the two lane halves hold different features of the same token.
template <class V, int NT>
__device__ __forceinline__ float normalize_stats(const f32x16 (&x)[2], f32x16 (&y)[2]);
template <class V, int NT>
__device__ __forceinline__ void normalize(const f32x16 (&x)[2], f32x16 (&y)[2]) {
    constexpr int K = V::KPT;
    const float rstd = normalize_stats<V, NT>(x, y);
    const f32x2 r2 = splat(rstd);
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
#pragma unroll
        for (int i = 0; i < K / 2; ++i) set_pair(y[rt], i, CF_PAIR(y[rt], i) * r2);
        y[rt][K - 1] *= rstd;
    }
}
// first half of normalize: y = x - mean, returns 1 / sqrt(var + eps) - the scaling is left to the caller (the products that
// consume the normalised features multiply each one right before the k-step that needs it, in the shadow of the MFMAs)
template <class V, int NT>
__device__ __forceinline__ float normalize_stats(const f32x16 (&x)[2], f32x16 (&y)[2]) {
    constexpr int K = V::KPT, NP = K / 2;
    static_assert(K % 2 == 1, "an odd number of valid registers per tile: pairs + one single");
#ifdef CF_ABL_VS_NOLN
    y[0] = x[0]; y[1] = x[1]; return 1.0f;
#endif
    f32x2 s2[2] = {splat(0.f), splat(0.f)};
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int i = 0; i < NP; ++i) s2[i & 1] += CF_PAIR(x[rt], i);
    s2[0] += s2[1];
    float s = s2[0].x + s2[0].y;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) s += x[rt][K - 1];
    const float mean = half_sum(s) * (1.0f / (float)(NT * V::C));
    const f32x2 nm = splat(-mean);
    f32x2 v2[2] = {splat(0.f), splat(0.f)};
    float v = 0.f;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const f32x2 d = CF_PAIR(x[rt], i) + nm;
            set_pair(y[rt], i, d);
            v2[i & 1] = pk_fma(d, d, v2[i & 1]);
        }
        const float d = x[rt][K - 1] - mean;
        y[rt][K - 1] = d;
        v = fmaf(d, d, v);
    }
    v2[0] += v2[1];
    const float var = half_sum(v + (v2[0].x + v2[0].y)) * (1.0f / (float)(NT * V::C)) + 1e-5f;
    float rstd = __builtin_amdgcn_rsqf(var);
    rstd = rstd * fmaf(-0.5f * var * rstd, rstd, 1.5f);            // one Newton step: 1 ulp -> rounding level
    return rstd;
}

// full LayerNorm of the two tiles: normalize, then y = y w + b (w, b: register-order vectors)
template <class V>
__device__ __forceinline__ void layernorm_affine(const f32x16 (&x)[2], f32x16 (&y)[2], const float* __restrict__ w,
                                                 const float* __restrict__ b, int lk) {
    f32x16 wv[2], bv[2];
    load_vec(wv, w, lk);
    load_vec(bv, b, lk);
    normalize<V, 2>(x, y);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int i = 0; i < V::KPT / 2; ++i) set_pair(y[rt], i, pk_fma(CF_PAIR(y[rt], i), CF_PAIR(wv[rt], i), CF_PAIR(bv[rt], i)));
        y[rt][V::KPT - 1] = fmaf(y[rt][V::KPT - 1], wv[rt][V::KPT - 1], bv[rt][V::KPT - 1]);
    }
}

// exact GELU v Phi(v), Phi(v) = (1 + erf(v / sqrt 2)) / 2 with erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7) on the hardware
// rcp / exp2: with q = P(t) e^{-v^2 / 2} / 2, t = 1 / (1 + p |v| / sqrt 2), Phi(v) = 1 - q for v >= 0 and q for v < 0, i.e.
// gelu(v) = max(v, 0) - |v| q.  Against the exact GELU: 1.0e-7 rms over [-8, 8], the same as torch's erf-based one.
__device__ __forceinline__ float abs_bits(float v) { return __int_as_float(__float_as_int(v) & 0x7fffffff); }
__device__ __forceinline__ float relu_bits(float v) { return __int_as_float(max(__float_as_int(v), 0)); }   // one integer max (fmaxf: two VALU here)
__device__ __forceinline__ float gelu1(float v) {
#ifdef CF_ABL_VS_NOGELU
    return v;
#endif
    const float av = abs_bits(v);
    const float t = __builtin_amdgcn_rcpf(fmaf(av, 0.3275911f * 0.70710678118654752f, 1.0f));
    float p = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((v * v) * -0.72134752044448170f);
    return fmaf(-av, (p * t) * e, relu_bits(v));
}

#ifdef CF_ABL_VS_NOBIAS
#define CF_BIAS(v, ptr) do { v[0] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; v[1] = v[0]; } while (0)
#else
#define CF_BIAS(v, ptr) load_vec(v, ptr, lk)
#endif
template <int K> __device__ __forceinline__ void add_tiles(f32x16 (&X)[2], const f32x16 (&a)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < K / 2; ++i) set_pair(X[t], i, CF_PAIR(X[t], i) + CF_PAIR(a[t], i));
        X[t][K - 1] += a[t][K - 1];
    }
}

template <int CTRL> __device__ __forceinline__ float quad(float v) {          // DPP quad permutation of the 4 tokens of a sample
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int M> __device__ __forceinline__ float tok_xor(float v) {
    if constexpr (M == 0) return v;
    else if constexpr (M == 1) return quad<0xB1>(v);      // [1,0,3,2]
    else if constexpr (M == 2) return quad<0x4E>(v);      // [2,3,0,1]
    else return quad<0x1B>(v);                            // [3,2,1,0]
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// x, z: (B, C, 8, 1).  ldj_acc[b] += H*W*log|det W| + sum logs + sum log_s.  hout (optional, tests): the conditioner's
// output un-patchified, (B, C, 8, 1) = [t | raw].
// DUMP (training forward at saturating batches): the residual stream at the depth + 1 layer boundaries goes to `xtape`,
// feature-major [boundary][feature < DIM][token < T] (token = 4 sample + n: 128 contiguous bytes per feature and half wave) -
// cf_vit_step_bwd_taped then walks back from these instead of running the six layers again.
#ifndef CF_VS_MINW
#define CF_VS_MINW 2          // waves per SIMD the register budget allows for (measured: 3 - with the DPP copies formed twice, 145 VGPRs - 3 899 us
#endif                        // per 524 288 samples against 3 385, 4 with spills 4 615: more resident waves only fight over the L1 / L2 weight traffic)
template <class V, bool DUMP = false>
__global__ __launch_bounds__(256, CF_VS_MINW) void k_vit_step(const float* __restrict__ x, float* __restrict__ z,
                                                     float* __restrict__ ldj_acc, const float* __restrict__ ws, int B,
                                                     int64_t xbs, int depth, float* __restrict__ hout,
                                                     float* __restrict__ xtape = nullptr, int64_t T = 0) {
    constexpr int C = V::C, CIN = V::CIN, HW = V::HW, K = V::KPT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int n = li & 3;                                   // token of the sample: positions 2n, 2n + 1
    const int smp = (blockIdx.x * 4 + wave) * 8 + (li >> 2);
    const bool live = smp < B;
    const float* xb = x + (int64_t)min(smp, B - 1) * xbs;
    const rsrc_t rs = make_rsrc(ws, ws_floats<V>(depth));
#ifdef CF_VS_TICKS
    const uint64_t tick0 = __builtin_readcyclecounter(), real0 = __builtin_amdgcn_s_memrealtime();
#endif
    // register r of this lane -> (position in the patch ii, channel c) of both tiles
    // (r is a constant after unrolling: both candidates fold at compile time, the lane half selects)
    auto pos_of = [&](int r) { const int j0 = logical_of_row<V>(trow(r, 0)), j1 = logical_of_row<V>(trow(r, 1)); return lk ? j1 / CIN : j0 / CIN; };
    auto chan_of = [&](int r) { const int j0 = logical_of_row<V>(trow(r, 0)), j1 = logical_of_row<V>(trow(r, 1)); return lk ? j1 % CIN : j0 % CIN; };

    // ================= Conv1x1 + ActNorm: [x0' | x1'] rows (ii, c), one k-step per input channel, lane half = position
    f32x16 y[2];
    load_vec(y, ws + V::OFF_B0, lk);
    {
        float xv[C];
#pragma unroll
        for (int k = 0; k < C; ++k) xv[k] = xb[k * HW + 2 * n + lk];
        gemm_regs<2, V::KS0>(y, rs, lane, V::OFF_A0, [&](int s) { return xv[s]; });
    }
    float* zb = z + (int64_t)smp * C * HW;
    if (live) {                                             // first half passes through (coupling.py:154)
#pragma unroll
        for (int r = 0; r < K; ++r) zb[chan_of(r) * HW + 2 * n + pos_of(r)] = y[0][r];
    }
    // ================= patch embedding: LN(pd) -> Linear -> LN(dim) + pos        (simple_vit.py:100-105,122)
    f32x16 X[2];
    {
        f32x16 u[2];
        normalize<V, 1>(y, u);
        load_vec(X, ws + V::OFF_BE, lk);
        gemm_regs<2, V::KS_IN>(X, rs, lane, V::OFF_WE, [&](int s) { return u[0][s]; });
        layernorm_affine<V>(X, X, ws + V::OFF_LN1, ws + V::OFF_LN1 + 64 * (1 + n), lk);
    }
    auto dump = [&](int bnd) {
        if constexpr (DUMP) {
            float* tb = xtape + (int64_t)bnd * V::DIM * T + (int64_t)(blockIdx.x * 4 + wave) * 32 + li;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < K; ++r) tb[(int64_t)(pos_of(r) * C + chan_of(r) + CIN * t) * T] = X[t][r];
        }
    };
    // (boundary 0, the embedding output, is rebuilt by the backward kernel together with the statistics it needs: not taped)
    // ================= transformer                                                 (simple_vit.py:56-88)
#pragma unroll 1
    for (int l = 0; l < depth; ++l) {
        const int wl = V::OFF_LAYER + l * V::L_STRIDE;
        {
            f32x16 u[2], g[2];
            // operands formed right before the k-step that consumes them: the vector instructions sit between the MFMAs of the
            // product (in their shadow) instead of in a block of their own in front of it
            const float rstd = normalize_stats<V, 2>(X, u);
            CF_BIAS(g, ws + wl + V::L_C1);
            gemm_regs<2, V::KS_RES>(g, rs, lane, wl + V::L_A1, [&](int s) { return u[s / K][s % K] *= rstd; });
            // scores of this token (query) against the 4 tokens of its sample (key = token ^ m), exact softmax; the 1 / 8 of
            // dim_head ** -0.5 sits in A1 / c1
            // (packed: the DPP-permuted copies of a register pair serve both the scores and the probability-weighted sum)
            f32x2 d0 = splat(0.f), d1 = splat(0.f), d2 = splat(0.f), d3 = splat(0.f);
            f32x2 k1[2][K / 2], k2[2][K / 2], k3[2][K / 2];
            float e1[2], e2[2], e3[2], s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int i = 0; i < K / 2; ++i) {
                    const f32x2 q = CF_PAIR(g[t], i), k = CF_PAIR(u[t], i);
                    k1[t][i] = f32x2{tok_xor<1>(k.x), tok_xor<1>(k.y)};
                    k2[t][i] = f32x2{tok_xor<2>(k.x), tok_xor<2>(k.y)};
                    k3[t][i] = f32x2{tok_xor<3>(k.x), tok_xor<3>(k.y)};
                    d0 = pk_fma(q, k, d0); d1 = pk_fma(q, k1[t][i], d1); d2 = pk_fma(q, k2[t][i], d2); d3 = pk_fma(q, k3[t][i], d3);
                }
                const float q = g[t][K - 1], k = u[t][K - 1];
                e1[t] = tok_xor<1>(k); e2[t] = tok_xor<2>(k); e3[t] = tok_xor<3>(k);
                s0 = fmaf(q, k, s0); s1 = fmaf(q, e1[t], s1); s2 = fmaf(q, e2[t], s2); s3 = fmaf(q, e3[t], s3);
            }
            s0 = half_sum(s0 + (d0.x + d0.y)); s1 = half_sum(s1 + (d1.x + d1.y));          // the lane halves hold different features
            s2 = half_sum(s2 + (d2.x + d2.y)); s3 = half_sum(s3 + (d3.x + d3.y));
            const float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
            float p0 = __expf(s0 - mx), p1 = __expf(s1 - mx), p2 = __expf(s2 - mx), p3 = __expf(s3 - mx);
            const float inv = 1.0f / ((p0 + p1) + (p2 + p3));
            p0 *= inv; p1 *= inv; p2 *= inv; p3 *= inv;
            // the block's output is summed on its own (bias + 26 k-steps) and meets the residual stream in ONE addition per
            // element, as in the reference (x = to_out(...) + x, simple_vit.py:84): accumulating the k-steps on top of X
            // would round every partial sum at the magnitude of the residual stream (tools/attribute_vit.py)
            f32x16 a[2];
            CF_BIAS(a, ws + wl + V::L_C2);
            gemm_regs<2, V::KS_RES>(a, rs, lane, wl + V::L_A2, [&](int s) {
                const int t = s / K, r = s % K;
                if (r == K - 1) return fmaf(p3, e3[t], fmaf(p2, e2[t], fmaf(p1, e1[t], p0 * u[t][r])));
                const f32x2 c1 = k1[t][r / 2], c2 = k2[t][r / 2], c3 = k3[t][r / 2];
                return fmaf(p3, r & 1 ? c3.y : c3.x, fmaf(p2, r & 1 ? c2.y : c2.x, fmaf(p1, r & 1 ? c1.y : c1.x, p0 * u[t][r]))); });
            add_tiles<K>(X, a);
        }
        {
            f32x16 u[2], h[2];
            const float rstd = normalize_stats<V, 2>(X, u);
            CF_BIAS(h, ws + wl + V::L_B1);
            gemm_regs<2, V::KS_RES>(h, rs, lane, wl + V::L_W1, [&](int s) { return u[s / K][s % K] * rstd; });
            f32x16 a[2];
            CF_BIAS(a, ws + wl + V::L_B2);                                   // W2 h + b2 on its own, then one add into the residual
            gemm_regs<2, V::KS_RES>(a, rs, lane, wl + V::L_W2, [&](int s) { return gelu1(h[s / K][s % K]); });
            add_tiles<K>(X, a);
        }
        dump(l + 1);
    }
    f32x16 hn[2];
    layernorm_affine<V>(X, hn, ws + off_lno<V>(depth), ws + off_lno<V>(depth) + 64, lk);     // transformer.norm: tile 0 = t, tile 1 = raw

    // ================= affine map, log-det, stores                                 (coupling.py:139-155)
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const int ii = pos_of(r), c = chan_of(r);            // register of both tiles -> (position ii, channel c)
        const float ls = 2.0f * tanhf(0.5f * hn[1][r]);
        const float z1 = fmaf(y[1][r], expf(ls), hn[0][r]);
        lsum += ls;
        if (live) {
            zb[(CIN + c) * HW + 2 * n + ii] = z1;
#ifndef CF_VS_TICKS
            if (hout) {
                float* hb = hout + (int64_t)smp * C * HW;
                hb[c * HW + 2 * n + ii] = hn[0][r];
                hb[(CIN + c) * HW + 2 * n + ii] = hn[1][r];
            }
#endif
        }
    }
    lsum += tok_xor<1>(lsum);
    lsum += tok_xor<2>(lsum);
    lsum = half_sum(lsum);
    if (live && lk == 0 && n == 0) ldj_acc[smp] += ws[0] + lsum;
#ifdef CF_VS_TICKS          // probe build (tools/dev/vit_clock.py): shader cycles and 100 MHz reference ticks this wave was alive
    if (hout && lane == 0) {
        float* o = hout + (int64_t)(blockIdx.x * 4 + wave) * 2;
        o[0] = (float)(__builtin_readcyclecounter() - tick0); o[1] = (float)(__builtin_amdgcn_s_memrealtime() - real0);
    }
#endif
}

using VS26 = VS<26>;

bool step_ok(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads) {
    return C == 26 && H == 8 && W == 1 && p1 == 2 && p2 == 1 && dim == 2 * C && dim_head == 64 && heads == 1;
}

}  // namespace

extern "C" {

int cf_vit_step_supported(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads) {
    return step_ok(C, H, W, p1, p2, dim, dim_head, heads) ? 1 : 0;
}

int64_t cf_vit_step_ws_bytes(int C, int depth) { return C == 26 ? (int64_t)ws_floats<VS26>(depth) * 4 : 0; }

int cf_vit_step_prepare(const float* Wm, const float* t, const float* logs, const float* flat_vit_params, const float* pos,
                        void* ws, int C, int depth, cf_stream_t stream) {
    CF_REQUIRE(Wm && t && logs && flat_vit_params && pos && ws && depth >= 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0);
    if (C != 26) { cf_set_error("cf_vit_step_prepare: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    float* w = (float*)ws;
    int rc = cf_slogdet_inverse(Wm, C, w + 1, nullptr, stream);
    if (rc) return rc;
    if (depth > 0) {
        VitFuseBatch fb{};
        fb.layers[0] = flat_vit_params + 2 * C + VS26::DIM * C + VS26::DIM + 2 * VS26::DIM;
        fb.scratch[0] = w + off_fuse<VS26>(depth);
        k_vit_fuse<VS26::DIM, VS26::HEAD><<<dim3(depth, 2, FUSE_SPLIT), dim3(256), 0, cf_s(stream)>>>(fb, depth);
    }
    k_vit_step_pack<VS26><<<dim3(64), dim3(256), 0, cf_s(stream)>>>(Wm, t, logs, flat_vit_params, pos, w, depth);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_vit_step_fwd(const float* x, float* z, float* ldj_acc, const void* ws, float* h_out, int B, int C, int depth,
                    int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && B > 0 && x_bstride >= (int64_t)C * 8);
    if (C != 26) { cf_set_error("cf_vit_step_fwd: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    k_vit_step<VS26><<<dim3((unsigned)((B + 31) / 32)), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, (const float*)ws, B,
                                                                                    x_bstride, depth, h_out);
    CF_LAUNCH_CHECK();
    return 0;
}

int64_t cf_vit_step_macs(int C, int depth, int what) {
    if (C != 26 || depth < 0) return 0;
    using V = VS26;
    const int64_t tok = V::NTOK, D = V::DIM, HD = V::HEAD;
    if (what == 0)           // the reference: Conv1x1, patch Linear, per layer qkv + out + the two MLP Linears, q.k^T and p.v (4 x 4 x 64 each)
        return (int64_t)C * C * V::HW + tok * (D * C + depth * (3 * HD * D + D * HD + 2 * D * D)) + depth * 2 * tok * tok * HD;
    if (what == 1)           // MFMAs of a wave (8 samples): 2 tiles x k-steps per product, 2 048 multiply-adds each
        return (int64_t)(2 * V::KS0 + 2 * V::KS_IN + depth * 4 * 2 * V::KS_RES) * 2048 / 8;
    return (int64_t)C * C * V::HW + tok * (D * C + depth * 4 * D * D);
}

// training forward at saturating batches: cf_vit_step_fwd that also writes the residual stream at the layer boundaries -
// xtape: cf_vit_step_tape_floats(B, C, depth) floats, [depth + 1][2C][T] with T = 4 * (B rounded up to 32) tokens.
int64_t cf_vit_step_tape_tokens(int B) { return 4ll * ((B + 31) / 32 * 32); }
int64_t cf_vit_step_tape_floats(int B, int C, int depth) { return C == 26 ? (int64_t)(depth + 1) * 2 * C * cf_vit_step_tape_tokens(B) : 0; }

int cf_vit_step_fwd_taped(const float* x, float* z, float* ldj_acc, const void* ws, float* xtape, int B, int C, int depth,
                          int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && xtape && B > 0 && x_bstride >= (int64_t)C * 8);
    if (C != 26) { cf_set_error("cf_vit_step_fwd_taped: C=%d unsupported", C); return CF_ERR_UNSUPPORTED; }
    k_vit_step<VS26, true><<<dim3((unsigned)((B + 31) / 32)), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, (const float*)ws, B,
                                                                                          x_bstride, depth, nullptr, xtape,
                                                                                          cf_vit_step_tape_tokens(B));
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
