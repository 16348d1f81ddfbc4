// Shared device code of the fused flow-step kernels (forward, inverse, backward): compile-time geometry,
// MFMA operand pipeline, wave-local staging, the conditioner (phases 1-3).  Included by cf_step.hip and
// cf_step_bwd.hip; everything lives in an anonymous namespace of the including translation unit.
#pragma once
#include <type_traits>
#include "cf_common.h"
#include <math.h>

namespace {


typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWaves = 4;

// ---- compile-time geometry of one supported shape ------------------------------------------------
// PIPE_: form of phase 2, the 3x3 (0 = compiler-scheduled per-tap loop, 1 = explicit two-stage operand pipeline, 2 = all nine
// taps unrolled with precomputed row / column offset parts, 3 = Winograd F(2x2,3x3): winograd_phase2)
// ABL_: timing-only ablations for tools/step_bench.py (1 = phase-2 operands are constants: no LDS / L2 loads there)
// PATCH_: (backward) every row of an LDS plane carries, behind its PIX pixel columns, the fold sums of the transposed
// reflect-padded 3x3 (cf_step_bwd.hip): PP slots per sample + one zero slot; RS = row stride of all LDS planes.
template <int C_, int H_, int W_, int SPW_, int PIPE_ = 1, int ABL_ = 0, int PATCH_ = 0>
struct Geo {
    static constexpr int C = C_, H = H_, W = W_, SPW = SPW_, PIPE = PIPE_, ABL = ABL_, PATCH = PATCH_;
    static constexpr int HW = H * W;
    static constexpr int PIX = SPW * HW;              // pixels per workgroup
    static constexpr int HALF = C / 2;                // conditioner channels
    static constexpr int HID = 2 * C;                 // hidden width of the coupling net
    static constexpr int NPT = PIX / 32;              // 32-pixel tiles per workgroup
    static constexpr int PTW = NPT / kWaves;          // pixel tiles per wave
    static constexpr int HP = HALF <= 16 ? 16 : 32;   // packed rows per channel half
    static constexpr int R03 = 2 * HP;                // packed rows of the C-channel results (y, h)
    static constexpr int RT03 = R03 / 32;
    static constexpr int RT1 = (HID + 31) / 32;       // row tiles of the hidden planes
    static constexpr int R1 = RT1 * 32;
    // k-steps (2 k per MFMA) and 4-step groups of every contraction
    static constexpr int KS0 = C / 2, KS1 = HALF / 2, KS3 = HID / 2;
    static constexpr int NG0 = (KS0 + 3) / 4, NG1 = (KS1 + 3) / 4, NG3 = (KS3 + 3) / 4;
    static constexpr int NCG = HID / 8;               // groups per 3x3 tap
    static constexpr int NG2 = 9 * NCG;
    // workspace layout (floats)
    static constexpr int OFF_B0 = 4, OFF_B1 = OFF_B0 + R03, OFF_B2 = OFF_B1 + R1, OFF_B3 = OFF_B2 + R1;
    static constexpr int OFF_A0 = OFF_B3 + R03;
    static constexpr int OFF_A1 = OFF_A0 + NG0 * RT03 * 256;
    static constexpr int OFF_A2 = OFF_A1 + NG1 * RT1 * 256;
    static constexpr int OFF_A3 = OFF_A2 + NG2 * RT1 * 256;
    // 16-row phases on v_mfma_f32_16x16x4_f32 (k_flow_step_small: C <= 16 on 16x16 images, one sample per workgroup).
    // With <= 16 real output rows a 32x32x2 tile spends half (or three quarters) of its cycles on padding rows; the
    // 16x16x4 form has the same flop/cycle and no padding.  Extra packed operands, 4 k-steps (16 k) per float4 group:
    //   S0 = e^{-logs} Wm (rows: packed y0 / y1), S3 = NN.4 (rows: packed t / raw); when the hidden planes are one
    //   16-row tile too (C = 8): S1 = NN.0, S2 = NN.2 (9 taps).  SB0 / SB3: the 16 packed-row biases.
    static constexpr bool SMALL = (H == 16 && W == 16 && SPW == 1 && C <= 16);
    static constexpr bool HID16 = SMALL && HID == 16;
    static constexpr int SG0 = (C + 15) / 16, SG3 = (HID + 15) / 16, SG1 = (HALF + 15) / 16;
    static constexpr int OFF_SB0 = OFF_A3 + NG3 * RT03 * 256;
    static constexpr int OFF_SB3 = OFF_SB0 + (SMALL ? 16 : 0);
    static constexpr int OFF_SA0 = OFF_SB3 + (SMALL ? 16 : 0);
    static constexpr int OFF_SA3 = OFF_SA0 + (SMALL ? SG0 * 256 : 0);
    static constexpr int OFF_SA1 = OFF_SA3 + (SMALL ? SG3 * 256 : 0);
    static constexpr int OFF_SA2 = OFF_SA1 + (HID16 ? SG1 * 256 : 0);
    // Winograd F(2x2,3x3) form of the 3x3 (PIPE == 3, winograd_phase2 below): U = G w G^T for the 16 positions, packed as
    // 16x16x4 A fragments: [position][16-row tile][group of 4 k-steps][lane][4]
    static constexpr bool WINO = (PIPE_ == 3 || PIPE_ == 4);
    // PIPE == 4: the Winograd-domain products on the bf16 matrix cores - every fp32 operand as three bf16 pieces, the six piece
    // products with i + j <= 2 accumulated in fp32 (winograd_phase2; tools/micro/bf16_split_gemm.hip: the error of the f32 MFMA)
    static constexpr bool BF16S = (PIPE_ == 4);
    // PIPE == 5: the DIRECT 3x3 on the bf16 matrix cores (direct_bf16_phases, below): h1 is split once, by its producer, into three
    // bf16 planes [pixel][channel] in LDS; a tap's B operand is one 16-byte read per piece and feeds 6 MFMAs per 16-row tile; no
    // transform arithmetic at all, and the matrix pipe of v_mfma_f32_16x16x32_bf16 runs beside the vector pipe
    static constexpr bool DBF = (PIPE_ == 5);
    static constexpr int RT16 = HID / 16, KG4 = HID / 16;         // row tiles / k-step groups of the 16x16x4 products over HID
    static constexpr int OFF_AW = OFF_SA2 + (HID16 ? 9 * 256 : 0);
    // one-sample-per-workgroup form of the 4x4 level for small batches (k_flow_step_rs16, cf_step.hip): 16x16x4 A fragments
    // of all four products in NATURAL row order, [row tile][group of 4 k-steps][lane][4], + their biases
    static constexpr bool RS16 = (C == 64 && H == 4 && W == 4);
    static constexpr int OFF_RB0 = OFF_AW + 16 * HID * HID, OFF_RB1 = OFF_RB0 + C, OFF_RB2 = OFF_RB1 + HID, OFF_RB3 = OFF_RB2 + HID;
    static constexpr int OFF_RA0 = OFF_RB3 + C, OFF_RA1 = OFF_RA0 + (C / 16) * (C / 16) * 256;
    static constexpr int OFF_RA2 = OFF_RA1 + (HID / 16) * (HALF / 16) * 256, OFF_RA3 = OFF_RA2 + (HID / 16) * 9 * (HID / 16) * 256;
    static constexpr int R16_END = OFF_RA3 + (C / 16) * (HID / 16) * 256;
    static constexpr int WS_END0 = RS16 ? R16_END : OFF_AW + 16 * HID * HID;
    // bf16 pieces of the Winograd-domain weights (BF16S): [position][16-row tile][32-channel block][piece][lane] 16 bytes =
    // 8 bf16: element j of lane l = piece of U[pos][16 rt + (l & 15)][32 kb + 4 j + (l >> 4)] (the hardware's k index (l >> 4, j)
    // stands for channel 4 j + (l >> 4): the eight values a lane of winograd_phase2 forms over eight consecutive k-steps)
    static constexpr int KB32 = HID / 32;
    static constexpr int OFF_AWB = WS_END0;
    // bf16 pieces of the direct 3x3 taps (DBF): [tap][16-row tile][32-channel block][piece][lane] 16 bytes = 8 bf16: element j of
    // lane l = piece of NN.2[16 rt + (l & 15)][32 kb + 8 (l >> 4) + j][tap] (the hardware's own k order)
    static constexpr int OFF_ADB = WS_END0 + 16 * RT16 * KB32 * 3 * 256;
    static constexpr int WS_FLOATS = OFF_ADB + 9 * RT16 * KB32 * 3 * 256;
    static constexpr int PP = 2 * W + 2 * H + 4;      // fold slots per sample: 2 patched rows, 2 patched columns, 4 corners
    static constexpr int RS = PATCH ? ((PIX + SPW * PP + 1 + 3) & ~3) : PIX;
    // C = 8 (HID = 16) in the Winograd form: all 16 Winograd-domain weight matrices are 16 KB - staged into LDS once per workgroup
    // (LDS_W floats behind the planes) instead of streamed per wave through L1 (that delivery costs 6.7 % of the level:
    // -DCF_ABL_FIXEDW); the larger levels fill their LDS with planes
#ifndef CF_WINO_LDSW
#define CF_WINO_LDSW 1
#endif
    static constexpr int LDS_W = (CF_WINO_LDSW && PIPE_ == 3 && HID == 16 && H == 16 && W == 16 && SPW == 1) ? 16 * HID * HID : 0;
    static constexpr int DBF_FLOATS = 3 * PIX * HID / 2;        // three bf16 planes [pixel][HID]; they alias the fp32 planes
    static constexpr int LDS_FLOATS = (DBF && DBF_FLOATS > (HALF + HID) * RS) ? DBF_FLOATS : (HALF + HID) * RS + LDS_W;
    // waves per SIMD the register allocator must leave room for = workgroups per CU the LDS footprint admits
    static constexpr int MINW = (160 * 1024) / (LDS_FLOATS * 4) >= 4 ? 4 : ((160 * 1024) / (LDS_FLOATS * 4) >= 2 ? 2 : 1);
    static_assert(PIX % 128 == 0 && PTW >= 1, "workgroup must own a multiple of 128 pixels");
    static_assert(HID % 8 == 0 && C % 4 == 0, "channel counts must fill whole k-steps");
    static_assert((HW & (HW - 1)) == 0 && (W & (W - 1)) == 0, "power-of-two images");
};

// packed row p of a C-channel result -> channel, or -1 for a padding row.  First-half channels sit
// in rows [0, HALF), second-half channels in rows [HP, HP + HALF): t / log_s / x1 of one channel
// then share a lane and differ by a fixed register offset.
template <class G> __host__ __device__ constexpr int chan_of_row(int p) {
    return ((p % G::HP) < G::HALF) ? (p / G::HP) * G::HALF + (p % G::HP) : -1;
}

// 16-row packing of the 16x16x4 phases: row 4g + j (g = lane >> 4 owns it in the result tile): j = 0,1 -> first-half
// channel 2g + j, j = 2,3 -> second-half channel HALF + 2g + (j - 2): t / raw / y1 of one channel again share a lane.
template <class G> __host__ __device__ constexpr int chan_of_row16(int p) {
    return (2 * (p >> 2) + (p & 1) < G::HALF) ? ((p & 2) ? G::HALF : 0) + 2 * (p >> 2) + (p & 1) : -1;
}

// ---- helpers -----------------------------------------------------------------------------------------
// accumulator tile initialised with the per-row bias: row(r, lk) = (r&3) + 8*(r>>2) + 4*lk
__device__ __forceinline__ f32x16 bias_tile(const float* __restrict__ bias32, int lk) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(bias32 + 8 * q + 4 * lk);
        a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
    return a;
}
__device__ __forceinline__ int tile_row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
// ReLU as ONE integer max on the bit pattern (negative floats are negative integers; -0 -> +0).  fmaxf(x, 0) costs two
// VALU instructions here (hipcc quiets a possible sNaN with v_max x,x first), and in the MFMA kernels every VALU
// instruction is time the matrix pipe of that SIMD stands still.
__device__ __forceinline__ float cf_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// log_s = 2 tanh(raw / 2) = 2 - 4 / (e^raw + 1)   (coupling.py:55-56) on the hardware exp / rcp path (v_exp_f32, v_rcp_f32:
// 1 ulp each): |abs err| <= ~3e-7 per element, i.e. ~1e-8 bits/dim after the per-sample sum (tolerance 1e-5; measured on
// the stress fixtures with |raw| up to 25: tests/test_gpu_parity.py::test_e2e_stress_regimes).  e^raw = inf gives 2,
// e^raw = 0 gives -2 exactly.  (`__fdividef` is an IEEE division on this toolchain: ~10 more VALU per element.)
__device__ __forceinline__ float cf_log_scale(float raw) { return fmaf(-4.0f, __builtin_amdgcn_rcpf(__expf(raw) + 1.0f), 2.0f); }
__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// Operands of one 4-k-step group: RT weight fragments (one 16-byte load each) and 4*PTW activation values.
template <int RT, int PTW>
struct GroupOps {
    float4 a[RT];
    float b[4][PTW];
};

template <int RT, int PTW>
__device__ __forceinline__ void group_mma(f32x16 (&acc)[RT][PTW], const GroupOps<RT, PTW>& o, int nsteps) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (e < nsteps) {
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(o.a[rt], e), o.b[e][q], acc[rt][q], 0, 0, 0);
        }
    }
}

// Packed weight fragments are fetched through a buffer resource: `buffer_load_dwordx4 v, v_lane16, s[rsrc], s_off offen`
// takes the group / row-tile / tap offset as a SCALAR, so no per-load 64-bit vector address arithmetic sits between the
// MFMAs (the flat form costs v_lshl_add_u64 / v_add_co / v_addc per fragment).
typedef __amdgpu_buffer_rsrc_t ws_rsrc_t;
__device__ __forceinline__ ws_rsrc_t ws_rsrc(const float* ws, int floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ws), 0, floats * 4, 0x00020000);
}
// fragment at float offset `foff` (wave-uniform) + this lane's 16 bytes
__device__ __forceinline__ float4 ws_frag(ws_rsrc_t rs, int lane, int foff) {
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, foff * 4, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

// One dense phase whose B operand is an LDS plane [k][PIX]:  acc[rt][q] += A_frag * plane
//   KS real k-steps, NG groups of 4, RT row tiles, frags = packed A of this phase.
// Software pipeline: the operands of group g+1 are requested BEFORE the MFMAs of group g are issued
// (sched_barrier pins the loads there: left alone, hipcc sinks them next to their first use and every
// group then stalls on an L2 round trip).
template <class G, int KS, int NG, int RT, class FRAG>
__device__ __forceinline__ void dense_phase_impl(f32x16 (&acc)[RT][G::PTW], FRAG frag,
                                                 const float* __restrict__ plane, const int (&pix)[G::PTW], int lane) {
    const int lk = lane >> 5;
    cf_wave_sync();                      // the operand plane was written by other lanes of this wave
    GroupOps<RT, G::PTW> ops[2];
    auto load = [&](int g, GroupOps<RT, G::PTW>& o) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) o.a[rt] = frag(g * RT + rt);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < G::PTW; ++q)
                o.b[e][q] = (4 * g + e < KS) ? plane[(2 * (4 * g + e) + lk) * G::RS + pix[q]] : 0.f;
    };
    load(0, ops[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) load(g + 1, ops[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        group_mma<RT, G::PTW>(acc, ops[g & 1], KS - 4 * g);
    }
}
template <class G, int KS, int NG, int RT>
__device__ __forceinline__ void dense_phase(f32x16 (&acc)[RT][G::PTW], const float4* __restrict__ frags,
                                            const float* __restrict__ plane, const int (&pix)[G::PTW], int lane) {
    dense_phase_impl<G, KS, NG, RT>(acc, [&](int i) { return frags[i * 64 + lane]; }, plane, pix, lane);
}
// same, fragments through a buffer resource at float offset `foff`
template <class G, int KS, int NG, int RT>
__device__ __forceinline__ void dense_phase(f32x16 (&acc)[RT][G::PTW], ws_rsrc_t rs, int foff,
                                            const float* __restrict__ plane, const int (&pix)[G::PTW], int lane) {
    dense_phase_impl<G, KS, NG, RT>(acc, [&](int i) { return ws_frag(rs, lane, foff + i * 256); }, plane, pix, lane);
}

// Streaming accesses of the step kernels (x is read once, z written once per step): non-temporal, so that they do not displace
// the weight fragments every workgroup re-reads from L1 / L2 (-DCF_STREAM_NT=0: plain accesses, A/B)
#ifndef CF_STREAM_NT
#define CF_STREAM_NT 1
#endif
typedef float cf_f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 cf_ld_stream(const float4* p) {
#if CF_STREAM_NT
    const cf_f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const cf_f32x4v*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *p;
#endif
}
__device__ __forceinline__ float cf_ldf_stream(const float* p) {
#if CF_STREAM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void cf_st_stream(float4* p, const float4 v) {
#if CF_STREAM_NT
    __builtin_nontemporal_store(cf_f32x4v{v.x, v.y, v.z, v.w}, reinterpret_cast<cf_f32x4v*>(p));
#else
    *p = v;
#endif
}

// ---- wave-local staging of activations with 16-byte global accesses ---------------------------------
// A wave owns WPX = 32*PTW pixel columns.  Item n = i*64 + lane enumerates its C x WPX block of x.
template <class G, bool SQ>
__device__ __forceinline__ void x_load(float4 (&xr)[G::C * G::PTW / 8], const float* __restrict__ x, int64_t xbs,
                                       int tile, int B, int wave, int lane) {
    constexpr int WPX = 32 * G::PTW, HW = G::HW, W = G::W;
    const int tb0 = tile * G::SPW;
#pragma unroll
    for (int i = 0; i < G::C * G::PTW / 8; ++i) {
        const int n = i * 64 + lane;
        if constexpr (!SQ) {
            const int ch = n / (WPX / 4), col = wave * WPX + 4 * (n % (WPX / 4));
            const int b = min(tb0 + col / HW, B - 1);
            xr[i] = cf_ld_stream(reinterpret_cast<const float4*>(x + (int64_t)b * xbs + ch * HW + col % HW));
        } else {
            // un-squeezed row (2y+i1), 4 consecutive floats = channels (4c'+2i1, +1) of squeezed pixels (x, x+1)
            const int cp = n / (WPX / 2), col = wave * WPX + 2 * (n % (WPX / 2));
            const int b = min(tb0 + col / HW, B - 1);
            const int p = col % HW, yy = p / W, xx = p % W;
            xr[i] = cf_ld_stream(reinterpret_cast<const float4*>(x + (int64_t)b * xbs + (cp >> 1) * 4 * HW + (2 * yy + (cp & 1)) * 2 * W + 2 * xx));
        }
    }
}

template <class G, bool SQ>
__device__ __forceinline__ void x_to_lds(const float4 (&xr)[G::C * G::PTW / 8], float* __restrict__ plane, int wave, int lane) {
    constexpr int WPX = 32 * G::PTW, PIX = G::RS;      // row stride of the plane
#pragma unroll
    for (int i = 0; i < G::C * G::PTW / 8; ++i) {
        const int n = i * 64 + lane;
        if constexpr (!SQ) {
            const int ch = n / (WPX / 4), col = wave * WPX + 4 * (n % (WPX / 4));
            *reinterpret_cast<float4*>(&plane[ch * PIX + col]) = make_float4(xr[i].x, xr[i].y, xr[i].z, xr[i].w);
        } else {
            const int cp = n / (WPX / 2), col = wave * WPX + 2 * (n % (WPX / 2));
            *reinterpret_cast<float2*>(&plane[(2 * cp) * PIX + col]) = make_float2(xr[i].x, xr[i].z);
            *reinterpret_cast<float2*>(&plane[(2 * cp + 1) * PIX + col]) = make_float2(xr[i].y, xr[i].w);
        }
    }
    cf_wave_sync();
}

// write NROWS channel rows (channels ch0 .. ch0+NROWS-1 of a (B,C,H,W) tensor) from an LDS plane [row][PIX]
// (this wave's columns), 16 bytes per lane
template <class G, int NROWS>
__device__ __forceinline__ void rows_store(float* __restrict__ z, const float* __restrict__ plane, int tb0, int ch0, int B,
                                           int wave, int lane) {
    constexpr int WPX = 32 * G::PTW, HW = G::HW, PIX = G::RS, C = G::C;
    cf_wave_sync();                      // rows written by other lanes of this wave
#pragma unroll
    for (int i = 0; i < (NROWS * G::PTW + 7) / 8; ++i) {
        const int n = i * 64 + lane;
        const int idx = n / (WPX / 4), col = wave * WPX + 4 * (n % (WPX / 4));
        const int b = tb0 + col / HW;
        if (idx < NROWS && b < B)
            cf_st_stream(reinterpret_cast<float4*>(z + (int64_t)b * C * HW + (int64_t)(ch0 + idx) * HW + col % HW),
                         *reinterpret_cast<const float4*>(&plane[idx * PIX + col]));
    }
    cf_wave_sync();                      // ... read back before anyone reuses the words
}
template <class G>
__device__ __forceinline__ void z_store(float* __restrict__ z, const float* __restrict__ plane, int tb0, int ch0, int B,
                                        int wave, int lane) {
    rows_store<G, G::HALF>(z, plane, tb0, ch0, B, wave, lane);
}

// rows_store for a target tensor with CT channels per sample, and its inverse (global -> this wave's columns of a plane).
// Both go through a buffer resource that covers exactly the VALID samples of this workgroup's tile of the tensor: the
// per-lane byte offset is one 32-bit register for every plane of that shape (item i adds a constant, which travels in the
// scalar offset operand), there is no 64-bit address arithmetic and no `b < B` compare - an access beyond the last sample
// is out of range for the resource: stores are dropped and loads return 0 by the hardware's range check.
template <class G, int CT>
__device__ __forceinline__ ws_rsrc_t tile_rsrc(const float* t, int tb0, int B) {
    const int nb = min(B - tb0, G::SPW);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t + (int64_t)tb0 * CT * G::HW), 0, nb * CT * G::HW * 4, 0x00020000);
}
// 16-byte store through a buffer resource with a SCALAR offset, as ONE unit with a wait state behind it.  Measured on
// gfx950: the VGPRs holding the data of a `buffer_store_dwordx4 ... s_off offen` must not be written by the VALU
// instruction right behind it (hipcc places one there when the registers become free - it treats a store with a
// register soffset as hazard-free - and the stored rows then carry the new register contents in some lanes:
// per-call checks of the weight-gradient inputs found 64-bit addresses inside the g_h2 plane; reproducer: tools/micro/store_hazard.hip).  Inline asm: the compiler cannot slide anything
// between the store and its s_nop.
typedef int cf_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cf_i32x4 tile_rsrc_words(const float* base, int bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    return cf_i32x4{(int)(uint32_t)p, (int)((uint32_t)(p >> 32) & 0xffffu), bytes, 0x00020000};
}
__device__ __forceinline__ void tile_store_b128(const cf_i32x4 rs, int voff, int soff, const float4 v) {
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const f32x4s d = {v.x, v.y, v.z, v.w};
#if CF_STREAM_NT      // planes of the tape / gradient planes: written once, read by another kernel - non-temporal, as the z rows
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" : : "v"(d), "v"(voff), "s"(rs), "s"(soff) : "memory");
#else
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(d), "v"(voff), "s"(rs), "s"(soff) : "memory");
#endif
}

template <class G, int NROWS, int CT>
__device__ __forceinline__ void rows_store_t(float* __restrict__ dst, const float* __restrict__ plane, int tb0, int B,
                                             int wave, int lane) {
    constexpr int WPX = 32 * G::PTW, HW = G::HW, PIX = G::RS, RPI = 256 / WPX;      // rows per item of 64 lanes x 16 bytes
    const cf_i32x4 rs = tile_rsrc_words(dst + (int64_t)tb0 * CT * HW, min(B - tb0, G::SPW) * CT * HW * 4);
    const int idx0 = lane / (WPX / 4), col = wave * WPX + 4 * (lane % (WPX / 4));
    const int voff = ((col / HW) * CT * HW + idx0 * HW + col % HW) * 4;
    cf_wave_sync();
#ifdef CF_ABL_NOSTORE                    // timing-only ablation (tools/dev/make_abl.py): no plane stores
    if (tb0 >= 0) return;
#endif
#pragma unroll
    for (int i = 0; i < (NROWS + RPI - 1) / RPI; ++i) {
        if (NROWS % RPI == 0 || idx0 + i * RPI < NROWS) {
            const float4 v = *reinterpret_cast<const float4*>(&plane[(idx0 + i * RPI) * PIX + col]);
            tile_store_b128(rs, voff, i * RPI * HW * 4, v);
        }
    }
    cf_wave_sync();
}
// the C rows of `plane` (this wave's own columns) to dst in the layout of the tensor BEFORE Squeeze((2,2)) (squeeze.py:10-11):
// squeezed channel c = 4 q + 2 dy + dx at pixel (y, x) is element (q, 2 y + dy, 2 x + dx) of the (C/4, 2H, 2W) sample.  A
// lane takes the channel pair (2 j, 2 j + 1) = (dx 0, 1) of q = j >> 1, dy = j & 1 at 4 pixels of one image row: 8
// consecutive floats of the destination, two 16-byte stores.
template <class G>
__device__ __forceinline__ void rows_store_unsq(float* __restrict__ dst, const float* __restrict__ plane, int tb0, int B,
                                                int wave, int lane) {
    constexpr int WPX = 32 * G::PTW, HW = G::HW, W = G::W, PIX = G::RS, C = G::C, Q4 = WPX / 4;
    static_assert(C % 4 == 0 && W % 4 == 0, "channel quadruples, 4 pixels of one image row");
    cf_wave_sync();
    for (int it = lane; it < (C / 2) * Q4; it += 64) {
        const int j = it / Q4, col = wave * WPX + 4 * (it - j * Q4);
        const int smp = tb0 + col / HW, p = col % HW, yy = p / W, xx = p - yy * W;
        if (smp < B) {
            const float4 a = *reinterpret_cast<const float4*>(&plane[(2 * j) * PIX + col]);
            const float4 b = *reinterpret_cast<const float4*>(&plane[(2 * j + 1) * PIX + col]);
            float* d = dst + (int64_t)smp * C * HW + (j >> 1) * 4 * HW + (2 * yy + (j & 1)) * 2 * W + 2 * xx;
            *reinterpret_cast<float4*>(d) = make_float4(a.x, b.x, a.y, b.y);
            *reinterpret_cast<float4*>(d + 4) = make_float4(a.z, b.z, a.w, b.w);
        }
    }
    cf_wave_sync();
}
template <class G, int NROWS, int CT>
__device__ __forceinline__ void rows_load_t(const float* __restrict__ src, float* __restrict__ plane, int tb0, int B,
                                            int wave, int lane) {
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    constexpr int WPX = 32 * G::PTW, HW = G::HW, PIX = G::RS, RPI = 256 / WPX;
    const ws_rsrc_t rs = tile_rsrc<G, CT>(src, tb0, B);
    const int idx0 = lane / (WPX / 4), col = wave * WPX + 4 * (lane % (WPX / 4));
    const int voff = ((col / HW) * CT * HW + idx0 * HW + col % HW) * 4;
    cf_wave_sync();                      // earlier readers of these words (other lanes) are done
#pragma unroll
    for (int i = 0; i < (NROWS + RPI - 1) / RPI; ++i) {
        if (NROWS % RPI == 0 || idx0 + i * RPI < NROWS) {
            const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * RPI * HW * 4, 0);
            *reinterpret_cast<float4*>(&plane[(idx0 + i * RPI) * PIX + col]) =
                make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
        }
    }
    cf_wave_sync();
}

// training tape of a step (forward with DUMP writes, backward with TAPED reads):
//   y0 (B, C/2, HW), h1 / h2 (B, 2C, HW; post-ReLU hidden planes of the conditioner): operands of the weight-gradient GEMMs;
//   ls, y1 (B, C/2, HW): log-scale and the second half of the step's Conv1x1+ActNorm output - with them the backward
//   kernel needs neither the step input nor a recompute of the first and last 1x1;
//   m1 / m2: the ReLU masks of h1 / h2 as bit words in the accumulator layout of the 32x32x2 tiles: word
//   [(T * RT1 + rt) * 64 + lane], T = global 32-pixel tile (sample-major pixel index / 32), bit r = row rt*32 + tile_row(r, lane>>5)
//   of pixel 32 T + (lane & 31).  Sized for the batch rounded up to 16 samples (cf_flow_step_tape_aux_bytes).
struct StepTape {
    float* y0;
    float* h1;
    float* h2;
    float* ls;
    float* y1;
    unsigned* m1;
    unsigned* m2;
};
constexpr StepTape kNoTape{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
// host: the four small tape items live in ONE caller-provided buffer `aux` = [ls | y1 | m1 | m2]
inline int64_t tape_mask_words(int B, int C, int H, int W) {
    return (int64_t)((B + 15) & ~15) * H * W / 32 * ((2 * C + 31) / 32) * 64;
}
inline int64_t tape_aux_bytes(int B, int C, int H, int W) {
    return (2 * (int64_t)B * (C / 2) * H * W + 2 * tape_mask_words(B, C, H, W)) * 4;
}
inline StepTape make_tape(float* y0, float* h1, float* h2, void* aux, int B, int C, int H, int W) {
    const int64_t nh = (int64_t)B * (C / 2) * H * W, nm = tape_mask_words(B, C, H, W);
    float* a = (float*)aux;
    return StepTape{y0, h1, h2, a, a + nh, (unsigned*)(a + 2 * nh), (unsigned*)(a + 2 * nh) + nm};
}

// ReLU mask words of RT x PTW accumulator tiles (pre- or post-ReLU values: the bit is `value > 0` either way)
template <class G, int RT>
__device__ __forceinline__ void mask_store(unsigned* __restrict__ m, const f32x16 (&acc)[RT][G::PTW], int tile, int wave, int lane) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < G::PTW; ++q) {
            unsigned b = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) b |= (acc[rt][q][r] > 0.f ? 1u : 0u) << r;
            m[((int64_t)(tile * G::NPT + wave * G::PTW + q) * RT + rt) * 64 + lane] = b;
        }
}

// ---- Winograd F(2x2,3x3) form of the reflect-padded 3x3 (phase 2 of PIPE == 3 geometries) ---------------------------
// The 3x3 holds 72 of the 80 C^2 HW multiply-adds of a step and the exact-fp32 matrix pipe is the bound of the whole path,
// so the way past it is fewer multiplications: per 2x2 output tile and channel pair, 16 products instead of 36,
//     Y = A^T [ sum_ci U[ci] (.) (B^T d[ci] B) ] A,     U = G w G^T (packed once, fp64 -> fp32, k_step_pack),
// d = the 4x4 input patch of the tile (reflect-padded), B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],
// A^T = [[1,1,1,0],[0,1,-1,-1]].  All entries are 0 / +-1 (the 1/2 of G lives in the packed weights): the transforms are
// additions, the error of the fp32 result is ~1.6x that of the direct sum (tests/dev_winograd_numerics.py: bits/dim of
// every end-to-end fixture unchanged to its last fp32 digit).
// Mapping: the 16 "positions" (xi, nu) are 16 independent (HID x HID) x (HID x tiles) products on v_mfma_f32_16x16x4_f32;
// a wave owns the 16 tiles (= 64 output pixels) of its own pixel columns and ALL output rows, so the B operand - one
// Winograd-domain value per lane and k-step = 4 LDS reads + 3 additions - serves HID/16 MFMAs, and h2 goes straight back to
// this wave's columns.  The output transform is folded position by position into the four Y accumulators (+-M).
// h1 layout (written by phase 1 of the same geometry): pixel (s, y, x) of row k sits at
//     k PIX + [ s HW + (y&1) HW/2 + (y>>1) W + (x&1) W/2 + (x>>1) ]  ^  (k&1) W/2
// - rows and columns split by parity, so the stride-2 patch reads of neighbouring tiles are consecutive words, and the
// XOR moves odd rows k to the other column-parity block: the 32 lanes of a half wave (16 tiles x 2 channels) hit 32 banks.
template <class G> __device__ __forceinline__ int wino_pix(int sHW, int y, int x) {
    int p = sHW + (y & 1) * (G::HW / 2) + (y >> 1) * G::W + (x & 1) * (G::W / 2) + (x >> 1);
    if constexpr (G::HW == 16) p ^= ((p >> 5) & 1) << 3;       // 4x4: four samples per wave - fold the sample bit that would alias
    return p;
}

typedef float f32x4w __attribute__((ext_vector_type(4)));
#ifndef CF_WINO_PEEL
#define CF_WINO_PEEL 1
#endif

// Loop form: the 16 positions run as a RUNTIME loop over xi (row of B^T / A^T on the vertical axis) around the four nu,
// unrolled.  (All 16 unrolled was measured first: 4x the code, and hipcc spilled 50-150 registers at 2-3 workgroups / CU:
// 561 / 448 / 414 us per 16384 samples at C = 16 / 32 / 64 against 447 / 372 / 406 us in this form; direct 3x3: 640 / 618 /
// 620 us.)  What depends on xi is data: the two patch rows (xi = 2 takes them swapped, so that the vertical transform is
// t1 + sigma t2 with sigma = +1 for xi = 1 and -1 otherwise), the fragment offset (a scalar), the coefficients A^T[i][xi].
// Operands of a group of 4 k-steps (RT16 weight fragments from L2, 4 patch values per Winograd-domain operand from LDS,
// k-steps paired so that the additions run as packed fp32) are requested one group ahead - the first group of the next xi
// during the last one of this xi - and sched_barrier keeps hipcc from sinking them to their first use.  k runs in chunks
// of <= 64 channels (outer loop): inside a chunk every operand address is `patch offset + immediate`; the output
// transform is linear, so each chunk's partial M is folded into Y.
template <class G>
__device__ __forceinline__ void winograd_phase2(float* __restrict__ lds, const float* __restrict__ wsl, ws_rsrc_t rs, int lane,
                                                     int wave) {
    constexpr int W = G::W, H = G::H, HW = G::HW, PIX = G::PIX, HALF = G::HALF, HID = G::HID, RT16 = G::RT16, KG4 = G::KG4;
    // Ownership.  PTW even: a wave takes the NT = PTW / 2 column tiles (16 output tiles = 64 pixels each) of its own pixel
    // columns and ALL output rows.  PTW = 1 (128 pixels per workgroup: half the LDS, two workgroups per CU): two waves share a
    // column tile and split the output ROWS (RSPLIT = 2) - the Winograd-domain operands are formed twice, h2 needs a workgroup
    // barrier before phase 3, but a second workgroup per CU covers the stalls of a lone wave per SIMD (C = 64: 406 -> see DESIGN).
    constexpr int RSPLIT = G::PTW == 1 ? 2 : 1, NT = G::PTW * RSPLIT / 2, NCW = 4 / RSPLIT, RTW = RT16 / RSPLIT;
    static_assert((G::PTW == 1 || G::PTW % 2 == 0) && HID % 16 == 0 && RT16 % RSPLIT == 0, "column tiles of 16 output tiles");
    const int cw = RSPLIT == 1 ? wave : wave % NCW;           // this wave's column-tile group ...
    const int rt0 = RSPLIT == 1 ? 0 : __builtin_amdgcn_readfirstlane((wave / NCW) * RTW);      // ... and first 16-row tile (a constant / a scalar)
    typedef float f32x2w __attribute__((ext_vector_type(2)));
    float* H1 = lds + HALF * PIX;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int TPS = HW / 4;
    int smp[NT], ty[NT], tx[NT], rpart[NT][4], cpart[NT][4];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const int tg = (cw * NT + ct) * 16 + l15, ti = tg % TPS;
        smp[ct] = tg / TPS; ty[ct] = ti / (W / 2); tx[ct] = ti % (W / 2);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int yy = 2 * ty[ct] - 1 + a, xx = 2 * tx[ct] - 1 + a;
            yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
            xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
            // BYTE offsets: an operand address is then one addition (row part + column part) in front of the ds_read's immediate
            rpart[ct][a] = 4 * (HALF * PIX + lg * PIX + wino_pix<G>(smp[ct] * HW, yy, 0));
            cpart[ct][a] = 4 * ((((xx ^ lg) & 1) * (W / 2)) + (xx >> 1));
        }
    }
    f32x4w Y[NT][2][2][RTW];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) Y[ct][i][j][rt] = f32x4w{0.f, 0.f, 0.f, 0.f};
    struct WFrag { float4 a[RTW]; };
    struct WPatch { f32x2w d[NT][2][4]; };
    constexpr int KGC = KG4 < 4 ? KG4 : 4, NCH = KG4 / KGC, NGX = 4 * KGC;       // groups per xi
    static_assert(NGX % 2 == 0, "static ping-pong");
    constexpr int B1[4] = {0, 1, 1, 1}, B2[4] = {2, 2, 2, 3};                       // patch columns of B^T row nu
    constexpr float S1[4] = {1.f, 1.f, -1.f, 1.f}, S2[4] = {-1.f, 1.f, 1.f, -1.f};
    constexpr float AT[2][4] = {{1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, -1.f}};
    // ---- column-shared form (HID <= 64: one k-chunk, no row split) ------------------------------------------------------
    // B^T d B for one vertical index xi: A[c] = d[r1][c] + sigma d[r2][c] for the four patch columns c, and the four
    // horizontal positions are  nu 0: A0 - A2,  1: A1 + A2,  2: A2 - A1,  3: A1 - A3.  A1 / A2 of every k-group are kept for
    // the whole xi (2 x 4 values per k-group), A0 / A3 are formed where nu = 0 / 3 uses them: 8 LDS reads + 8 VALU per
    // channel and xi instead of 16 + 12 in the per-position form below, same register footprint (no extra accumulators).
    // At HID = 32 / 16 a transformed operand feeds only 2 / 1 MFMAs: the per-position form kept the CU's LDS pipe 50 / 100 %
    // busy with patch reads (round 2 PMC: SQ_WAIT_INST_LDS 7x the C = 32 level's) and spilled 5-7 registers at 4 workgroups
    // per CU.  Measured (tools/dev/ab_step.py, 65536 samples): C = 8  0.73 -> 0.68 ms, C = 16  1.76 -> 1.64 ms (no spills),
    // C = 32  1.50 -> 1.47 ms.  (Also measured: the same arithmetic on scalar floats with -fno-slp-vectorize, as the
    // micro-architecture guide suggests for VALU beside bf16 MFMAs - no gain at C = 16, 2 % slower at C = 32 / 64: the fp32
    // MFMA shares the fp32 lanes with the VALU, so what counts here is the NUMBER of vector instructions, and packed ones
    // halve it.)
#ifndef CF_WINO_VF_MAXRT
#define CF_WINO_VF_MAXRT 4
#endif
    constexpr bool VF = RT16 <= CF_WINO_VF_MAXRT && NCH == 1 && RSPLIT == 1;
    if constexpr (VF) {
        const float* base = lds;
        constexpr int frc = G::OFF_AW;
        auto rows_of = [&](int xi, int (&r1)[NT], int (&r2)[NT]) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                r1[ct] = xi == 0 ? rpart[ct][0] : (xi == 2 ? rpart[ct][2] : rpart[ct][1]);
                r2[ct] = xi == 2 ? rpart[ct][1] : (xi == 3 ? rpart[ct][3] : rpart[ct][2]);
            }
        };
        // operands of group g = nu KGC + kk: the weight fragments and the raw patch columns the group has to fold
        // (nu 0: columns 0 and 2, nu 1: column 1, nu 2: none, nu 3: column 3); k-steps paired for packed arithmetic
        struct WRaw { f32x2w d[NT][2][2][2]; };               // [ct][slot][row r1 / r2][k-step pair]
        WFrag fa[2];
        WRaw raw[2];
        auto load = [&](int xi, const int (&r1)[NT], const int (&r2)[NT], int g, WFrag& fo, WRaw& o) {
            const int nu = g / KGC, kk = g % KGC;
#ifdef CF_ABL_FIXEDW                       // timing-only probe: every group reads the SAME fragments (L1-resident; wrong results)
            const int fr = frc + (0 * xi * nu * kk) * 256;
#else
            const int fr = frc + ((xi * 4 + nu) * RT16 * KG4 + kk) * 256;
#endif
            if constexpr (G::LDS_W > 0) {            // staged by the kernel (k_flow_step_small) behind the planes, same fragment order
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt)
                    fo.a[rt] = *reinterpret_cast<const float4*>(lds + (HALF + HID) * G::RS + (fr - frc) + (rt0 + rt) * KG4 * 256 + lane * 4);
            } else if constexpr (!G::BF16S) {
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) fo.a[rt] = ws_frag(rs, lane, fr + (rt0 + rt) * KG4 * 256);
            }
            constexpr int NC[4] = {2, 1, 0, 1}, C0[4] = {0, 1, 0, 3}, C1[4] = {2, 0, 0, 0};
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
                    if (sl < NC[nu]) {
                        const int c = sl == 0 ? C0[nu] : C1[nu];
                        const int o1 = r1[ct] + cpart[ct][c], o2 = r2[ct] + cpart[ct][c];
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const char* q0 = reinterpret_cast<const char*>(base + (16 * kk + 8 * e2) * PIX);
                            const char* q1 = q0 + 4 * PIX * 4;
                            auto at = [](const char* q, int ob) { return *reinterpret_cast<const float*>(q + ob); };
                            o.d[ct][sl][0][e2] = f32x2w{at(q0, o1), at(q1, o1)};
                            o.d[ct][sl][1][e2] = f32x2w{at(q0, o2), at(q1, o2)};
                        }
                    }
        };
        f32x2w A1[KGC][NT][2], A2[KGC][NT][2];
        float4 wpb[2][RTW][3];             // BF16S: the weight pieces of the current and of the next 32-channel block
        auto wp_load = [&](float4 (&dst)[RTW][3], int xi_, int nu_, int kb_) {
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                for (int pz = 0; pz < 3; ++pz)
#ifdef CF_ABL_FIXEDW
                    dst[rt][pz] = ws_frag(rs, lane, G::OFF_AWB + ((((0 * xi_ * nu_ * kb_) * RT16 + rt0 + rt) * G::KB32) * 3 + pz) * 256);
#else
                    dst[rt][pz] = ws_frag(rs, lane, G::OFF_AWB + ((((xi_ * 4 + nu_) * RT16 + rt0 + rt) * G::KB32 + kb_) * 3 + pz) * 256);
#endif
        };
        if constexpr (G::BF16S) wp_load(wpb[0], 0, 0, 0);
        static_assert(!G::BF16S || (4 * G::KB32) % 2 == 0, "static ping-pong of the weight pieces across xi");
        f32x2w vb[2][NT][2];               // BF16S: the Winograd-domain values of its two k-groups
        static_assert(!G::BF16S || KGC % 2 == 0, "32-channel blocks");
        int r1[NT], r2[NT], r1n[NT], r2n[NT];
        rows_of(0, r1, r2);
        load(0, r1, r2, 0, fa[0], raw[0]);
        // the vertical coefficients A^T[0][xi] = (1, 1, 1, 0) and A^T[1][xi] = (0, 1, -1, -1): xi = 0 and xi = 3 are peeled so that
        // their zero halves of the output transform are not executed (CF_WINO_PEEL; the loop keeps xi = 1, 2)
        auto xi_body = [&](const int xi, auto has0, auto has1) {
            const int xn = xi < 3 ? xi + 1 : 3;
            rows_of(xn, r1n, r2n);
            const float sigma = xi == 1 ? 1.f : -1.f;
            const float c0 = xi < 3 ? 1.f : 0.f, c1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);      // A^T[0][xi], A^T[1][xi]
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                f32x4w M[NT][RTW];
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int rt = 0; rt < RTW; ++rt) M[ct][rt] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KGC; ++kk) {
                    const int g = nu * KGC + kk;
                    if (g + 1 < NGX) load(xi, r1, r2, g + 1, fa[(g + 1) & 1], raw[(g + 1) & 1]);
                    else load(xn, r1n, r2n, 0, fa[0], raw[0]);           // first group of the next xi (after the last xi: harmless)
                    __builtin_amdgcn_sched_barrier(0);
                    const WFrag& oa = fa[g & 1];
                    const WRaw& o = raw[g & 1];
                    f32x2w v[NT][2];
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            if (nu == 0) {
                                const f32x2w a0 = o.d[ct][0][0][e2] + sigma * o.d[ct][0][1][e2];
                                A2[kk][ct][e2] = o.d[ct][1][0][e2] + sigma * o.d[ct][1][1][e2];
                                v[ct][e2] = a0 - A2[kk][ct][e2];
                            } else if (nu == 1) {
                                A1[kk][ct][e2] = o.d[ct][0][0][e2] + sigma * o.d[ct][0][1][e2];
                                v[ct][e2] = A1[kk][ct][e2] + A2[kk][ct][e2];
                            } else if (nu == 2) {
                                v[ct][e2] = A2[kk][ct][e2] - A1[kk][ct][e2];
                            } else {
                                const f32x2w a3 = o.d[ct][0][0][e2] + sigma * o.d[ct][0][1][e2];
                                v[ct][e2] = A1[kk][ct][e2] - a3;
                            }
                        }
                    if constexpr (!G::BF16S) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                                for (int ct = 0; ct < NT; ++ct)
                                    M[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(oa.a[rt], e), v[ct][e >> 1][e & 1], M[ct][rt], 0, 0, 0);
                    } else {
                        // two k-groups = one 32-channel block: the lane's eight Winograd-domain values (channel 4 j + lg, j = 4 (kk & 1)
                        // + 2 e2 + h) are split into three bf16 pieces each and meet the pre-split weight pieces in six MFMAs per tile
                        (void)oa;
                        // the weight pieces of a 32-channel block are requested one BLOCK ahead of their use (double buffer: an L2
                        // round trip is longer than a block's vector work); block index q counts (nu, kb) within xi, then the next xi
                        constexpr int NBX = 4 * G::KB32;                        // blocks per xi
                        const int q = nu * G::KB32 + (kk >> 1);
                        if (!(kk & 1)) {
                            const int qn = q + 1 < NBX ? q + 1 : 0, xq = q + 1 < NBX ? xi : xn;
                            wp_load(wpb[(q + 1) & 1], xq, qn / G::KB32, qn % G::KB32);
                        }
                        auto& wp = wpb[q & 1];
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                            for (int e2 = 0; e2 < 2; ++e2) vb[kk & 1][ct][e2] = v[ct][e2];
                        if (kk & 1) {
                            typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
                            float4 bp[NT][3];
#pragma unroll
                            for (int ct = 0; ct < NT; ++ct) {
                                unsigned p0[8], p1[8], p2[8];
#pragma unroll
                                for (int j = 0; j < 8; ++j) {             // (scalar on purpose: packed subtractions measured 1.7 % slower)
                                    const float xv = vb[j >> 2][ct][(j >> 1) & 1][j & 1];
#ifdef CF_ABL_NOSPLIT                                                     // timing-only probe: no split arithmetic (wrong results)
                                    p0[j] = p1[j] = p2[j] = __float_as_uint(xv);
                                    continue;
#endif
                                    const unsigned u0 = __float_as_uint(xv) & 0xffff0000u;
                                    const float r1 = xv - __uint_as_float(u0);
                                    const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
                                    const float r2 = r1 - __uint_as_float(u1);
                                    p0[j] = u0; p1[j] = u1; p2[j] = __float_as_uint(r2);
                                }
                                auto pack = [](const unsigned (&p)[8]) {   // element j in bits 16 (j & 1) .. of dword j / 2
                                    return make_float4(__uint_as_float(__builtin_amdgcn_perm(p[1], p[0], 0x07060302u)),
                                                       __uint_as_float(__builtin_amdgcn_perm(p[3], p[2], 0x07060302u)),
                                                       __uint_as_float(__builtin_amdgcn_perm(p[5], p[4], 0x07060302u)),
                                                       __uint_as_float(__builtin_amdgcn_perm(p[7], p[6], 0x07060302u)));
                                };
                                bp[ct][0] = pack(p0); bp[ct][1] = pack(p1); bp[ct][2] = pack(p2);
                            }
                            constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // small terms first
#ifndef CF_ABL_NPAIRS
#define CF_ABL_NPAIRS 6                                                     // timing-only probe: fewer piece pairs (wrong results)
#endif
#pragma unroll
                            for (int t = 6 - CF_ABL_NPAIRS; t < 6; ++t)
#pragma unroll
                                for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                                    for (int ct = 0; ct < NT; ++ct)
                                        M[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, wp[rt][order[t][0]]),
                                                                                            __builtin_bit_cast(bf16x8w, bp[ct][order[t][1]]), M[ct][rt], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // Y[i][j] += A^T[i][xi] A^T[j][nu] M
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (AT[j][nu] != 0.f) {
                        const float k0 = c0 * AT[j][nu], k1 = c1 * AT[j][nu];
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                            for (int rt = 0; rt < RTW; ++rt) {
                                if constexpr (decltype(has0)::value) Y[ct][0][j][rt] += k0 * M[ct][rt];
                                if constexpr (decltype(has1)::value) Y[ct][1][j][rt] += k1 * M[ct][rt];
                            }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) { r1[ct] = r1n[ct]; r2[ct] = r2n[ct]; }
        };
        if constexpr (CF_WINO_PEEL == 2) {      // all four bodies (measured: see DESIGN)
            xi_body(0, std::true_type{}, std::false_type{});
            xi_body(1, std::true_type{}, std::true_type{});
            xi_body(2, std::true_type{}, std::true_type{});
            xi_body(3, std::false_type{}, std::true_type{});
        } else if constexpr (CF_WINO_PEEL) {
            xi_body(0, std::true_type{}, std::false_type{});
#pragma unroll 1
            for (int xi = 1; xi < 3; ++xi) xi_body(xi, std::true_type{}, std::true_type{});
            xi_body(3, std::false_type{}, std::true_type{});
        } else {
#pragma unroll 1
            for (int xi = 0; xi < 4; ++xi) xi_body(xi, std::true_type{}, std::true_type{});
        }
    } else {
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        const float* base = lds + ch * (16 * KGC) * PIX;
        const int frc = G::OFF_AW + ch * KGC * 256;
        WFrag fa[2];
        WPatch pd[2];
        auto rows_of = [&](int xi, int (&r1)[NT], int (&r2)[NT]) {          // xi is wave-uniform: selects, not branches
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                r1[ct] = xi == 0 ? rpart[ct][0] : (xi == 2 ? rpart[ct][2] : rpart[ct][1]);
                r2[ct] = xi == 2 ? rpart[ct][1] : (xi == 3 ? rpart[ct][3] : rpart[ct][2]);
            }
        };
        auto load = [&](int xi, const int (&r1)[NT], const int (&r2)[NT], int g, WFrag& fo, WPatch& o) {    // g = nu * KGC + kk (static)
            const int nu = g / KGC, kk = g % KGC;
            const int fr = frc + ((xi * 4 + nu) * RT16 * KG4 + kk) * 256;
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) fo.a[rt] = ws_frag(rs, lane, fr + (rt0 + rt) * KG4 * 256);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const int o11 = r1[ct] + cpart[ct][B1[nu]], o12 = r1[ct] + cpart[ct][B2[nu]];
                const int o21 = r2[ct] + cpart[ct][B1[nu]], o22 = r2[ct] + cpart[ct][B2[nu]];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const char* q0 = reinterpret_cast<const char*>(base + (16 * kk + 8 * e2) * PIX);
                    const char* q1 = q0 + 4 * PIX * 4;
                    auto at = [](const char* q, int ob) { return *reinterpret_cast<const float*>(q + ob); };
                    o.d[ct][e2][0] = f32x2w{at(q0, o11), at(q1, o11)}; o.d[ct][e2][1] = f32x2w{at(q0, o12), at(q1, o12)};
                    o.d[ct][e2][2] = f32x2w{at(q0, o21), at(q1, o21)}; o.d[ct][e2][3] = f32x2w{at(q0, o22), at(q1, o22)};
                }
            }
        };
        int r1[NT], r2[NT], r1n[NT], r2n[NT];
        rows_of(0, r1, r2);
        load(0, r1, r2, 0, fa[0], pd[0]);
        auto xi_body = [&](const int xi, auto has0, auto has1) {          // xi = 0 / 3 peeled, as in the column-shared form
            const int xn = xi < 3 ? xi + 1 : 3;
            rows_of(xn, r1n, r2n);
            const float sigma = xi == 1 ? 1.f : -1.f;
            const float c0 = xi < 3 ? 1.f : 0.f, c1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);      // A^T[0][xi], A^T[1][xi]
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                f32x4w M[NT][RTW];
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int rt = 0; rt < RTW; ++rt) M[ct][rt] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KGC; ++kk) {
                    const int g = nu * KGC + kk;
                    if (g + 1 < NGX) load(xi, r1, r2, g + 1, fa[(g + 1) & 1], pd[(g + 1) & 1]);
                    else load(xn, r1n, r2n, 0, fa[0], pd[0]);            // first group of the next xi (after the last xi: harmless)
                    __builtin_amdgcn_sched_barrier(0);
                    const WFrag& oa = fa[g & 1];
                    const WPatch& o = pd[g & 1];
                    f32x2w v[NT][2];
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const f32x2w t1 = S1[nu] * o.d[ct][e2][0] + S2[nu] * o.d[ct][e2][1];
                            const f32x2w t2 = S1[nu] * o.d[ct][e2][2] + S2[nu] * o.d[ct][e2][3];
                            v[ct][e2] = t1 + sigma * t2;
                        }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
                            for (int ct = 0; ct < NT; ++ct)
                                M[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(oa.a[rt], e), v[ct][e >> 1][e & 1], M[ct][rt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // Y[i][j] += A^T[i][xi] A^T[j][nu] M
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (AT[j][nu] != 0.f) {
                        const float k0 = c0 * AT[j][nu], k1 = c1 * AT[j][nu];
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                            for (int rt = 0; rt < RTW; ++rt) {
                                if constexpr (decltype(has0)::value) Y[ct][0][j][rt] += k0 * M[ct][rt];
                                if constexpr (decltype(has1)::value) Y[ct][1][j][rt] += k1 * M[ct][rt];
                            }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) { r1[ct] = r1n[ct]; r2[ct] = r2n[ct]; }
        };
        if constexpr (CF_WINO_PEEL == 2) {      // all four bodies (measured: see DESIGN)
            xi_body(0, std::true_type{}, std::false_type{});
            xi_body(1, std::true_type{}, std::true_type{});
            xi_body(2, std::true_type{}, std::true_type{});
            xi_body(3, std::false_type{}, std::true_type{});
        } else if constexpr (CF_WINO_PEEL) {
            xi_body(0, std::true_type{}, std::false_type{});
#pragma unroll 1
            for (int xi = 1; xi < 3; ++xi) xi_body(xi, std::true_type{}, std::true_type{});
            xi_body(3, std::false_type{}, std::true_type{});
        } else {
#pragma unroll 1
            for (int xi = 0; xi < 4; ++xi) xi_body(xi, std::true_type{}, std::true_type{});
        }
    }
    }       // per-position form
    __syncthreads();                 // every wave has finished reading h1
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
        const float4 b = *reinterpret_cast<const float4*>(wsl + G::OFF_B2 + (rt0 + rt) * 16 + 4 * lg);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int i = 0; i < 2; ++i) {            // the two pixels (j = 0, 1) of an output row of the tile: one 8-byte store
                float* dst = H1 + ((rt0 + rt) * 16 + 4 * lg) * PIX + smp[ct] * HW + (2 * ty[ct] + i) * W + 2 * tx[ct];
                const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<float2*>(dst + r * PIX) =
                        make_float2(cf_relu(Y[ct][i][0][rt][r] + bb[r]), cf_relu(Y[ct][i][1][rt][r] + bb[r]));
            }
    }
    if constexpr (RSPLIT > 1) __syncthreads();      // phase 3 reads all rows of this wave's own columns: both row halves in place
    else cf_wave_sync();
}

// ---- direct 3x3 on the bf16 matrix cores (phases 1 + 2 of the PIPE == 5 geometry: 16x16 images, HID = 32) -----------------------
// `v_mfma_f32_*_f32` runs at the vector rate on the pipe the vector instructions use; `v_mfma_f32_16x16x32_bf16` has 16x that
// rate on a pipe of its own.  An fp32 value is exactly three bf16 pieces (truncation split), piece products are exact in fp32,
// and the six pairs with i + j <= 2 accumulated in fp32 carry the error of the f32 MFMA (tools/micro/bf16_split_gemm.hip).  The
// Winograd form would have to split every TRANSFORMED operand at its read (5.5 vector instructions per value: that, not the
// matrix pipe, bounded the PIPE == 4 kernel); the direct form splits h1 ONCE, where phase 1 produces it, and a tap is pure
// data movement: 9 taps x 6 piece pairs = 54 bf16 MFMAs of K = 32 per 16-row x 16-pixel tile = 864 cycles, against 16 positions x 8
// f32 MFMAs = 4096 cycles / 4 tiles = 1024 for the Winograd form - with no transform arithmetic beside them.
// LDS: plane pz at byte pz * PIX * HID * 2; pixel p = 16 y + x holds its 32 channels in 64 bytes, the 16-byte chunk c (channels
// 8 c ..) at slot c ^ ((x >> 1) & 2): a tap's B operand (lane = (pixel column l & 15, chunk l >> 4), pixel shifted by the tap with
// reflection) is then one conflict-free ds_read_b128 for every dx.  The planes alias Y0 / the x plane (barrier before they are
// written) and h2 (fp32 [row][pixel], natural order, this wave's own columns) overwrites them after a barrier.
template <class G>
__device__ __forceinline__ void direct_bf16_phases(float* __restrict__ lds, const float* __restrict__ wsl, ws_rsrc_t rs, int lane, int wave) {
    static_assert(G::DBF && G::H == 16 && G::W == 16 && G::SPW == 1 && G::HID == 32 && G::PTW == 2, "the 16x16 level at C = 16");
    constexpr int PIX = G::PIX, HALF = G::HALF, HID = G::HID, PTW = G::PTW, RT16 = G::RT16;
    constexpr int PLANE = PIX * HID * 2;            // bytes of one bf16 plane
    typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
    float* Y0 = lds;
    float* H1 = lds + HALF * PIX;
    char* const ldsb = reinterpret_cast<char*>(lds);
    const int li = lane & 31, lk = lane >> 5, l15 = lane & 15, lg = lane >> 4;
    // ================= phase 1: h1 = relu(NN.0 y0 + b) on 32x32x2 tiles (K = 8), this wave's own columns of Y0
    int pix[PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) pix[q] = (wave * PTW + q) * 32 + li;
    f32x16 acc1[1][PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) acc1[0][q] = bias_tile(wsl + G::OFF_B1, lk);
    dense_phase<G, G::KS1, G::NG1, 1>(acc1, rs, G::OFF_A1, Y0, pix, lane);
    // weight pieces of the first tap: requested before the barrier
    float4 wp[2][RT16][3];
    auto wp_load = [&](float4 (&dst)[RT16][3], int tap) {
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) dst[rt][pz] = ws_frag(rs, lane, G::OFF_ADB + ((tap * RT16 + rt) * 3 + pz) * 256);
    };
    wp_load(wp[0], 0);
    __syncthreads();                 // every wave is done with Y0 and the x plane: the bf16 planes alias them
#pragma unroll
    for (int q = 0; q < PTW; ++q) {
        const int xx = pix[q] & 15;
#pragma unroll
        for (int a = 0; a < 4; ++a) {               // registers 4 a .. 4 a + 3 = channels 8 a + 4 lk + (0..3): half a chunk
            unsigned p[3][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = cf_relu(acc1[0][q][4 * a + i]);
#ifdef CF_ABL_DB_NOSPLIT                                                     // timing-only probe: no split arithmetic (wrong results)
                p[0][i] = p[1][i] = p[2][i] = __float_as_uint(v);
                continue;
#endif
                const unsigned u0 = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(u0);
                const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(u1);
                p[0][i] = u0; p[1][i] = u1; p[2][i] = __float_as_uint(r2);
            }
            char* dst = ldsb + pix[q] * (HID * 2) + ((a ^ ((xx >> 1) & 2)) * 16) + lk * 8;
#pragma unroll
            for (int pz = 0; pz < 3; ++pz)
                *reinterpret_cast<uint2*>(dst + pz * PLANE) = make_uint2(__builtin_amdgcn_perm(p[pz][1], p[pz][0], 0x07060302u),
                                                                         __builtin_amdgcn_perm(p[pz][3], p[pz][2], 0x07060302u));
        }
    }
    __syncthreads();                 // h1 pieces complete: the taps read neighbouring waves' image rows
    // ================= phase 2: h2 = relu(NN.2 (*) h1 + b), 3x3, reflect padding   (coupling.py:27)
    int colpart[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int xx = l15 + d - 1;
        xx = xx < 0 ? -xx : (xx >= 16 ? 30 - xx : xx);
        colpart[d] = xx * (HID * 2) + ((lg ^ ((xx >> 1) & 2)) * 16);
    }
    f32x4w acc[4][RT16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) acc[ct][rt] = f32x4w{0.f, 0.f, 0.f, 0.f};
    constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // (weight piece, h1 piece): small terms first
    // B operands are requested one (tap, image row) ahead of the 12 MFMAs that consume them, the weight pieces one tap ahead
    float4 bq[2][3];
    auto b_load = [&](float4 (&dst)[3], int tap, int ct) {
        int yy = wave * 4 + ct + tap / 3 - 1;                            // wave-uniform: a column tile is one image row
        yy = yy < 0 ? -yy : (yy >= 16 ? 30 - yy : yy);
        const char* src = ldsb + yy * (16 * HID * 2) + colpart[tap % 3];
#pragma unroll
        for (int pz = 0; pz < 3; ++pz) dst[pz] = *reinterpret_cast<const float4*>(src + pz * PLANE);
    };
    b_load(bq[0], 0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) wp_load(wp[(tap + 1) & 1], tap + 1);
        auto& w = wp[tap & 1];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int it = tap * 4 + ct;
            if (it + 1 < 36) b_load(bq[(it + 1) & 1], (it + 1) / 4, (it + 1) % 4);
            __builtin_amdgcn_sched_barrier(0);
            auto& b = bq[it & 1];
#ifndef CF_ABL_DB_NPAIRS
#define CF_ABL_DB_NPAIRS 6                 // timing-only probe: fewer piece pairs (wrong results)
#endif
#pragma unroll
            for (int t = 6 - CF_ABL_DB_NPAIRS; t < 6; ++t)
#pragma unroll
                for (int rt = 0; rt < RT16; ++rt)
                    acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, w[rt][order[t][0]]),
                                                                          __builtin_bit_cast(bf16x8w, b[order[t][1]]), acc[ct][rt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                 // every wave has finished reading the h1 pieces: h2 overwrites them
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) {
        const float4 bv = *reinterpret_cast<const float4*>(wsl + G::OFF_B2 + rt * 16 + 4 * lg);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) H1[(rt * 16 + 4 * lg + r) * PIX + wave * 64 + 16 * ct + l15] = cf_relu(acc[ct][rt][r] + bb[r]);
    }
    cf_wave_sync();
}

// ---- the conditioner: phases 1-3 of a step (shared by the forward and the inverse kernel) --------------
// In: Y0 = y0 plane (this wave's columns).  Out: acc3 = NN.4 output tiles (t rows / raw rows, packed-row
// layout of chan_of_row).  Uses the H region of LDS for h1 / h2; two workgroup barriers.
// CTX (specialist mode, coupling.py:39-47): 0 none; 1 per-sample bias sb[sample][C] added to the net OUTPUT
// (h = NN(x0) + CN(c), contextflow); 2 per-sample bias sb[sample][HID] added before the first ReLU (CN(c) concatenated
// to the net input = W[:, D:] CN(c) through the first 1x1).  soff[q] = this lane's row offset into sb.
// DUMP (training): the post-ReLU h1 / h2 planes go to the tape (16-byte stores of this wave's own columns).
// UPTO = 2: stop once h2 sits in LDS (k_flow_step_small runs the last 1x1 on 16x16x4 tiles itself; acc3 untouched).
template <class G, int CTX = 0, bool DUMP = false, int UPTO = 3>
__device__ __forceinline__ void conditioner_net(f32x16 (&acc3)[G::RT03][G::PTW], float* __restrict__ lds,
                                                const float* __restrict__ wsl, const int (&pix)[G::PTW],
                                                const int (&pin)[G::PTW], int lane, int tid, float* __restrict__ dbg,
                                                int64_t dbg_cols, int tile, const float* __restrict__ sb = nullptr,
                                                const int* soff = nullptr, StepTape tp = kNoTape, int B = 0) {
    constexpr int C = G::C, W = G::W, H = G::H, PIX = G::PIX, HALF = G::HALF, HID = G::HID;
    constexpr int PTW = G::PTW, RT03 = G::RT03, RT1 = G::RT1;
    float* Y0 = lds;
    float* H1 = lds + HALF * PIX;
    const int lk = lane >> 5;
    const ws_rsrc_t rs = ws_rsrc(wsl, G::WS_FLOATS);
    // ================= phase 1: h1 = relu(NN.0 y0 + b)                      (coupling.py:26)
    {
        f32x16 acc[RT1][PTW];
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) acc[rt][q] = bias_tile(wsl + G::OFF_B1 + rt * 32, lk);
        if constexpr (CTX == 2) {
#pragma unroll
            for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
                for (int q = 0; q < PTW; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rt * 32 + tile_row(r, lk);
                        if (row < HID) acc[rt][q][r] += sb[soff[q] + row];
                    }
        }
        dense_phase<G, G::KS1, G::NG1, RT1>(acc, rs, G::OFF_A1, Y0, pix, lane);
        // the parity-split h1 order of the Winograd form scatters a wave's pixels over its whole SAMPLE: where a sample spans
        // several waves (16x16), every wave must be done with its columns of what the H region held before (x / z plane)
        if constexpr (G::WINO && G::HW > 32 * PTW) __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + tile_row(r, lk);
                    if constexpr (G::WINO) {     // parity-split pixel order of winograd_phase2 (row & 1 = r & 1)
                        const int pw = wino_pix<G>(pix[q] - pin[q], pin[q] / W, pin[q] % W) ^ ((r & 1) * (W / 2));
                        if (row < HID) H1[row * PIX + pw] = cf_relu(acc[rt][q][r]);
                    } else {
                        if (row < HID) H1[row * PIX + pix[q]] = cf_relu(acc[rt][q][r]);
                    }
                }
        if constexpr (DUMP && G::WINO) {
            // the LDS plane is in the parity-split order: the tape's h1 (natural order, operand of the 3x3 weight gradient)
            // goes out from the accumulators, 128 contiguous bytes per row and half wave
#pragma unroll
            for (int q = 0; q < PTW; ++q) {
                const int b = tile * G::SPW + (pix[q] - pin[q]) / G::HW;
                if (b < B) {
                    float* dst = tp.h1 + (int64_t)b * HID * G::HW + pin[q];
#pragma unroll
                    for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = rt * 32 + tile_row(r, lk);
                            if (row < HID) dst[row * G::HW] = cf_relu(acc[rt][q][r]);
                        }
                }
            }
            mask_store<G, RT1>(tp.m1, acc, tile, tid >> 6, lane);
        } else if constexpr (DUMP) {
            cf_wave_sync();
            rows_store_t<G, HID, HID>(tp.h1, H1, tile * G::SPW, B, tid >> 6, lane);
            mask_store<G, RT1>(tp.m1, acc, tile, tid >> 6, lane);
        }
    }
    __syncthreads();                 // h1 complete: the 3x3 taps read neighbouring waves' columns
    if (dbg) {
        float* d = dbg + (int64_t)C * dbg_cols;
        for (int e = tid; e < HID * PIX; e += 256) d[(int64_t)(e / PIX) * dbg_cols + (int64_t)tile * PIX + (e % PIX)] = H1[e];
    }

    // ================= phase 2: h2 = relu(NN.2 (*) h1 + b), 3x3, reflect padding   (coupling.py:27)
    if constexpr (G::WINO) {
        winograd_phase2<G>(lds, wsl, rs, lane, tid >> 6);
        if constexpr (DUMP) {            // h2 sits in LDS in natural order (own columns): plane store + mask words from the plane
            rows_store_t<G, HID, HID>(tp.h2, H1, tile * G::SPW, B, tid >> 6, lane);
#pragma unroll
            for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
                for (int q = 0; q < PTW; ++q) {
                    unsigned bits = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rt * 32 + tile_row(r, lk);
                        if (row < HID) bits |= (H1[row * PIX + pix[q]] > 0.f ? 1u : 0u) << r;
                    }
                    tp.m2[((int64_t)(tile * G::NPT + (tid >> 6) * PTW + q) * RT1 + rt) * 64 + lane] = bits;
                }
        }
    } else {
        f32x16 acc[RT1][PTW];
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) acc[rt][q] = bias_tile(wsl + G::OFF_B2 + rt * 32, lk);
        // reflect-padded source pixel of a tap, as an index into lds[] (offsets, not pointers: a pointer array
        // loses the LDS address space and hipcc falls back to flat loads)
        auto tap_src = [&](int tap, int (&src)[PTW]) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;              // wave-uniform
#pragma unroll
            for (int q = 0; q < PTW; ++q) {
                int yy = pin[q] / W + dy, xx = pin[q] % W + dx;
                yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);       // reflect (padding_mode='reflect')
                xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
                src[q] = HALF * PIX + (pix[q] - pin[q]) + yy * W + xx + lk * PIX;
            }
        };
        if constexpr (G::PIPE == 2) {
            // every tap unrolled (compiler-scheduled inside).  The reflect-padded source offset of a tap splits into a row
            // part per (dy, pixel tile) and a column part per dx - the column inside the image row is the same for every
            // tile because W divides 32 - so the whole 3x3 costs 3 + 3 PTW reflections and one add per (tap, tile),
            // instead of a reflection pair per tap and tile inside the MFMA stream (27 VALU per tap at C = 16).
            static_assert(32 % W == 0, "a 32-pixel tile must hold whole image rows");
            int yo[3][PTW], xo[3];
            const int xin = pin[0] % W;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int xx = xin + d - 1;
                xo[d] = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
#pragma unroll
                for (int q = 0; q < PTW; ++q) {
                    int yy = pin[q] / W + d - 1;
                    yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
                    yo[d][q] = HALF * PIX + (pix[q] - pin[q]) + yy * W + lk * PIX;
                }
            }
            float4 a_cur[RT1], a_nxt[RT1];
#pragma unroll
            for (int rt = 0; rt < RT1; ++rt) a_cur[rt] = ws_frag(rs, lane, G::OFF_A2 + rt * 256);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                int src[PTW];
#pragma unroll
                for (int q = 0; q < PTW; ++q) src[q] = yo[tap / 3][q] + xo[tap % 3];
#pragma unroll
                for (int cg = 0; cg < G::NCG; ++cg) {
                    const int g = tap * G::NCG + cg;
                    const int gn = g + 1 < G::NG2 ? g + 1 : G::NG2 - 1;
#pragma unroll
                    for (int rt = 0; rt < RT1; ++rt) a_nxt[rt] = ws_frag(rs, lane, G::OFF_A2 + (gn * RT1 + rt) * 256);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int q = 0; q < PTW; ++q) {
                            const float bv = lds[src[q] + (8 * cg + 2 * e) * PIX];
#pragma unroll
                            for (int rt = 0; rt < RT1; ++rt)
                                acc[rt][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a_cur[rt], e), bv, acc[rt][q], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < RT1; ++rt) a_cur[rt] = a_nxt[rt];
                }
            }
        } else if constexpr (G::PIPE == 0) {
            // compiler-scheduled form: one tap per loop trip, all NCG groups of the tap unrolled
            float4 a_cur[RT1], a_nxt[RT1];
#pragma unroll
            for (int rt = 0; rt < RT1; ++rt) a_cur[rt] = ws_frag(rs, lane, G::OFF_A2 + rt * 256);
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                int src[PTW];
                tap_src(tap, src);
#pragma unroll
                for (int cg = 0; cg < G::NCG; ++cg) {
                    const int g = tap * G::NCG + cg;
                    const int gn = min(g + 1, G::NG2 - 1);
#pragma unroll
                    for (int rt = 0; rt < RT1; ++rt)
                        a_nxt[rt] = G::ABL == 1 ? make_float4(0.5f, 0.25f, -0.5f, 0.125f + cg) : ws_frag(rs, lane, G::OFF_A2 + (gn * RT1 + rt) * 256);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int q = 0; q < PTW; ++q) {
                            const float bv = G::ABL == 1 ? 0.001f * (src[q] + e) : lds[src[q] + (8 * cg + 2 * e) * PIX];
#pragma unroll
                            for (int rt = 0; rt < RT1; ++rt)
                                acc[rt][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a_cur[rt], e), bv, acc[rt][q], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < RT1; ++rt) a_cur[rt] = a_nxt[rt];
                }
            }
        } else {
            // group g = tap*NCG + cg covers tap (dy,dx) and input channels 8cg .. 8cg+7.  The operands of group
            // g+1 are requested before the MFMAs of group g (two register sets, loads pinned by sched_barrier);
            // the source pixel of a tap is computed once per tap, one tap ahead.
            GroupOps<RT1, PTW> ops[2];
            auto load = [&](int fr, const int (&src)[PTW], int cg, GroupOps<RT1, PTW>& o) {   // fr: float offset of the group
                if constexpr (G::ABL == 1) {                      // timing ablation: operands from registers only
#pragma unroll
                    for (int rt = 0; rt < RT1; ++rt) o.a[rt] = make_float4(0.5f, 0.25f, -0.5f, 0.125f + cg);
#pragma unroll
                    for (int q = 0; q < PTW; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o.b[e][q] = 0.001f * (src[q] + e);
                    return;
                }
#pragma unroll
                for (int rt = 0; rt < RT1; ++rt) o.a[rt] = ws_frag(rs, lane, fr + rt * 256);
#pragma unroll
                for (int q = 0; q < PTW; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.b[e][q] = lds[src[q] + (8 * cg + 2 * e) * PIX];
            };
            int src_cur[PTW], src_nxt[PTW];
            tap_src(0, src_cur);
            load(G::OFF_A2, src_cur, 0, ops[0]);
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                tap_src(min(tap + 1, 8), src_nxt);
                const int fr = G::OFF_A2 + tap * G::NCG * RT1 * 256;
#pragma unroll
                for (int cg = 0; cg < G::NCG; ++cg) {                       // NCG is even: static ping-pong
                    const int fn = fr + (cg + 1) * RT1 * 256;               // fragments of the next group
                    if (cg + 1 < G::NCG) load(fn, src_cur, cg + 1, ops[(cg + 1) & 1]);
                    else load(tap < 8 ? fn : fr, src_nxt, 0, ops[0]);       // first group of the next tap
                    __builtin_amdgcn_sched_barrier(0);
                    group_mma<RT1, PTW>(acc, ops[cg & 1], 4);
                }
#pragma unroll
                for (int q = 0; q < PTW; ++q) src_cur[q] = src_nxt[q];
            }
        }
        __syncthreads();                 // every wave has finished reading h1 (taps cross pixel tiles)
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + tile_row(r, lk);
                    if (row < HID) H1[row * PIX + pix[q]] = cf_relu(acc[rt][q][r]);
                }
        cf_wave_sync();              // h2 rows of all lanes in place (read below by other lanes of this wave)
        if constexpr (DUMP) {
            rows_store_t<G, HID, HID>(tp.h2, H1, tile * G::SPW, B, tid >> 6, lane);
            mask_store<G, RT1>(tp.m2, acc, tile, tid >> 6, lane);
        }
    }
    // no workgroup barrier: phase 3 reads only this wave's own pixel columns of h2
    if (dbg) {
        __syncthreads();
        float* d = dbg + (int64_t)(C + HID) * dbg_cols;
        for (int e = tid; e < HID * PIX; e += 256) d[(int64_t)(e / PIX) * dbg_cols + (int64_t)tile * PIX + (e % PIX)] = H1[e];
    }

    if constexpr (UPTO < 3) return;
    // ================= phase 3: h = NN.4 h2 + b ; affine map ; log-det        (coupling.py:28,52-66)
#pragma unroll
    for (int rt = 0; rt < RT03; ++rt)
#pragma unroll
        for (int q = 0; q < PTW; ++q) acc3[rt][q] = bias_tile(wsl + G::OFF_B3 + rt * 32, lk);
    if constexpr (CTX == 1) {
#pragma unroll
        for (int rt = 0; rt < RT03; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = chan_of_row<G>(rt * 32 + (r & 3) + 8 * (r >> 2)) ;      // lk = 0 row; lk = 1 is +4
                    const int ch1 = chan_of_row<G>(rt * 32 + (r & 3) + 8 * (r >> 2) + 4);
                    const int c = lk ? ch1 : ch;
                    if ((lk ? ch1 : ch) >= 0) acc3[rt][q][r] += sb[soff[q] + c];
                }
    }
    dense_phase<G, G::KS3, G::NG3, RT03>(acc3, rs, G::OFF_A3, H1, pix, lane);
}

// ---- dispatch ------------------------------------------------------------------------------------------
// Tile / loop form per shape, chosen by measurement on MI355X (tools/step_bench.py, B = 16384):
//   C16: 1 sample (256 px, 40 KB LDS, 3-4 workgroups/CU), compiler-scheduled tap loop    113 TFLOP/s
//   C32: 4 samples (256 px, 80 KB, 2/CU), explicit operand pipeline                        130 TFLOP/s
//   C64: 16 samples (256 px, 160 KB = the whole LDS, 1/CU), explicit pipeline, 4x2 tiles   136 TFLOP/s
using G8 = Geo<8, 16, 16, 1, 0>;
using G16 = Geo<16, 16, 16, 1, 0>;
using G32 = Geo<32, 8, 8, 4, 1>;
using G64 = Geo<64, 4, 4, 16, 1>;
// alternates kept for tools/step_bench.py, reachable only through cf_flow_step_fwd_debug (flags bits 16..19)
using G16s = Geo<16, 16, 16, 1, 2>;      // k_flow_step_small: 16x16x4 tiles for the 16-row phases, taps unrolled
using G8s = Geo<8, 16, 16, 1, 2>;
using G16v1 = Geo<16, 16, 16, 1, 1>;
using G16v2 = Geo<16, 16, 16, 2, 1>;
using G32v1 = Geo<32, 8, 8, 4, 0>;
using G32v2 = Geo<32, 8, 8, 2, 1>;
using G32v3 = Geo<32, 8, 8, 8, 0>;
using G64v1 = Geo<64, 4, 4, 16, 1, 1>;
using G64v2 = Geo<64, 4, 4, 8, 1>;
using G64v3 = Geo<64, 4, 4, 16, 0>;
using G8w = Geo<8, 16, 16, 1, 3>;
using G16wb = Geo<16, 16, 16, 1, 4>;     // ... with the Winograd-domain products as bf16-piece MFMAs (CONTEXTFLOW_BF16_SPLIT=1)
using G16db = Geo<16, 16, 16, 1, 5>;     // direct 3x3 on the bf16 matrix cores, h1 split by its producer (CONTEXTFLOW_BF16_SPLIT=2)
using G16w = Geo<16, 16, 16, 1, 3>;      // Winograd F(2x2,3x3) form of the 3x3 (winograd_phase2); 16x16: in k_flow_step_small
using G32w = Geo<32, 8, 8, 4, 3>;
using G64w = Geo<64, 4, 4, 16, 3>;
using G64w2 = Geo<64, 4, 4, 8, 3>;       // 8 samples per workgroup, 2 workgroups / CU: two waves per column tile split the rows


int shape_id(int C, int H, int W) {
    if (C == 8 && H == 16 && W == 16) return 0;
    if (C == 16 && H == 16 && W == 16) return 1;
    if (C == 32 && H == 8 && W == 8) return 2;
    if (C == 64 && H == 4 && W == 4) return 3;
    return -1;
}

}  // namespace
