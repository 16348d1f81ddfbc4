// Backward of one flow step (Conv1x1 -> ActNorm -> Coupling) for the training step that follows the density
// path (contextflow/experiment_cl.py:130-136 `cost.backward()`), as ONE gfx950 kernel per step.
//
// What the data-gradient chain needs of the forward - y1, the log-scale and the two ReLU masks - comes from the tape the
// training forward wrote (TAPED, the default: cf_flow_step_fwd_taped; no step input, no recompute), or is recomputed from
// the step input in LDS with the forward's own code (invertible flow: nothing else stored).  Then, on the fp32 matrix
// cores with TRANSPOSED weight fragments:
//     g_h   = [ g_z1 ,  (g_z1 * y1 * e^{ls} + g_ld) * (1 - (ls/2)^2) ]           affine map + log-det
//     g_h2  = (NN.4^T g_h)            * [h2 > 0]
//     g_h1  = (NN.2^T (*) g_h2)       * [h1 > 0]     3x3 transposed conv = adjoint of the reflect-padded gather
//     g_y0  =  NN.0^T g_h1 + g_z0 ;   g_y1 = g_z1 * e^{ls}
//     g_x   = (e^{-logs} Wm)^T g_y
// and writes, next to g_x, the gradient planes the weight gradients contract over (g_h, g_h2, g_h1, g_y; in the recompute
// form also y0, h1, h2).  The weight gradients themselves are split-K MFMA GEMMs over (sample, pixel): cf_wgrad.hip,
// called from contextflow_amd/layers/autograd.py.  ReLU masks are 16-bit lane masks in registers.
// The adjoint of the reflected gather reads, per tap, ONE source per operand: every LDS plane row carries the fold sums of
// its border rows / columns behind its pixels (PATCH geometries, see patch_build below); 4x4 images keep the class-by-class
// weighted sums (adj_tap<NS>).
#include "cf_step_common.h"

namespace {

template <class G> struct GeoBwd {
    static constexpr int RTI = (G::C + 31) / 32;                  // natural-order C rows
    static constexpr int OFF_A3T = 0;                             // rows HID, K = C        g_h2 = NN.4^T g_h
    static constexpr int OFF_A2T = OFF_A3T + G::NG0 * G::RT1 * 256;   // per tap: rows ci, K = co
    static constexpr int OFF_A1T = OFF_A2T + G::NG2 * G::RT1 * 256;   // rows HALF (1 tile), K = HID
    static constexpr int OFF_A0T = OFF_A1T + G::NG3 * 1 * 256;        // rows C, K = C          g_x = W'^T g_y
    static constexpr int WS32_END = OFF_A0T + G::NG0 * RTI * 256;
    // one-sample-per-workgroup form of the 4x4 level for small batches (k_flow_step_bwd_rs16): the same four transposed
    // matrices as 16x16x4 A fragments in NATURAL row / k order, [row tile][group of 4 k-steps][lane][4]
    static constexpr int OFF_R3T = WS32_END;                                                        // rows HID, K = C
    static constexpr int OFF_R2T = OFF_R3T + (G::HID / 16) * (G::C / 16) * 256;                     // rows ci, group = tap * HID/16 + co group
    static constexpr int OFF_R1T = OFF_R2T + (G::HID / 16) * 9 * (G::HID / 16) * 256;               // rows HALF, K = HID
    static constexpr int OFF_R0T = OFF_R1T + (G::HALF / 16) * (G::HID / 16) * 256;                  // rows C (k_in), K = C: [g_y0 ; g_y1]
    static constexpr int R16_END = OFF_R0T + (G::C / 16) * (G::C / 16) * 256;
    static constexpr int WS_FLOATS = G::RS16 ? R16_END : WS32_END;
};

// blockIdx.y = flow step of a batch (cf_flow_step_bwd_prepare_batch)
constexpr int kPrepBatchBwd = 16;
struct StepPackBwdBatch {
    const float *Wm[kPrepBatchBwd], *logs[kPrepBatchBwd], *w1[kPrepBatchBwd], *w2[kPrepBatchBwd], *w3[kPrepBatchBwd];
    float* wsb[kPrepBatchBwd];
};
template <class G>
__global__ __launch_bounds__(256) void k_step_pack_bwd(const StepPackBwdBatch pb) {
    const int bi = blockIdx.y;
    const float* __restrict__ Wm = pb.Wm[bi]; const float* __restrict__ logs = pb.logs[bi]; const float* __restrict__ w1 = pb.w1[bi];
    const float* __restrict__ w2 = pb.w2[bi]; const float* __restrict__ w3 = pb.w3[bi];
    float* __restrict__ wsb = pb.wsb[bi];
    using Bw = GeoBwd<G>;
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    auto split = [](int e, int RT, int& g, int& rt, int& lane, int& j) {
        j = e & 3; lane = (e >> 2) & 63; const int q = e >> 8; rt = q % RT; g = q / RT;
    };
    int g, rt, lane, j;
    for (int e = gtid; e < G::NG0 * G::RT1 * 256; e += gsz) {               // A3T[j_hid][ch] = w3[ch][j_hid]
        split(e, G::RT1, g, rt, lane, j);
        const int row = rt * 32 + (lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
        wsb[Bw::OFF_A3T + e] = (row < G::HID && k < G::C) ? w3[k * G::HID + row] : 0.f;
    }
    for (int e = gtid; e < G::NG2 * G::RT1 * 256; e += gsz) {               // A2T[tap][ci][co] = w2[co][ci][tap]
        split(e, G::RT1, g, rt, lane, j);
        const int row = rt * 32 + (lane & 31);
        const int tap = g / G::NCG, co = 8 * (g % G::NCG) + 2 * j + (lane >> 5);
        wsb[Bw::OFF_A2T + e] = (row < G::HID) ? w2[(co * G::HID + row) * 9 + tap] : 0.f;
    }
    for (int e = gtid; e < G::NG3 * 256; e += gsz) {                         // A1T[c][j_hid] = w1[j_hid][c]
        split(e, 1, g, rt, lane, j);
        const int row = lane & 31, k = 2 * (4 * g + j) + (lane >> 5);
        wsb[Bw::OFF_A1T + e] = (row < G::HALF && k < G::HID) ? w1[k * G::HALF + row] : 0.f;
    }
    for (int e = gtid; e < G::NG0 * Bw::RTI * 256; e += gsz) {               // A0T[k_in][c] = e^{-logs[c]} Wm[c][k_in]
        split(e, Bw::RTI, g, rt, lane, j);
        const int row = rt * 32 + (lane & 31), kk = 2 * (4 * g + j) + (lane >> 5);
        const int k = kk < G::HALF ? kk + G::HALF : kk - G::HALF;          // the g_y plane sits in LDS as [g_y1 ; g_y0]
        wsb[Bw::OFF_A0T + e] = (row < G::C && kk < G::C) ? expf(-logs[k]) * Wm[k * G::C + row] : 0.f;
    }
    if constexpr (G::RS16) {
        // element ((rt * NG + gi) * 64 + lane) * 4 + j = A[16 rt + (lane & 15)][4 (4 gi + j) + (lane >> 4)]
        constexpr int C = G::C, HID = G::HID, HALF = G::HALF;
        // NG groups per row tile, the k index runs over the KG = NG / taps groups of one tap
        auto split16 = [](int e, int NG, int KG, int& gi, int& row, int& k) {
            const int jj = e & 3, ln = (e >> 2) & 63, q = e >> 8;
            gi = q % NG; row = 16 * (q / NG) + (ln & 15); k = 4 * (4 * (gi % KG) + jj) + (ln >> 4);
        };
        int gi, row, k;
        for (int e = gtid; e < (HID / 16) * (C / 16) * 256; e += gsz) {                  // A3T[hid][c] = w3[c][hid]
            split16(e, C / 16, C / 16, gi, row, k);
            wsb[Bw::OFF_R3T + e] = w3[k * HID + row];
        }
        for (int e = gtid; e < (HID / 16) * 9 * (HID / 16) * 256; e += gsz) {            // A2T[tap][ci][co] = w2[co][ci][tap]
            split16(e, 9 * (HID / 16), HID / 16, gi, row, k);
            wsb[Bw::OFF_R2T + e] = w2[(k * HID + row) * 9 + gi / (HID / 16)];
        }
        for (int e = gtid; e < (HALF / 16) * (HID / 16) * 256; e += gsz) {               // A1T[c][hid] = w1[hid][c]
            split16(e, HID / 16, HID / 16, gi, row, k);
            wsb[Bw::OFF_R1T + e] = w1[k * HALF + row];
        }
        for (int e = gtid; e < (C / 16) * (C / 16) * 256; e += gsz) {                    // A0T[k_in][c] = e^{-logs[c]} Wm[c][k_in]
            split16(e, C / 16, C / 16, gi, row, k);
            wsb[Bw::OFF_R0T + e] = expf(-logs[k]) * Wm[k * C + row];
        }
    }
}

template <class G, int RT>
__device__ __forceinline__ void tiles_to_plane(const f32x16 (&acc)[RT][G::PTW], float* __restrict__ plane, int nrows,
                                               const int (&pix)[G::PTW], int lk) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < G::PTW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + tile_row(r, lk);
                if (row < nrows) plane[row * G::RS + pix[q]] = acc[rt][q][r];
            }
    cf_wave_sync();                      // other lanes of this wave read these rows next
}

// ---- adjoint of the reflect-padded 3x3 gather (g_h2 -> g_h1) --------------------------------------------------
// Output pixel p of tap d = (dy,dx) collects every source o with reflect(o + d) = p: the main source o = p - d (if
// inside the image) plus, per axis, the border pixel whose reflection lands on p (row 0 for p in row 1 with dy = -1,
// row H-1 for row H-2 with dy = +1; same for columns).  The number of candidate sources is a property of the tap:
// 1 (centre), 2 (edge taps) or 4 (corner taps) - one code path per class (NS) instead of four masked reads for
// every tap.  VALU work never overlaps the MFMAs of its SIMD (tools/micro/mfma_issue.hip), so the 0/1-weighted sums
// run as ONE batch per 4-k-step group, and the LDS reads of group cg+1 are issued before the MFMAs of group cg.
template <class G, int NS>
__device__ __forceinline__ void adj_issue(float (&raw)[G::PTW][4][NS], float4 (&a)[G::RT1], ws_rsrc_t rs, int fr,
                                          const float* __restrict__ lds, const int (&off)[G::PTW][NS], int cg, int lane) {
#pragma unroll
    for (int rt = 0; rt < G::RT1; ++rt) a[rt] = ws_frag(rs, lane, fr + (cg * G::RT1 + rt) * 256);
#pragma unroll
    for (int q = 0; q < G::PTW; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NS; ++j) raw[q][e][j] = lds[off[q][j] + (8 * cg + 2 * e) * G::RS];
}

template <class G, int NS>
__device__ __forceinline__ void adj_combine(GroupOps<G::RT1, G::PTW>& o, const float (&raw)[G::PTW][4][NS],
                                            const float4 (&a)[G::RT1], const float (&wgt)[G::PTW][NS]) {
#pragma unroll
    for (int rt = 0; rt < G::RT1; ++rt) o.a[rt] = a[rt];
#pragma unroll
    for (int q = 0; q < G::PTW; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (NS == 1) o.b[e][q] = raw[q][e][0];
            else {
                float v = wgt[q][0] * raw[q][e][0];
#pragma unroll
                for (int j = 1; j < NS; ++j) v = fmaf(wgt[q][j], raw[q][e][j], v);
                o.b[e][q] = v;
            }
        }
}

template <class G, int NS>
__device__ __forceinline__ void adj_tap(f32x16 (&acc)[G::RT1][G::PTW], ws_rsrc_t rs, int fr,
                                        const float* __restrict__ lds, const int (&off)[G::PTW][NS],
                                        const float (&wgt)[G::PTW][NS], int lane) {
    float raw[G::PTW][4][NS];
    float4 a[2][G::RT1];
    adj_issue<G, NS>(raw, a[0], rs, fr, lds, off, 0, lane);
#pragma unroll
    for (int cg = 0; cg < G::NCG; ++cg) {
        GroupOps<G::RT1, G::PTW> o;
        adj_combine<G, NS>(o, raw, a[cg & 1], wgt);
        if (cg + 1 < G::NCG) adj_issue<G, NS>(raw, a[(cg + 1) & 1], rs, fr, lds, off, cg + 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        group_mma<G::RT1, G::PTW>(acc, o, 4);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- PATCH geometries (16x16 and 8x8 images): the folds as precomputed SOURCES --------------------------------------
// Along one axis of size N the adjoint of the reflected gather for shift d reads a PATCHED copy of g at s - d:
//   d = -1:  G'[j] = g[j], except G'[2] = g[2] + g[0] and G'[N] = 0;    d = +1:  G'[N-3] = g[N-3] + g[N-1], G'[-1] = 0
// (the border element whose reflection lands on s shares its output pixel with the regular source next to it).  In two
// dimensions only one row (y = 2 / H-3), one column (x = 2 / W-3) and their crossing differ from g per direction, so every
// plane row carries PP = 2W + 2H + 4 fold sums per sample behind its pixels (+ one zero for out-of-range sources), built
// once per tile; the tap loop then reads ONE source per operand through a per-lane offset - no VALU between the MFMAs,
// 9 instead of 25 LDS reads per k-step (the class-by-class weighted sums of adj_tap<NS> remain for 4x4 images, where
// the fold slots would outgrow the planes).
// Slot order per sample: [0,W) row y=2 (+ y=0), [W,2W) row H-3 (+ H-1), [2W,2W+H) column x=2 (+ x=0), [2W+H,2W+2H) column
// W-3 (+ W-1), then the corners (dy<0,dx<0), (dy<0,dx>0), (dy>0,dx<0), (dy>0,dx>0).
template <class G>
__device__ __forceinline__ void patch_build(float* __restrict__ plane, int tid) {
    constexpr int W = G::W, H = G::H, HW = G::HW, PP = G::PP, SLOTS = G::SPW * PP, GROUPS = 256 / SLOTS;
    static_assert(GROUPS >= 1, "one thread per fold slot");
    if (tid < G::HID) plane[tid * G::RS + G::PIX + SLOTS] = 0.f;              // the zero slot of every row
    if (tid >= GROUPS * SLOTS) return;
    const int j = tid % SLOTS, grp = tid / SLOTS, smp = j / PP, t = j % PP;
    int ya, yb = -1, xa, xb = -1;
    if (t < W) { ya = 2; yb = 0; xa = t; }
    else if (t < 2 * W) { ya = H - 3; yb = H - 1; xa = t - W; }
    else if (t < 2 * W + H) { ya = t - 2 * W; xa = 2; xb = 0; }
    else if (t < 2 * W + 2 * H) { ya = t - 2 * W - H; xa = W - 3; xb = W - 1; }
    else {
        const int c = t - 2 * W - 2 * H;
        ya = (c >> 1) ? H - 3 : 2; yb = (c >> 1) ? H - 1 : 0;
        xa = (c & 1) ? W - 3 : 2;  xb = (c & 1) ? W - 1 : 0;
    }
    const int o0 = smp * HW + ya * W + xa;
    const int o1 = xb >= 0 ? smp * HW + ya * W + xb : o0, o2 = yb >= 0 ? smp * HW + yb * W + xa : o0;
    const int o3 = (xb >= 0 && yb >= 0) ? smp * HW + yb * W + xb : o0;
    const float w1 = xb >= 0 ? 1.f : 0.f, w2 = yb >= 0 ? 1.f : 0.f, w3 = w1 * w2;
#pragma unroll 4
    for (int k = grp; k < G::HID; k += GROUPS) {
        float* row = plane + k * G::RS;
        row[G::PIX + j] = fmaf(w3, row[o3], fmaf(w2, row[o2], fmaf(w1, row[o1], row[o0])));
    }
}

// candidate sources of output coordinate c along one axis of size N for tap shift d: [0] main, [1] reflected border
__device__ __forceinline__ void adj_axis(int c, int d, int N, int (&src)[2], bool (&ok)[2]) {
    const int m = c - d;
    ok[0] = m >= 0 && m < N;
    src[0] = ok[0] ? m : 0;
    const int e = (d == -1 && c == 1) ? 0 : ((d == 1 && c == N - 2) ? N - 1 : -1);
    ok[1] = e >= 0;
    src[1] = ok[1] ? e : 0;
}

// CTX = 1: specialist coupling under contextflow (coupling.py:44): the conditioner output carries a per-sample bias
// sb (B, C) = CN(c); only its log-scale half matters for the recompute (t does not enter any gradient); d/d sb is the
// per-sample row sum of the s_gh plane, taken by the caller.  The generalist's weights are frozen in that mode, so the
// six operand planes of the weight-gradient GEMMs are not written at all (156 of 172 KB per sample at C = 16).
// TAPED: s_y0 / s_h1 / s_h2 are INPUTS written by the training forward (cf_flow_step_fwd_taped): h1 / h2 are loaded for
// their ReLU masks and as the operand of phase 3, the two big contractions of the recompute (phases 1, 2) are skipped.
template <class G, bool SQ, int CTX = 0, bool TAPED = false>
__global__ __launch_bounds__(256, (G::C <= 16 ? 3 : 2)) void k_flow_step_bwd(
    const float* __restrict__ x, const float* __restrict__ gz, const float* __restrict__ gld,
    const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ gx,
    float* __restrict__ s_y0, float* __restrict__ s_h1, float* __restrict__ s_h2, float* __restrict__ s_gh,
    float* __restrict__ s_gh2, float* __restrict__ s_gh1, float* __restrict__ s_gy, int B, int64_t xbs,
    const float* __restrict__ sb, StepTape tp, int gx_unsq) {
    using Bw = GeoBwd<G>;
    constexpr int C = G::C, HW = G::HW, W = G::W, H = G::H, RS = G::RS, HALF = G::HALF, HID = G::HID;   // RS: row stride of the LDS planes
    constexpr int PTW = G::PTW, RT03 = G::RT03, RT1 = G::RT1, NR = (HALF <= 16 ? 8 : 16);
    constexpr int XI = C * PTW / 8;
    extern __shared__ __align__(16) float lds[];
    float* Y0 = lds;                    // [HALF][RS]
    float* H1 = lds + HALF * RS;       // [HID][RS]  (HID = 2C rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int tile = blockIdx.x, b0 = tile * G::SPW;
    int pix[PTW], pin[PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) { pix[q] = (wave * PTW + q) * 32 + li; pin[q] = pix[q] % HW; }
    // packed weights through buffer resources: fragment offsets are scalars (cf_step_common.h)
    const ws_rsrc_t rsw = ws_rsrc(ws, G::WS_FLOATS), rsb = ws_rsrc(wsb, Bw::WS_FLOATS);

    // ---------------------------------------------------------------- what the data-gradient chain needs of the forward
    // y1 (second half of the Conv1x1 + ActNorm output), ls (log-scale) in the packed-row register layout, and the ReLU
    // masks of h1 / h2, one bit per accumulator register.  TAPED: all four were written by the training forward
    // (StepTape: ls / y1 as (B, C/2, HW) planes read 128 contiguous bytes per row and half wave, the masks as words in
    // exactly this layout) - no step input, no recompute.  Otherwise: the forward is re-run from x in LDS.
    float y1[PTW][NR], ls[PTW][NR];
    unsigned m1[RT1][PTW], m2[RT1][PTW];
    // the upstream gradient of the transformed half, g_z1, in the same register layout (128 contiguous bytes per row and
    // half wave), and d L / d ld1 of this lane's samples - requested first, consumed after the tape / recompute below
    float g1v[PTW][NR], glv[PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) {
        const int smp = min(b0 + pix[q] / HW, B - 1);
        const float* gzp = gz + ((int64_t)smp * C + HALF) * HW + pin[q];
        glv[q] = gld[smp];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int idx = tile_row(r, lk);
            g1v[q][r] = (HALF >= 16 || idx < HALF) ? gzp[idx * HW] : 0.f;
        }
    }
    if constexpr (TAPED) {
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) {
                const int64_t w = ((int64_t)(tile * G::NPT + wave * PTW + q) * RT1 + rt) * 64 + lane;
                m1[rt][q] = tp.m1[w];
                m2[rt][q] = tp.m2[w];
            }
#pragma unroll
        for (int q = 0; q < PTW; ++q) {
            const int64_t o = (int64_t)min(b0 + pix[q] / HW, B - 1) * HALF * HW + pin[q];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int idx = tile_row(r, lk);
                const bool ok = HALF >= 16 || idx < HALF;
                ls[q][r] = ok ? tp.ls[o + idx * HW] : 0.f;
                y1[q][r] = ok ? tp.y1[o + idx * HW] : 0.f;
            }
        }
    } else {
    {
        float4 xr[XI];
        x_load<G, SQ>(xr, x, xbs, tile, B, wave, lane);
        x_to_lds<G, SQ>(xr, H1, wave, lane);
        f32x16 acc0[RT03][PTW];
#pragma unroll
        for (int rt = 0; rt < RT03; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) acc0[rt][q] = bias_tile(ws + G::OFF_B0 + rt * 32, lk);
        dense_phase<G, G::KS0, G::NG0, RT03>(acc0, rsw, G::OFF_A0, H1, pix, lane);
#pragma unroll
        for (int q = 0; q < PTW; ++q)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int idx = tile_row(r, lk);
                if constexpr (!TAPED) { if (idx < HALF) Y0[idx * RS + pix[q]] = acc0[0][q][r]; }   // operand of phase 1
                y1[q][r] = (HALF <= 16) ? acc0[0][q][r + 8] : acc0[RT03 - 1][q][r];
            }
        if constexpr (CTX == 0 && !TAPED) rows_store_t<G, HALF, HALF>(s_y0, Y0, b0, B, wave, lane);   // weight-gradient operand plane
    }
    {   // phase 1
        f32x16 acc[RT1][PTW];
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) acc[rt][q] = bias_tile(ws + G::OFF_B1 + rt * 32, lk);
        dense_phase<G, G::KS1, G::NG1, RT1>(acc, rsw, G::OFF_A1, Y0, pix, lane);
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) {
                unsigned m = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = cf_relu(acc[rt][q][r]);
                    m |= (acc[rt][q][r] > 0.f ? 1u : 0u) << r;
                    acc[rt][q][r] = v;
                }
                asm volatile("" : "+v"(m));
                m1[rt][q] = m;
            }
        tiles_to_plane<G, RT1>(acc, H1, HID, pix, lk);
        if constexpr (CTX == 0) rows_store_t<G, HID, HID>(s_h1, H1, b0, B, wave, lane);   // weight-gradient operand plane
    }
    __syncthreads();                     // h1 complete (taps cross waves)
    {   // phase 2 (compiler-scheduled form; the backward is not yet tuned per shape)
        f32x16 acc[RT1][PTW];
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) acc[rt][q] = bias_tile(ws + G::OFF_B2 + rt * 32, lk);
        GroupOps<RT1, PTW> ops[2];
        auto tap_src = [&](int tap, int (&src)[PTW]) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int q = 0; q < PTW; ++q) {
                int yy = pin[q] / W + dy, xx = pin[q] % W + dx;
                yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
                xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
                src[q] = HALF * RS + (pix[q] - pin[q]) + yy * W + xx + lk * RS;
            }
        };
        auto load = [&](int fr, const int (&src)[PTW], int cg, GroupOps<RT1, PTW>& o) {
#pragma unroll
            for (int rt = 0; rt < RT1; ++rt) o.a[rt] = ws_frag(rsw, lane, fr + rt * 256);
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) o.b[e][q] = lds[src[q] + (8 * cg + 2 * e) * RS];
        };
        int src_cur[PTW], src_nxt[PTW];
        tap_src(0, src_cur);
        load(G::OFF_A2, src_cur, 0, ops[0]);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            tap_src(min(tap + 1, 8), src_nxt);
            const int fr = G::OFF_A2 + tap * G::NCG * RT1 * 256;
#pragma unroll
            for (int cg = 0; cg < G::NCG; ++cg) {
                const int fn = fr + (cg + 1) * RT1 * 256;
                if (cg + 1 < G::NCG) load(fn, src_cur, cg + 1, ops[(cg + 1) & 1]);
                else load(tap < 8 ? fn : fr, src_nxt, 0, ops[0]);
                __builtin_amdgcn_sched_barrier(0);
                group_mma<RT1, PTW>(acc, ops[cg & 1], 4);
            }
#pragma unroll
            for (int q = 0; q < PTW; ++q) src_cur[q] = src_nxt[q];
        }
        __syncthreads();                 // everyone done reading h1
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) {
                unsigned m = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    m |= (acc[rt][q][r] > 0.f ? 1u : 0u) << r;
                    acc[rt][q][r] = cf_relu(acc[rt][q][r]);
                }
                asm volatile("" : "+v"(m));
                m2[rt][q] = m;
            }
        tiles_to_plane<G, RT1>(acc, H1, HID, pix, lk);
        if constexpr (CTX == 0) rows_store_t<G, HID, HID>(s_h2, H1, b0, B, wave, lane);   // weight-gradient operand plane
    }
    // phase 3 -> t, raw
    {
        f32x16 acc3[RT03][PTW];
#pragma unroll
        for (int rt = 0; rt < RT03; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) acc3[rt][q] = bias_tile(ws + G::OFF_B3 + rt * 32, lk);
        dense_phase<G, G::KS3, G::NG3, RT03>(acc3, rsw, G::OFF_A3, H1, pix, lane);
#pragma unroll
        for (int q = 0; q < PTW; ++q)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float raw = (HALF <= 16) ? acc3[0][q][r + 8] : acc3[RT03 - 1][q][r];
                if constexpr (CTX == 1) {
                    const int idx = tile_row(r, lk);
                    if (idx < HALF) raw += sb[(int64_t)min(b0 + pix[q] / HW, B - 1) * C + HALF + idx];
                }
                ls[q][r] = cf_log_scale(raw);
            }
    }

    }

    // ---------------------------------------------------------------- backward
    // g_y1 = g_z1 e^{ls} is parked in the Y0 region (dead since phase 1) until the last phase; g_z0 is fetched separately
    // for the last phases: neither stays in registers across the transposed 3x3
    {
        float* GH = H1 + C * RS;                     // g_h plane: rows [0,HALF) = g_t, [HALF,C) = g_raw
#pragma unroll
        for (int q = 0; q < PTW; ++q) {
            const float gl = glv[q];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int idx = tile_row(r, lk);
                if (idx < HALF) {
                    const float g1 = g1v[q][r];
                    const float e = __expf(ls[q][r]);
                    Y0[idx * RS + pix[q]] = g1 * e;                                 // g_y1 = d z1 / d y1
                    const float gls = g1 * y1[q][r] * e + gl;                        // d/d log_s (+ the log-det path)
                    GH[idx * RS + pix[q]] = g1;                                     // d z1 / d t
                    GH[(HALF + idx) * RS + pix[q]] = gls * (1.0f - 0.25f * ls[q][r] * ls[q][r]);   // d log_s / d raw
                }
            }
        }
        rows_store_t<G, C, C>(s_gh, GH, b0, B, wave, lane);
        if constexpr (CTX == 0) rows_store_t<G, HALF, C>(s_gy + HALF * HW, Y0, b0, B, wave, lane);   // g_y1 rows of the g_y plane
        // g_h2 = (NN.4^T g_h) * [h2 > 0]
        f32x16 acc[RT1][PTW];
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][q][r] = 0.f;
        dense_phase<G, G::KS0, G::NG0, RT1>(acc, rsb, Bw::OFF_A3T, GH, pix, lane);
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!((m2[rt][q] >> r) & 1u)) acc[rt][q][r] = 0.f;
        tiles_to_plane<G, RT1>(acc, H1, HID, pix, lk);       // g_h2 plane over the whole H region (own columns)
        if constexpr (CTX == 0) rows_store_t<G, HID, HID>(s_gh2, H1, b0, B, wave, lane);   // weight-gradient operand plane
    }
    // g_z0, the start value of the g_y0 accumulators two phases on (rows of that single tile = channels 0..31 in natural
    // order; 128 contiguous bytes per row and half wave): in flight during the transposed 3x3
    f32x16 accy[1][PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) {
        const float* gzp = gz + (int64_t)min(b0 + pix[q] / HW, B - 1) * C * HW + pin[q];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tile_row(r, lk);
            accy[0][q][r] = (HALF >= 32 || row < HALF) ? gzp[row * HW] : 0.f;
        }
    }
    __syncthreads();                     // g_h2 complete: the transposed 3x3 reads neighbouring waves' columns
    {   // g_h1 = (NN.2^T (*) g_h2) * [h1 > 0]: adjoint of the reflect-padded gather.  Output pixel p of tap (dy,dx)
        // collects every o with reflect(o + d) = p: o = p - d, plus the border pixel whose reflection lands on p.
        f32x16 acc[RT1][PTW];
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][q][r] = 0.f;
        constexpr int frags = Bw::OFF_A2T;
        constexpr int TAPF = G::NCG * RT1 * 256;             // fragment floats per tap
        int py[PTW], px[PTW], base[PTW];
#pragma unroll
        for (int q = 0; q < PTW; ++q) { py[q] = pin[q] / W; px[q] = pin[q] % W; base[q] = HALF * RS + (pix[q] - pin[q]) + lk * RS; }
        if constexpr (G::PATCH) {
            patch_build<G>(H1, tid);
            __syncthreads();             // fold slots in place
            constexpr int PP = G::PP;
            // one source per operand: the same two-stage operand pipeline as the forward 3x3 (the first group of the next tap
            // is requested before the last group of this tap runs; NCG is even: static ping-pong)
            auto tap_off = [&](int tap, int (&off)[PTW]) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;                 // scalars
                const int ysp = dy < 0 ? 2 : H - 3, xsp = dx < 0 ? 2 : W - 3;
#pragma unroll
                for (int q = 0; q < PTW; ++q) {
                    const int jy = py[q] - dy, jx = px[q] - dx;
                    const bool oob = jy < 0 || jy >= H || jx < 0 || jx >= W;
                    const bool fy = dy != 0 && jy == ysp, fx = dx != 0 && jx == xsp;
                    const int slots = G::PIX + ((pix[q] - pin[q]) / HW) * PP;
                    int o = (pix[q] - pin[q]) + jy * W + jx;
                    if (fy) o = slots + (dy > 0 ? W : 0) + jx;
                    if (fx) o = slots + 2 * W + (dx > 0 ? H : 0) + jy;
                    if (fy && fx) o = slots + 2 * W + 2 * H + (dy > 0 ? 2 : 0) + (dx > 0 ? 1 : 0);
                    if (oob) o = G::PIX + G::SPW * PP;
                    off[q] = HALF * RS + lk * RS + o;
                }
            };
            GroupOps<RT1, PTW> ops[2];
            auto load = [&](int fr, const int (&off)[PTW], int cg, GroupOps<RT1, PTW>& o) {
#pragma unroll
                for (int rt = 0; rt < RT1; ++rt) o.a[rt] = ws_frag(rsb, lane, fr + rt * 256);
#pragma unroll
                for (int q = 0; q < PTW; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.b[e][q] = lds[off[q] + (8 * cg + 2 * e) * RS];
            };
            int off_cur[PTW], off_nxt[PTW];
            tap_off(0, off_cur);
            load(frags, off_cur, 0, ops[0]);
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                tap_off(min(tap + 1, 8), off_nxt);
                const int fr = frags + tap * TAPF;
#pragma unroll
                for (int cg = 0; cg < G::NCG; ++cg) {
                    const int fn = fr + (cg + 1) * RT1 * 256;
                    if (cg + 1 < G::NCG) load(fn, off_cur, cg + 1, ops[(cg + 1) & 1]);
                    else load(tap < 8 ? fn : fr, off_nxt, 0, ops[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    group_mma<RT1, PTW>(acc, ops[cg & 1], 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < PTW; ++q) off_cur[q] = off_nxt[q];
            }
        } else {
        {   // centre tap: the pixel itself
            int off[PTW][1];
            float wgt[PTW][1];
#pragma unroll
            for (int q = 0; q < PTW; ++q) { off[q][0] = base[q] + pin[q]; wgt[q][0] = 1.f; }
            adj_tap<G, 1>(acc, rsb, frags + 4 * TAPF, lds, off, wgt, lane);
        }
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {                        // edge taps 1, 3, 5, 7: main + one reflected source
            const int tap = 2 * k + 1, dy = tap / 3 - 1, dx = tap % 3 - 1;
            int off[PTW][2];
            float wgt[PTW][2];
#pragma unroll
            for (int q = 0; q < PTW; ++q) {
                int ys[2], xs[2];
                bool yv[2], xv[2];
                adj_axis(py[q], dy, H, ys, yv);
                adj_axis(px[q], dx, W, xs, xv);
                const bool vert = dy != 0;                   // scalar
                off[q][0] = base[q] + ys[0] * W + xs[0];
                wgt[q][0] = (yv[0] && xv[0]) ? 1.f : 0.f;
                off[q][1] = base[q] + (vert ? ys[1] : ys[0]) * W + (vert ? xs[0] : xs[1]);
                wgt[q][1] = (vert ? (yv[1] && xv[0]) : (yv[0] && xv[1])) ? 1.f : 0.f;
            }
            adj_tap<G, 2>(acc, rsb, frags + tap * TAPF, lds, off, wgt, lane);
        }
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {                        // corner taps 0, 2, 6, 8: up to four sources
            const int tap = (k >> 1) * 6 + (k & 1) * 2, dy = tap / 3 - 1, dx = tap % 3 - 1;
            int off[PTW][4];
            float wgt[PTW][4];
#pragma unroll
            for (int q = 0; q < PTW; ++q) {
                int ys[2], xs[2];
                bool yv[2], xv[2];
                adj_axis(py[q], dy, H, ys, yv);
                adj_axis(px[q], dx, W, xs, xv);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        off[q][2 * a + b] = base[q] + ys[a] * W + xs[b];
                        wgt[q][2 * a + b] = (yv[a] && xv[b]) ? 1.f : 0.f;
                    }
            }
            adj_tap<G, 4>(acc, rsb, frags + tap * TAPF, lds, off, wgt, lane);
        }
        }
        __syncthreads();                 // everyone done reading g_h2
#pragma unroll
        for (int rt = 0; rt < RT1; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!((m1[rt][q] >> r) & 1u)) acc[rt][q][r] = 0.f;
        tiles_to_plane<G, RT1>(acc, H1, HID, pix, lk);       // g_h1 plane
        if constexpr (CTX == 0) rows_store_t<G, HID, HID>(s_gh1, H1, b0, B, wave, lane);   // weight-gradient operand plane
    }
    {   // g_y0 = NN.0^T g_h1 + g_z0: the accumulators start from g_z0 (requested before the transposed 3x3)
        f32x16 (&acc)[1][PTW] = accy;
        dense_phase<G, G::KS3, G::NG3, 1>(acc, rsb, Bw::OFF_A1T, H1, pix, lane);
        // g_y plane in LDS: rows [0, HALF) = g_y1 (the Y0 region, parked above), rows [HALF, C) = g_y0 (first rows of the H
        // region: g_h1 is dead for this wave now); A0T is packed with its k in that order
        float* GY = lds;
#pragma unroll
        for (int q = 0; q < PTW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tile_row(r, lk);
                if (row < HALF) H1[row * RS + pix[q]] = acc[0][q][r];
            }
        if constexpr (CTX == 0) rows_store_t<G, HALF, C>(s_gy, H1, b0, B, wave, lane);   // g_y0 rows (weight-gradient operand plane)
        // g_x = (e^{-logs} Wm)^T g_y
        f32x16 ax[Bw::RTI][PTW];
#pragma unroll
        for (int rt = 0; rt < Bw::RTI; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) ax[rt][q][r] = 0.f;
        dense_phase<G, G::KS0, G::NG0, Bw::RTI>(ax, rsb, Bw::OFF_A0T, GY, pix, lane);
        float* GX = H1 + C * RS;
        tiles_to_plane<G, Bw::RTI>(ax, GX, C, pix, lk);
        if (gx_unsq) rows_store_unsq<G>(gx, GX, b0, B, wave, lane);      // d/dx in the layout of the tensor BEFORE Squeeze((2,2))
        else rows_store_t<G, C, C>(gx, GX, b0, B, wave, lane);
    }
}


// ---- the 4x4 level at small batches: ONE sample per workgroup on 16-column tiles ------------------------------------------------
// The backward's analogue of k_flow_step_rs16 (cf_step.hip).  k_flow_step_bwd<B64> puts 8 samples into a workgroup: 32
// workgroups at the reference's batch of 256, every wave running all four 32-row tiles of its two samples - 86 us per step,
// 15 % of the captured training step.  Here a workgroup is one sample = the 16 columns of v_mfma_f32_16x16x4_f32 tiles, the
// four waves split the OUTPUT rows (16-row tiles: 2 / 2 / 1 (half of K) / 1 per wave in the four products), planes
// [channel][16 pixels] in LDS in natural row order.  The adjoint of the reflect-padded gather is taken on the OPERAND: per tap
// row dy the three planes  GA[dx][co][s] = sum over { p : reflect(p + d) = s } of g_h2[co][p]  are formed once (two row reads,
// one fold of a float4 per dx), after which a tap is a plain product.  TAPED form only (log-scale, y1 and the ReLU mask words
// of the tape in the layout of the 32x32x2 kernels: one word holds the four rows of a lane's accumulator).
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class G>
__global__ __launch_bounds__(256) void k_flow_step_bwd_rs16(const float* __restrict__ gz, const float* __restrict__ gld,
                                                            const float* __restrict__ wsb, float* __restrict__ gx,
                                                            float* __restrict__ s_gh, float* __restrict__ s_gh2,
                                                            float* __restrict__ s_gh1, float* __restrict__ s_gy, int B,
                                                            StepTape tp, int gx_unsq) {
    static_assert(G::RS16, "C = 64 on 4x4 images");
    using Bw = GeoBwd<G>;
    constexpr int C = G::C, HW = 16, W = 4, HALF = G::HALF, HID = G::HID, P = 16, RT1 = G::RT1;
    constexpr int NGA = C / 16, NGT = HID / 16, NGC = HID / 32, NGD = C / 16;      // groups of 4 k-steps per product (C: half of K)
    __shared__ __align__(16) float lds[(C + C + HID + 3 * HID) * P];
    float* GH = lds;                 // [C][16]      g_t | g_raw
    float* GY = GH + C * P;          // [C][16]      g_y0 | g_y1
    float* H2 = GY + C * P;          // [HID][16]    g_h2, then g_h1, then g_x
    float* GA = H2 + HID * P;        // [3][HID][16] the gathered planes of one tap row; then the two K-halves of g_y0
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const ws_rsrc_t rs = ws_rsrc(wsb, Bw::WS_FLOATS);
    // fragments of the three small products, requested before anything else; the 3x3 through a ring of four groups
    float4 fa[2][NGA], fc[NGC], fd[NGD];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gi = 0; gi < NGA; ++gi) fa[t][gi] = ws_frag(rs, lane, Bw::OFF_R3T + ((2 * w + t) * NGA + gi) * 256);
#pragma unroll
    for (int gi = 0; gi < NGC; ++gi) fc[gi] = ws_frag(rs, lane, Bw::OFF_R1T + ((w & 1) * NGT + (w >> 1) * NGC + gi) * 256);
#pragma unroll
    for (int gi = 0; gi < NGD; ++gi) fd[gi] = ws_frag(rs, lane, Bw::OFF_R0T + (w * NGD + gi) * 256);
    auto frag2 = [&](int j, float4 (&o)[2]) {
        const int jj = j < 9 * NGT ? j : 9 * NGT - 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) o[t] = ws_frag(rs, lane, Bw::OFF_R2T + ((2 * w + t) * 9 * NGT + jj) * 256);
    };
    float4 ring[4][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) frag2(j, ring[j]);
    // the mask words of this lane's rows (16-row tile 2 w + t = half t of the 32-row tile w; bits 4 (2 t + g / 2) + i)
    unsigned mw1[2], mw2[2];
    {
        const int64_t wi = ((int64_t)(b >> 1) * RT1 + w) * 64 + (g & 1) * 32 + (b & 1) * 16 + col;
        const unsigned a1 = tp.m1[wi], a2 = tp.m2[wi];
#pragma unroll
        for (int t = 0; t < 2; ++t) { mw1[t] = a1 >> (4 * (2 * t + (g >> 1))); mw2[t] = a2 >> (4 * (2 * t + (g >> 1))); }
    }
    __builtin_amdgcn_sched_barrier(0);
    auto mma = [](float a, float bv, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0); };
    const float* gzb = gz + (int64_t)b * C * HW;
    // ---- affine map + log-det: g_h = [g_z1, (g_z1 y1 e^{ls} + g_ld) (1 - (ls/2)^2)], g_y1 = g_z1 e^{ls}; two elements per thread
    float gz0[HALF * P / 256];
    {
        const float gl = gld[b];
#pragma unroll
        for (int i = 0; i < HALF * P / 256; ++i) {
            const int e = tid + 256 * i;
            const float g1 = gzb[HALF * HW + e], ls = tp.ls[(int64_t)b * HALF * HW + e], y1 = tp.y1[(int64_t)b * HALF * HW + e];
            gz0[i] = gzb[e];
            const float ex = __expf(ls);
            GY[HALF * P + e] = g1 * ex;
            const float gls = g1 * y1 * ex + gl;
            GH[e] = g1;
            GH[HALF * P + e] = gls * (1.0f - 0.25f * ls * ls);
        }
    }
    __syncthreads();
    // weight-gradient operand planes leave as whole 16-byte rows: (B, rows, 16) is the LDS plane itself
    auto plane_out = [&](float* __restrict__ dst, const float* __restrict__ src, int floats) {
        for (int e = 4 * tid; e < floats; e += 1024) *reinterpret_cast<float4*>(dst + e) = *reinterpret_cast<const float4*>(src + e);
    };
    plane_out(s_gh + (int64_t)b * C * HW, GH, C * P);
    plane_out(s_gy + (int64_t)b * C * HW + HALF * HW, GY + HALF * P, HALF * P);
    // ---- g_h2 = (NN.4^T g_h) * [h2 > 0]: row tiles 2 w, 2 w + 1
    {
        f32x4 a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 4 * NGA; ++s) {
            const float bv = GH[(4 * s + g) * P + col];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = mma(f4e(fa[t][s >> 2], s & 3), bv, a[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) H2[(16 * (2 * w + t) + 4 * g + r) * P + col] = ((mw2[t] >> r) & 1u) ? a[t][r] : 0.f;
    }
    __syncthreads();
    plane_out(s_gh2 + (int64_t)b * HID * HW, H2, HID * P);
    // ---- g_h1 = (NN.2^T (*) g_h2) * [h1 > 0]
    f32x4 a2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
    for (int dyi = 0; dyi < 3; ++dyi) {
        if (dyi) __syncthreads();                                  // the previous tap row's planes have been consumed
        // rows py with reflect(py + dy) = sy:  dy = -1: {1}, {0, 2}, {3}, {};  dy = 0: {sy};  dy = +1: {}, {0}, {1, 3}, {2}
#pragma unroll
        for (int i = 0; i < HID * 4 / 256; ++i) {
            const int it = tid + 256 * i, k = it >> 2, sy = it & 3;
            int p0, p1;
            if (dyi == 1) { p0 = sy; p1 = -1; }
            else if (dyi == 0) { p0 = sy == 0 ? 1 : sy == 1 ? 0 : sy == 2 ? 3 : -1; p1 = sy == 1 ? 2 : -1; }
            else { p0 = sy == 0 ? -1 : sy == 1 ? 0 : sy == 2 ? 1 : 2; p1 = sy == 2 ? 3 : -1; }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p0 >= 0) v = *reinterpret_cast<const float4*>(&H2[k * P + 4 * p0]);
            if (p1 >= 0) {
                const float4 u = *reinterpret_cast<const float4*>(&H2[k * P + 4 * p1]);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            // columns px with reflect(px + dx) = sx, the same sets along x
            *reinterpret_cast<float4*>(&GA[(0 * HID + k) * P + 4 * sy]) = make_float4(v.y, v.x + v.z, v.w, 0.f);     // dx = -1
            *reinterpret_cast<float4*>(&GA[(1 * HID + k) * P + 4 * sy]) = v;                                         // dx = 0
            *reinterpret_cast<float4*>(&GA[(2 * HID + k) * P + 4 * sy]) = make_float4(0.f, v.x, v.y + v.w, v.z);     // dx = +1
        }
        __syncthreads();
        float bv[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[0][e] = GA[(4 * e + g) * P + col];
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
            const int tap = 3 * dyi + dxi;
            const float* cur = GA + dxi * HID * P + g * P + col;
            const float* nxt = GA + (dxi < 2 ? dxi + 1 : 2) * HID * P + g * P + col;
#pragma unroll
            for (int c = 0; c < NGT; ++c) {                            // NGT = 8: static ring slots
                frag2(tap * NGT + c + 3, ring[(c + 3) & 3]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    bv[(c + 1) & 1][e] = (c + 1 < NGT) ? cur[(16 * (c + 1) + 4 * e) * P] : nxt[4 * e * P];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < 2; ++t) a2[t] = mma(f4e(ring[c & 3][t], e), bv[c & 1][e], a2[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // g_h2 was last read when the third tap row's planes were formed (a barrier ago): its region takes g_h1
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) H2[(16 * (2 * w + t) + 4 * g + r) * P + col] = ((mw1[t] >> r) & 1u) ? a2[t][r] : 0.f;
    __syncthreads();
    plane_out(s_gh1 + (int64_t)b * HID * HW, H2, HID * P);
    // ---- g_y0 = NN.0^T g_h1 + g_z0: row tile w & 1, K half w >> 1; the halves meet in the (dead) gathered planes
    {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kh = w >> 1, rc = w & 1;
#pragma unroll
        for (int s = 0; s < 4 * NGC; ++s) a = mma(f4e(fc[s >> 2], s & 3), H2[(4 * (4 * NGC * kh + s) + g) * P + col], a);
#pragma unroll
        for (int r = 0; r < 4; ++r) GA[(kh * HALF + 16 * rc + 4 * g + r) * P + col] = a[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < HALF * P / 256; ++i) {
        const int e = tid + 256 * i;
        const float v = GA[e] + GA[HALF * P + e] + gz0[i];
        GY[e] = v;
        s_gy[(int64_t)b * C * HW + e] = v;
    }
    __syncthreads();
    // ---- g_x = (e^{-logs} Wm)^T g_y: row tile w
    {
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int s = 0; s < 4 * NGD; ++s) {
            const float bv = GY[(4 * s + g) * P + col];
            if (s & 1) a1 = mma(f4e(fd[s >> 2], s & 3), bv, a1); else a0 = mma(f4e(fd[s >> 2], s & 3), bv, a0);
        }
        a0 += a1;
#pragma unroll
        for (int r = 0; r < 4; ++r) H2[(16 * w + 4 * g + r) * P + col] = a0[r];      // g_h1 is dead (two barriers ago)
    }
    __syncthreads();
    float* gxb = gx + (int64_t)b * C * HW;
    if (!gx_unsq) plane_out(gxb, H2, C * P);
    else if (tid < (C / 2) * 4) {        // d/dx in the layout of the tensor BEFORE Squeeze((2,2)) (rows_store_unsq)
        const int j = tid >> 2, yy = tid & 3;
        const float4 p = *reinterpret_cast<const float4*>(&H2[(2 * j) * P + 4 * yy]);
        const float4 q = *reinterpret_cast<const float4*>(&H2[(2 * j + 1) * P + 4 * yy]);
        float* d = gxb + (j >> 1) * 4 * HW + (2 * yy + (j & 1)) * 2 * W;
        *reinterpret_cast<float4*>(d) = make_float4(p.x, q.x, p.y, q.y);
        *reinterpret_cast<float4*>(d + 4) = make_float4(p.z, q.z, p.w, q.w);
    }
}

#ifndef CF_BWD_RS16_MAXB
#define CF_BWD_RS16_MAXB 1536       // tools/bwd_bench.py: 17 / 30 / 54 / 100 us at 256 / 512 / 1024 / 2048 samples against 86 / 87 / 88 / 95
#endif

template <class G>
int launch_prepare_bwd(const StepPackBwdBatch& pb, int n, hipStream_t s) {
    int blocks = (GeoBwd<G>::WS_FLOATS + 255) / 256;
    if (blocks > 512) blocks = 512;
    k_step_pack_bwd<G><<<dim3(blocks, n), dim3(256), 0, s>>>(pb);
    return 0;
}

template <class G, bool SQ, int CTX = 0, bool TAPED = false>
int launch_step_bwd(const float* x, const float* gz, const float* gld, const float* ws, const float* wsb, float* gx,
                    float* s_y0, float* s_h1, float* s_h2, float* s_gh, float* s_gh2, float* s_gh1, float* s_gy, int B,
                    int64_t xbs, hipStream_t s, const float* sb = nullptr, StepTape tp = kNoTape, int gx_unsq = 0) {
    constexpr size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
    if (lds_bytes > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_flow_step_bwd<G, SQ, CTX, TAPED>, 160 * 1024, raised, __func__)) return rc_;
    }
    k_flow_step_bwd<G, SQ, CTX, TAPED><<<dim3((B + G::SPW - 1) / G::SPW), dim3(256), lds_bytes, s>>>(
        x, gz, gld, ws, wsb, gx, s_y0, s_h1, s_h2, s_gh, s_gh2, s_gh1, s_gy, B, xbs, sb, tp, gx_unsq);
    return 0;
}

// the backward uses the smaller tiles (2 workgroups/CU at C = 64): its register footprint is larger than the forward's
// 16x16 / 8x8: fold slots behind every plane row (PATCH); 8x8 with 2 samples per workgroup so that two workgroups share a CU
using B8 = Geo<8, 16, 16, 1, 1, 0, 1>;
using B16 = Geo<16, 16, 16, 1, 1, 0, 1>;
using B32 = Geo<32, 8, 8, 2, 1, 0, 1>;
using B64 = Geo<64, 4, 4, 8, 1>;

}  // namespace

extern "C" {

int64_t cf_flow_step_bwd_ws_bytes(int C, int H, int W) {
    switch (shape_id(C, H, W)) {
        case 0: return (int64_t)GeoBwd<B8>::WS_FLOATS * 4;
        case 1: return (int64_t)GeoBwd<B16>::WS_FLOATS * 4;
        case 2: return (int64_t)GeoBwd<B32>::WS_FLOATS * 4;
        case 3: return (int64_t)GeoBwd<B64>::WS_FLOATS * 4;
    }
    return 0;
}

int cf_flow_step_bwd_prepare_batch(int n, const float* const* Wm, const float* const* logs, const float* const* w1,
                                   const float* const* w2, const float* const* w3, void* const* wsb, int C, int H, int W,
                                   cf_stream_t stream) {
    CF_REQUIRE(n >= 0 && Wm && logs && w1 && w2 && w3 && wsb);
    const int sid = shape_id(C, H, W);
    if (sid < 0) { cf_set_error("cf_flow_step_bwd_prepare: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED; }
    for (int i0 = 0; i0 < n; i0 += kPrepBatchBwd) {
        const int m = n - i0 < kPrepBatchBwd ? n - i0 : kPrepBatchBwd;
        StepPackBwdBatch pb{};
        for (int i = 0; i < m; ++i) {
            const int j = i0 + i;
            CF_REQUIRE(Wm[j] && logs[j] && w1[j] && w2[j] && w3[j] && wsb[j] && (reinterpret_cast<uintptr_t>(wsb[j]) & 15) == 0);
            pb.Wm[i] = Wm[j]; pb.logs[i] = logs[j]; pb.w1[i] = w1[j]; pb.w2[i] = w2[j]; pb.w3[i] = w3[j]; pb.wsb[i] = (float*)wsb[j];
        }
        switch (sid) {
            case 0: launch_prepare_bwd<B8>(pb, m, cf_s(stream)); break;
            case 1: launch_prepare_bwd<B16>(pb, m, cf_s(stream)); break;
            case 2: launch_prepare_bwd<B32>(pb, m, cf_s(stream)); break;
            default: launch_prepare_bwd<B64>(pb, m, cf_s(stream)); break;
        }
        CF_LAUNCH_CHECK();
    }
    return 0;
}

int cf_flow_step_bwd_prepare(const float* Wm, const float* logs, const float* w1, const float* w2, const float* w3,
                             void* wsb, int C, int H, int W, cf_stream_t stream) {
    CF_REQUIRE(Wm && logs && w1 && w2 && w3 && wsb && (reinterpret_cast<uintptr_t>(wsb) & 15) == 0);
    return cf_flow_step_bwd_prepare_batch(1, &Wm, &logs, &w1, &w2, &w3, &wsb, C, H, W, stream);
}

// backward of a step whose forward was cf_flow_step_fwd_taped: the kernel reads ls / y1 / the two ReLU masks from the
// tape's aux buffer - it needs neither the step input nor the forward's packed weights, and runs no recompute.  (The y0 /
// h1 / h2 planes of the tape are operands of cf_wgrad only.)  gx comes out in the (B, C, H, W) layout of the step.
int cf_flow_step_bwd_taped(const float* gz, const float* gld, const void* wsb, const void* t_aux, float* gx, float* s_gh,
                           float* s_gh2, float* s_gh1, float* s_gy, int B, int C, int H, int W, int gx_unsqueezed,
                           cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(gz && gld && wsb && t_aux && gx && s_gh && s_gh2 && s_gh1 && s_gy && (!gx_unsqueezed || C % 4 == 0));
    CF_REQUIRE((reinterpret_cast<uintptr_t>(gz) & 15) == 0 && (reinterpret_cast<uintptr_t>(gx) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(t_aux) & 15) == 0);
    const float* wb = (const float*)wsb;
    const StepTape tp = make_tape(nullptr, nullptr, nullptr, const_cast<void*>(t_aux), B, C, H, W);
    int rc = 0;
#define CF_BWDT(G) rc = launch_step_bwd<G, false, 0, true>(nullptr, gz, gld, nullptr, wb, gx, nullptr, nullptr, nullptr, s_gh, s_gh2, \
                                                           s_gh1, s_gy, B, (int64_t)C * H * W, cf_s(stream), nullptr, tp, gx_unsqueezed != 0)
    switch (shape_id(C, H, W)) {
        case 0: CF_BWDT(B8); break;
        case 1: CF_BWDT(B16); break;
        case 2: CF_BWDT(B32); break;
        case 3:
            if (B <= CF_BWD_RS16_MAXB)      // one sample per workgroup: B workgroups instead of B / 8
                k_flow_step_bwd_rs16<B64><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(gz, gld, wb, gx, s_gh, s_gh2, s_gh1, s_gy, B, tp,
                                                                                  gx_unsqueezed != 0);
            else CF_BWDT(B64);
            break;
        default: cf_set_error("cf_flow_step_bwd_taped: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED;
    }
#undef CF_BWDT
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

// backward of the specialist coupling under contextflow (cf_flow_step_fwd_ctx, mode 1): same kernel, the recompute
// adds the per-sample bias sbias (B, C) to the conditioner output.  d/d sbias[b, c] = sum_p s_gh[b, c, p].
int cf_flow_step_bwd_ctx(const float* x, const float* gz, const float* gld, const void* ws, const void* wsb,
                         const float* sbias, float* gx, float* s_y0, float* s_h1, float* s_h2, float* s_gh, float* s_gh2,
                         float* s_gh1, float* s_gy, int B, int C, int H, int W, int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && gz && gld && ws && wsb && sbias && gx && s_gh);     // the other planes are not written in this mode
    CF_REQUIRE(x_bstride >= (int64_t)C * H * W && x_bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(gz) & 15) == 0 && (reinterpret_cast<uintptr_t>(gx) & 15) == 0);
    const float* w = (const float*)ws;
    const float* wb = (const float*)wsb;
    int rc = 0;
#define CF_BWDC(G) rc = launch_step_bwd<G, false, 1>(x, gz, gld, w, wb, gx, s_y0, s_h1, s_h2, s_gh, s_gh2, s_gh1, s_gy, B, x_bstride, cf_s(stream), sbias)
    switch (shape_id(C, H, W)) {
        case 0: CF_BWDC(B8); break;
        case 1: CF_BWDC(B16); break;
        case 2: CF_BWDC(B32); break;
        case 3: CF_BWDC(B64); break;
        default: cf_set_error("cf_flow_step_bwd_ctx: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED;
    }
#undef CF_BWDC
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
