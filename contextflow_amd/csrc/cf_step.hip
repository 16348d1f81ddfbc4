// One flow step — Conv1x1 -> ActNorm -> Coupling(1x1, ReLU, 3x3 reflect, ReLU, 1x1; affine map;
// log-det) — as ONE gfx950 kernel.  Reference: contextflow/model.py:129-147 (the per-step triple),
// layers/conv1x1.py:52-57, layers/actnorm.py:53-60, layers/coupling.py:26-29,52-66.
//
// Design (MI355X):
//  * every contraction runs on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32): the weights
//    are the A operand (row = output channel), the activations the B operand (column = pixel), so
//    a result tile has the PIXEL on the lane and the CHANNELS in the 16 accumulator registers.
//    That makes every LDS access of the activation planes [channel][pixel] a run of 32 consecutive
//    floats per half-wave (bank-conflict free, also under the 3x3 tap shifts with reflection), and
//    puts t, log_s and x1 of one (channel, pixel) in the SAME lane for the affine epilogue;
//  * a workgroup (4 waves) owns PIX = 128/256 pixels = SPW whole samples; activations never leave
//    the CU between the five contractions: x -> [MFMA] y=W'x+b' (ActNorm folded into the 1x1
//    matrix) -> LDS -> h1 -> LDS -> 3x3 -> LDS -> h -> epilogue.  HBM traffic per step = read x once + write z once.
//    The 3x3 - 72 of the 80 C^2 HW multiply-adds of a step - is an implicit GEMM over the 9 taps in the direct-form
//    geometries, and Winograd F(2x2,3x3) on 16x16x4 tiles (16 of 36 products per output tile; cf_step_common.h:
//    winograd_phase2) in the PIPE = 3 geometries that cf_flow_step_fwd dispatches at benchmark batch sizes;
//  * weights are pre-packed per call by cf_flow_step_prepare into MFMA-fragment order (one 16-byte
//    load per lane = A fragments of 4 k-steps, 1 KiB per wave-instruction, L2-resident) and
//    software-prefetched one group ahead of the MFMAs that use them;
//  * log-det: lane-local sum over channels, shuffle-reduce over the lanes of one sample, one LDS
//    hop across waves, then a plain read-modify-write of ldj_acc[b] by the single owner of sample
//    b — no float atomics, bitwise reproducible.
#include "cf_step_common.h"
#include <cstdlib>

namespace {

// ---- weight packing (cf_flow_step_prepare) ---------------------------------------------------------
// A fragment of k-step s, row tile rt, lane l: A[row = rt*32 + (l&31)][k = 2s + (l>>5)]; 4 k-steps
// per float4: element ((g*RT + rt)*64 + l)*4 + e holds k-step 4g+e.
// blockIdx.y = flow step of a batch (cf_flow_step_prepare_batch: the steps of one resolution level pack side by side)
constexpr int kPrepBatch = 16;
struct StepPackBatch {
    const float *Wm[kPrepBatch], *t[kPrepBatch], *logs[kPrepBatch], *w1[kPrepBatch], *b1[kPrepBatch], *w2[kPrepBatch], *b2[kPrepBatch],
        *w3[kPrepBatch], *b3[kPrepBatch];
    float* ws[kPrepBatch];
    int pieces;                          // also the bf16 pieces of the Winograd-domain weights (CONTEXTFLOW_BF16_SPLIT=1)
};
template <class G>
__global__ __launch_bounds__(256) void k_step_pack(const StepPackBatch pb) {
    const int bi = blockIdx.y;
    const float* __restrict__ Wm = pb.Wm[bi]; const float* __restrict__ t = pb.t[bi]; const float* __restrict__ logs = pb.logs[bi];
    const float* __restrict__ w1 = pb.w1[bi]; const float* __restrict__ b1 = pb.b1[bi]; const float* __restrict__ w2 = pb.w2[bi];
    const float* __restrict__ b2 = pb.b2[bi]; const float* __restrict__ w3 = pb.w3[bi]; const float* __restrict__ b3 = pb.b3[bi];
    float* __restrict__ ws = pb.ws[bi];
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    if (gtid == 0) {       // ldj_const = H*W*log|det Wm| + sum_c logs  (ws[1] holds log|det| from k_slogdet)
        float s = 0.f;
        for (int c = 0; c < G::C; ++c) s += logs[c];
        ws[0] = (float)G::HW * ws[1] + s;                    // conv1x1.py:53 + actnorm.py:58
    }
    for (int p = gtid; p < G::R03; p += gsz) {
        const int ch = chan_of_row<G>(p);
        ws[G::OFF_B0 + p] = ch >= 0 ? -t[ch] * expf(-logs[ch]) : 0.f;     // (x - t) e^{-logs} = e^{-logs} x - t e^{-logs}
        ws[G::OFF_B3 + p] = ch >= 0 ? b3[ch] : 0.f;
    }
    for (int p = gtid; p < G::R1; p += gsz) {
        ws[G::OFF_B1 + p] = p < G::HID ? b1[p] : 0.f;
        ws[G::OFF_B2 + p] = p < G::HID ? b2[p] : 0.f;
    }
    auto split = [](int e, int RT, int& g, int& rt, int& lane, int& j) {
        j = e & 3; lane = (e >> 2) & 63; const int q = e >> 8; rt = q % RT; g = q / RT;
    };
    int g, rt, lane, j;
    for (int e = gtid; e < G::NG0 * G::RT03 * 256; e += gsz) {            // phase 0: e^{-logs} Wm
        split(e, G::RT03, g, rt, lane, j);
        const int ch = chan_of_row<G>(rt * 32 + (lane & 31)), k = 2 * (4 * g + j) + (lane >> 5);
        ws[G::OFF_A0 + e] = (ch >= 0 && k < G::C) ? expf(-logs[ch]) * Wm[ch * G::C + k] : 0.f;
    }
    for (int e = gtid; e < G::NG1 * G::RT1 * 256; e += gsz) {             // phase 1: NN.0 (HID x HALF)
        split(e, G::RT1, g, rt, lane, j);
        const int row = rt * 32 + (lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
        ws[G::OFF_A1 + e] = (row < G::HID && k < G::HALF) ? w1[row * G::HALF + k] : 0.f;
    }
    for (int e = gtid; e < G::NG2 * G::RT1 * 256; e += gsz) {             // phase 2: NN.2, k = tap*HID + ci
        split(e, G::RT1, g, rt, lane, j);
        const int row = rt * 32 + (lane & 31);
        const int tap = g / G::NCG, ci = 8 * (g % G::NCG) + 2 * j + (lane >> 5);
        ws[G::OFF_A2 + e] = (row < G::HID) ? w2[(row * G::HID + ci) * 9 + tap] : 0.f;
    }
    for (int e = gtid; e < G::NG3 * G::RT03 * 256; e += gsz) {            // phase 3: NN.4 (C x HID), packed rows
        split(e, G::RT03, g, rt, lane, j);
        const int ch = chan_of_row<G>(rt * 32 + (lane & 31)), k = 2 * (4 * g + j) + (lane >> 5);
        ws[G::OFF_A3 + e] = (ch >= 0 && k < G::HID) ? w3[ch * G::HID + k] : 0.f;
    }
    {   // Winograd-domain weights of NN.2: U[xi][nu] = G w G^T (fp64, rounded once), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        constexpr int HID = G::HID;
        const double Gm[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
        for (int e = gtid; e < 16 * HID * HID; e += gsz) {
            const int jj = e & 3, ln = (e >> 2) & 63, q = e >> 8;
            const int kg = q % G::KG4, rt16 = (q / G::KG4) % G::RT16, pos = q / (G::KG4 * G::RT16);
            const int co = rt16 * 16 + (ln & 15), ci = 4 * (4 * kg + jj) + (ln >> 4), xi = pos >> 2, nu = pos & 3;
            const float* wk = w2 + ((int64_t)co * HID + ci) * 9;
            double u = 0.0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) u += Gm[xi][a] * Gm[nu][b] * (double)wk[a * 3 + b];
            ws[G::OFF_AW + e] = (float)u;
        }
        if (G::KB32 > 0 && G::HID == 32 && pb.pieces) {     // (the 16x16 level is the only consumer so far)
            // the same values as three bf16 pieces each (truncation split: exact, u = p0 + p1 + p2), in the operand layout of
            // v_mfma_f32_16x16x32_bf16 with k index (lane >> 4, j) = channel 4 j + (lane >> 4) of a 32-channel block
            unsigned short* wb = reinterpret_cast<unsigned short*>(ws + G::OFF_AWB);
            for (int e = gtid; e < 16 * G::RT16 * G::KB32 * 64 * 8; e += gsz) {
                const int j = e & 7, ln = (e >> 3) & 63, q = e >> 9;
                const int kb = q % G::KB32, rt16 = (q / G::KB32) % G::RT16, pos = q / (G::KB32 * G::RT16);
                const int co = rt16 * 16 + (ln & 15), ci = 32 * kb + 4 * j + (ln >> 4), xi = pos >> 2, nu = pos & 3;
                const float* wk = w2 + ((int64_t)co * HID + ci) * 9;
                double u = 0.0;
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) u += Gm[xi][a] * Gm[nu][b] * (double)wk[a * 3 + b];
                const float uf = (float)u;
                const unsigned u0 = __float_as_uint(uf) & 0xffff0000u;
                const float r1 = uf - __uint_as_float(u0);
                const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(u1);
                const unsigned pc[3] = {u0, u1, __float_as_uint(r2)};
#pragma unroll
                for (int pz = 0; pz < 3; ++pz)
                    wb[((((int64_t)(pos * G::RT16 + rt16) * G::KB32 + kb) * 3 + pz) * 64 + ln) * 8 + j] = (unsigned short)(pc[pz] >> 16);
            }
        }
        if (G::KB32 > 0 && G::HID == 32 && pb.pieces) {
            // the taps of NN.2 themselves as three bf16 pieces each, operand layout of v_mfma_f32_16x16x32_bf16 in the hardware's own k
            // order (direct_bf16_phases): element j of lane l = NN.2[16 rt + (l & 15)][32 kb + 8 (l >> 4) + j][tap]
            unsigned short* wb = reinterpret_cast<unsigned short*>(ws + G::OFF_ADB);
            for (int e = gtid; e < 9 * G::RT16 * G::KB32 * 64 * 8; e += gsz) {
                const int j = e & 7, ln = (e >> 3) & 63, q = e >> 9;
                const int kb = q % G::KB32, rt16 = (q / G::KB32) % G::RT16, tap = q / (G::KB32 * G::RT16);
                const int co = rt16 * 16 + (ln & 15), ci = 32 * kb + 8 * (ln >> 4) + j;
                const float uf = w2[((int64_t)co * HID + ci) * 9 + tap];
                const unsigned u0 = __float_as_uint(uf) & 0xffff0000u;
                const float r1 = uf - __uint_as_float(u0);
                const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(u1);
                const unsigned pc[3] = {u0, u1, __float_as_uint(r2)};
#pragma unroll
                for (int pz = 0; pz < 3; ++pz)
                    wb[((((int64_t)(tap * G::RT16 + rt16) * G::KB32 + kb) * 3 + pz) * 64 + ln) * 8 + j] = (unsigned short)(pc[pz] >> 16);
            }
        }
    }
    if constexpr (G::RS16) {
        // k_flow_step_rs16: natural row order, element ((rt * NG + gi) * 64 + lane) * 4 + j = A[16 rt + (lane & 15)][4 (4 gi + j) + (lane >> 4)]
        constexpr int C = G::C, HID = G::HID, HALF = G::HALF;
        for (int p = gtid; p < C; p += gsz) { ws[G::OFF_RB0 + p] = -t[p] * expf(-logs[p]); ws[G::OFF_RB3 + p] = b3[p]; }
        for (int p = gtid; p < HID; p += gsz) { ws[G::OFF_RB1 + p] = b1[p]; ws[G::OFF_RB2 + p] = b2[p]; }
        auto split16 = [](int e, int NG, int& rt, int& gi, int& row, int& kk, int& j) {
            j = e & 3; const int ln = (e >> 2) & 63, q = e >> 8; gi = q % NG; rt = q / NG; row = 16 * rt + (ln & 15); kk = ln >> 4;
        };
        int rt, gi, row, kk, j;
        for (int e = gtid; e < (C / 16) * (C / 16) * 256; e += gsz) {                     // e^{-logs} Wm
            split16(e, C / 16, rt, gi, row, kk, j);
            const int k = 4 * (4 * gi + j) + kk;
            ws[G::OFF_RA0 + e] = expf(-logs[row]) * Wm[row * C + k];
        }
        for (int e = gtid; e < (HID / 16) * (HALF / 16) * 256; e += gsz) {                // NN.0
            split16(e, HALF / 16, rt, gi, row, kk, j);
            ws[G::OFF_RA1 + e] = w1[row * HALF + 4 * (4 * gi + j) + kk];
        }
        for (int e = gtid; e < (HID / 16) * 9 * (HID / 16) * 256; e += gsz) {             // NN.2: group = tap * (HID / 16) + channel group
            split16(e, 9 * (HID / 16), rt, gi, row, kk, j);
            const int tap = gi / (HID / 16), ci = 4 * (4 * (gi % (HID / 16)) + j) + kk;
            ws[G::OFF_RA2 + e] = w2[(row * HID + ci) * 9 + tap];
        }
        for (int e = gtid; e < (C / 16) * (HID / 16) * 256; e += gsz) {                   // NN.4
            split16(e, HID / 16, rt, gi, row, kk, j);
            ws[G::OFF_RA3 + e] = w3[row * HID + 4 * (4 * gi + j) + kk];
        }
    }
    if constexpr (G::SMALL) {
        // operands of the 16x16x4 phases: A fragment of k-step s, lane l = A[row = l & 15][k = 4 s + (l >> 4)];
        // element e = (group * 64 + lane) * 4 + j holds k-step 4 group + j
        for (int p = gtid; p < 16; p += gsz) {
            const int ch = chan_of_row16<G>(p);
            ws[G::OFF_SB0 + p] = ch >= 0 ? -t[ch] * expf(-logs[ch]) : 0.f;
            ws[G::OFF_SB3 + p] = ch >= 0 ? b3[ch] : 0.f;
        }
        for (int e = gtid; e < G::SG0 * 256; e += gsz) {
            const int jj = e & 3, ln = (e >> 2) & 63, gg = e >> 8, ch = chan_of_row16<G>(ln & 15), k = 4 * (4 * gg + jj) + (ln >> 4);
            ws[G::OFF_SA0 + e] = (ch >= 0 && k < G::C) ? expf(-logs[ch]) * Wm[ch * G::C + k] : 0.f;
        }
        for (int e = gtid; e < G::SG3 * 256; e += gsz) {
            const int jj = e & 3, ln = (e >> 2) & 63, gg = e >> 8, ch = chan_of_row16<G>(ln & 15), k = 4 * (4 * gg + jj) + (ln >> 4);
            ws[G::OFF_SA3 + e] = (ch >= 0 && k < G::HID) ? w3[ch * G::HID + k] : 0.f;
        }
        if constexpr (G::HID16) {
            for (int e = gtid; e < G::SG1 * 256; e += gsz) {               // NN.0: 16 rows x HALF
                const int jj = e & 3, ln = (e >> 2) & 63, gg = e >> 8, row = ln & 15, k = 4 * (4 * gg + jj) + (ln >> 4);
                ws[G::OFF_SA1 + e] = k < G::HALF ? w1[row * G::HALF + k] : 0.f;
            }
            for (int e = gtid; e < 9 * 256; e += gsz) {                    // NN.2: group = tap, k = input channel
                const int jj = e & 3, ln = (e >> 2) & 63, tap = e >> 8, row = ln & 15, ci = 4 * jj + (ln >> 4);
                ws[G::OFF_SA2 + e] = w2[(row * G::HID + ci) * 9 + tap];
            }
        }
    }
}

// ---- the step kernel ---------------------------------------------------------------------------------
// One workgroup per tile of SPW samples.  Per wave (all wave-local unless noted; a wave owns the pixel
// columns of its PTW tiles):
//   x (16-byte loads) -> registers -> LDS plane Xp[ch][pix]
//   phase 0  y = W'x + b'        (MFMA, B operand from Xp)   -> y0 -> LDS Y0, y1 stays in registers
//            z[:, :C/2] = y0     written from Y0 with 16-byte stores
//   phase 1  h1 = relu(NN.0 y0)  -> LDS H1 ;  BARRIER (3x3 taps read neighbouring waves' columns)
//   phase 2  3x3 reflect conv    ;  BARRIER (everyone done reading h1) ;  h2 -> LDS (in place)
//   phase 3  h = NN.4 h2 ; affine map ; z[:, C/2:] via LDS with 16-byte stores ; per-sample log-det
// SQ: x is the UN-squeezed tensor (B, C/4, 2H, 2W); Squeeze((2,2)) (squeeze.py:10-11) is folded into the x
// staging: one 16-byte load along the un-squeezed row = channels (4c'+2i1, +1) of squeezed pixels (x, x+1).
// dbg (optional, tests only): dumps of y0, h1, h2, h as [rows][tiles*PIX].
// CTX: per-sample bias of the specialist coupling (see conditioner_net); sb = (B, C) or (B, 2C) floats.
// DUMP (training): y0 and the post-ReLU h1 / h2 planes are also written to the tape `tp`; the backward kernel then
// loads them instead of recomputing phases 1 and 2 (cf_step_bwd.hip, TAPED).
// DBG: the test-only instantiation that honours `dbg`; the production kernels carry no dump branches at all.
template <class G, bool SQ, int CTX = 0, bool DUMP = false, bool DBG = false>
__global__ __launch_bounds__(256, G::MINW) void k_flow_step(const float* __restrict__ x, float* __restrict__ z,
                                                   float* __restrict__ ldj_acc, const float* __restrict__ ws, int B,
                                                   int64_t xbs, float* __restrict__ dbg_arg, int flags,
                                                   const float* __restrict__ sb, StepTape tp) {
    float* const dbg = DBG ? dbg_arg : nullptr;
    constexpr int C = G::C, HW = G::HW, W = G::W, H = G::H, PIX = G::PIX, HALF = G::HALF, HID = G::HID;
    constexpr int PTW = G::PTW, RT03 = G::RT03, RT1 = G::RT1;
    constexpr int WPX = 32 * PTW;                     // pixel columns owned by one wave
    constexpr int XI = C * PTW / 8;                   // 16-byte x items per lane and tile
    extern __shared__ __align__(16) float lds[];      // G::LDS_FLOATS floats (up to 160 KiB: dynamic)
    float* Y0 = lds;                    // [HALF][PIX]   y0 (phase 0 -> 1), later the z1 staging plane
    float* H1 = lds + HALF * PIX;       // [HID][PIX]    x plane (rows < C), then h1, then h2 in place

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int ntiles = (B + G::SPW - 1) / G::SPW;
    const int64_t dbg_cols = (int64_t)ntiles * PIX;
    (void)flags;

    int pix[PTW], pin[PTW];             // this lane's pixel column inside the workgroup / inside its sample
#pragma unroll
    for (int q = 0; q < PTW; ++q) {
        pix[q] = (wave * PTW + q) * 32 + li;
        pin[q] = pix[q] % HW;
    }

    // One tile per workgroup (grid = ntiles).  A persistent tile loop with the next tile's x prefetched into
    // registers was tried: hipcc then keeps 50-110 more VGPRs live across the loop (occupancy 4 -> 2 at C=16) and
    // it measured slower; co-resident workgroups hide the x-load latency instead.
    float4 xr[XI];
    const int tile = blockIdx.x;
    x_load<G, SQ>(xr, x, xbs, tile, B, wave, lane);
    {
        const float* __restrict__ wsl = ws;
        const int b0 = tile * G::SPW;
        int smp[PTW];
        bool live[PTW];
#pragma unroll
        for (int q = 0; q < PTW; ++q) { smp[q] = b0 + pix[q] / HW; live[q] = smp[q] < B; }

        x_to_lds<G, SQ>(xr, H1, wave, lane);
        cf_wave_sync();                 // x plane: written 16 bytes per lane, read as MFMA operands by other lanes

        // ================= phase 0: y = (e^{-logs} Wm) x - t e^{-logs}        (conv1x1.py:54 + actnorm.py:59)
        f32x16 acc0[RT03][PTW];
#pragma unroll
        for (int rt = 0; rt < RT03; ++rt)
#pragma unroll
            for (int q = 0; q < PTW; ++q) acc0[rt][q] = bias_tile(wsl + G::OFF_B0 + rt * 32, lk);
        dense_phase<G, G::KS0, G::NG0, RT03>(acc0, ws_rsrc(wsl, G::WS_FLOATS), G::OFF_A0, H1, pix, lane);
        // first half: conditioner input -> LDS, and it is also the first half of the output (coupling.py:65);
        // second half (x1 after Conv1x1+ActNorm) stays in registers until the epilogue
        float y1[PTW][HALF <= 16 ? 8 : 16];
#pragma unroll
        for (int q = 0; q < PTW; ++q)
#pragma unroll
            for (int r = 0; r < (HALF <= 16 ? 8 : 16); ++r) {
                const int idx = tile_row(r, lk);                 // channel index inside its half
                if (idx < HALF) Y0[idx * PIX + pix[q]] = acc0[0][q][r];
                y1[q][r] = (HALF <= 16) ? acc0[0][q][r + 8] : acc0[RT03 - 1][q][r];
            }
        cf_wave_sync();                 // y0 plane complete for this wave's columns
        z_store<G>(z, Y0, b0, 0, B, wave, lane);
        if constexpr (DUMP) rows_store_t<G, HALF, HALF>(tp.y0, Y0, b0, B, wave, lane);
        if (dbg) {
            __syncthreads();
            for (int e = tid; e < HALF * PIX; e += 256) dbg[(int64_t)(e / PIX) * dbg_cols + (int64_t)tile * PIX + (e % PIX)] = Y0[e];
            __syncthreads();
        }

        f32x16 acc3[RT03][PTW];
        if constexpr (CTX == 0) {
            conditioner_net<G, 0, DUMP>(acc3, lds, wsl, pix, pin, lane, tid, dbg, dbg_cols, tile, nullptr, nullptr, tp, B);
        } else {
            int soff[PTW];
#pragma unroll
            for (int q = 0; q < PTW; ++q) soff[q] = min(smp[q], B - 1) * (CTX == 1 ? C : HID);
            conditioner_net<G, CTX, DUMP>(acc3, lds, wsl, pix, pin, lane, tid, dbg, dbg_cols, tile, sb, soff, tp, B);
        }

        float lsum[PTW];
#pragma unroll
        for (int q = 0; q < PTW; ++q) {
            lsum[q] = 0.f;
#pragma unroll
            for (int r = 0; r < (HALF <= 16 ? 8 : 16); ++r) {
                const int idx = tile_row(r, lk);
                if (idx < HALF) {
                    const float tt = acc3[0][q][r];
                    const float raw = (HALF <= 16) ? acc3[0][q][r + 8] : acc3[RT03 - 1][q][r];
                    const float ls = cf_log_scale(raw);
                    Y0[idx * PIX + pix[q]] = fmaf(y1[q][r], __expf(ls), tt);      // coupling.py:63 (z1, staged in LDS)
                    lsum[q] += ls;
                    if constexpr (DUMP) {           // tape: 128 contiguous bytes per row and half wave
                        if (live[q]) {
                            const int64_t o = ((int64_t)smp[q] * HALF + idx) * HW + pin[q];
                            tp.ls[o] = ls;
                            tp.y1[o] = y1[q][r];
                        }
                    }
                    if (dbg) {
                        float* d = dbg + (int64_t)(C + 2 * HID) * dbg_cols + (int64_t)tile * PIX + pix[q];
                        d[(int64_t)idx * dbg_cols] = tt;
                        d[(int64_t)(HALF + idx) * dbg_cols] = raw;
                    }
                }
            }
        }
        cf_wave_sync();                 // z1 plane complete for this wave's columns
        z_store<G>(z, Y0, b0, HALF, B, wave, lane);
        cf_wave_sync();                 // ... and read back, before the log-det scratch below reuses those words

        // per-sample reduction of log_s: lanes of one sample inside a 32-pixel tile first (shuffles) ...
        constexpr int SEG = HW < 32 ? HW : 32;            // lanes (pixels) of one sample inside a tile
        float v[PTW];
#pragma unroll
        for (int q = 0; q < PTW; ++q) {
            v[q] = lsum[q];
#pragma unroll
            for (int o = 1; o < SEG; o <<= 1) v[q] += __shfl_xor(v[q], o, 64);
            v[q] += __shfl_xor(v[q], 32, 64);
        }
        if constexpr (HW <= 32 * PTW) {
            // ... whole samples live inside this wave: finish in registers, the first lane of a sample owns ldj_acc[b]
            constexpr int TPS = HW >= 32 ? HW / 32 : 1;   // tiles per sample
#pragma unroll
            for (int q = 0; q < PTW; q += TPS) {
                float sum = v[q];
#pragma unroll
                for (int i = 1; i < TPS; ++i) sum += v[q + i];
                if (lk == 0 && (li % SEG) == 0 && live[q]) ldj_acc[smp[q]] += wsl[0] + sum;
            }
        } else {
            // ... a sample spans several waves: one LDS hop through this wave's own columns of Y0 (its z1 values have
            // been read back already); the second barrier keeps the next tile's y0 from overwriting the scratch
            constexpr int TPS = HW / 32;
#pragma unroll
            for (int q = 0; q < PTW; ++q)
                if (lane == 0) Y0[wave * WPX + q] = v[q];
            __syncthreads();
            if (tid < G::SPW && b0 + tid < B) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < TPS; ++i) {
                    const int t = tid * TPS + i;                       // tile index inside the workgroup
                    sum += Y0[(t / PTW) * WPX + (t % PTW)];
                }
                ldj_acc[b0 + tid] += wsl[0] + sum;
            }
            __syncthreads();
        }
    }
}

// ---- small-channel forward step (C = 8 / 16 on 16x16 images: first resolution level of mnist / cifar10) ------------
// Same data flow as k_flow_step, one sample per workgroup, but the phases with <= 16 output rows run on
// v_mfma_f32_16x16x4_f32 tiles (same flop/cycle as 32x32x2, no padding rows):
//   phase 0  [y0 | y1] = W'x + b'   16 packed rows (chan_of_row16)                        K = C
//   phase 1  h1 = relu(NN.0 y0)     C = 16: 32 rows on 32x32x2 (conditioner_net);  C = 8: 16 rows on 16x16x4
//   phase 2  3x3 reflect conv       C = 16: 32x32x2, taps unrolled;                 C = 8: 16x16x4, a column tile is one
//                                   image row, so the reflected source row of a tap is a scalar
//   phase 3  [t | raw] = NN.4 h2    16 packed rows                                        K = 2C
// Result tile of the 16x16x4 form: column (pixel) = lane & 15, rows 4 (lane >> 4) + r in the 4 registers; the packing
// puts t, raw and y1 of channels 2g, 2g+1 into lane group g, so the affine epilogue is lane-local again.
// In the 16-row phases lanes l and l + 16 read the same bank (rows k, k+1 of a [row][256] plane): a 2-way conflict on
// ~100 ds_read_b32 per wave, i.e. a few hundred LDS cycles against ~22 000 MFMA cycles.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// DUMP (training): y0 and the post-ReLU h1 / h2 planes also go to the tape `tp` (see k_flow_step).
// CF_G16W_MINW: waves per SIMD the C = 16 Winograd geometry is compiled for (A/B builds of tools/dev/make_abl.py)
#ifndef CF_G16W_MINW
#define CF_G16W_MINW 4
#endif
// CTX: per-sample bias of the specialist coupling (see conditioner_net): 1 sb (B, C) on the conditioner output, 2 sb (B, 2C)
// before the first ReLU.
// (x and z carry no __restrict__: the chained form below runs steps 2.. IN PLACE on the z of the step before)
// timing-only probe (-DCF_ABL_NOCONF): the B-operand reads of the 16-row phases without their bank conflict (rows 4s + lg of a
// 256-float plane share a bank): the lane group shifts the COLUMN instead - wrong data, conflict-free
#ifdef CF_ABL_NOCONF
#define CF_ROWIDX(r, lg) ((r) * PIX + ((lg) * 16 & 63))
#else
#define CF_ROWIDX(r, lg) (((r) + (lg)) * PIX)
#endif
template <class G, bool SQ, bool DBG = false, bool DUMP = false, int CTX = 0>
__device__ __forceinline__ void flow_step_small_body(const float* x, float* z, float* __restrict__ ldj_acc, const float* __restrict__ ws,
                                                     int B, int64_t xbs, float* __restrict__ dbg, StepTape tp,
                                                     const float* __restrict__ sb = nullptr) {
    static_assert(G::SMALL, "16x16 images, one sample per workgroup, C <= 16");
    static_assert(CTX == 0 || !G::HID16, "the per-sample bias is wired into the 32-row conditioner phases (C = 16)");
    constexpr int C = G::C, W = 16, H = 16, PIX = 256, HALF = G::HALF, HID = G::HID, PTW = G::PTW;
    constexpr int XI = C * PTW / 8;
    extern __shared__ __align__(16) float lds[];
    float* Y0 = lds;                    // [HALF][PIX]
    float* H1 = lds + HALF * PIX;       // [HID][PIX]: x plane, then h1, then h2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x;        // = sample index
    const int colb = wave * 64 + l15;   // this lane's pixel column in column tile 0 (tile ct: + 16 ct)

    float4 xr[XI];
    x_load<G, SQ>(xr, x, xbs, tile, B, wave, lane);
    const ws_rsrc_t rs = ws_rsrc(ws, G::WS_FLOATS);
    x_to_lds<G, SQ>(xr, H1, wave, lane);
    if constexpr (G::LDS_W > 0) {
        // the Winograd-domain weights of the 3x3 into LDS (16 KB, fragment order as packed): read by every wave of phase 2,
        // two workgroup barriers from here
        float* WL = lds + (HALF + HID) * PIX;
#pragma unroll
        for (int i = 0; i < G::LDS_W / 1024; ++i)
            *reinterpret_cast<float4*>(WL + (i * 256 + tid) * 4) = ws_frag(rs, tid, G::OFF_AW + i * 1024);
    }
    cf_wave_sync();

    // ================= phase 0: [y0 | y1] = (e^{-logs} Wm) x - t e^{-logs}   (conv1x1.py:54 + actnorm.py:59)
    f32x4 acc0[4];
    {
        const float4 b = *reinterpret_cast<const float4*>(ws + G::OFF_SB0 + 4 * lg);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { acc0[ct][0] = b.x; acc0[ct][1] = b.y; acc0[ct][2] = b.z; acc0[ct][3] = b.w; }
#pragma unroll
        for (int g = 0; g < G::SG0; ++g) {
            const float4 a = ws_frag(rs, lane, G::OFF_SA0 + g * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * (4 * g + e) < C) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a, e), H1[CF_ROWIDX(4 * (4 * g + e), lg) + colb + 16 * ct],
                                                                        acc0[ct], 0, 0, 0);
                }
        }
    }
    // first half -> LDS (conditioner input and first half of the output, coupling.py:65); y1 stays in acc0[.][2..3]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (2 * lg + j < HALF) Y0[(2 * lg + j) * PIX + colb + 16 * ct] = acc0[ct][j];
    cf_wave_sync();
    z_store<G>(z, Y0, tile, 0, B, wave, lane);
    if constexpr (DUMP) rows_store_t<G, HALF, HALF>(tp.y0, Y0, tile, B, wave, lane);
    const int64_t dbg_cols = (int64_t)B * PIX;       // test-only dumps (DBG): planes as [rows][B * PIX], see k_flow_step
    auto dump = [&](const float* plane, int rows, int row0) {
        __syncthreads();
        for (int e = tid; e < rows * PIX; e += 256) dbg[(int64_t)(row0 + e / PIX) * dbg_cols + (int64_t)tile * PIX + (e % PIX)] = plane[e];
        __syncthreads();
    };
    if constexpr (DBG) dump(Y0, HALF, 0);
    // C = 8 (16 hidden rows on 16x16x4 tiles): the tape's ReLU mask words are defined in the 32x32x2 accumulator layout
    // (StepTape) - taken from the plane in LDS, this wave's own columns (rows_store_t before it ends with a wave fence)
    auto plane_mask_store = [&](unsigned* __restrict__ m) {
        const int li = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int q = 0; q < PTW; ++q) {
            unsigned b = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) b |= (H1[tile_row(r, lk) * PIX + (wave * PTW + q) * 32 + li] > 0.f ? 1u : 0u) << r;
            m[((int64_t)tile * G::NPT + wave * PTW + q) * 64 + lane] = b;
        }
    };
    (void)plane_mask_store;

    // ================= phases 1, 2: h2 = relu(NN.2 (*) relu(NN.0 y0 + b) + b)   (coupling.py:26-27)
    if constexpr (G::DBF) {
        static_assert(!DUMP && !DBG && CTX != 2, "evaluation form only");
        direct_bf16_phases<G>(lds, ws, rs, lane, wave);
    } else if constexpr (!G::HID16) {
        const int li = lane & 31;
        int pix[PTW], pin[PTW];
#pragma unroll
        for (int q = 0; q < PTW; ++q) { pix[q] = (wave * PTW + q) * 32 + li; pin[q] = pix[q]; }
        f32x16 unused[G::RT03][PTW];
        int soff[PTW];
#pragma unroll
        for (int q = 0; q < PTW; ++q) soff[q] = tile * HID;
        conditioner_net<G, CTX == 2 ? 2 : 0, DUMP, 2>(unused, lds, ws, pix, pin, lane, tid, DBG ? dbg : nullptr, dbg_cols, tile,
                                                      CTX == 2 ? sb : nullptr, soff, tp, B);
    } else {
        f32x4 a1[4];
        {
            const float4 b = *reinterpret_cast<const float4*>(ws + G::OFF_B1 + 4 * lg);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { a1[ct][0] = b.x; a1[ct][1] = b.y; a1[ct][2] = b.z; a1[ct][3] = b.w; }
            const float4 a = ws_frag(rs, lane, G::OFF_SA1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * e < HALF) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        a1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a, e), Y0[(4 * e + lg) * PIX + colb + 16 * ct], a1[ct], 0, 0, 0);
                }
            if constexpr (G::WINO) {
                // Winograd form of the 3x3 (cf_step_common.h: winograd_phase2): h1 goes to LDS in its parity-split pixel
                // order, which scatters a wave's pixels over the whole sample - every wave must be done with the x plane
                static_assert(!DUMP && !DBG, "the tape / the dumps are written by the direct-form geometry");
                __syncthreads();
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int pw = wino_pix<G>(0, wave * 4 + ct, l15);      // a column tile is one image row
#pragma unroll
                    for (int r = 0; r < 4; ++r) H1[(4 * lg + r) * PIX + (pw ^ ((r & 1) * (W / 2)))] = cf_relu(a1[ct][r]);
                }
            } else {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) H1[(4 * lg + r) * PIX + colb + 16 * ct] = cf_relu(a1[ct][r]);
            }
            if constexpr (DUMP) { rows_store_t<G, HID, HID>(tp.h1, H1, tile, B, wave, lane); plane_mask_store(tp.m1); }
        }
        __syncthreads();                 // h1 complete: the 3x3 taps read neighbouring waves' image rows
        if constexpr (DBG) dump(H1, HID, C);
        if constexpr (G::WINO) {
            winograd_phase2<G>(lds, ws, rs, lane, wave);       // ends with h2 (ReLU, bias) in natural order, own columns
        } else {
        f32x4 a2[4];
        {
            const float4 b = *reinterpret_cast<const float4*>(ws + G::OFF_B2 + 4 * lg);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { a2[ct][0] = b.x; a2[ct][1] = b.y; a2[ct][2] = b.z; a2[ct][3] = b.w; }
            int xo[3];                   // reflected source column per dx, plus this lane's k row of the h1 plane
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int xx = l15 + d - 1;
                xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
                xo[d] = HALF * PIX + lg * PIX + xx;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float4 a = ws_frag(rs, lane, G::OFF_SA2 + tap * 256);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    int yy = wave * 4 + ct + tap / 3 - 1;            // wave-uniform: a column tile is one image row
                    yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
                    const int src = xo[tap % 3] + yy * W;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        a2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a, e), lds[src + 4 * e * PIX], a2[ct], 0, 0, 0);
                }
            }
        }
        __syncthreads();                 // every wave has finished reading h1
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) H1[(4 * lg + r) * PIX + colb + 16 * ct] = cf_relu(a2[ct][r]);
        if constexpr (DUMP) { rows_store_t<G, HID, HID>(tp.h2, H1, tile, B, wave, lane); plane_mask_store(tp.m2); }
        if constexpr (DBG) dump(H1, HID, C + HID);
        }
    }
    cf_wave_sync();                      // h2: every lane's rows in place before other lanes read them as operands

    // ================= phase 3: [t | raw] = NN.4 h2 + b   (coupling.py:28); reads this wave's own columns of h2 only
    f32x4 acc3[4];
    {
        const float4 b = *reinterpret_cast<const float4*>(ws + G::OFF_SB3 + 4 * lg);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { acc3[ct][0] = b.x; acc3[ct][1] = b.y; acc3[ct][2] = b.z; acc3[ct][3] = b.w; }
#pragma unroll
        for (int g = 0; g < G::SG3; ++g) {
            const float4 a = ws_frag(rs, lane, G::OFF_SA3 + g * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * (4 * g + e) < HID) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc3[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a, e), H1[CF_ROWIDX(4 * (4 * g + e), lg) + colb + 16 * ct],
                                                                        acc3[ct], 0, 0, 0);
                }
        }
    }
    if constexpr (CTX == 1) {            // h = NN(x0) + CN(c): the sample's bias per packed row (coupling.py:39-42)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = chan_of_row16<G>(4 * lg + r);
            const float add = ch >= 0 ? sb[(int64_t)tile * C + ch] : 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc3[ct][r] += add;
        }
    }
    // ================= affine map and log-det   (coupling.py:52-66)
    float lsum = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (2 * lg + j < HALF) {
                const float ls = cf_log_scale(acc3[ct][j + 2]);
                Y0[(2 * lg + j) * PIX + colb + 16 * ct] = fmaf(acc0[ct][j + 2], __expf(ls), acc3[ct][j]);     // z1, staged in LDS
                lsum += ls;
                if constexpr (DUMP) {
                    const int64_t o = ((int64_t)tile * HALF + 2 * lg + j) * PIX + colb + 16 * ct;
                    tp.ls[o] = ls;
                    tp.y1[o] = acc0[ct][j + 2];
                }
                if constexpr (DBG) {
                    float* d = dbg + (int64_t)(C + 2 * HID) * dbg_cols + (int64_t)tile * PIX + colb + 16 * ct;
                    d[(int64_t)(2 * lg + j) * dbg_cols] = acc3[ct][j];
                    d[(int64_t)(HALF + 2 * lg + j) * dbg_cols] = acc3[ct][j + 2];
                }
            }
    cf_wave_sync();
    z_store<G>(z, Y0, tile, HALF, B, wave, lane);
    cf_wave_sync();
    lsum = cf_wave_sum(lsum);
    if (lane == 0) Y0[wave * 64] = lsum;          // own column (its z1 value has been read back by z_store already)
    __syncthreads();
    if (tid == 0 && tile < B) ldj_acc[tile] += ws[0] + ((Y0[0] + Y0[64]) + (Y0[128] + Y0[192]));
}
#ifndef CF_G16WB_MINW
#define CF_G16WB_MINW 3
#endif
#ifndef CF_G16DB_MINW
#define CF_G16DB_MINW 3
#endif
template <class G, bool SQ, bool DBG = false, bool DUMP = false, int CTX = 0>
__global__ __launch_bounds__(256, G::DBF ? CF_G16DB_MINW : G::BF16S ? CF_G16WB_MINW : (G::WINO && DUMP) ? 3 : (G::WINO && G::C == 16) ? CF_G16W_MINW : G::MINW) void k_flow_step_small(const float* __restrict__ x, float* __restrict__ z,
                                                                  float* __restrict__ ldj_acc, const float* __restrict__ ws,
                                                                  int B, int64_t xbs, float* __restrict__ dbg, StepTape tp,
                                                                  const float* __restrict__ sb = nullptr) {
    flow_step_small_body<G, SQ, DBG, DUMP, CTX>(x, z, ldj_acc, ws, B, xbs, dbg, tp, sb);
}
// Chained form for small batches (cf_flow_step_fwd_chain): the n <= kChain consecutive flow steps of one resolution level in ONE
// launch.  A workgroup owns whole samples end to end, so step k + 1 can read what step k wrote as soon as the workgroup has passed
// a barrier; steps 2.. run in place on z.  At a batch of 256 a forward is ~20 launches of 10-35 us that a replayed graph issues
// at 7-12 us per node whatever they do: 12 step launches become 3.
constexpr int kChain = 4;
struct WsChain { const float* ws[kChain]; };
template <class G, bool SQ>
__global__ __launch_bounds__(256, (G::WINO && G::C == 16) ? CF_G16W_MINW : G::MINW) void k_flow_step_small_chain(const float* x, float* z, float* __restrict__ ldj_acc,
                                                                                   const WsChain wc, int nsteps, int B, int64_t xbs) {
    flow_step_small_body<G, SQ>(x, z, ldj_acc, wc.ws[0], B, xbs, nullptr, kNoTape);
    for (int st = 1; st < nsteps; ++st) {
        __syncthreads();                 // this workgroup's z of the step before is complete and visible to all of its waves
        flow_step_small_body<G, false>(z, z, ldj_acc, wc.ws[st], B, (int64_t)G::C * G::HW, nullptr, kNoTape);
    }
}
template <class G, bool SQ>
int launch_step_small_chain(const float* x, float* z, float* ldj, const WsChain& wc, int n, int B, int64_t xbs, hipStream_t s) {
    k_flow_step_small_chain<G, SQ><<<dim3(B), dim3(256), (size_t)G::LDS_FLOATS * sizeof(float), s>>>(x, z, ldj, wc, n, B, xbs);
    return 0;
}

template <class G, bool SQ, bool DBG = false, bool DUMP = false, int CTX = 0>
int launch_step_small(const float* x, float* z, float* ldj, const float* ws, int B, int64_t xbs, hipStream_t s,
                      float* dbg = nullptr, StepTape tp = kNoTape, const float* sb = nullptr) {
    k_flow_step_small<G, SQ, DBG, DUMP, CTX><<<dim3(B), dim3(256), (size_t)G::LDS_FLOATS * sizeof(float), s>>>(x, z, ldj, ws, B, xbs, dbg, tp, sb);
    return 0;
}

// ---- row-split forward step for SMALL batches (C = 32 on 8x8, C = 64 on 4x4) -------------------------------------------
// At the reference's batch sizes (64 / 256, config.py:10) a launch has far fewer workgroups than the chip has CUs, and
// the time of a step is the serial work of ONE workgroup: in k_flow_step a wave runs every row tile of its own pixel
// columns (C = 64: 4 tiles x 576 k-steps in the 3x3 alone).  Here the 4 waves of a workgroup share PIXR = 32 NPT pixel
// columns (2 samples at 4x4, 1 sample at 8x8) and split the OUTPUT ROWS: in phases 1 and 2 each wave owns one
// (row tile, pixel tile) pair, in phase 3 the two halves of K go to two waves whose partial sums meet in the epilogue.
// 4x the workgroups, a quarter of the serial MFMA chain per workgroup; the planes are exchanged through LDS with a
// workgroup barrier per phase.  Same packed workspace as k_flow_step.  Used below ~2 workgroups per CU (cf_flow_step_fwd).
// DUMP (training at small batches): the tape of cf_flow_step_fwd_taped - y0 / h1 / h2 planes, log-scale and y1, the ReLU
// mask words of this wave's (row tile, pixel tile) pairs, in the layout k_flow_step writes.
// KS = 2 (round 3): eight waves - waves 4..7 take the second half of the input channels of every tap of the 3x3, their
// partial sums meet waves 0..3's through the (then idle) T region.  What it buys is a second wave per SIMD to cover the
// first one's waits (35.0 -> 32.3 us at C = 64, 20.1 -> 19.1 at C = 32, batch of 256), NOT a shorter chain: the 2 304 MFMAs of a
// workgroup's 3x3 share the four matrix pipes of ONE CU whoever issues them (tools/dev/rs_ticks.py: phase 2 = 45.6 K of the
// 65 K cycles of a C = 64 step against 36.9 K of pure MFMA time).  Going below that needs more CUs per sample, i.e. 16-column
// tiles (one sample per workgroup at 4x4).
// timing-only probe build (tools/dev/make_abl.py ticks -DCF_RS_TICKS; read with tools/dev/rs_ticks.py): workgroup 0 leaves the
// cycles between the phase boundaries of k_flow_step_rs in z[wave * 16 + k] (the results of that build are garbage)
#ifdef CF_RS_TICKS
#define RS_TICK_INIT long long tacc_[12]; for (int k_ = 0; k_ < 12; ++k_) tacc_[k_] = 0; long long tprev_ = __builtin_readcyclecounter();
#define RS_TICK(k) { const long long tn_ = __builtin_readcyclecounter(); tacc_[k] += tn_ - tprev_; tprev_ = tn_; }
#define RS_TICK_DUMP if (blockIdx.x == 0 && lane == 0) for (int k_ = 0; k_ < 12; ++k_) z[wave * 16 + k_] = (float)tacc_[k_];
#else
#define RS_TICK_INIT
#define RS_TICK(k)
#define RS_TICK_DUMP
#endif
template <class G, int NPT, bool SQ, bool DUMP = false, int KS = 2>
__device__ __forceinline__ void flow_step_rs_body(const float* x, float* z, float* __restrict__ ldj_acc, const float* __restrict__ ws, int B,
                                                  int64_t xbs, StepTape tp = kNoTape) {
    constexpr int C = G::C, HW = G::HW, W = G::W, H = G::H, HALF = G::HALF, HID = G::HID, HP = G::HP;
    constexpr int PIXR = 32 * NPT, SPWR = PIXR / HW, RT1 = G::RT1, RT03 = G::RT03;
    static_assert(RT1 * NPT == 4 && RT03 * NPT == 2 && HP == HALF && PIXR % HW == 0, "row-split geometry");
    __shared__ __align__(16) float lds[(2 * C + HID + 2 * C) * PIXR];
    float* XP = lds;                         // [C][PIXR]    x plane
    float* Y = XP + C * PIXR;                // [C][PIXR]    rows [0, HALF) = y0, [HALF, C) = y1
    float* H1 = Y + C * PIXR;                // [HID][PIXR]  h1, then h2 in place
    float* T = H1 + HID * PIXR;              // [2][C][PIXR] the two K-halves of [t | raw]
    static_assert(KS == 2, "eight waves: the only form that is instantiated and tested");
    constexpr int NTH = 256 * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int w4 = wave & 3, kh2 = wave >> 2;          // phase 2: (row tile, pixel tile) pair of waves w4 and w4 + 4, channel half
    const int b0 = blockIdx.x * SPWR;
    const ws_rsrc_t rs = ws_rsrc(ws, G::WS_FLOATS);
    RS_TICK_INIT
    // the weight fragments of the three small products, requested before anything else (same reason as the ring of phase 2)
    const int rt1_ = (wave & 3) % RT1, it3_ = wave & 1, kh3_ = (wave >> 1) & 1, rt3_ = it3_ % RT03;
    constexpr int NGH3 = G::NG3 / 2;
    float4 f0[G::NG0], f1[G::NG1], f3[NGH3];
    if (wave < 2) {
#pragma unroll
        for (int g = 0; g < G::NG0; ++g) f0[g] = ws_frag(rs, lane, G::OFF_A0 + (g * RT03 + wave % RT03) * 256);
    }
    if (wave < 4) {
#pragma unroll
        for (int g = 0; g < G::NG1; ++g) f1[g] = ws_frag(rs, lane, G::OFF_A1 + (g * RT1 + rt1_) * 256);
#pragma unroll
        for (int g = 0; g < NGH3; ++g) f3[g] = ws_frag(rs, lane, G::OFF_A3 + ((kh3_ * NGH3 + g) * RT03 + rt3_) * 256);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- x -> LDS (element-wise: small batches are not bandwidth-bound; Squeeze folded into the index)
#pragma unroll
    for (int i = 0; i < C * PIXR / NTH; ++i) {
        const int e = tid + NTH * i, ch = e / PIXR, col = e - ch * PIXR, sm = col / HW, pp = col - sm * HW;
        const int64_t bb = min(b0 + sm, B - 1);
        int src;
        if constexpr (!SQ) src = ch * HW + pp;
        else { const int yy = pp / W, xx = pp - yy * W; src = (ch >> 2) * 4 * HW + (2 * yy + ((ch >> 1) & 1)) * 2 * W + 2 * xx + (ch & 1); }
        XP[e] = x[bb * xbs + src];
    }
    __syncthreads();
    RS_TICK(0)
    // ---- phase 0: [y0 | y1] = W' x + b'   (waves 0, 1: one (row tile, pixel tile) pair each)
    if (wave < 2) {
        const int rt = wave % RT03, q = wave / RT03, col = q * 32 + li;
        f32x16 acc = bias_tile(ws + G::OFF_B0 + rt * 32, lk);
#pragma unroll
        for (int g = 0; g < G::NG0; ++g) {
            const float4 a = f0[g];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < G::KS0)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a, e), XP[(2 * (4 * g + e) + lk) * PIXR + col], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[(rt * 32 + tile_row(r, lk)) * PIXR + col] = acc[r];
    }
    __syncthreads();
    RS_TICK(1)
    // first half of the output = y0 (coupling.py:65)
#pragma unroll
    for (int i = 0; i < HALF * PIXR / NTH; ++i) {
        const int e = tid + NTH * i, ch = e / PIXR, col = e - ch * PIXR, sm = col / HW, pp = col - sm * HW;
        if (b0 + sm < B) {
            z[(int64_t)(b0 + sm) * C * HW + ch * HW + pp] = Y[e];
            if constexpr (DUMP) tp.y0[((int64_t)(b0 + sm) * HALF + ch) * HW + pp] = Y[e];
        }
    }
    const int rt1 = w4 % RT1, q1 = w4 / RT1, col1 = q1 * 32 + li;              // this wave's pair in phases 1, 2
    // tape helpers: the mask word of this wave's accumulator tile (global 32-column tile = blockIdx NPT + q1), and a
    // cooperative copy of the [HID][PIXR] plane in LDS to its (B, HID, HW) place
    auto mask_word = [&](unsigned* __restrict__ m, const f32x16& a) {
        unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) bits |= (a[r] > 0.f ? 1u : 0u) << r;
        m[((int64_t)(blockIdx.x * NPT + q1) * RT1 + rt1) * 64 + lane] = bits;
    };
    auto plane_dump = [&](float* __restrict__ dst) {
#pragma unroll
        for (int i = 0; i < HID * PIXR / NTH; ++i) {
            const int e = tid + NTH * i, ch = e / PIXR, col = e - ch * PIXR, sm = col / HW, pp = col - sm * HW;
            if (b0 + sm < B) dst[((int64_t)(b0 + sm) * HID + ch) * HW + pp] = H1[e];
        }
    };
    RS_TICK(2)
    // ---- phase 1: h1 = relu(NN.0 y0 + b)
    if (wave < 4) {
        f32x16 acc = bias_tile(ws + G::OFF_B1 + rt1 * 32, lk);
#pragma unroll
        for (int g = 0; g < G::NG1; ++g) {
            const float4 a = f1[g];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < G::KS1)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a, e), Y[(2 * (4 * g + e) + lk) * PIXR + col1], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) H1[(rt1 * 32 + tile_row(r, lk)) * PIXR + col1] = cf_relu(acc[r]);
        if constexpr (DUMP) mask_word(tp.m1, acc);
    }
    __syncthreads();
    RS_TICK(3)
    if constexpr (DUMP) plane_dump(tp.h1);
    // ---- phase 2: h2 = relu(NN.2 (*) h1 + b), 3x3 reflect: K = 9 taps x HID channels, fragments one group ahead
    f32x16 acc2 = bias_tile(ws + G::OFF_B2 + rt1 * 32, lk);
    if (kh2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    }
    {
        const int sm = col1 / HW, pp = col1 - sm * HW, py = pp / W, px = pp - py * W;
        // a wave has ONE row tile here: 4 MFMAs (256 cycles) per weight fragment, less than an L2 round trip under load -
        // the fragments run in a ring of 4, fetched three groups ahead.  This wave's sequence of groups: for every tap the NH
        // channel groups of its half, cg0 .. cg0 + NH - 1
        constexpr int NH = G::NCG / KS;
        // ring depth: at a batch of 256 the packed weights of a step (0.16-0.66 MB, 6 MB over the twelve steps of a forward)
        // come from the Infinity Cache, not from L2 - measured ~1900 cycles per fragment under the load of 128 workgroups
        // streaming the same 590 KB (tools/dev/rs_ticks.py: 634 cycles per 256-cycle MFMA group with three loads in
        // flight).  Seven in flight cover it.
        constexpr int RD = NH % 8 == 0 ? 8 : 4;
        static_assert(NH % RD == 0, "fragment ring");
        const int cg0 = kh2 * NH;
        auto seq_frag = [&](int tap, int c) {              // fragment offset of group (tap, cg0 + c); c may run past NH into the next tap
            const int t2 = tap + c / NH, c2 = c % NH;
            const int g = (t2 < 9 ? t2 : 8) * G::NCG + cg0 + (t2 < 9 ? c2 : NH - 1);
            return G::OFF_A2 + (g * RT1 + rt1) * 256;
        };
        float4 a[RD];
#pragma unroll
        for (int j = 0; j < RD - 1; ++j) a[j] = ws_frag(rs, lane, seq_frag(0, j));
        // ... and the B operands (LDS) one group ahead: with a single accumulator chain per wave nothing else hides the
        // LDS latency in front of each group's first MFMA
        auto tap_src = [&](int tap) -> const float* {
            int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
            yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
            xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
            return H1 + sm * HW + yy * W + xx + (lk + 8 * cg0) * PIXR;      // channel 8 cg0 + lk of the source pixel
        };
        float bv[2][4];
        const float* src = tap_src(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[0][e] = src[2 * e * PIXR];
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const float* nsrc = tap_src(tap < 8 ? tap + 1 : 8);
#pragma unroll
            for (int c = 0; c < NH; ++c) {
                a[(c + RD - 1) & (RD - 1)] = ws_frag(rs, lane, seq_frag(tap, c + RD - 1));
#pragma unroll
                for (int e = 0; e < 4; ++e)       // next group: same tap, next 8 channels - or the first 8 of the next tap
                    bv[(c + 1) & 1][e] = (c + 1 < NH) ? src[(8 * (c + 1) + 2 * e) * PIXR] : nsrc[2 * e * PIXR];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a[c & (RD - 1)], e), bv[c & 1][e], acc2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            src = nsrc;
        }
    }
    RS_TICK(4)
    __syncthreads();                 // every wave has finished reading h1
    RS_TICK(5)
    if constexpr (KS > 1) {          // the two channel halves meet (T is idle until phase 3)
        if (kh2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(rt1 * 32 + tile_row(r, lk)) * PIXR + col1] = acc2[r];
        }
        __syncthreads();
        if (!kh2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] += T[(rt1 * 32 + tile_row(r, lk)) * PIXR + col1];
        }
    }
    if (!kh2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) H1[(rt1 * 32 + tile_row(r, lk)) * PIXR + col1] = cf_relu(acc2[r]);
        if constexpr (DUMP) mask_word(tp.m2, acc2);
    }
    __syncthreads();
    RS_TICK(6)
    if constexpr (DUMP) plane_dump(tp.h2);
    // ---- phase 3: [t | raw] = NN.4 h2 + b: pair = wave & 1, K half = wave >> 1 (the bias rides with half 0)
    if (wave < 4) {
        const int it = wave & 1, kh = wave >> 1, rt = it % RT03, q = it / RT03, col = q * 32 + li;
        f32x16 acc = bias_tile(ws + G::OFF_B3 + rt * 32, lk);
        if (kh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        constexpr int NGH = G::NG3 / 2;
        static_assert(G::NG3 % 2 == 0 && G::KS3 % 8 == 0, "phase-3 K halves");
#pragma unroll
        for (int gg = 0; gg < NGH; ++gg) {
            const int g = kh * NGH + gg;
            const float4 a = f3[gg];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a, e), H1[(2 * (4 * g + e) + lk) * PIXR + col], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) T[(kh * C + rt * 32 + tile_row(r, lk)) * PIXR + col] = acc[r];
    }
    __syncthreads();
    RS_TICK(7)
    // ---- affine map and log-det over the first 256 threads: element e = (channel, column); a thread's column is fixed
    float lsum = 0.f;
    if (wave < 4) {
    const int colE = tid % PIXR, smE = colE / HW, ppE = colE - smE * HW;
#pragma unroll
    for (int i = 0; i < HALF * PIXR / 256; ++i) {
        const int ch = (tid + 256 * i) / PIXR;
        const float tt = T[ch * PIXR + colE] + T[(C + ch) * PIXR + colE];
        const float raw = T[(HP + ch) * PIXR + colE] + T[(C + HP + ch) * PIXR + colE];
        const float ls = cf_log_scale(raw);
        const float y1v = Y[(HP + ch) * PIXR + colE];
        const float z1 = fmaf(y1v, __expf(ls), tt);
        lsum += ls;
        if (b0 + smE < B) {
            z[(int64_t)(b0 + smE) * C * HW + (HALF + ch) * HW + ppE] = z1;
            if constexpr (DUMP) {
                const int64_t o = ((int64_t)(b0 + smE) * HALF + ch) * HW + ppE;
                tp.ls[o] = ls;
                tp.y1[o] = y1v;
            }
        }
    }
    // per-sample sum: the lanes of a sample inside the wave (columns are lane % PIXR), then across the waves through LDS
    constexpr int LPS = HW < 64 ? HW : 64;            // consecutive lanes of one sample
#pragma unroll
    for (int o = 1; o < LPS; o <<= 1) lsum += __shfl_xor(lsum, o, 64);
    if (PIXR < 64 && HW < 32) lsum += __shfl_xor(lsum, 32, 64);    // PIXR = 32: lanes l and l + 32 hold the same column
    }
    __syncthreads();                                  // T is dead: reuse its first words
    constexpr int LPS2 = HW < 64 ? HW : 64;
    if (wave < 4 && (lane % LPS2) == 0 && (PIXR >= 64 || lane < 32)) T[wave * 4 + (lane % PIXR) / HW] = lsum;
    __syncthreads();
    if (tid < SPWR && b0 + tid < B) ldj_acc[b0 + tid] += ws[0] + ((T[tid] + T[4 + tid]) + (T[8 + tid] + T[12 + tid]));
    RS_TICK(8)
    RS_TICK_DUMP
}

template <class G, int NPT, bool SQ, bool DUMP = false, int KS = 2>
__global__ __launch_bounds__(256 * KS) void k_flow_step_rs(const float* __restrict__ x, float* __restrict__ z,
                                                      float* __restrict__ ldj_acc, const float* __restrict__ ws, int B,
                                                      int64_t xbs, StepTape tp = kNoTape) {
    flow_step_rs_body<G, NPT, SQ, DUMP, KS>(x, z, ldj_acc, ws, B, xbs, tp);
}
template <class G, int NPT, bool SQ, int KS = 2>
__global__ __launch_bounds__(256 * KS) void k_flow_step_rs_chain(const float* x, float* z, float* __restrict__ ldj_acc, const WsChain wc,
                                                            int nsteps, int B, int64_t xbs) {
    flow_step_rs_body<G, NPT, SQ, false, KS>(x, z, ldj_acc, wc.ws[0], B, xbs);
    for (int st = 1; st < nsteps; ++st) {
        __syncthreads();
        flow_step_rs_body<G, NPT, false, false, KS>(z, z, ldj_acc, wc.ws[st], B, (int64_t)G::C * G::HW);
    }
}
template <class G, int NPT, bool SQ>
int launch_step_rs_chain(const float* x, float* z, float* ldj, const WsChain& wc, int n, int B, int64_t xbs, hipStream_t s) {
    constexpr int SPWR = 32 * NPT / G::HW;
    k_flow_step_rs_chain<G, NPT, SQ, 2><<<dim3((B + SPWR - 1) / SPWR), dim3(512), 0, s>>>(x, z, ldj, wc, n, B, xbs);
    return 0;
}

template <class G, int NPT, bool SQ, bool DUMP = false>
int launch_step_rs(const float* x, float* z, float* ldj, const float* ws, int B, int64_t xbs, hipStream_t s, StepTape tp = kNoTape) {
    constexpr int SPWR = 32 * NPT / G::HW;
    constexpr int KS = 2;
    k_flow_step_rs<G, NPT, SQ, DUMP, KS><<<dim3((B + SPWR - 1) / SPWR), dim3(256 * KS), 0, s>>>(x, z, ldj, ws, B, xbs, tp);
    return 0;
}

#ifndef CF_RS_MAXB_C64
#define CF_RS_MAXB_C64 1024          // evaluation forward of the 4x4 level: row-split / one-sample kernels up to here, Winograd above
#endif
#ifndef CF_RS16_MAXB
#define CF_RS16_MAXB 1024
#endif
// ---- the 4x4 level at small batches: ONE sample per workgroup on 16-column tiles -------------------------------------------
// tools/dev/rs_ticks.py: at a batch of 256 the row-split kernel above spends 45.6 K of its 65 K cycles in the 3x3 - against
// 36.9 K of pure MFMA time: the 2 304 MFMAs of a workgroup's 3x3 (2 samples = 32 pixel columns of a 32x32x2 tile) share the
// four matrix pipes of ONE CU, and the launch has 128 workgroups for 256 CUs.  v_mfma_f32_16x16x4_f32 tiles hold exactly one
// 4x4 sample in their 16 columns: twice the workgroups, half the matrix work each.  The four waves split the output rows
// (16-row tiles: 1 / 2 / 2 / 1 per wave in the four products), planes [channel][16 pixels] in LDS (20 KB), natural row
// order (t / raw / y1 of a channel meet in the LDS epilogue), the fragments of the three small products requested at
// kernel start, those of the 3x3 through a ring of four groups.
// DUMP (training): the tape of cf_flow_step_fwd_taped - the planes are the LDS planes themselves (whole 16-byte rows), a mask
// word of the 32x32x2 layout is put together from the four 4-bit groups two lanes (g, g + 2) hold of it.
template <class G, bool SQ, bool DUMP = false>
__device__ __forceinline__ void flow_step_rs16_body(const float* x, float* z, float* __restrict__ ldj_acc, const float* __restrict__ ws, int B,
                                                    int64_t xbs, StepTape tp = kNoTape) {
    static_assert(G::RS16, "C = 64 on 4x4 images");
    constexpr int C = G::C, HW = 16, W = 4, H = 4, HALF = G::HALF, HID = G::HID, P = 16;
    constexpr int NG0 = C / 16, NG1 = HALF / 16, NGT = HID / 16, NG3 = HID / 16;            // groups of 4 k-steps: per product / per tap
    __shared__ __align__(16) float lds[(C + C + HID + C) * P + 4];
    float* XP = lds;                 // [C][16]    x
    float* YP = XP + C * P;          // [C][16]    y0 | y1
    float* H1 = YP + C * P;          // [HID][16]  h1, then h2 in place
    float* TP = H1 + HID * P;        // [C][16]    t | raw
    float* red = TP + C * P;         // [4]
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const ws_rsrc_t rs = ws_rsrc(ws, G::WS_FLOATS);
    float4 f0[NG0], f1[2][NG1], f3[NG3];
#pragma unroll
    for (int gi = 0; gi < NG0; ++gi) f0[gi] = ws_frag(rs, lane, G::OFF_RA0 + (w * NG0 + gi) * 256);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gi = 0; gi < NG1; ++gi) f1[t][gi] = ws_frag(rs, lane, G::OFF_RA1 + ((2 * w + t) * NG1 + gi) * 256);
#pragma unroll
    for (int gi = 0; gi < NG3; ++gi) f3[gi] = ws_frag(rs, lane, G::OFF_RA3 + (w * NG3 + gi) * 256);
    // the 3x3: group sequence j = tap * NGT + channel group, two row tiles per group, ring of four groups
    auto frag2 = [&](int j, float4 (&o)[2]) {
        const int jj = j < 9 * NGT ? j : 9 * NGT - 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) o[t] = ws_frag(rs, lane, G::OFF_RA2 + ((2 * w + t) * 9 * NGT + jj) * 256);
    };
    float4 ring[4][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) frag2(j, ring[j]);
    __builtin_amdgcn_sched_barrier(0);
    const float* xb = x + (int64_t)b * xbs;
#pragma unroll
    for (int i = 0; i < C * P / 256; ++i) {
        const int e = tid + 256 * i, ch = e / P, pp = e - ch * P;
        int src;
        if constexpr (!SQ) src = ch * HW + pp;
        else { const int yy = pp / W, xx = pp - yy * W; src = (ch >> 2) * 4 * HW + (2 * yy + ((ch >> 1) & 1)) * 2 * W + 2 * xx + (ch & 1); }
        XP[e] = xb[src];
    }
    __syncthreads();
    auto mma = [](float a, float bv, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0); };
    auto bias4 = [&](int off, int rt) {
        const float4 v = *reinterpret_cast<const float4*>(ws + off + 16 * rt + 4 * g);
        return f32x4{v.x, v.y, v.z, v.w};
    };
    // ---- phase 0: [y0 | y1] = e^{-logs} (Wm x - t), rows 16 w ..
    {
        f32x4 a0 = bias4(G::OFF_RB0, w), a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4 * NG0; ++s) {
            const float bv = XP[(4 * s + g) * P + col];
            if (s & 1) a1 = mma(f4e(f0[s >> 2], s & 3), bv, a1); else a0 = mma(f4e(f0[s >> 2], s & 3), bv, a0);
        }
        a0 += a1;
#pragma unroll
        for (int r = 0; r < 4; ++r) YP[(16 * w + 4 * g + r) * P + col] = a0[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < HALF * P / 256; ++i) {                                 // first half of the output = y0 (coupling.py:65)
        const int e = tid + 256 * i;
        z[(int64_t)b * C * HW + e] = YP[e];
        if constexpr (DUMP) tp.y0[(int64_t)b * HALF * HW + e] = YP[e];
    }
    // tape helpers: this lane's bits of the mask word of 32-row tile w (rows 16 t + 4 g + r -> bit 4 (2 t + g / 2) + r of the
    // word of lane (g & 1) * 32 + column), and the copy of a [HID][16] plane to its (B, HID, 16) place
    auto mask_word = [&](unsigned* __restrict__ m, const f32x4 (&a)[2]) {
        unsigned bits = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) bits |= (a[t][r] > 0.f ? 1u : 0u) << (4 * (2 * t + (g >> 1)) + r);
        bits |= (unsigned)__shfl_xor((int)bits, 32);
        if (g < 2) m[((int64_t)(b >> 1) * G::RT1 + w) * 64 + g * 32 + (b & 1) * 16 + col] = bits;
    };
    auto plane_dump = [&](float* __restrict__ dst) {
        for (int e = 4 * tid; e < HID * P; e += 1024)
            *reinterpret_cast<float4*>(dst + (int64_t)b * HID * HW + e) = *reinterpret_cast<const float4*>(H1 + e);
    };
    // ---- phase 1: h1 = relu(NN.0 y0 + b), row tiles 2 w, 2 w + 1
    {
        f32x4 a[2] = {bias4(G::OFF_RB1, 2 * w), bias4(G::OFF_RB1, 2 * w + 1)};
#pragma unroll
        for (int s = 0; s < 4 * NG1; ++s) {
            const float bv = YP[(4 * s + g) * P + col];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = mma(f4e(f1[t][s >> 2], s & 3), bv, a[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) H1[(16 * (2 * w + t) + 4 * g + r) * P + col] = cf_relu(a[t][r]);
        if constexpr (DUMP) mask_word(tp.m1, a);
    }
    __syncthreads();
    if constexpr (DUMP) plane_dump(tp.h1);
    // ---- phase 2: h2 = relu(NN.2 (*) h1 + b), 3x3 with reflect padding: 9 taps x HID channels
    f32x4 a2[2] = {bias4(G::OFF_RB2, 2 * w), bias4(G::OFF_RB2, 2 * w + 1)};
    {
        const int py = col / W, px = col - py * W;
        auto tap_src = [&](int tap) {
            int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
            yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
            xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
            return g * P + yy * W + xx;                                        // + 4 s P per k-step
        };
        float bv[2][4];
        int src = tap_src(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[0][e] = H1[src + 4 * e * P];
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int nsrc = tap_src(tap < 8 ? tap + 1 : 8);
#pragma unroll
            for (int c = 0; c < NGT; ++c) {                                    // NGT = 8: static ring slots
                frag2(tap * NGT + c + 3, ring[(c + 3) & 3]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    bv[(c + 1) & 1][e] = (c + 1 < NGT) ? H1[src + (16 * (c + 1) + 4 * e) * P] : H1[nsrc + 4 * e * P];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < 2; ++t) a2[t] = mma(f4e(ring[c & 3][t], e), bv[c & 1][e], a2[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
            src = nsrc;
        }
    }
    __syncthreads();                 // every wave has finished reading h1
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) H1[(16 * (2 * w + t) + 4 * g + r) * P + col] = cf_relu(a2[t][r]);
    if constexpr (DUMP) mask_word(tp.m2, a2);
    __syncthreads();
    if constexpr (DUMP) plane_dump(tp.h2);
    // ---- phase 3: [t | raw] = NN.4 h2 + b, rows 16 w ..
    {
        f32x4 a0 = bias4(G::OFF_RB3, w), a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4 * NG3; ++s) {
            const float bv = H1[(4 * s + g) * P + col];
            if (s & 1) a1 = mma(f4e(f3[s >> 2], s & 3), bv, a1); else a0 = mma(f4e(f3[s >> 2], s & 3), bv, a0);
        }
        a0 += a1;
#pragma unroll
        for (int r = 0; r < 4; ++r) TP[(16 * w + 4 * g + r) * P + col] = a0[r];
    }
    __syncthreads();
    // ---- affine map and log-det (coupling.py:52-66): element (channel, pixel), two per thread
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < HALF * P / 256; ++i) {
        const int e = tid + 256 * i;
        const float ls = cf_log_scale(TP[HALF * P + e]);
        z[(int64_t)b * C * HW + HALF * HW + e] = fmaf(YP[HALF * P + e], __expf(ls), TP[e]);
        if constexpr (DUMP) { tp.ls[(int64_t)b * HALF * HW + e] = ls; tp.y1[(int64_t)b * HALF * HW + e] = YP[HALF * P + e]; }
        lsum += ls;
    }
    lsum = cf_block_sum<4>(lsum, red);
    if (tid == 0) ldj_acc[b] += ws[0] + lsum;
}
template <class G, bool SQ, bool DUMP = false>
__global__ __launch_bounds__(256) void k_flow_step_rs16(const float* __restrict__ x, float* __restrict__ z,
                                                        float* __restrict__ ldj_acc, const float* __restrict__ ws, int B,
                                                        int64_t xbs, StepTape tp) {
    flow_step_rs16_body<G, SQ, DUMP>(x, z, ldj_acc, ws, B, xbs, tp);
}
template <class G, bool SQ>
__global__ __launch_bounds__(256) void k_flow_step_rs16_chain(const float* x, float* z, float* __restrict__ ldj_acc, const WsChain wc,
                                                              int nsteps, int B, int64_t xbs) {
    flow_step_rs16_body<G, SQ>(x, z, ldj_acc, wc.ws[0], B, xbs);
    for (int st = 1; st < nsteps; ++st) {
        __syncthreads();
        flow_step_rs16_body<G, false>(z, z, ldj_acc, wc.ws[st], B, (int64_t)G::C * G::HW);
    }
}

extern "C" int cf_slogdet_inverse_batch(int n, const float* const* Wm, int C, float* const* logabsdet, float* const* inv, cf_stream_t stream);

// n <= kPrepBatch steps of one shape: ONE factorisation launch (log|det Wm| into ws[1]; winv: also Wm^-1, training) and ONE
// packing launch
template <class G>
int launch_prepare(const StepPackBatch& pb, int n, float* const* winv, hipStream_t s) {
    float* lad[kPrepBatch];
    for (int i = 0; i < n; ++i) lad[i] = pb.ws[i] + 1;
    int rc = cf_slogdet_inverse_batch(n, pb.Wm, G::C, lad, winv, (cf_stream_t)s);
    if (rc) return rc;
    int blocks = (G::WS_FLOATS + 255) / 256;
    if (blocks > 512) blocks = 512;
    k_step_pack<G><<<dim3(blocks, n), dim3(256), 0, s>>>(pb);
    return 0;
}

template <class G, bool SQ, int CTX = 0, bool DUMP = false, bool DBG = false>
int launch_step(const float* x, float* z, float* ldj, const float* ws, int B, int64_t xbs, float* dbg, int flags,
                hipStream_t s, const float* sb = nullptr, StepTape tp = kNoTape) {
    constexpr size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
    if (lds_bytes > 64 * 1024) {          // one-time opt-in to > 64 KiB of dynamic LDS (immutable afterwards)
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_flow_step<G, SQ, CTX, DUMP, DBG>, 160 * 1024, raised, __func__)) return rc_;
    }
    const int grid = (B + G::SPW - 1) / G::SPW;
    k_flow_step<G, SQ, CTX, DUMP, DBG><<<dim3(grid), dim3(256), lds_bytes, s>>>(x, z, ldj, ws, B, xbs, dbg, flags, sb, tp);
    return 0;
}

// ---- inverse step: x = Conv1x1^-1(ActNorm^-1(Coupling^-1(z)))   (coupling.py:68-73, actnorm.py:78, conv1x1.py:72)
// z0 conditions the same three contractions as in the forward direction; y1 = (z1 - t) e^{-log_s}; then
// x = (Wm^-1 diag(e^{logs})) [z0 | y1] + Wm^-1 t  as one more MFMA phase with natural channel rows.
template <class G> struct GeoInv {
    static constexpr int RTI = (G::C + 31) / 32;                 // row tiles of the natural-order output
    static constexpr int OFF_WINV = 0;                           // C*C floats: Wm^-1 (scratch of the prepare step)
    static constexpr int OFF_BI = ((G::C * G::C + 3) / 4) * 4;
    static constexpr int OFF_AI = OFF_BI + RTI * 32;
    static constexpr int WS_FLOATS = OFF_AI + G::NG0 * RTI * 256;
};

template <class G>
__global__ __launch_bounds__(256) void k_step_pack_inv(const float* __restrict__ t, const float* __restrict__ logs,
                                                       float* __restrict__ wsi) {
    using I = GeoInv<G>;
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    const float* Winv = wsi + I::OFF_WINV;
    for (int c = gtid; c < I::RTI * 32; c += gsz) {
        float s = 0.f;
        if (c < G::C)
            for (int k = 0; k < G::C; ++k) s = fmaf(Winv[c * G::C + k], t[k], s);
        wsi[I::OFF_BI + c] = s;
    }
    for (int e = gtid; e < G::NG0 * I::RTI * 256; e += gsz) {
        const int j = e & 3, lane = (e >> 2) & 63, q = e >> 8, rt = q % I::RTI, g = q / I::RTI;
        const int row = rt * 32 + (lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
        wsi[I::OFF_AI + e] = (row < G::C && k < G::C) ? Winv[row * G::C + k] * expf(logs[k]) : 0.f;
    }
}

template <class G>
#ifndef CF_INV16_MINW
#define CF_INV16_MINW 3          // (4 waves per SIMD: 12 spilled registers, 6.55 vs 6.49 ms per cifar10 sample(16384))
#endif
__global__ __launch_bounds__(256, (G::WINO && G::C == 16) ? CF_INV16_MINW : G::MINW) void k_flow_step_inv(const float* __restrict__ z, float* __restrict__ x,
                                                       const float* __restrict__ ws, const float* __restrict__ wsi, int B,
                                                       int64_t zbs, int x_unsq) {
    using I = GeoInv<G>;
    constexpr int C = G::C, HW = G::HW, PIX = G::PIX, HALF = G::HALF;
    constexpr int PTW = G::PTW, RT03 = G::RT03, NR = (HALF <= 16 ? 8 : 16);
    constexpr int XI = C * PTW / 8;
    extern __shared__ __align__(16) float lds[];
    float* Y0 = lds;
    float* H1 = lds + HALF * PIX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int tile = blockIdx.x, b0 = tile * G::SPW;
    int pix[PTW], pin[PTW];
#pragma unroll
    for (int q = 0; q < PTW; ++q) { pix[q] = (wave * PTW + q) * 32 + li; pin[q] = pix[q] % HW; }

    {   // conditioner input: the first half of z, this wave's columns, straight into the Y0 plane (16-byte accesses)
        float4 zr[XI];
        x_load<G, false>(zr, z, zbs, tile, B, wave, lane);
        x_to_lds<G, false>(zr, H1, wave, lane);                  // z plane: rows [0,HALF) = z0, [HALF,C) = z1
        if constexpr (G::LDS_W > 0) {
            // C = 8: the Winograd-domain weights of the 3x3 into LDS behind the planes (winograd_phase2 reads them there; two
            // workgroup barriers of the conditioner lie between this and their first use)
            const ws_rsrc_t rsw = ws_rsrc(ws, G::WS_FLOATS);
            float* WL = lds + (HALF + G::HID) * PIX;
#pragma unroll
            for (int i = 0; i < G::LDS_W / 1024; ++i)
                *reinterpret_cast<float4*>(WL + (i * 256 + tid) * 4) = ws_frag(rsw, tid, G::OFF_AW + i * 1024);
        }
        cf_wave_sync();
#pragma unroll
        for (int q = 0; q < PTW; ++q)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int idx = tile_row(r, lk);
                if (idx < HALF) Y0[idx * PIX + pix[q]] = H1[idx * PIX + pix[q]];
            }
        cf_wave_sync();
    }
    f32x16 acc3[RT03][PTW];
    conditioner_net<G>(acc3, lds, ws, pix, pin, lane, tid, nullptr, 0, tile);
    cf_wave_sync();                      // phase 3 has read h2: its words are reused for the y plane
    // y = [z0 | (z1 - t) e^{-log_s}] as the operand plane of the last phase (this wave's columns of the H region).  z is read a
    // SECOND time here (L2-resident: this workgroup read it a few microseconds ago) instead of being carried in registers
    // across the conditioner: 2 NR PTW registers per lane had cost the Winograd geometries half of their occupancy
    // (140 / 256 / 256 VGPRs at C = 16 / 32 / 64: 2 / 1 / 1 waves per SIMD where the forward kernels run 4 / 2 / 2).
    {
        float4 zr[XI];
        x_load<G, false>(zr, z, zbs, tile, B, wave, lane);
        x_to_lds<G, false>(zr, H1, wave, lane);
        cf_wave_sync();
    }
#pragma unroll
    for (int q = 0; q < PTW; ++q)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int idx = tile_row(r, lk);
            if (idx < HALF) {
                const float tt = acc3[0][q][r];
                const float raw = (HALF <= 16) ? acc3[0][q][r + 8] : acc3[RT03 - 1][q][r];
                const float ls = cf_log_scale(raw);
                float* y1 = &H1[(HALF + idx) * PIX + pix[q]];
                *y1 = (*y1 - tt) * __expf(-ls);
            }
        }
    cf_wave_sync();
    f32x16 acc[I::RTI][PTW];
#pragma unroll
    for (int rt = 0; rt < I::RTI; ++rt)
#pragma unroll
        for (int q = 0; q < PTW; ++q) acc[rt][q] = bias_tile(wsi + I::OFF_BI + rt * 32, lk);
    dense_phase<G, G::KS0, G::NG0, I::RTI>(acc, reinterpret_cast<const float4*>(wsi + I::OFF_AI), H1, pix, lane);
    // x tiles -> LDS rows [C, 2C) of the H region (own columns) -> 16-byte stores
    float* Xp = H1 + C * PIX;
#pragma unroll
    for (int rt = 0; rt < I::RTI; ++rt)
#pragma unroll
        for (int q = 0; q < PTW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + tile_row(r, lk);
                if (row < C) Xp[row * PIX + pix[q]] = acc[rt][q][r];
            }
    cf_wave_sync();
    // x_unsq: the step sits behind a Squeeze((2,2)) in the flow - its inverse is an index map of these stores (squeeze.py:13-14)
    if (x_unsq) rows_store_unsq<G>(x, Xp, b0, B, wave, lane);
    else rows_store<G, C>(x, Xp, b0, 0, B, wave, lane);
}

template <class G>
int launch_prepare_inv(const float* Wm, const float* t, const float* logs, float* wsi, hipStream_t s) {
    using I = GeoInv<G>;
    int rc = cf_slogdet_inverse(Wm, G::C, wsi + I::WS_FLOATS, wsi + I::OFF_WINV, (cf_stream_t)s);   // log|det| lands past the tables
    if (rc) return rc;
    k_step_pack_inv<G><<<dim3(16), dim3(256), 0, s>>>(t, logs, wsi);
    return 0;
}

static int g_bf16_split = -1;             // -1: the environment decides (CONTEXTFLOW_BF16_SPLIT=1|2); 0 / 1 / 2: set by cf_bf16_split
static int bf16_split_mode() {            // 0 off, 1 Winograd-domain products as bf16 pieces (G16wb), 2 direct 3x3 on bf16 pieces (G16db)
    static const int v = [] { const char* e = getenv("CONTEXTFLOW_BF16_SPLIT"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }();
    return g_bf16_split < 0 ? v : g_bf16_split;
}
static bool bf16_split_enabled() { return bf16_split_mode() != 0; }

template <class G>
int launch_step_inv(const float* z, float* x, const float* ws, const float* wsi, int B, int64_t zbs, int x_unsq, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
    if (lds_bytes > 64 * 1024) {
        static std::atomic<uint64_t> raised{0};
        if (int rc_ = cf_raise_dynamic_lds((const void*)k_flow_step_inv<G>, 160 * 1024, raised, __func__)) return rc_;
    }
    k_flow_step_inv<G><<<dim3((B + G::SPW - 1) / G::SPW), dim3(256), lds_bytes, s>>>(z, x, ws, wsi, B, zbs, x_unsq);
    return 0;
}


}  // namespace

static bool direct_conv_only() {
    static const bool v = [] { const char* e = getenv("CONTEXTFLOW_DIRECT_CONV"); return e && e[0] == '1'; }();
    return v;
}

extern "C" {

int cf_flow_step_supported(int C, int H, int W, int kh, int kw) {
    return (kh == 3 && kw == 3 && shape_id(C, H, W) >= 0) ? 1 : 0;
}

int64_t cf_flow_step_ws_bytes(int C, int H, int W) {
    switch (shape_id(C, H, W)) {
        case 0: return (int64_t)G8::WS_FLOATS * 4;
        case 1: return (int64_t)G16::WS_FLOATS * 4;
        case 2: return (int64_t)G32::WS_FLOATS * 4;
        case 3: return (int64_t)G64::WS_FLOATS * 4;
    }
    return 0;
}

int cf_flow_step_prepare_batch(int n, const float* const* Wm, const float* const* t, const float* const* logs, const float* const* w1,
                               const float* const* b1, const float* const* w2, const float* const* b2, const float* const* w3,
                               const float* const* b3, void* const* ws, float* const* winv, int C, int H, int W, cf_stream_t stream) {
    CF_REQUIRE(n >= 0 && Wm && t && logs && w1 && b1 && w2 && b2 && w3 && b3 && ws);
    const int sid = shape_id(C, H, W);
    if (sid < 0) { cf_set_error("cf_flow_step_prepare: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED; }
    for (int i0 = 0; i0 < n; i0 += kPrepBatch) {
        const int m = n - i0 < kPrepBatch ? n - i0 : kPrepBatch;
        StepPackBatch pb{};
        pb.pieces = bf16_split_enabled() ? 1 : 0;
        for (int i = 0; i < m; ++i) {
            const int j = i0 + i;
            CF_REQUIRE(Wm[j] && t[j] && logs[j] && w1[j] && b1[j] && w2[j] && b2[j] && w3[j] && b3[j] && ws[j] &&
                       (reinterpret_cast<uintptr_t>(ws[j]) & 15) == 0 && (!winv || winv[j]));
            pb.Wm[i] = Wm[j]; pb.t[i] = t[j]; pb.logs[i] = logs[j]; pb.w1[i] = w1[j]; pb.b1[i] = b1[j]; pb.w2[i] = w2[j]; pb.b2[i] = b2[j];
            pb.w3[i] = w3[j]; pb.b3[i] = b3[j]; pb.ws[i] = (float*)ws[j];
        }
        int rc = 0;
        switch (sid) {
            case 0: rc = launch_prepare<G8>(pb, m, winv ? winv + i0 : nullptr, cf_s(stream)); break;
            case 1: rc = launch_prepare<G16>(pb, m, winv ? winv + i0 : nullptr, cf_s(stream)); break;
            case 2: rc = launch_prepare<G32>(pb, m, winv ? winv + i0 : nullptr, cf_s(stream)); break;
            default: rc = launch_prepare<G64>(pb, m, winv ? winv + i0 : nullptr, cf_s(stream)); break;
        }
        if (rc) return rc;
        CF_LAUNCH_CHECK();
    }
    return 0;
}

int cf_flow_step_prepare(const float* Wm, const float* t, const float* logs, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3, void* ws, int C, int H, int W,
                         cf_stream_t stream) {
    CF_REQUIRE(Wm && t && logs && w1 && b1 && w2 && b2 && w3 && b3 && ws && (reinterpret_cast<uintptr_t>(ws) & 15) == 0);
    return cf_flow_step_prepare_batch(1, &Wm, &t, &logs, &w1, &b1, &w2, &b2, &w3, &b3, &ws, nullptr, C, H, W, stream);
}

int cf_flow_step_prepare_train(const float* Wm, const float* t, const float* logs, const float* w1, const float* b1,
                               const float* w2, const float* b2, const float* w3, const float* b3, void* ws, float* winv,
                               int C, int H, int W, cf_stream_t stream) {
    CF_REQUIRE(Wm && t && logs && w1 && b1 && w2 && b2 && w3 && b3 && ws && winv && (reinterpret_cast<uintptr_t>(ws) & 15) == 0);
    return cf_flow_step_prepare_batch(1, &Wm, &t, &logs, &w1, &b1, &w2, &b2, &w3, &b3, &ws, &winv, C, H, W, stream);
}

int64_t cf_flow_step_inv_ws_bytes(int C, int H, int W) {
    switch (shape_id(C, H, W)) {
        case 0: return (int64_t)(GeoInv<G8>::WS_FLOATS + 4) * 4;
        case 1: return (int64_t)(GeoInv<G16>::WS_FLOATS + 4) * 4;
        case 2: return (int64_t)(GeoInv<G32>::WS_FLOATS + 4) * 4;
        case 3: return (int64_t)(GeoInv<G64>::WS_FLOATS + 4) * 4;
    }
    return 0;
}

int cf_flow_step_inv_prepare(const float* Wm, const float* t, const float* logs, void* wsi, int C, int H, int W,
                             cf_stream_t stream) {
    CF_REQUIRE(Wm && t && logs && wsi && (reinterpret_cast<uintptr_t>(wsi) & 15) == 0);
    int rc;
    float* w = (float*)wsi;
    switch (shape_id(C, H, W)) {
        case 0: rc = launch_prepare_inv<G8>(Wm, t, logs, w, cf_s(stream)); break;
        case 1: rc = launch_prepare_inv<G16>(Wm, t, logs, w, cf_s(stream)); break;
        case 2: rc = launch_prepare_inv<G32>(Wm, t, logs, w, cf_s(stream)); break;
        case 3: rc = launch_prepare_inv<G64>(Wm, t, logs, w, cf_s(stream)); break;
        default: cf_set_error("cf_flow_step_inv_prepare: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_flow_step_inv(const float* z, float* x, const void* ws, const void* wsi, int B, int C, int H, int W,
                     int64_t z_bstride, int x_unsqueezed, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(z && x && ws && wsi && B >= 0 && z_bstride >= (int64_t)C * H * W && z_bstride % 4 == 0);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(z) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0);
    int rc;
    const float* w = (const float*)ws;
    const float* wi = (const float*)wsi;
    switch (shape_id(C, H, W)) {
        case 0: rc = direct_conv_only() ? launch_step_inv<G8>(z, x, w, wi, B, z_bstride, x_unsqueezed, cf_s(stream))
                                        : launch_step_inv<G8w>(z, x, w, wi, B, z_bstride, x_unsqueezed, cf_s(stream)); break;
        // the conditioner is the forward's: Winograd form of its 3x3 unless CONTEXTFLOW_DIRECT_CONV=1
        case 1: rc = direct_conv_only() ? launch_step_inv<G16>(z, x, w, wi, B, z_bstride, x_unsqueezed, cf_s(stream))
                                        : launch_step_inv<G16w>(z, x, w, wi, B, z_bstride, x_unsqueezed, cf_s(stream)); break;
        case 2: rc = direct_conv_only() ? launch_step_inv<G32>(z, x, w, wi, B, z_bstride, x_unsqueezed, cf_s(stream))
                                        : launch_step_inv<G32w>(z, x, w, wi, B, z_bstride, x_unsqueezed, cf_s(stream)); break;
        case 3: rc = direct_conv_only() ? launch_step_inv<G64>(z, x, w, wi, B, z_bstride, x_unsqueezed, cf_s(stream))
                                        : launch_step_inv<G64w2>(z, x, w, wi, B, z_bstride, x_unsqueezed, cf_s(stream)); break;
        default: cf_set_error("cf_flow_step_inv: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

// test hook (not part of the public header): same as cf_flow_step_fwd plus per-phase dumps
int cf_flow_step_fwd_debug(const float* x, float* z, float* ldj_acc, const void* ws, int B, int C, int H, int W,
                           int64_t x_bstride, int in_squeeze, float* dbg, int flags, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && z && ldj_acc && ws && B >= 0 && x_bstride >= (int64_t)C * H * W);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && x_bstride % 4 == 0);
    if (B == 0) return 0;
    const float* w = (const float*)ws;
    int rc = 0;
#define CF_STEP(G) rc = in_squeeze ? launch_step<G, true>(x, z, ldj_acc, w, B, x_bstride, dbg, flags, cf_s(stream)) \
                                   : launch_step<G, false>(x, z, ldj_acc, w, B, x_bstride, dbg, flags, cf_s(stream))
    const int variant = (flags >> 16) & 15;
    if (dbg != nullptr && variant == 3 && shape_id(C, H, W) <= 1 && shape_id(C, H, W) >= 0) {     // dumps of k_flow_step_small
        if (shape_id(C, H, W) == 0)
            rc = in_squeeze ? launch_step_small<G8s, true, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), dbg)
                            : launch_step_small<G8s, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), dbg);
        else
            rc = in_squeeze ? launch_step_small<G16s, true, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), dbg)
                            : launch_step_small<G16s, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), dbg);
        if (rc) return rc;
        CF_LAUNCH_CHECK();
        return 0;
    }
    if (dbg != nullptr) {                  // per-phase dumps (tests): the default geometry of each shape only
        CF_REQUIRE(variant == 0);
#define CF_STEPD(G) rc = in_squeeze ? launch_step<G, true, 0, false, true>(x, z, ldj_acc, w, B, x_bstride, dbg, flags, cf_s(stream)) \
                                    : launch_step<G, false, 0, false, true>(x, z, ldj_acc, w, B, x_bstride, dbg, flags, cf_s(stream))
        switch (shape_id(C, H, W)) {
            case 0: CF_STEPD(G8); break;
            case 1: CF_STEPD(G16); break;
            case 2: CF_STEPD(G32); break;
            case 3: CF_STEPD(G64); break;
            default: cf_set_error("cf_flow_step_fwd_debug: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED;
        }
#undef CF_STEPD
        if (rc) return rc;
        CF_LAUNCH_CHECK();
        return 0;
    }
    switch (shape_id(C, H, W) * 8 + variant) {
        case 0: CF_STEP(G8); break;
        case 3: rc = in_squeeze ? launch_step_small<G8s, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream))
                                : launch_step_small<G8s, false>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream)); break;
        case 8: CF_STEP(G16); break;
        case 9: CF_STEP(G16v1); break;
        case 10: CF_STEP(G16v2); break;
        case 11: rc = in_squeeze ? launch_step_small<G16s, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream))
                                : launch_step_small<G16s, false>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream)); break;
        case 16: CF_STEP(G32); break;
        case 17: CF_STEP(G32v1); break;
        case 18: CF_STEP(G32v2); break;
        case 19: CF_STEP(G32v3); break;
        case 24: CF_STEP(G64); break;
        case 25: CF_STEP(G64v1); break;
        case 26: CF_STEP(G64v2); break;
        case 27: CF_STEP(G64v3); break;
        case 4: rc = in_squeeze ? launch_step_small<G8w, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream))           // variant 4 at C = 8
                               : launch_step_small<G8w, false>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream)); break;
        case 14: rc = in_squeeze ? launch_step_small<G16wb, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream))        // variant 6: bf16-piece form of variant 4
                                 : launch_step_small<G16wb, false>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream)); break;
        case 15: rc = in_squeeze ? launch_step_small<G16db, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream))        // variant 7: direct 3x3 on bf16 pieces
                                 : launch_step_small<G16db, false>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream)); break;
        case 12: rc = in_squeeze ? launch_step_small<G16w, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream))         // variant 4:
                                : launch_step_small<G16w, false>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream)); break;  // Winograd form of the 3x3
        case 20: CF_STEP(G32w); break;
        case 28: CF_STEP(G64w); break;
        case 29: CF_STEP(G64w2); break;           // variant 5: row-split Winograd at 128 pixels per workgroup
        default: cf_set_error("cf_flow_step_fwd: shape (%d,%d,%d) variant %d unsupported", C, H, W, variant); return CF_ERR_UNSUPPORTED;
    }
#undef CF_STEP
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

// n <= 4 consecutive flow steps of one shape in ONE launch, for the batch sizes at which a step is one of the small-batch
// kernels anyway (cf_flow_step_chain_max_batch: 16x16 images up to 1 024 samples, 8x8 up to 512, 4x4 up to 1 024); ws: HOST array
// of the n packed tables; in_squeeze applies to the first step.  z receives the output of the LAST step (the intermediate
// activations live in z too: the steps after the first run in place).  Same numbers as n calls of cf_flow_step_fwd, bit for bit.
int cf_flow_step_chain_max_batch(int C, int H, int W) {
    switch (shape_id(C, H, W)) {
        case 0: case 1: return direct_conv_only() ? 0 : 1024;
        case 2: return 512;                                // k_flow_step_rs's range in cf_flow_step_fwd
        case 3: return CF_RS16_MAXB < CF_RS_MAXB_C64 ? CF_RS16_MAXB : CF_RS_MAXB_C64;
    }
    return 0;
}

int cf_flow_step_fwd_chain(const float* x, float* z, float* ldj_acc, const void* const* ws, int n, int B, int C, int H, int W,
                           int64_t x_bstride, int in_squeeze, cf_stream_t stream) {
    if (B == 0 || n == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && n >= 1 && n <= kChain && B > 0 && x_bstride >= (int64_t)C * H * W);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && x_bstride % 4 == 0);
    const int sid = shape_id(C, H, W);
    if (sid < 0 || B > cf_flow_step_chain_max_batch(C, H, W)) {
        cf_set_error("cf_flow_step_fwd_chain: shape (%d,%d,%d) at a batch of %d is not a chained case", C, H, W, B);
        return CF_ERR_UNSUPPORTED;
    }
    WsChain wc{};
    for (int i = 0; i < n; ++i) { CF_REQUIRE(ws[i]); wc.ws[i] = (const float*)ws[i]; }
    hipStream_t s = cf_s(stream);
    switch (sid) {
        case 0: in_squeeze ? launch_step_small_chain<G8w, true>(x, z, ldj_acc, wc, n, B, x_bstride, s)
                           : launch_step_small_chain<G8w, false>(x, z, ldj_acc, wc, n, B, x_bstride, s); break;
        case 1: in_squeeze ? launch_step_small_chain<G16w, true>(x, z, ldj_acc, wc, n, B, x_bstride, s)
                           : launch_step_small_chain<G16w, false>(x, z, ldj_acc, wc, n, B, x_bstride, s); break;
        case 2: in_squeeze ? launch_step_rs_chain<G32, 2, true>(x, z, ldj_acc, wc, n, B, x_bstride, s)
                           : launch_step_rs_chain<G32, 2, false>(x, z, ldj_acc, wc, n, B, x_bstride, s); break;
        default:
            if (in_squeeze) k_flow_step_rs16_chain<G64, true><<<dim3(B), dim3(256), 0, s>>>(x, z, ldj_acc, wc, n, B, x_bstride);
            else k_flow_step_rs16_chain<G64, false><<<dim3(B), dim3(256), 0, s>>>(x, z, ldj_acc, wc, n, B, x_bstride);
    }
    CF_LAUNCH_CHECK();
    return 0;
}

// on = 0 / 1: switch the bf16-piece form of the 16x16 level off / on for the tables packed and the steps launched from now on
// (overrides CONTEXTFLOW_BF16_SPLIT); on < 0: query.  Returns the setting in force.
int cf_bf16_split(int on) {
    if (on >= 0) g_bf16_split = on > 2 ? 1 : on;
    return bf16_split_mode();
}

int cf_flow_step_fwd(const float* x, float* z, float* ldj_acc, const void* ws, int B, int C, int H, int W,
                     int64_t x_bstride, int in_squeeze, cf_stream_t stream) {
    // Small batches are latency-bound by the serial work of ONE workgroup (a launch of < 256 workgroups leaves CUs
    // idle anyway): pick the geometry with half the samples per workgroup (same packed-weight workspace).
    int flags = 0;
    const int sid = shape_id(C, H, W);
    if ((sid == 2 && B < 256 * G32::SPW) || (sid == 3 && B < 256 * G64::SPW)) flags = 2 << 16;
    if (sid == 0 || sid == 1) flags = 3 << 16;      // 16x16 images: k_flow_step_small
    // Winograd F(2x2,3x3) form of the 3x3 (winograd_phase2): 40 instead of 80 C^2 HW multiply-adds per sample and step.
    // CONTEXTFLOW_DIRECT_CONV=1 keeps the direct form (A/B measurements, tools/step_bench.py).
    const bool direct_only = direct_conv_only();
    if (!direct_only && (sid == 0 || sid == 1 || (sid == 2 && B >= 256 * G32::SPW))) flags = 4 << 16;
    if (!direct_only && sid == 3 && B >= 256 * G64w2::SPW) flags = 5 << 16;      // 4x4: 8 samples per workgroup, rows split over wave pairs
    // CONTEXTFLOW_BF16_SPLIT=1 (off by default): the 16x16 level's Winograd-domain products as bf16-piece MFMAs (G16wb)
    // (mode 2, the direct bf16-piece form: from 1024 samples per launch - below that the chained launches of the fp32 form run)
    if (!direct_only && sid == 1 && bf16_split_mode() == 1) flags = 6 << 16;
    if (!direct_only && sid == 1 && bf16_split_mode() == 2 && B >= 1024) flags = 7 << 16;
    // very small batches: the row-split kernel (a quarter of the serial chain per workgroup, 4x the workgroups)
    if ((sid == 2 && B <= 512) || (sid == 3 && B <= CF_RS_MAXB_C64)) {
        CF_REQUIRE(x && z && ldj_acc && ws && B >= 0 && x_bstride >= (int64_t)C * H * W);
        if (B == 0) return 0;
        const float* w = (const float*)ws;
        if (sid == 3 && B <= CF_RS16_MAXB) {   // 4x4: one sample per workgroup on 16-column tiles (twice the workgroups of the row-split kernel)
            if (in_squeeze) k_flow_step_rs16<G64, true><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, w, B, x_bstride, kNoTape);
            else k_flow_step_rs16<G64, false><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, w, B, x_bstride, kNoTape);
            CF_LAUNCH_CHECK();
            return 0;
        }
        if (sid == 2) { if (in_squeeze) launch_step_rs<G32, 2, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream));
                        else launch_step_rs<G32, 2, false>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream)); }
        else          { if (in_squeeze) launch_step_rs<G64, 1, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream));
                        else launch_step_rs<G64, 1, false>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream)); }
        CF_LAUNCH_CHECK();
        return 0;
    }
    return cf_flow_step_fwd_debug(x, z, ldj_acc, ws, B, C, H, W, x_bstride, in_squeeze, nullptr, flags, stream);
}

// Multiply-adds per sample the matrix pipe EXECUTES for one step at this batch size (bench.py's executed-flop roofline):
// pass 0 = cf_flow_step_fwd, 1 = cf_flow_step_fwd_taped, 2 = cf_flow_step_bwd_taped (direct transposed 3x3), 3 = cf_flow_step_inv
// (the forward's conditioner + W^-1 instead of W: the same count; Winograd form at every level).  The direct
// form runs C^2 (Conv1x1) + C^2 + 36 C^2 + 2 C^2 = 40 C^2 per pixel; the Winograd form of the 3x3 runs 16 instead of 36
// C^2.  The conditions below restate the dispatch of the two entry points above / below - change them together.
int64_t cf_flow_step_macs(int B, int C, int H, int W, int pass) {
    const int sid = shape_id(C, H, W);
    if (sid < 0 || pass < 0 || pass > 3) return 0;
    const int64_t direct = 40ll * C * C * H * W, wino = 20ll * C * C * H * W;
    if (pass == 2 || direct_conv_only()) return direct;
    if (pass == 3) return wino;
    bool w;
    if (sid == 0) w = pass == 0;                                  // mnist's C = 8 level: evaluation only
    else if (sid == 1) w = true;                                  // 16x16, C = 16: every batch size
    else if (sid == 2) w = B >= 256 * G32::SPW;
    else w = B >= 256 * G64w2::SPW;
    return w ? wino : direct;
}

// training forward: the same step, and the conditioner's intermediate planes y0 (B, C/2, H, W), h1, h2 (B, 2C, H, W;
// post-ReLU) go to the caller's tape.  cf_flow_step_bwd_taped consumes them: it skips the recompute of the two big
// contractions and uses the planes directly as the operands of the weight-gradient GEMMs.
int64_t cf_flow_step_tape_aux_bytes(int B, int C, int H, int W) {
    return shape_id(C, H, W) < 0 ? 0 : tape_aux_bytes(B, C, H, W);
}

int cf_flow_step_fwd_taped(const float* x, float* z, float* ldj_acc, const void* ws, float* t_y0, float* t_h1, float* t_h2,
                           void* t_aux, int B, int C, int H, int W, int64_t x_bstride, int in_squeeze, cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && t_y0 && t_h1 && t_h2 && t_aux && x_bstride >= (int64_t)C * H * W && x_bstride % 4 == 0);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(t_y0) & 15) == 0 && (reinterpret_cast<uintptr_t>(t_h1) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(t_h2) & 15) == 0 && (reinterpret_cast<uintptr_t>(t_aux) & 15) == 0);
    const float* w = (const float*)ws;
    const StepTape tp = make_tape(t_y0, t_h1, t_h2, t_aux, B, C, H, W);
    const bool direct_only = direct_conv_only();
    int rc = 0;
#define CF_STEPT(G) rc = in_squeeze ? launch_step<G, true, 0, true>(x, z, ldj_acc, w, B, x_bstride, nullptr, 0, cf_s(stream), nullptr, tp) \
                                    : launch_step<G, false, 0, true>(x, z, ldj_acc, w, B, x_bstride, nullptr, 0, cf_s(stream), nullptr, tp)
    // small batches (the reference trains with 256 samples): the half-size workgroup geometry, as in cf_flow_step_fwd
    switch (shape_id(C, H, W)) {
        case 0: rc = in_squeeze ? launch_step_small<G8s, true, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), nullptr, tp)
                                : launch_step_small<G8s, false, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), nullptr, tp); break;
        // as in cf_flow_step_fwd: Winograd form of the 3x3 on 16x16 images and at saturating batches (h1 reaches the tape from
        // the accumulators there, its LDS plane is in the parity-split order)
        case 1: if (direct_only) rc = in_squeeze ? launch_step_small<G16s, true, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), nullptr, tp)
                                                 : launch_step_small<G16s, false, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), nullptr, tp);
                else rc = in_squeeze ? launch_step_small<G16w, true, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), nullptr, tp)
                                     : launch_step_small<G16w, false, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), nullptr, tp);
                break;
        // very small batches: the row-split kernel, as in cf_flow_step_fwd (a quarter of the serial chain per workgroup)
        case 2: if (B <= 512) rc = in_squeeze ? launch_step_rs<G32, 2, true, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), tp)
                                              : launch_step_rs<G32, 2, false, true>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), tp);
                else if (B < 256 * G32::SPW) CF_STEPT(G32v2); else if (direct_only) CF_STEPT(G32); else CF_STEPT(G32w); break;
        case 3: if (B <= CF_RS16_MAXB) {    // one sample per workgroup, as the evaluation forward
                    if (in_squeeze) k_flow_step_rs16<G64, true, true><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, w, B, x_bstride, tp);
                    else k_flow_step_rs16<G64, false, true><<<dim3(B), dim3(256), 0, cf_s(stream)>>>(x, z, ldj_acc, w, B, x_bstride, tp);
                }
                else if (!direct_only && B >= 256 * G64w2::SPW) CF_STEPT(G64w2); else if (B < 256 * G64::SPW) CF_STEPT(G64v2); else CF_STEPT(G64); break;
        default: cf_set_error("cf_flow_step_fwd_taped: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED;
    }
#undef CF_STEPT
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

// specialist coupling (coupling.py:39-47): the same fused step with a per-sample bias from the CN net.
// mode 1: sbias (B, C) added to the conditioner OUTPUT (contextflow); mode 2: sbias (B, 2C) added before the first ReLU
// (CN(c) concatenated to the conditioner input).  The caller packs `ws` with cf_flow_step_prepare; with an identity
// matrix / zero ActNorm there, the kernel is the Coupling layer alone (per-sample Conv1x1 / ActNorm run before it).
int cf_flow_step_fwd_ctx(const float* x, float* z, float* ldj_acc, const void* ws, const float* sbias, int mode, int B, int C,
                         int H, int W, int64_t x_bstride, cf_stream_t stream) {
    if (B == 0) return 0;
    const bool keep_direct = (mode & 4) != 0;   // mode | 4: the direct form of the 3x3 whatever the batch size - the TRAINING forward under
    mode &= 3;                                  // contextflow, whose backward (cf_flow_step_bwd_ctx) rebuilds the conditioner in that form
    CF_REQUIRE(x && z && ldj_acc && ws && sbias && (mode == 1 || mode == 2) && B >= 0 && x_bstride >= (int64_t)C * H * W);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && x_bstride % 4 == 0);
    const float* w = (const float*)ws;
    int rc = 0;
#define CF_STEPC(G) rc = mode == 1 ? launch_step<G, false, 1>(x, z, ldj_acc, w, B, x_bstride, nullptr, 0, cf_s(stream), sbias) \
                                   : launch_step<G, false, 2>(x, z, ldj_acc, w, B, x_bstride, nullptr, 0, cf_s(stream), sbias)
    const bool wino = !direct_conv_only() && !keep_direct;      // as cf_flow_step_fwd: Winograd form of the 3x3 (16x16 always, 8x8 / 4x4 at saturating batches)
    switch (shape_id(C, H, W)) {
        case 0: CF_STEPC(G8); break;
        case 1: if (!wino) CF_STEPC(G16);      // one sample per workgroup, as the generalist's 16x16 level (k_flow_step_small)
                else rc = mode == 1 ? launch_step_small<G16w, false, false, false, 1>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), nullptr, kNoTape, sbias)
                                    : launch_step_small<G16w, false, false, false, 2>(x, z, ldj_acc, w, B, x_bstride, cf_s(stream), nullptr, kNoTape, sbias);
                break;
        case 2: if (wino && B >= 256 * G32::SPW) CF_STEPC(G32w); else CF_STEPC(G32); break;
        case 3: if (wino && B >= 256 * G64w2::SPW) CF_STEPC(G64w2); else CF_STEPC(G64); break;
        default: cf_set_error("cf_flow_step_fwd_ctx: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED;
    }
#undef CF_STEPC
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

// training forward of the specialist coupling WITHOUT contextflow (mode 2: CN(c) enters before the first ReLU and every
// parameter trains): cf_flow_step_fwd_ctx that also writes the tape planes of cf_flow_step_fwd_taped.  The per-sample
// bias only shapes h1, which the backward loads - cf_flow_step_bwd_taped needs no context argument.
int cf_flow_step_fwd_ctx_taped(const float* x, float* z, float* ldj_acc, const void* ws, const float* sbias, float* t_y0,
                               float* t_h1, float* t_h2, void* t_aux, int B, int C, int H, int W, int64_t x_bstride,
                               cf_stream_t stream) {
    if (B == 0) return 0;
    CF_REQUIRE(x && z && ldj_acc && ws && sbias && t_y0 && t_h1 && t_h2 && t_aux && x_bstride >= (int64_t)C * H * W && x_bstride % 4 == 0);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(t_y0) & 15) == 0 && (reinterpret_cast<uintptr_t>(t_h1) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(t_h2) & 15) == 0 && (reinterpret_cast<uintptr_t>(t_aux) & 15) == 0);
    const float* w = (const float*)ws;
    const StepTape tp = make_tape(t_y0, t_h1, t_h2, t_aux, B, C, H, W);
    int rc = 0;
#define CF_STEPCT(G) rc = launch_step<G, false, 2, true>(x, z, ldj_acc, w, B, x_bstride, nullptr, 0, cf_s(stream), sbias, tp)
    switch (shape_id(C, H, W)) {
        case 0: CF_STEPCT(G8); break;
        case 1: CF_STEPCT(G16); break;
        case 2: CF_STEPCT(G32); break;
        case 3: CF_STEPCT(G64); break;
        default: cf_set_error("cf_flow_step_fwd_ctx_taped: shape (%d,%d,%d) unsupported", C, H, W); return CF_ERR_UNSUPPORTED;
    }
#undef CF_STEPCT
    if (rc) return rc;
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
