// Fused attention matrices of the one-kernel transformer steps (cf_vit_step.hip, cf_vit_rs.hip), formed ONCE per parameter
// version in fp64 and rounded to fp32 once.  A single head of HEAD = 64 on a DIM = 52-wide stream makes the score and the
// value / output maps factor through the stream (simple_vit.py:56-68): with u = g (.) n + b the attention pre-norm output,
//     q_i . k_j / 8  =  (M1 n_i + c1) . n_j  + terms without n_j (equal for the four keys of a query: they leave the softmax),
//         M1 = diag(g) Wk^T Wq diag(g) / 8,   c1 = diag(g) Wk^T Wq b / 8,
//     to_out(sum_j p_ij Wv u_j)  =  M2 (sum_j p_ij n_j) + c2,     M2 = Wout Wv diag(g),   c2 = Wout Wv b     (sum_j p_ij = 1).
// k_vit_fuse writes, per layer, [M1 (DIM x DIM, row-major) | c1 (DIM) | M2 (DIM x DIM) | c2 (DIM)] floats; the pack kernels
// of the two step kernels copy entries of these into their own fragment orders.  (Forming the entries inside the pack kernels -
// one 64-term fp64 chain per fragment element, on 64 workgroups - took 142 us per flow step; a training step at a batch of 256
// packs 8 flow steps per update.)
#pragma once
#include "cf_common.h"

namespace {

template <int DIM, int HEAD> struct VitFuse {
    static constexpr int LAYER_FLOATS = 2 * DIM * DIM + 2 * DIM;                     // scratch per layer
    static constexpr int M1 = 0, C1 = DIM * DIM, M2 = C1 + DIM, C2 = M2 + DIM * DIM;
    // flat parameter layout of one transformer layer (TransCoupling._flat_params): norm w, b | to_qkv | to_out | ff norm w, b | W1 b1 W2 b2
    static constexpr int P_GA = 0, P_BA = DIM, P_WQ = 2 * DIM, P_WK = P_WQ + HEAD * DIM, P_WV = P_WK + HEAD * DIM, P_WO = P_WV + HEAD * DIM;
    static constexpr int P_STRIDE = P_WO + DIM * HEAD + 2 * DIM + 2 * (DIM * DIM + DIM);
};

// grid (depth, 2, FUSE_SPLIT): blocks (l, 0, *) form M1 / c1 of layer l, blocks (l, 1, *) M2 / c2, each a row range.  Both factors
// go through LDS first (coalesced loads; the 64-term fp64 chains then read LDS: straight from global memory, on 12 workgroups, the
// kernel took 98 us - more than the backward kernel of a flow step at a batch of 256).  `layers` = the flat parameters at layer 0.
constexpr int FUSE_SPLIT = 4;
constexpr int kVitPrepBatch = 16;
struct VitFuseBatch {                                          // per flow step of a batch: its layers' flat parameters, its scratch
    const float* layers[kVitPrepBatch];
    float* scratch[kVitPrepBatch];
};
// blockIdx.x = layer + depth * (flow step of the batch)
template <int DIM, int HEAD>
__global__ __launch_bounds__(256) void k_vit_fuse(const VitFuseBatch fb, int depth) {
    using F = VitFuse<DIM, HEAD>;
    __shared__ float sA[HEAD * DIM], sB[HEAD * DIM];          // out[r][b] = sum_h sA[h][r] sB[h][b]
    __shared__ double tb[HEAD];
    const int bi = blockIdx.x / depth, layer = blockIdx.x - bi * depth;
    const float* p = fb.layers[bi] + (size_t)layer * F::P_STRIDE;
    float* out = fb.scratch[bi] + (size_t)layer * F::LAYER_FLOATS;
    const float *ga = p + F::P_GA, *ba = p + F::P_BA;
    const bool scores = blockIdx.y == 0;
    for (int i = threadIdx.x; i < HEAD * DIM; i += 256) {
        const int h = i / DIM, k = i - h * DIM;
        sA[i] = scores ? p[F::P_WK + i] : p[F::P_WO + k * HEAD + h];          // Wk[h][r]  |  Wout[r][h] transposed
        sB[i] = scores ? p[F::P_WQ + i] : p[F::P_WV + i];                     // Wq[h][b]  |  Wv[h][b]
    }
    __syncthreads();
    if (threadIdx.x < HEAD) {                                  // Wq b (scores) or Wv b (value path), b = the norm's bias
        double a = 0.0;
        for (int k = 0; k < DIM; ++k) a += (double)sB[threadIdx.x * DIM + k] * (double)ba[k];
        tb[threadIdx.x] = a;
    }
    __syncthreads();
    constexpr int RPB = (DIM + FUSE_SPLIT - 1) / FUSE_SPLIT;   // rows per block
    const int r0 = blockIdx.z * RPB, r1 = min(DIM, r0 + RPB);
    for (int i = threadIdx.x; i < (r1 - r0) * (DIM + 1); i += 256) {
        const int r = r0 + i / (DIM + 1), b = i % (DIM + 1);   // consecutive threads: consecutive b (conflict-free sB rows); b == DIM: the bias entry
        double s = 0.0;
        if (b < DIM) {
#pragma unroll 8
            for (int h = 0; h < HEAD; ++h) s += (double)sA[h * DIM + r] * (double)sB[h * DIM + b];
            if (scores) out[F::M1 + r * DIM + b] = (float)(0.125 * s * (double)ga[r] * (double)ga[b]);
            else out[F::M2 + r * DIM + b] = (float)(s * (double)ga[b]);
        } else {
#pragma unroll 8
            for (int h = 0; h < HEAD; ++h) s += (double)sA[h * DIM + r] * tb[h];
            if (scores) out[F::C1 + r] = (float)(0.125 * s * (double)ga[r]);
            else out[F::C2 + r] = (float)s;
        }
    }
}

}  // namespace
