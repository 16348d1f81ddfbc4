// TransCoupling in ONE kernel: patchify -> SimpleViT (LN, Linear, 6 x [LN, qkv, 1-head attention,
// out-proj, LN, MLP(GELU)], LN) -> un-patchify -> affine coupling map -> log-det.
// Reference: contextflow/layers/coupling.py:100-159, layers/simple_vit.py:18-127.
//
// MI355X design
//  * every Linear runs on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) with the WEIGHTS as
//    the A operand (row = output feature) and the activations as the B operand (column = token), so
//    a result tile has the TOKEN on the lane and the FEATURES in the accumulator registers:
//    LayerNorm / GELU / residuals are lane-local, and the B operand of the next Linear is a
//    conflict-free run of 32 consecutive floats of an LDS plane [feature][token];
//  * a wave owns 32 token columns = 32/N whole samples (N tokens per sample, a power of two), and
//    attention only mixes the tokens of one sample: q.k and p.v go through lane shuffles with an
//    online softmax.  Nothing crosses a wave -> the kernel has NO workgroup barrier;
//  * two LDS planes per workgroup (residual stream X, operand scratch Y; 64 features x 128 tokens
//    each, 64 KiB total, 2 workgroups/CU); weights arrive as pre-packed 16-byte MFMA fragments
//    (cf_vit_prepare, L2 resident) through the same two-stage operand pipeline as cf_step.hip.
// Limits: dim <= 64, patch_dim <= 64, dim_head == 64, heads == 1, tokens per sample in {1,2,4,8,16,32}.
#include "cf_common.h"
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int VR = 128;          // token columns per workgroup (4 waves x 32)
constexpr int VP = 64;           // feature rows per plane

__host__ __device__ inline int ngroups(int k) { return (k + 7) / 8; }        // groups of 4 k-steps (8 features)

// workspace layout (floats).  LN parameter blocks are [w(64) | b(64)], bias blocks 64 floats,
// fragment blocks ngroups(K) * RT * 256 floats.
struct VitLayout {
    int ln0, we, be, ln1, layer0, layer_stride, lnA, wqkv, wout, lnF, w1, b1, w2, b2, lnO, total;
};
__host__ __device__ inline VitLayout vit_layout(int pd, int dim, int depth) {
    VitLayout L;
    int o = 0;
    L.ln0 = o; o += 128;
    L.we = o; o += ngroups(pd) * 2 * 256;
    L.be = o; o += 64;
    L.ln1 = o; o += 128;
    L.layer0 = o;
    int p = 0;
    L.lnA = p; p += 128;
    L.wqkv = p; p += ngroups(dim) * 6 * 256;
    L.wout = p; p += ngroups(64) * 2 * 256;
    L.lnF = p; p += 128;
    L.w1 = p; p += ngroups(dim) * 2 * 256;
    L.b1 = p; p += 64;
    L.w2 = p; p += ngroups(dim) * 2 * 256;
    L.b2 = p; p += 64;
    L.layer_stride = p;
    o += depth * p;
    L.lnO = o; o += 128;
    L.total = o;
    return L;
}

// flat parameter order expected by cf_vit_prepare (host concatenates the reference's tensors in this order):
//   ln0.w(pd) ln0.b(pd) We(dim x pd) be(dim) ln1.w(dim) ln1.b(dim)
//   depth x [ lnA.w lnA.b Wqkv(192 x dim) Wout(dim x 64) lnF.w lnF.b W1(dim x dim) b1 W2(dim x dim) b2 ]
//   lnO.w lnO.b
__host__ __device__ inline int flat_layer_size(int dim) { return 2 * dim + 192 * dim + dim * 64 + 2 * dim + 2 * (dim * dim + dim); }

__device__ inline void pack_ln(float* dst, const float* w, const float* b, int n, int gtid, int gsz) {
    for (int i = gtid; i < 128; i += gsz) {
        const int f = i & 63;
        dst[i] = f < n ? (i < 64 ? w[f] : b[f]) : 0.f;
    }
}
__device__ inline void pack_vec(float* dst, const float* v, int n, int gtid, int gsz) {
    for (int i = gtid; i < 64; i += gsz) dst[i] = i < n ? v[i] : 0.f;
}
// Wt: [rows][K] row-major (nn.Linear weight).  Fragment element ((g*RT + rt)*64 + lane)*4 + e =
// W[rt*32 + (lane&31)][2*(4g+e) + (lane>>5)]
__device__ inline void pack_frags(float* dst, const float* Wt, int rows, int K, int RT, int gtid, int gsz) {
    const int n = ngroups(K) * RT * 256;
    for (int i = gtid; i < n; i += gsz) {
        const int e = i & 3, lane = (i >> 2) & 63, q = i >> 8, rt = q % RT, g = q / RT;
        const int row = rt * 32 + (lane & 31), k = 2 * (4 * g + e) + (lane >> 5);
        dst[i] = (row < rows && k < K) ? Wt[row * K + k] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_vit_pack(const float* __restrict__ flat, float* __restrict__ ws, int pd, int dim,
                                                  int depth) {
    const VitLayout L = vit_layout(pd, dim, depth);
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    const float* p = flat;
    pack_ln(ws + L.ln0, p, p + pd, pd, gtid, gsz); p += 2 * pd;
    pack_frags(ws + L.we, p, dim, pd, 2, gtid, gsz); p += dim * pd;
    pack_vec(ws + L.be, p, dim, gtid, gsz); p += dim;
    pack_ln(ws + L.ln1, p, p + dim, dim, gtid, gsz); p += 2 * dim;
    for (int l = 0; l < depth; ++l) {
        float* w = ws + L.layer0 + l * L.layer_stride;
        pack_ln(w + L.lnA, p, p + dim, dim, gtid, gsz); p += 2 * dim;
        pack_frags(w + L.wqkv, p, 192, dim, 6, gtid, gsz); p += 192 * dim;
        pack_frags(w + L.wout, p, dim, 64, 2, gtid, gsz); p += dim * 64;
        pack_ln(w + L.lnF, p, p + dim, dim, gtid, gsz); p += 2 * dim;
        pack_frags(w + L.w1, p, dim, dim, 2, gtid, gsz); p += dim * dim;
        pack_vec(w + L.b1, p, dim, gtid, gsz); p += dim;
        pack_frags(w + L.w2, p, dim, dim, 2, gtid, gsz); p += dim * dim;
        pack_vec(w + L.b2, p, dim, gtid, gsz); p += dim;
    }
    pack_ln(ws + L.lnO, p, p + dim, dim, gtid, gsz);
}

// ---- device helpers ------------------------------------------------------------------------------
__device__ __forceinline__ int trow(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
__device__ __forceinline__ float f4c(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

template <int RT> struct VOps { float4 a[RT]; float b[4]; };

template <int RT>
__device__ __forceinline__ void vload(VOps<RT>& o, const float4* __restrict__ fr, int g, const float* __restrict__ plane,
                                      int col, int lane) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) o.a[rt] = fr[(g * RT + rt) * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) o.b[e] = plane[(8 * g + 2 * e + (lane >> 5)) * VR + col];
}
template <int RT>
__device__ __forceinline__ void vmma(f32x16 (&acc)[RT], const VOps<RT>& o) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(o.a[rt], e), o.b[e], acc[rt], 0, 0, 0);
}
// acc[rt] += W_frag * plane over ng groups (operands of group g+1 requested before the MFMAs of group g)
template <int RT>
__device__ __forceinline__ void vgemm(f32x16 (&acc)[RT], const float* __restrict__ frags, int ng,
                                      const float* __restrict__ plane, int col, int lane) {
    const float4* fr = reinterpret_cast<const float4*>(frags);
    cf_wave_sync();                      // the operand plane was written by other lanes of this wave (cf_common.h)
    VOps<RT> o0, o1;
    vload<RT>(o0, fr, 0, plane, col, lane);
#pragma unroll 1
    for (int g = 0; g < ng; g += 2) {
        if (g + 1 < ng) vload<RT>(o1, fr, g + 1, plane, col, lane);
        __builtin_amdgcn_sched_barrier(0);
        vmma<RT>(acc, o0);
        if (g + 1 < ng) {
            if (g + 2 < ng) vload<RT>(o0, fr, g + 2, plane, col, lane);
            __builtin_amdgcn_sched_barrier(0);
            vmma<RT>(acc, o1);
        }
    }
}
template <int RT>
__device__ __forceinline__ void store_tiles(const f32x16 (&acc)[RT], float* __restrict__ plane, int rows, int col, int lk) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + trow(r, lk);
            if (row < rows) plane[row * VR + col] = acc[rt][r];
        }
    cf_wave_sync();
}
// acc = plane (+ bias)
__device__ __forceinline__ void load_tiles2(f32x16 (&acc)[2], const float* __restrict__ plane, const float* __restrict__ bias,
                                            int col, int lk) {
    cf_wave_sync();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + trow(r, lk);
            acc[rt][r] = (plane ? plane[row * VR + col] : 0.f) + (bias ? bias[row] : 0.f);
        }
}
// LayerNorm of one token column over D features (biased variance, eps 1e-5): dst = LN(src)*w + b (+ extra).
// The two half-waves (lanes l and l+32 hold the same token) split the features.
__device__ __forceinline__ void col_layernorm(const float* __restrict__ src, float* __restrict__ dst, int D,
                                              const float* __restrict__ ln, const float* __restrict__ extra, int col, int lk) {
    cf_wave_sync();                      // src rows were written by the other lane half of this token
    float s = 0.f;
    for (int f = lk; f < D; f += 2) s += src[f * VR + col];
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)D;
    float v = 0.f;
    for (int f = lk; f < D; f += 2) { const float d = src[f * VR + col] - mean; v = fmaf(d, d, v); }
    v += __shfl_xor(v, 32, 64);
    const float rstd = 1.0f / sqrtf(v / (float)D + 1e-5f);
    for (int f = lk; f < D; f += 2) {
        float o = (src[f * VR + col] - mean) * rstd * ln[f] + ln[64 + f];
        if (extra) o += extra[f];
        dst[f * VR + col] = o;
    }
    cf_wave_sync();
}

// ---- the kernel ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_vit_coupling(const float* __restrict__ x, float* __restrict__ z,
                                                         float* __restrict__ ldj, const float* __restrict__ ws,
                                                         const float* __restrict__ pos, int B, int C, int H, int W,
                                                         int p1, int p2, int dim, int depth, int64_t xbs, int inverse) {
    __shared__ __align__(16) float lds[2 * VP * VR];
    float* X = lds;                   // residual stream   [feature][token]
    float* Y = lds + VP * VR;         // operand scratch   [feature][token]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
    const int col = wave * 32 + li;
    const int cin = C / 2, gh = H / p1, gw = W / p2, ntok = gh * gw, pp = p1 * p2, pd = cin * pp, HW = H * W;
    const VitLayout L = vit_layout(pd, dim, depth);
    const int64_t gtok = (int64_t)blockIdx.x * VR + col;
    const int b = (int)(gtok / ntok), n = (int)(gtok % ntok);
    const bool live = b < B;
    const float* xb = x + (int64_t)min(b, B - 1) * xbs;
    const int py0 = (n / gw) * p1, px0 = (n % gw) * p2;            // top-left pixel of this token's patch

    for (int f = lk; f < VP; f += 2) { X[f * VR + col] = 0.f; Y[f * VR + col] = 0.f; }
    // patchify: feature f = (i1*p2 + i2)*cin + c   (simple_vit.py:101)
    for (int f = lk; f < pd; f += 2) {
        const int c = f % cin, ii = f / cin;
        Y[f * VR + col] = xb[(int64_t)c * HW + (py0 + ii / p2) * W + px0 + ii % p2];
    }
    cf_wave_sync();
    col_layernorm(Y, Y, pd, ws + L.ln0, nullptr, col, lk);                    // to_patch_embedding.1
    {
        f32x16 acc[2];
        load_tiles2(acc, nullptr, ws + L.be, col, lk);
        vgemm<2>(acc, ws + L.we, ngroups(pd), Y, col, lane);                   // to_patch_embedding.2
        store_tiles<2>(acc, X, dim, col, lk);
    }
    col_layernorm(X, X, dim, ws + L.ln1, pos + n * dim, col, lk);             // to_patch_embedding.3 + pos-emb

    const float scale = 0.125f;                                               // dim_head ** -0.5, dim_head = 64
    for (int l = 0; l < depth; ++l) {
        const float* wl = ws + L.layer0 + l * L.layer_stride;
        // ---- attention block: x = to_out(softmax(q k^T * scale) v) + x        (simple_vit.py:56-68,84)
        col_layernorm(X, Y, dim, wl + L.lnA, nullptr, col, lk);
        f32x16 o[2];
        {
            f32x16 qkv[6];
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) qkv[t][r] = 0.f;
            vgemm<6>(qkv, wl + L.wqkv, ngroups(dim), Y, col, lane);
            // online softmax over the ntok tokens of this sample: partner token = lane ^ m
            float mx = -INFINITY, sum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
            for (int m = 0; m < ntok; ++m) {
                float d = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) d = fmaf(qkv[t][r], __shfl_xor(qkv[2 + t][r], m, 64), d);
                d += __shfl_xor(d, 32, 64);                                   // the two halves hold different features
                d *= scale;
                const float nm = fmaxf(mx, d);
                const float c = expf(mx - nm), p = expf(d - nm);
                sum = sum * c + p;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] = fmaf(p, __shfl_xor(qkv[4 + t][r], m, 64), o[t][r] * c);
                mx = nm;
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= inv;
        }
        store_tiles<2>(o, Y, 64, col, lk);
        {
            // product first, ONE addition into the residual stream afterwards (as the reference: simple_vit.py:84).  Summing
            // the k-steps on top of X rounds every partial sum at the residual's magnitude: 2-3x the reference's fp32 error
            f32x16 acc[2], res[2];
            load_tiles2(acc, nullptr, nullptr, col, lk);
            vgemm<2>(acc, wl + L.wout, ngroups(64), Y, col, lane);
            load_tiles2(res, X, nullptr, col, lk);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] += res[t];
            store_tiles<2>(acc, X, dim, col, lk);
        }
        // ---- MLP block: x = W2 gelu(W1 LN(x) + b1) + b2 + x                   (simple_vit.py:30-40,86)
        col_layernorm(X, Y, dim, wl + L.lnF, nullptr, col, lk);
        {
            f32x16 acc[2];
            load_tiles2(acc, nullptr, wl + L.b1, col, lk);
            vgemm<2>(acc, wl + L.w1, ngroups(dim), Y, col, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[t][r]; acc[t][r] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
            store_tiles<2>(acc, Y, dim, col, lk);
        }
        {
            f32x16 acc[2], res[2];
            load_tiles2(acc, nullptr, wl + L.b2, col, lk);
            vgemm<2>(acc, wl + L.w2, ngroups(dim), Y, col, lane);
            load_tiles2(res, X, nullptr, col, lk);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] += res[t];
            store_tiles<2>(acc, X, dim, col, lk);
        }
    }
    col_layernorm(X, Y, dim, ws + L.lnO, nullptr, col, lk);                   // transformer.norm
    cf_wave_sync();

    // ---- un-patchify (simple_vit.py:115) + affine coupling map (coupling.py:139-155) + log-det
    // net output channel ch of pixel (i1,i2) of this token = feature (i1*p2+i2)*C + ch; t = ch < C/2, raw = ch >= C/2
    float lsum = 0.f;
    float* zb = z + (int64_t)b * C * HW;
    for (int c = lk; c < cin; c += 2)
        for (int ii = 0; ii < pp; ++ii) {
            const int pix = (py0 + ii / p2) * W + px0 + ii % p2;
            const float tt = Y[(ii * C + c) * VR + col];
            const float raw = Y[(ii * C + cin + c) * VR + col];
            const float ls = 2.0f * tanhf(raw * 0.5f);
            const float x0 = xb[(int64_t)c * HW + pix], x1 = xb[(int64_t)(cin + c) * HW + pix];
            const float z1 = inverse ? (x1 - tt) / expf(ls) : x1 * expf(ls) + tt;
            lsum += ls;
            if (live) { zb[(int64_t)c * HW + pix] = x0; zb[(int64_t)(cin + c) * HW + pix] = z1; }
        }
    for (int o = 1; o < ntok; o <<= 1) lsum += __shfl_xor(lsum, o, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (!inverse && ldj != nullptr && live && lk == 0 && n == 0) ldj[b] = lsum;
}

bool vit_ok(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads) {
    if (C < 2 || C % 2 || p1 < 1 || p2 < 1 || H % p1 || W % p2) return false;
    const int ntok = (H / p1) * (W / p2), pd = (C / 2) * p1 * p2;
    if (heads != 1 || dim_head != 64 || dim > 64 || dim < 2 || pd > 64 || dim != C * p1 * p2) return false;
    return ntok >= 1 && ntok <= 32 && (ntok & (ntok - 1)) == 0;
}

}  // namespace

extern "C" {

int cf_vit_supported(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads) {
    return vit_ok(C, H, W, p1, p2, dim, dim_head, heads) ? 1 : 0;
}

int64_t cf_vit_ws_bytes(int patch_dim, int dim, int depth) { return (int64_t)vit_layout(patch_dim, dim, depth).total * 4; }

int64_t cf_vit_flat_params(int patch_dim, int dim, int depth) {
    return 2 * patch_dim + (int64_t)dim * patch_dim + dim + 2 * dim + (int64_t)depth * flat_layer_size(dim) + 2 * dim;
}

int cf_vit_prepare(const float* flat_params, void* ws, int patch_dim, int dim, int depth, cf_stream_t stream) {
    CF_REQUIRE(flat_params && ws && patch_dim > 0 && patch_dim <= 64 && dim > 0 && dim <= 64 && depth >= 0);
    CF_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
    k_vit_pack<<<dim3(64), dim3(256), 0, cf_s(stream)>>>(flat_params, (float*)ws, patch_dim, dim, depth);
    CF_LAUNCH_CHECK();
    return 0;
}

int cf_vit_coupling(const float* x, float* z, float* ldj, const void* ws, const float* pos, int B, int C, int H, int W,
                    int p1, int p2, int dim, int depth, int64_t x_bstride, int inverse, cf_stream_t stream) {
    if (B == 0) return 0;                       // empty batch: nothing to do (pointers may be null)
    CF_REQUIRE(x && z && ws && pos && B >= 0 && (inverse || ldj) && x_bstride >= (int64_t)C * H * W);
    if (!vit_ok(C, H, W, p1, p2, dim, 64, 1)) {
        cf_set_error("cf_vit_coupling: geometry C=%d H=%d W=%d p=(%d,%d) dim=%d unsupported", C, H, W, p1, p2, dim);
        return CF_ERR_UNSUPPORTED;
    }
    if (B == 0) return 0;
    const int64_t tokens = (int64_t)B * (H / p1) * (W / p2);
    k_vit_coupling<<<dim3((unsigned)((tokens + VR - 1) / VR)), dim3(256), 0, cf_s(stream)>>>(
        x, z, ldj, (const float*)ws, pos, B, C, H, W, p1, p2, dim, depth, x_bstride, inverse);
    CF_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
