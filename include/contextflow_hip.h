/*
 * contextflow_hip.h — C ABI of libcontextflow_hip.so (MI355X / gfx950).
 *
 * The reference (gudovskiy/contextflow) has no FFI: its "plugin API" is the Python package
 * `layers` (contextflow/layers/__init__.py:1-14) whose classes implement
 * FlowLayer.forward/reverse/logdet (contextflow/layers/flowlayer.py:7-24).  This library is the
 * seam *underneath* those classes: each entry point below replaces the torch arithmetic of the
 * reference lines it cites.  The modules in `contextflow_amd/layers/` bind it with ctypes
 * (contextflow_amd/layers/_hip.py); INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to contiguous fp32 (NCHW for activations) unless noted;
 *     `*_bstride` arguments are batch strides in elements, so a channel slice of a larger tensor
 *     (SplitPrior halves, Augment concat target) can be passed without a copy;
 *   - the caller allocates every output and workspace; the library never allocates, frees or
 *     retains caller memory; work is enqueued asynchronously on `stream` (a hipStream_t);
 *   - return 0 on success, a positive hipError_t or a negative CF_ERR_* otherwise; the message
 *     is in cf_last_error() (thread-local).  No exceptions cross the ABI, no global state.
 */
#ifndef CONTEXTFLOW_HIP_H
#define CONTEXTFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_ABI_VERSION 13
#define CF_ERR_ARG (-1)          /* bad argument (shape, null pointer, unsupported size) */
#define CF_ERR_UNSUPPORTED (-2)  /* shape not covered by this kernel; caller uses the generic path */

typedef void* cf_stream_t;       /* hipStream_t */

int cf_abi_version(void);
const char* cf_last_error(void);

/* ---- pre-processing (layers/dequantize.py:14-17, normalize.py:27-49, transforms.py:11-18) ---- */
/* y = x + u                                                     Dequantization.forward           */
int cf_dequant_fwd(const float* x, const float* u, float* y, int64_t n, cf_stream_t stream);
/* inverse=0: y = x/scale + translation ; inverse=1: y = (x - translation)*scale   Normalization   */
int cf_affine(const float* x, float* y, int64_t n, float translation, float scale, int inverse, cf_stream_t stream);
/* y = log x - log(1-x), ldj[b] = sum(-log x - log(1-x))         LogitTransform.forward/logdet    */
int cf_logit_fwd(const float* x, float* y, float* ldj, int B, int N, cf_stream_t stream);
/* y = sigmoid(x)                                                LogitTransform.reverse           */
int cf_sigmoid(const float* x, float* y, int64_t n, cf_stream_t stream);
/* y = floor(x)                                                  Dequantization.reverse           */
int cf_floor(const float* x, float* y, int64_t n, cf_stream_t stream);
/* the tail of FlowSequential.sample / inverse (flowsequential.py:32-39) in one pass: [Augment.reverse, augment.py] ->
 * LogitTransform.reverse -> Normalization.reverse x 2 (normalize.py:40) -> Dequantization.reverse:
 * x[b,i] = floor(((sigmoid(z[b,i]) - t2) * s2 - t1) * s1), i < n_keep; z_bstride > n_keep drops the augmented channels.
 * Bitwise the result of the single calls (cf_sigmoid, cf_affine inverse x 2, cf_floor). */
int cf_postprocess_inv(const float* z, float* x, int B, int n_keep, int64_t z_bstride, float t2, float s2, float t1, float s1,
                       cf_stream_t stream);
/* Fused layers 0-3 of the image flows (model.py:97-100): v = ((x+u)/s1 + t1)/s2 + t2,
 * y = logit(v) written with batch stride y_bstride, ldj[b] = ldj_const + sum(-log v - log(1-v)).   */
int cf_preprocess_fwd(const float* x, const float* u, float* y, float* ldj, int B, int N, int64_t y_bstride,
                      float t1, float s1, float t2, float s2, float ldj_const, cf_stream_t stream);
/* the same fused pre-processing with the noise drawn inside the kernel (Philox4x32-10; u ~ U[0,1) per element,
 * eps ~ N(0,1) for the `aug_n` elements of the Augment channel appended after the N image elements, whose
 * -log q(eps) is added to ldj).  rng_state: one uint64 on the device = the stream position of THIS call (the Philox
 * counter's high half); seed: the Philox key.  advance != 0: the call also increments rng_state[0] on the stream (a
 * self-advancing stream; graph replays then draw fresh noise); advance == 0: the caller supplies a fresh position per
 * call (FlowSequential draws it from torch's CUDA generator).  N, aug_n, y_bstride multiples of 4.              */
int cf_preprocess_rng_fwd(const float* x, float* y, float* ldj, uint64_t* rng_state, uint64_t seed, int B, int N,
                          int aug_n, int64_t y_bstride, float t1, float s1, float t2, float s2, float ldj_const,
                          int advance, cf_stream_t stream);
/* out[b] = 0.5*sum(eps^2) + 0.5*N*log(2 pi)  = -log N(eps;0,I)  Augment ldj (augment.py:14-18,
 * distributions/gaussian.py:50-54); eps rows have stride eps_bstride.                              */
int cf_std_normal_nll(const float* eps, float* out, int B, int N, int64_t eps_bstride, cf_stream_t stream);

/* ---- index-only layers (layers/squeeze.py:10-14) ---------------------------------------------- */
/* inverse=0: 'b c (h p1)(w p2) -> b (c p1 p2) h w' with x = (B,C,H,W); inverse=1: the opposite map,
 * x = (B, C*p1*p2, H/p1, W/p2) -> y = (B,C,H,W).  C,H,W always describe the UNsqueezed tensor.     */
int cf_squeeze(const float* x, float* y, int B, int C, int H, int W, int p1, int p2,
               int64_t x_bstride, int64_t y_bstride, int inverse, cf_stream_t stream);

/* ---- Conv1x1 (layers/conv1x1.py:52-57,72) ------------------------------------------------------ */
/* z[b,o,p] = sum_i Wm[o,i] x[b,i,p] (+ bias[o] if bias != NULL); C <= 192.                         */
int cf_conv1x1_fwd(const float* x, const float* Wm, const float* bias, float* z, int B, int C, int HW,
                   int64_t x_bstride, int64_t z_bstride, cf_stream_t stream);
/* logabsdet[0] = log|det Wm| (LU with partial pivoting; fp64 in registers up to C = 128, an fp32 LDS copy with
 * fp64 accumulation of log|pivot| up to 192); if inv != NULL also Wm^-1 (C x C).                          */
int cf_slogdet_inverse(const float* Wm, int C, float* logabsdet, float* inv, cf_stream_t stream);
/* ... of n matrices of one width in ONE launch (C <= 64; wider ones run one at a time).  The pointer arrays are HOST arrays of
 * device pointers (they travel as kernel arguments); inv may be NULL (no inverses) or hold n pointers.                     */
int cf_slogdet_inverse_batch(int n, const float* const* Wm, int C, float* const* logabsdet, float* const* inv, cf_stream_t stream);

/* ---- ActNorm (layers/actnorm.py:28-35,53-60,78) ------------------------------------------------- */
/* data-dependent init: t[c] = mean, logs[c] = log(unbiased_std + 1e-8) over (B,H,W).
 * ws: caller workspace of cf_actnorm_stats_ws_bytes(C) bytes.                                       */
int64_t cf_actnorm_stats_ws_bytes(int C);
int cf_actnorm_stats(const float* x, float* t, float* logs, void* ws, int B, int C, int HW, int64_t x_bstride,
                     cf_stream_t stream);
/* The two halves of cf_actnorm_stats, for a batch sharded over data-parallel ranks: every rank computes
 * sums[0:C] = sum x, sums[C:2C] = sum x^2 (fp64) of ITS shard, the ranks all-reduce the 2C doubles and their element
 * counts, and cf_actnorm_from_sums turns the global sums into t / logs - the statistics actnorm.py:28-35 takes from the
 * whole batch.  count = elements per channel behind the sums; count_dev (device, may be NULL) overrides it.          */
int cf_actnorm_sums(const float* x, double* sums, void* ws, int B, int C, int HW, int64_t x_bstride, cf_stream_t stream);
int cf_actnorm_from_sums(const double* sums, const double* count_dev, double count, float* t, float* logs, int C,
                         cf_stream_t stream);
/* inverse=0: z = (x - t)*exp(-logs), ldj_scalar[0] = sum_c logs (reference quirk: no H*W factor);
 * inverse=1: z = x*exp(logs) + t (ldj_scalar may be NULL).                                          */
int cf_actnorm(const float* x, const float* t, const float* logs, float* z, float* ldj_scalar,
               int B, int C, int HW, int inverse, cf_stream_t stream);

/* ---- Coupling (layers/coupling.py:26-29,39-73) -------------------------------------------------- */
/* generic k x k convolution, stride 1, reflect padding (ph,pw), optional ReLU; fp32 direct form.     */
int cf_conv2d_reflect(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                      int H, int W, int kh, int kw, int ph, int pw, int relu, int64_t x_bstride, cf_stream_t stream);
/* same implicit-GEMM kernel with ZERO padding (ph, pw): y (B, Cout, Hi + 2 ph - kh + 1, Wi + 2 pw - kw + 1).  With the
 * weights flipped in space and transposed in the channels and padding (kh-1, kw-1) this is the transposed convolution
 * of the backward; cf_reflect_pad_adjoint then folds the padded border (B*C, H + 2 ph, W + 2 pw) back onto the pixels it
 * was reflected from: together the data gradient of cf_conv2d_reflect.                                              */
int cf_conv2d_zero(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Hi, int Wi,
                   int kh, int kw, int ph, int pw, int relu, int64_t x_bstride, cf_stream_t stream);
int cf_reflect_pad_adjoint(const float* gpad, float* gx, int BC, int H, int W, int ph, int pw, cf_stream_t stream);
/* affine map from the net output h (B,C,HW): t = h[:, :C/2], log_s = 2 tanh(h[:, C/2:]/2).
 * inverse=0: z = [x0 | x1*exp(log_s)+t], ldj[b] = sum log_s ;  inverse=1: z = [x0 | (x1-t)/exp(log_s)] */
int cf_coupling_apply(const float* x, const float* h, float* z, float* ldj, int B, int C, int HW, int inverse,
                      cf_stream_t stream);

/* ---- GaussianMixtureDistribution.log_prob (layers/distributions/gaussian.py:138-161) ------------ */
/* parameter transform, once per call: a = 1/softplus(sG), nm = -mG (both (M*K, D)),
 * cst[m,k] = log_softmax(wG[m])[k] - sum_d log softplus(sG) - D/2 log(2 pi).                         */
int cf_gmm_prepare(const float* mG, const float* sG, const float* wG, float* a, float* nm, float* cst,
                   int M, int K, int D, cf_stream_t stream);
/* out[b,m] (+)= logsumexp_k( cst[m,k] - 0.5 * sum_d ((x[b,d] + nm)*a)^2 );  K <= 16.  (x - mu first, as the reference does:
 * the difference is exact for fitted means; gaussian.py:142-161.)
 * accumulate != 0 adds into out (used to fold SplitPrior's ldj into the running (B,M) log-det).
 * ws: optional workspace of cf_gmm_ws_bytes(B,M,K,D) bytes; when given (and non-zero sized) the
 * D axis is split over workgroups to fill the chip at small B; NULL = single pass.                  */
int64_t cf_gmm_ws_bytes(int B, int M, int K, int D);
int cf_gmm_logprob(const float* x, const float* a, const float* nm, const float* cst, float* out, void* ws,
                   int B, int M, int K, int D, int64_t x_bstride, int accumulate, cf_stream_t stream);

/* All mixtures of one flow in two launches (small batches, where a launch costs more than the arithmetic: the Split priors
 * of flowsequential.py:18-27 + the final prior).  Level l < n <= 4: x[l] (B rows of D[l] floats, row stride x_bstride[l])
 * with its prepared tables a[l], nm[l] (M*K, D[l]) and cst[l] (M*K).
 *   out[b,m] = (ldM[b,m] +) sum_l log p_l(x_l[b] | m) (+ ld1[b])        ldM, ld1: optional (NULL)
 * - the chain of cf_gmm_logprob(..., accumulate) calls followed by cf_logdet_combine, bit for bit (same D splits, same
 * order of sums).  Requires M*K <= 256, D[l] % 4 == 0, x_bstride[l] % 4 == 0, 16-byte aligned x / a / nm.
 * ws: cf_gmm_levels_ws_bytes(n, D, B, M, K) bytes (always > 0; -1 for n outside 1..4).                                     */
int64_t cf_gmm_levels_ws_bytes(int n, const int* D, int B, int M, int K);
int cf_gmm_logprob_levels(int n, const float* const* x, const float* const* a, const float* const* nm,
                          const float* const* cst, const int* D, const int64_t* x_bstride, const float* ldM,
                          const float* ld1, float* out, void* ws, int B, int M, int K, cf_stream_t stream);

/* Mixtures whose context shifts are embedding lookups (model.py:157,162; gaussian.py:142-158): a sample's parameters are
 * one of Um x Us sets - means mG + shift[km] (rounded once, as `self.mG + cond_mean`), scales softplus(sG + shift[ks]).
 * With the samples bucketed by (ks, km) the 128 samples of a workgroup share their rows: the register-tiled kernel of
 * cf_gmm_logprob with gathered x rows (7x the per-sample form of cf_gmm_ctx_logprob_tab at saturating batches).
 * x: B rows of D floats (D = channels x pixels); a_tab (Us, M*K, D) = 1/sigma; nm_tab (Um, M*K, D) = -(mu + shift);
 * cst_tab (Us, M*K); key_s (B) int32: scale key per sample; order (B) int32: sample indices grouped by (ks, km);
 * tiles (T, 4) int32 rows [ks, km, first position in order, count <= 128 (0 = unused tile)].
 * out[b,m] (+)= logsumexp_k(cst_tab[key_s[b]] - 0.5 sum_d ((x + nm) a)^2).  16 < M*K <= 256, 80 % K == 0, D % 4 == 0,
 * x_bstride % 4 == 0, 16-byte aligned x / tables / tiles.  ws: cf_gmm_keyed_ws_bytes(T, B, M, K, D) bytes.              */
int64_t cf_gmm_keyed_ws_bytes(int T, int B, int M, int K, int D);
int cf_gmm_logprob_keyed(const float* x, const float* a_tab, const float* nm_tab, const float* cst_tab, const int* key_s,
                         const int* tiles, const int* order, float* out, void* ws, int T, int B, int M, int K, int D,
                         int64_t x_bstride, int accumulate, cf_stream_t stream);

/* q[b, m*K+k] = sum_d ((x[b,d] + nm)*a)^2 only (B x M*K, dense): the backward pass rebuilds the responsibilities
 * softmax_k(cst - q/2) from it.                                                                        */
int cf_gmm_quad(const float* x, const float* a, const float* nm, float* q, int B, int M, int K, int D,
                int64_t x_bstride, cf_stream_t stream);
/* backward, first half in one call: r (B, M*K) = softmax_k(cst - q/2) * g[b, m] - the component responsibilities times the
 * upstream gradient g (B, M); the reduction over D is split over workgroups when B alone does not fill the chip.
 * ws: cf_gmm_resp_ws_bytes(...) bytes.                                                                          */
/* elementwise pieces of the mixture backward: A2 = a a, AB = a a nm (M*K, D);  gx = -(x G1 + G2) with G1 = r A2, G2 = r AB
 * (B, D);  g_mu / g_sG (M*K, D) from the batch sums S0 = sum_b r (M*K), S1 = r^T x, S2 = r^T x^2 (M*K, D):
 * g_mu = a^2 (S1 + nm S0), g_sG = a (a^2 (S2 + 2 nm S1 + nm^2 S0) - S0) sigmoid(sG).                              */
int cf_gmm_bwd_coeffs(const float* a, const float* nm, float* A2, float* AB, int MK, int D, int transposed,
                      cf_stream_t stream);      /* transposed != 0: A2, AB are written as (D, M*K): the Wt operand of cf_linear */
int cf_gmm_bwd_gx(const float* x, const float* G1, const float* G2, float* gx, int B, int D, int64_t x_bstride,
                  cf_stream_t stream);
int cf_gmm_bwd_params(const float* a, const float* nm, const float* sG, const float* S0, const float* S1, const float* S2,
                      float* gmu, float* gsig, int MK, int D, cf_stream_t stream);
/* Parameter sums of the mixture backward in one product: S0[mk] = sum_b r[b,mk], S1[mk,d] = sum_b r[b,mk] x[b,d],
 * S2[mk,d] = sum_b r[b,mk] x[b,d]^2 (r (B, MK) dense from cf_gmm_resp; x: B rows of D floats, row stride x_bstride < 2^20).
 * MK = 80 and D % 32 == 0 (cf_gmm_bwd_sums_supported; other shapes: cf_linear_wgrad / cf_linear_wgrad_x2).  Batch slices
 * are summed in slice order - no float atomics.  ws: cf_gmm_bwd_sums_ws_bytes(B, MK, D) bytes.                            */
int cf_gmm_bwd_sums_supported(int MK, int D);
int64_t cf_gmm_bwd_sums_ws_bytes(int B, int MK, int D);
int cf_gmm_bwd_sums(const float* x, const float* r, float* S0, float* S1, float* S2, void* ws, int B, int MK, int D,
                    int64_t x_bstride, cf_stream_t stream);
/* the same + the mixture-weight gradient gw (M, K) = S0 - gcol softmax(wG) (gaussian.py:149-153: the log-weights enter through
 * log_softmax; gcol (M) = column sums of the upstream gradient (B, M)) in the same launch */
int cf_gmm_bwd_params_w(const float* a, const float* nm, const float* sG, const float* S0, const float* S1, const float* S2,
                        const float* wG, const float* gcol, float* gmu, float* gsig, float* gw, int M, int K, int D,
                        cf_stream_t stream);
int64_t cf_gmm_resp_ws_bytes(int B, int M, int K, int D);
int cf_gmm_resp(const float* x, const float* a, const float* nm, const float* cst, const float* g, float* r, void* ws, int B,
                int M, int K, int D, int64_t x_bstride, cf_stream_t stream);
/* prior sampling (gaussian.py:163-169): out[n,:] = mG[rows[n],:] + softplus(sG[rows[n],:]) * eps[n,:];
 * rows[n] = m*K + k_n (int64, component drawn by the caller), eps ~ N(0,1) supplied by the caller.      */
int cf_gmm_sample(const float* mG, const float* sG, const int64_t* rows, const float* eps, float* out, int N, int D,
                  cf_stream_t stream);

/* ---- fused flow step: Conv1x1 -> ActNorm -> Coupling(conv net) in ONE kernel, fp32 MFMA ---------
 * (model.py:129-147 per-step triple; coupling.py:26-29 net; conv1x1.py:52-57; actnorm.py:53-60)
 * Supported (C,H,W): (8,16,16) (16,16,16) (32,8,8) (64,4,4) with 3x3 reflect conv; others return
 * CF_ERR_UNSUPPORTED from cf_flow_step_supported().                                                  */
int cf_flow_step_supported(int C, int H, int W, int kh, int kw);
int64_t cf_flow_step_ws_bytes(int C, int H, int W);
/* pack the step's parameters into MFMA-fragment order (device side, every call, no host sync):
 * folds ActNorm into the 1x1 matrix, computes ldj_const[0] = H*W*log|det Wm| + sum_c logs.          */
int cf_flow_step_prepare(const float* Wm, const float* t, const float* logs,
                         const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* w3, const float* b3, void* ws, int C, int H, int W, cf_stream_t stream);
/* training form: the same packing, and Wm^-1 (C, C; the gradient of log|det Wm| is Wm^-T, conv1x1.py:52-57) from the one
 * factorisation the prepare step runs anyway */
int cf_flow_step_prepare_train(const float* Wm, const float* t, const float* logs, const float* w1, const float* b1,
                               const float* w2, const float* b2, const float* w3, const float* b3, void* ws, float* winv,
                               int C, int H, int W, cf_stream_t stream);
/* The tables of n flow steps of ONE shape (e.g. the four steps of a resolution level) in one factorisation launch + one packing
 * launch instead of 2 n (batches of 256: a training step re-packs every step's tables per update, and 36 serial launches of
 * 15-68 us were a fifth of a cifar10 step).  HOST arrays of n device pointers each; winv NULL or n pointers (training).
 * cf_flow_step_bwd_prepare_batch: the same for the backward kernel's transposed fragments.                                 */
int cf_flow_step_prepare_batch(int n, const float* const* Wm, const float* const* t, const float* const* logs, const float* const* w1,
                               const float* const* b1, const float* const* w2, const float* const* b2, const float* const* w3,
                               const float* const* b3, void* const* ws, float* const* winv, int C, int H, int W, cf_stream_t stream);
int cf_flow_step_bwd_prepare_batch(int n, const float* const* Wm, const float* const* logs, const float* const* w1,
                                   const float* const* w2, const float* const* w3, void* const* wsb, int C, int H, int W,
                                   cf_stream_t stream);

/* The 16x16 level's 3x3 as exact bf16-piece MFMAs (default: the environment variable CONTEXTFLOW_BF16_SPLIT=1|2,
 * else off): 1 = the Winograd-domain products on bf16 pieces, 2 = the DIRECT 3x3 with h1 split once by its producer (from 1024
 * samples per launch).  on = 0 / 1 / 2 sets it for the tables packed and the steps launched from now on, on < 0 queries; returns the
 * setting in force.  Tables packed while it was off do not hold the weight pieces: prepare again after switching on.  Same fp32
 * results as the default (fp32 Winograd) kernels to their rounding; measured: no gain for a whole flow (power), see DESIGN.md. */
int cf_bf16_split(int on);

/* z = step(x); ldj_acc[b] += ldj_const + sum log_s   (ldj_acc is the running per-sample log-det).
 * in_squeeze != 0: x is the UN-squeezed (B, C/4, 2H, 2W) tensor and Squeeze((2,2)) (squeeze.py:10-11)
 * is folded into the kernel's operand addressing (no separate index kernel, no extra HBM pass).
 * The 3x3 of the coupling net runs in Winograd F(2x2,3x3) form (fp32; weights G w G^T packed by
 * cf_flow_step_prepare) on 16x16 images and from 1024 (8x8) / 2048 (4x4) samples per call, in direct form
 * otherwise - same results to fp32 rounding; the environment variable CONTEXTFLOW_DIRECT_CONV=1 keeps
 * the direct form everywhere.  The same holds for _fwd_taped, _fwd_ctx and _inv below.                 */
int cf_flow_step_fwd(const float* x, float* z, float* ldj_acc, const void* ws, int B, int C, int H, int W,
                     int64_t x_bstride, int in_squeeze, cf_stream_t stream);
/* Chained form for small batches (ABI 7): n <= 4 consecutive flow steps of ONE shape in ONE launch - a workgroup owns whole samples
 * end to end, so the steps after the first run in place on z behind a workgroup barrier.  ws: HOST array of the n packed tables
 * (cf_flow_step_prepare); in_squeeze applies to the first step; z holds the output of the last step.  Only for batches at which
 * cf_flow_step_fwd takes its small-batch kernels (B <= cf_flow_step_chain_max_batch(C, H, W); 0 = never): the numbers are bit for
 * bit those of n single calls.  At a batch of 256 a forward is launch-bound: 12 step launches become 3.                      */
int cf_flow_step_chain_max_batch(int C, int H, int W);
int cf_flow_step_fwd_chain(const float* x, float* z, float* ldj_acc, const void* const* ws, int n, int B, int C, int H, int W,
                           int64_t x_bstride, int in_squeeze, cf_stream_t stream);


/* the fused step with a per-sample bias from the specialist coupling's CN net (coupling.py:39-47):
 * mode 1: sbias (B,C) added to the conditioner output (contextflow); mode 2: sbias (B,2C) added before the
 * first ReLU (CN(c) concatenated to the conditioner input).  mode | 4: keep the direct form of the 3x3 at every batch
 * size (the training forward under contextflow: cf_flow_step_bwd_ctx rebuilds the conditioner in that form).       */
int cf_flow_step_fwd_ctx(const float* x, float* z, float* ldj_acc, const void* ws, const float* sbias, int mode, int B, int C,
                         int H, int W, int64_t x_bstride, cf_stream_t stream);

/* inverse of the fused step (coupling.py:68-73, actnorm.py:78, conv1x1.py:72): x = step^-1(z) in ONE kernel.
 * `ws` is the forward table of cf_flow_step_prepare (the conditioner is the same); `wsi` holds
 * Wm^-1 diag(e^{logs}) in fragment order and Wm^-1 t, built by cf_flow_step_inv_prepare.               */
int64_t cf_flow_step_inv_ws_bytes(int C, int H, int W);
int cf_flow_step_inv_prepare(const float* Wm, const float* t, const float* logs, void* wsi, int C, int H, int W,
                             cf_stream_t stream);
/* x_unsqueezed != 0 (ABI 7): the step sits behind a Squeeze((2,2)) in the flow; x is written in the (B, C/4, 2H, 2W) layout of the
 * tensor BEFORE that Squeeze (squeeze.py:13-14), i.e. Squeeze.reverse is folded into the kernel's stores.                 */
int cf_flow_step_inv(const float* z, float* x, const void* ws, const void* wsi, int B, int C, int H, int W,
                     int64_t z_bstride, int x_unsqueezed, cf_stream_t stream);

/* ---- backward of the fused step (training: experiment_cl.py:130-136) ------------------------------
 * cf_flow_step_bwd_prepare packs the TRANSPOSED weight fragments of the data-gradient chain (workspace of
 * cf_flow_step_bwd_ws_bytes bytes).  The generalist's backward is the taped pair below; a caller that kept only the step
 * input re-runs cf_flow_step_fwd_taped into a scratch tape at backward time (same kernel => the very masks of the
 * forward).  gz: dL/dz (B,C,H,W) dense; gld: dL/d(log-det) (B,).                                               */
int64_t cf_flow_step_bwd_ws_bytes(int C, int H, int W);
int cf_flow_step_bwd_prepare(const float* Wm, const float* logs, const float* w1, const float* w2, const float* w3,
                             void* wsb, int C, int H, int W, cf_stream_t stream);

/* Multiply-adds per sample that the matrix pipe EXECUTES for one step at batch size B (the dispatch picks the Winograd
 * form of the 3x3 - 16 instead of 36 C^2 HW - by shape and batch size): pass 0 = cf_flow_step_fwd, 1 = cf_flow_step_fwd_taped,
 * 2 = cf_flow_step_bwd_taped, 3 = cf_flow_step_inv; cf_step_wgrads_macs: the four weight gradients of cf_step_wgrads.  The ALGORITHMIC count
 * (the reference's direct convolutions) is 40 C^2 HW for each of them.  Host-only, no stream.           */
int64_t cf_flow_step_macs(int B, int C, int H, int W, int pass);
int64_t cf_step_wgrads_macs(int B, int C, int H, int W);

/* Taped training pair (experiment_cl.py:130-136: forward + cost.backward()).
 * cf_flow_step_fwd_taped = cf_flow_step_fwd that also writes the tape of the step:
 *   t_y0 (B, C/2, H*W), t_h1, t_h2 (B, 2C, H*W; post-ReLU planes of Coupling.NN, coupling.py:26-27): operands of cf_wgrad;
 *   t_aux (cf_flow_step_tape_aux_bytes(B, C, H, W) bytes, 16-byte aligned; opaque): log-scale and second output half of
 *   Conv1x1+ActNorm per (sample, channel, pixel) and the two ReLU masks as bit words.
 * cf_flow_step_bwd_taped = the data-gradient chain of the step from dL/dz (gz), dL/d ld1 (gld) and t_aux alone - no step
 * input, no recompute; writes dL/dx (gx, in the step's (B, C, H, W) layout) and the gradient planes s_gh (B, C, H*W),
 * s_gh2, s_gh1 (B, 2C, H*W), s_gy (B, C, H*W) that cf_wgrad contracts with t_h2 / t_h1 / t_y0 / the step input. */
int64_t cf_flow_step_tape_aux_bytes(int B, int C, int H, int W);
int cf_flow_step_fwd_taped(const float* x, float* z, float* ldj_acc, const void* ws, float* t_y0, float* t_h1, float* t_h2,
                           void* t_aux, int B, int C, int H, int W, int64_t x_bstride, int in_squeeze, cf_stream_t stream);
/* specialist coupling without contextflow (coupling.py:45-47: CN(c) concatenated to the conditioner input = per-sample
 * bias sbias (B, 2C) before the first ReLU): training forward with the same tape; backward = cf_flow_step_bwd_taped. */
int cf_flow_step_fwd_ctx_taped(const float* x, float* z, float* ldj_acc, const void* ws, const float* sbias, float* t_y0,
                               float* t_h1, float* t_h2, void* t_aux, int B, int C, int H, int W, int64_t x_bstride,
                               cf_stream_t stream);
/* gx_unsqueezed != 0: dL/dx is written in the layout of the tensor BEFORE the Squeeze((2,2)) in front of the step,
 * (B, C/4, 2H, 2W) - the index map of squeeze.py:10-11 folded into the kernel's stores (C % 4 == 0).                       */
int cf_flow_step_bwd_taped(const float* gz, const float* gld, const void* wsb, const void* t_aux, float* gx, float* s_gh,
                           float* s_gh2, float* s_gh1, float* s_gy, int B, int C, int H, int W, int gx_unsqueezed,
                           cf_stream_t stream);

/* Conv1x1 / ActNorm parameter gradients of a fused step from the gradients of its folded matrix / bias (gWp (C,C) and
 * gbp (C) = the wgrad of the g_y plane against the step input): gNN = diag(s) gWp + G H W Wm^-T, gt = -s gbp,
 * glogs = -rowsum(gWp o diag(s) Wm) + gbp t s + G, s = exp(-logs), G = gld_sum[0] = sum_b d/d ld1[b] (device scalar). */
int cf_step_param_grads(const float* gWp, const float* gbp, const float* Wm, const float* t, const float* logs,
                        const float* winv, const float* gld_sum, int HW, float* gNN, float* gt, float* glogs, int C,
                        cf_stream_t stream);
/* n flow steps of one width (the steps of a resolution level) in ONE launch: every operand is a host array of n device pointers,
 * gld_sum is shared (small batches: the parameter work of a backward pass is a chain of launches of a few microseconds
 * each - layers/autograd.py::_step_param_part_batch).  Bitwise equal to n cf_step_param_grads calls.                        */
int cf_step_param_grads_batch(int n, const float* const* gWp, const float* const* gbp, const float* const* Wm, const float* const* t,
                              const float* const* logs, const float* const* winv, const float* gld_sum, int HW, float* const* gNN,
                              float* const* gt, float* const* glogs, int C, cf_stream_t stream);

/* weight gradient as a split-K MFMA GEMM over (sample, pixel):
 *   gw[t][m][n] = sum_{b,p} A[b][m][p] * Bm[b][n][src_t(p)],  t < taps (1, or 9 = 3x3 reflect-shifted pixels)
 *   gbias[m]    = sum_{b,p} A[b][m][p]                         (optional)
 * A: (B, MR, H*W), Bm: (B, NR, H*W) dense (planes written by cf_flow_step_bwd_taped / the step tape); gw: (taps, MR, NR) fp32.
 * ws: cf_wgrad_ws_bytes(...) bytes for the split-K partials, summed in a fixed order (reproducible).
 * H x W in {16x16, 8x8, 4x4}, NR <= 128.                                                               */
int64_t cf_wgrad_ws_bytes(int B, int MR, int NR, int H, int W, int taps);
int cf_wgrad(const float* A, const float* Bm, float* gw, float* gbias, void* ws, int B, int MR, int NR, int H, int W,
             int taps, cf_stream_t stream);

/* The four weight gradients of one flow step in one call (layers/autograd.py::step_backward; reference: autograd through
 * Conv1x1 / ActNorm / Coupling.NN, coupling.py:26-28, conv1x1.py:52-57): NN.4 = s_gh x t_h2 -> gw3 (1, C, 2C), gb3 (C);
 * NN.2 (3x3) = s_gh2 x t_h1 -> gw2 (2C, 2C, 3, 3: the reference's weight layout), gb2 (2C); NN.0 = s_gh1 x t_y0 -> gw1 (1, 2C, C/2), gb1 (2C); folded
 * Conv1x1 / ActNorm matrix = s_gy x xs -> gwp (1, C, C), gbp (C).  Planes as written by cf_flow_step_bwd[_taped] /
 * cf_flow_step_fwd_taped.  xs = the step input, read in place: B samples at a stride of xs_bstride floats, each
 * (C, H*W) - or, with xs_unsqueezed != 0, the (C/4, 2H, 2W) tensor in front of the step's Squeeze((2,2)), read through the
 * index map of squeeze.py:10-11 (no squeezed / contiguous copy is made).  Four split-K launches + ONE reduce launch; results
 * bitwise equal to four cf_wgrad calls on dense planes. */
int64_t cf_step_wgrads_ws_bytes(int B, int C, int H, int W);
int cf_step_wgrads(const float* s_gh, const float* s_gh2, const float* s_gh1, const float* s_gy, const float* t_h2,
                   const float* t_h1, const float* t_y0, const float* xs, float* gw3, float* gb3, float* gw2, float* gb2,
                   float* gw1, float* gb1, float* gwp, float* gbp, void* ws, int B, int C, int H, int W, int64_t xs_bstride,
                   int xs_unsqueezed, cf_stream_t stream);
/* The same for n flow steps of one shape (host arrays of n device pointers / strides / flags; ws[i]: cf_step_wgrads_ws_bytes
 * each): per product ONE launch over all steps (the Conv1x1 product once per distinct (xs_bstride, xs_unsqueezed): the first
 * step of a level reads the tensor in front of its Squeeze) and ONE reduce launch - 5 or 6 launches instead of 5 n.  Bitwise
 * equal to n cf_step_wgrads calls.                                                                                          */
int cf_step_wgrads_batch(int n, const float* const* s_gh, const float* const* s_gh2, const float* const* s_gh1,
                         const float* const* s_gy, const float* const* t_h2, const float* const* t_h1, const float* const* t_y0,
                         const float* const* xs, float* const* gw3, float* const* gb3, float* const* gw2, float* const* gb2,
                         float* const* gw1, float* const* gb1, float* const* gwp, float* const* gbp, void* const* ws, int B, int C,
                         int H, int W, const int64_t* xs_bstride, const int* xs_unsqueezed, cf_stream_t stream);

/* ---- SimpleViT conditioner of TransCoupling (layers/simple_vit.py:18-127, coupling.py:100-159) --- */
/* y[r,n] = act(sum_k x[r,k] Wt[n,k] + bias[n]) + res[r,n]; fp32 MFMA; bias/res may be NULL; any K, N.
 * act: 0 none, 1 exact (erf) GELU, 2 ReLU.   nn.Linear of patch embedding / to_qkv / to_out / FeedForward; CN nets. */
int cf_linear(const float* x, const float* Wt, const float* bias, const float* res, float* y,
              int rows, int K, int N, int act, cf_stream_t stream);
/* data gradient of cf_linear: y[r,n] = sum_k x[r,k] W[k,n] with W the layer's weight as stored ((K, N) row-major here =
 * nn.Linear's (out, in)): gx = gy W without a transposed copy of the weight. */
int cf_linear_tn(const float* x, const float* W, float* y, int rows, int K, int N, cf_stream_t stream);
/* n Linears over the same rows in ONE launch (the CN nets of a specialist flow: reference conv1x1.py:31-50, actnorm.py:40-51,
 * coupling.py:30-47 evaluate them layer by layer): y_g = act_g(x_g W_g^T + b_g), x_g (rows, K), W_g (N_g, K), b_g (N_g) or NULL,
 * act_g 0 | 2 (ReLU); host arrays of n device pointers / ints.  With ctx != NULL the inputs are formed from the integer context
 * (rows, nctx) while they are staged - the uniform dequantisation of the context encoders (model.py:30-90, dequantize.py:55-64):
 * x_g[r,k] = (code(ctx[r], k) + u_g[r,k]) / qbins_g[k]; x[] then holds the uniforms u_g (rows, K), q[] the qbins (K); onehot != 0: code
 * = concatenated one-hot code with cardinalities card (device int64[nctx]), else the context itself (K == nctx).  c_out (NULL, or n
 * pointers of which any may be NULL): the formed codes x_g (rows, K) are stored there too - the training forward keeps them. */
int cf_linear_group(int n, const float* const* x, const float* const* q, const float* const* W, const float* const* b, float* const* y,
                    float* const* c_out, const int* N, const int* act, const int64_t* ctx, const int64_t* card, int nctx, int onehot,
                    int rows, int K, cf_stream_t stream);
/* backward of cf_linear w.r.t. its parameters: gW (N,K) = gy^T x, gb (N) = column sums of gy (gb may be NULL); split-K
 * fp32-MFMA GEMM over the rows, partials in ws (cf_linear_wgrad_ws_bytes) summed in a fixed order.  Any K, N (wide
 * problems run as column blocks of at most 128 outputs x 255 inputs).  cf_linear_wgrad_x2: the same with x squared
 * element by element while it is staged (gW = gy^T x^2: second-moment sums of the mixture backward).                */
int64_t cf_linear_wgrad_ws_bytes(int rows, int K, int N);
int cf_linear_wgrad(const float* x, const float* gy, float* gW, float* gb, void* ws, int rows, int K, int N,
                    cf_stream_t stream);
int cf_linear_wgrad_x2(const float* x, const float* gy, float* gW, void* ws, int rows, int K, int N, cf_stream_t stream);
/* A group of up to 32 such weight gradients in ONE launch pair (the Linears of a transformer flow step are far too small
 * to fill the chip one by one): member i: gW[i] (N[i], K[i]) = gy[i]^T x[i] over rows[i] rows, gb[i] (N[i]; may be NULL) =
 * column sums of gy[i]; x[i] (rows[i], K[i]) and gy[i] (rows[i], N[i]) dense.  N <= 192, K + 1 <= (12 - ceil(N/32)) * 32.
 * The pointer / size arrays live on the HOST; ws: cf_linear_wgrad_group_ws_bytes bytes on the device.                  */
int64_t cf_linear_wgrad_group_ws_bytes(const int* rows, const int* K, const int* N, int n);
int cf_linear_wgrad_group(const float* const* x, const float* const* gy, float* const* gW, float* const* gb, const int* rows,
                          const int* K, const int* N, int n, void* ws, cf_stream_t stream);
/* y[r,:] = LayerNorm(x[r,:])*w + b (+ pos[r % ntok,:] if pos != NULL); biased variance, eps.          */
int cf_layernorm(const float* x, const float* w, const float* b, const float* pos, float* y,
                 int rows, int dim, int ntok, float eps, cf_stream_t stream);
/* single-head attention over N tokens per sample; qkv rows = [q|k|v] (3*dh): out = softmax(qk^T*scale)v */
int cf_attention(const float* qkv, float* out, int B, int N, int dh, float scale, cf_stream_t stream);
/* inverse=0: tokens[b,(h w),(p1 p2 c)] <- image[b,c,(h p1),(w p2)] (image batch stride img_bstride);
 * inverse=1: image <- tokens.  C,H,W describe the image.                                              */
int cf_patchify(const float* src, float* dst, int B, int C, int H, int W, int p1, int p2,
                int64_t img_bstride, int inverse, cf_stream_t stream);

/* ---- fused TransCoupling: patchify -> SimpleViT -> un-patchify -> affine map -> log-det in ONE kernel ---
 * (layers/coupling.py:100-159 + layers/simple_vit.py:18-127).  Limits: heads 1, dim_head 64,
 * dim = C*p1*p2 <= 64, patch_dim = (C/2)*p1*p2 <= 64, tokens per sample a power of two <= 32.        */
int cf_vit_supported(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads);
int64_t cf_vit_ws_bytes(int patch_dim, int dim, int depth);
/* number of floats of the flat parameter vector cf_vit_prepare expects, in this order:
 *   patch LN w,b | patch Linear W (dim x patch_dim), b | embed LN w,b |
 *   depth x [attn LN w,b | to_qkv W (192 x dim) | to_out W (dim x 64) | ff LN w,b | ff W1 (dim x dim), b1 | W2, b2] |
 *   final LN w,b                                                                                      */
int64_t cf_vit_flat_params(int patch_dim, int dim, int depth);
/* pack the flat parameters into MFMA-fragment order (device side, every call).                        */
int cf_vit_prepare(const float* flat_params, void* ws, int patch_dim, int dim, int depth, cf_stream_t stream);
/* x: (B,C,H,W) batch stride x_bstride; z: (B,C,H,W) dense; pos: (tokens, dim) sin/cos table.
 * inverse=0: z = [x0 | x1*exp(log_s)+t], ldj[b] = sum log_s (assigned); inverse=1: z = [x0 | (x1-t)/exp(log_s)] */
int cf_vit_coupling(const float* x, float* z, float* ldj, const void* ws, const float* pos, int B, int C, int H, int W,
                    int p1, int p2, int dim, int depth, int64_t x_bstride, int inverse, cf_stream_t stream);

/* One transformer-coupling flow step (model.py:129-147 with --coupling trans: Conv1x1 -> ActNorm -> TransCoupling) as ONE
 * kernel, activations register-resident between x and z (csrc/cf_vit_step.hip).  Geometry: C = 26 channels on 8 x 1
 * windows, patch (2,1), dim = 2C, one head of 64 (SMAP; cf_vit_step_supported).  prepare: Wm (C,C), ActNorm t / logs (C),
 * the ViT parameters flattened in the order of cf_vit_prepare, pos (4, dim); ws: cf_vit_step_ws_bytes(C, depth) bytes.
 * fwd: z (B,C,8,1); ldj_acc[b] += 8 log|det Wm| + sum logs + sum log_s (conv1x1.py:53, actnorm.py:58, coupling.py:151);
 * h_out (optional, may be NULL): the conditioner's output [t | raw], (B,C,8,1).                                         */
int cf_vit_step_supported(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads);
int64_t cf_vit_step_ws_bytes(int C, int depth);
int cf_vit_step_prepare(const float* Wm, const float* t, const float* logs, const float* flat_vit_params, const float* pos,
                        void* ws, int C, int depth, cf_stream_t stream);
int cf_vit_step_fwd(const float* x, float* z, float* ldj_acc, const void* ws, float* h_out, int B, int C, int depth,
                    int64_t x_bstride, cf_stream_t stream);
/* Multiply-adds PER SAMPLE of one cf_vit_step_fwd launch: what = 0 the reference's count (Conv1x1 + every Linear of the
 * SimpleViT + q.k^T / p.v, SURVEY.md 8d: 472 384 for C = 26, depth 6); 1 = what the matrix pipe executes, padding rows of
 * its 32-row tiles included (1 326 v_mfma_f32_32x32x2_f32 per wave of 8 samples); 2 = the useful part of that (the fused
 * 52 x 52 products of the round-4 kernel: q.k through Wq^T Wk, value / output through Wout Wv).  Host-only, no stream. */
int64_t cf_vit_step_macs(int C, int depth, int what);


/* The same step for SMALL batches (the reference's operating point is 256 samples, config.py:10), row-split: the four waves
 * of a workgroup share 16 token columns (4 samples) and split the output rows of every Linear (v_mfma_f32_16x16x4_f32,
 * planes through LDS; csrc/cf_vit_rs.hip) - an eighth of the serial chain of cf_vit_step_fwd per workgroup, at 8x the
 * workgroups.  Its own workspace layout (LayerNorm weights folded into the packed Linears); same arguments and results. */
int cf_vit_step_rs_supported(int C, int H, int W, int p1, int p2, int dim, int dim_head, int heads);
int64_t cf_vit_step_rs_ws_bytes(int C, int depth);
int cf_vit_step_rs_prepare(const float* Wm, const float* t, const float* logs, const float* flat_vit_params, const float* pos,
                           void* ws, int C, int depth, cf_stream_t stream);
/* The row-split tables of n flow steps in one factorisation, one fuse and one packing launch (HOST arrays of n device
 * pointers; pos is shared).  winv: NULL or n (C, C) outputs for Wm^-1; wsb: NULL or n backward workspaces
 * (cf_vit_step_bwd_ws_bytes) that also receive the backward kernel's tables (cf_vit_step_bwd_prepare_batch).            */
int cf_vit_step_rs_prepare_batch(int n, const float* const* Wm, const float* const* t, const float* const* logs,
                                 const float* const* flat_vit_params, const float* pos, void* const* ws, float* const* winv,
                                 void* const* wsb, int C, int depth, cf_stream_t stream);
int cf_vit_step_bwd_prepare_batch(int n, const float* const* Wm, const float* const* logs, const float* const* flat_vit_params,
                                  void* const* wsb, int C, int depth, cf_stream_t stream);

int cf_vit_step_rs_fwd(const float* x, float* z, float* ldj_acc, const void* ws, float* h_out, int B, int C, int depth,
                       int64_t x_bstride, cf_stream_t stream);

/* Backward of that step as ONE kernel (training step of the anomaly-detection flows, experiment_ad.py:204-213;
 * csrc/cf_vit_rs_bwd.hip): from the step INPUT x, dL/dz (gz, dense (B,C,8,1)) and dL/d(log-det) (gld (B,)) it re-runs the
 * step in the row-split form, walks back and writes dL/dx (gx, dense) plus
 *   planes: the operands of every weight gradient as token-major matrices, Bp = B rounded up to 4, R = 4 Bp token rows,
 *           P = 8 Bp position rows, in floats from the start:
 *             [ x^T (P,26) | g_y (P,26) | u0 (R,26) | g_e (R,52) | per layer l: u1 (R,52) | g_qkv (R,192) | o (R,64) | g_xmid (R,52) |
 *               u2 (R,52) | g_hpre (R,52) | h (R,52) | g_xout (R,52) ]
 *           i.e. gW(Conv1x1+ActNorm folded) = g_y^T x^T, gW(embed) = g_e^T u0, gW(qkv) = g_qkv^T u1, gW(out) = g_xmid^T o,
 *           gW(fc1) = g_hpre^T u2, gW(fc2) = g_xout^T h (cf_linear_wgrad_group contracts them in one launch);
 *   ln_partials: ceil(B/4) rows of cf_vit_step_bwd_ln_floats / ceil(B/4) floats: per-workgroup sums of the LayerNorm
 *           weight / bias gradients: [LN(pd): g 32 | b 32][LN(dim) of the embedding: g 64 | b 64][per layer: attention norm
 *           g 64 | b 64 | feed-forward norm g 64 | b 64][transformer.norm: g 64 | b 64]; the column sums are the gradients.
 * ws: cf_vit_step_rs_prepare; wsb: cf_vit_step_bwd_prepare (cf_vit_step_bwd_ws_bytes).  depth <= 6.                    */
int64_t cf_vit_step_bwd_ws_bytes(int C, int depth);
int64_t cf_vit_step_bwd_plane_floats(int B, int C, int depth);
int64_t cf_vit_step_bwd_ln_floats(int B, int C, int depth);
int cf_vit_step_bwd_prepare(const float* Wm, const float* logs, const float* flat_vit_params, void* wsb, int C, int depth,
                            cf_stream_t stream);
int cf_vit_step_bwd(const float* x, const float* gz, const float* gld, float* gx, const void* ws, const void* wsb, float* planes,
                    float* ln_partials, int B, int C, int depth, int64_t x_bstride, cf_stream_t stream);

/* Taped pair for saturating batches (the register-resident forward): cf_vit_step_fwd_taped = cf_vit_step_fwd that also writes
 * the residual stream of the conditioner at its depth + 1 layer boundaries to xtape - cf_vit_step_tape_floats(B, C, depth)
 * floats, feature-major [boundary][2C][T] (slot 0, the embedding output, is left unwritten: the backward rebuilds it), T = cf_vit_step_tape_tokens(B) = 4 x (B rounded up to 32), token = 4 sample + n
 * (5.8 KB per sample at C = 26, depth 6).  cf_vit_step_bwd_taped = cf_vit_step_bwd that starts from that tape: the Conv1x1 /
 * ActNorm / patch embedding are re-run (their statistics are needed on the way back), the transformer layers are not - a
 * quarter of the kernel's work.  Same outputs to fp32 rounding.                                                          */
int64_t cf_vit_step_tape_tokens(int B);
int64_t cf_vit_step_tape_floats(int B, int C, int depth);
int cf_vit_step_fwd_taped(const float* x, float* z, float* ldj_acc, const void* ws, float* xtape, int B, int C, int depth,
                          int64_t x_bstride, cf_stream_t stream);
int cf_vit_step_rs_fwd_taped(const float* x, float* z, float* ldj_acc, const void* ws, float* xtape, int B, int C, int depth,
                             int64_t x_bstride, cf_stream_t stream);
/* n <= cf_vit_step_rs_chain_max_steps() CONSECUTIVE transformer flow steps in one launch (evaluation at small batches; ws: host
 * array of n packed tables; a workgroup owns its samples end to end, steps 2.. run in place on z).  Bitwise equal to n
 * cf_vit_step_rs_fwd calls.                                                                                                  */
int cf_vit_step_rs_chain_max_steps(void);
int cf_vit_step_rs_fwd_chain(const float* x, float* z, float* ldj_acc, const void* const* ws, int n, int B, int C, int depth,
                             int64_t x_bstride, cf_stream_t stream);       /* the row-split (small-batch) forward, same tape */
int cf_vit_step_bwd_taped(const float* x, const float* gz, const float* gld, float* gx, const void* ws, const void* wsb,
                          float* planes, float* ln_partials, const float* xtape, int B, int C, int depth, int64_t x_bstride,
                          cf_stream_t stream);

/* ---- SplineActivation: monotone rational-quadratic spline, linear tails (layers/activations.py:120-211,
 * layers/splines/rational_quadratic.py:21-176) ----------------------------------------------------- */
/* knot tables: P parameter sets (1 = shared weights, C*H*W = individual_weights) of K bins;
 * uw, uh: (P,K), ud: (P,K-1); table: cf_spline_table_floats(P,K) floats.                               */
int64_t cf_spline_table_floats(int P, int K);
int cf_spline_prepare(const float* uw, const float* uh, const float* ud, float* table, int P, int K, float tail_bound,
                      cf_stream_t stream);
/* inverse=0: y = spline(x), ldj[b] = sum log|dy/dx| ; inverse=1: y = spline^-1(x).  x,y: (B,N) dense.  */
int cf_spline(const float* x, const float* table, float* y, float* ldj, int B, int N, int P, int K, float tail_bound,
              int inverse, cf_stream_t stream);

/* ---- backward of the layer-by-layer path (training step, experiment_ad.py:207-213: loss.backward()) ---------
 * SimpleViT conditioner of TransCoupling (simple_vit.py:30-127) and the generic ActNorm / affine coupling map.
 * The Linear layers' backward GEMMs (gX = gY W, gW = gY^T X) are plain library GEMMs on the host side.        */
/* LayerNorm backward: gx (rows,dim); partial[cf_layernorm_bwd_parts()][2*dim] = per-workgroup sums of
 * (gy*xhat | gy) in a fixed order - the host adds the rows to get d/dweight, d/dbias.  dim <= 128.            */
int cf_layernorm_bwd_parts(void);
int cf_layernorm_bwd(const float* x, const float* w, const float* gy, float* gx, float* partial, int rows, int dim,
                     float eps, cf_stream_t stream);
/* single-head attention backward (simple_vit.py:56-68): qkv (B*N, 3*dh) rows [q|k|v], go (B*N, dh) -> gqkv   */
int cf_attention_bwd(const float* qkv, const float* go, float* gqkv, int B, int N, int dh, float scale,
                     cf_stream_t stream);
/* exact GELU (nn.GELU default, simple_vit.py:36): backward=0: out = gelu(x); backward=1: out = gy * gelu'(x)  */
int cf_gelu(const float* x, const float* gy, float* out, int64_t n, int backward, cf_stream_t stream);
/* affine coupling map backward (coupling.py:52-66): x (B,C,HW) by stride, h = [t|raw] dense, gz by stride,
 * gld (B) = d/d ldj -> gx = [gz0 | gz1*s] dense, gh = [d/dt | d/draw] dense                                   */
int cf_coupling_apply_bwd(const float* x, const float* h, const float* gz, const float* gld, float* gx, float* gh, int B,
                          int C, int HW, int64_t x_bstride, int64_t gz_bstride, cf_stream_t stream);
/* out[c] = sum_{b,p} a[b,c,p]; out[C+c] = sum_{b,p} a*b2 (b2 may be null): ActNorm's d/dt, d/dlogs reductions
 * (actnorm.py:53-60); two passes over batch slices, fixed order; ws: cf_channel_sums_ws_bytes(B, C) bytes         */
int64_t cf_channel_sums_ws_bytes(int B, int C);
int cf_channel_sums(const float* a, const float* b2, float* out, void* ws, int B, int C, int HW, int64_t a_bstride,
                    int64_t b_bstride, cf_stream_t stream);

/* ---- specialist (context-conditioned) branches: SURVEY 8(f) rank 2 ------------------------------------------
 * ContextEncoder (model.py:30-90) = OneHotEncoder | EyeEncoder (rtdl/nn/_embeddings.py:76-150) followed by
 * UniformCatDequantization (dequantize.py:55-64): out (B,width) = (code(ctx) + u) / qbins.  ctx (B,nctx) int64,
 * card (nctx) int64 cardinalities (one-hot only).  onehot = 2: ArgmaxCatDequantization (dequantize.py:236-262):
 * card = bits per variable, out = u * (2 bit - 1) over the big-endian binary codes (zero pad column if odd).  */
int cf_ctx_encode(const int64_t* ctx, const float* u, const float* qbins, const int64_t* card, float* out, int B, int nctx,
                  int width, int onehot, cf_stream_t stream);
/* Conv1x1 with a context net (conv1x1.py:34-50): m (B, C*C) = CN(c); W_b = tril(m,-1) + diag(exp(diag m))
 * [+ Wm - I when Wm != NULL: contextflow]; z[b] = W_b x[b]; ldj[b] = H*W*sum(diag m).  C <= 64.                */
int cf_conv1x1_ctx(const float* x, const float* m, const float* Wm, float* z, float* ldj, int B, int C, int HW,
                   int64_t x_bstride, cf_stream_t stream);
/* Conv1x1 + ActNorm of a specialist step in one pass (evaluation; conv1x1.py:34-50, actnorm.py:40-60): m1 (B, C*C) =
 * Conv1x1.CN(c), m2 (B, 2C) = ActNorm.CN(c'); Wm (C, C) / t, logs (C): the shared parameters under contextflow, else NULL.
 *   z[b] = (W_b x[b] - t_b) exp(-logs_b);   ldj[b] (+)= H W (sum diag m1[b] + lad[0]) + sum_c logs_b + cadd
 * (the two layers' own log-dets, reference quirks included: H W log|det NN| for Conv1x1, no H W factor for ActNorm).
 * lad: device scalar log|det NN| or NULL; cadd: a host constant (the encoders' constant log-densities).
 * in_squeeze != 0: x is the un-squeezed (B, C/4, 2H, 2W) tensor, Squeeze((2,2)) folded into the reads.                
 * m1_blocked != 0 (offered where cf_affine_ctx_blocked_floats(C, H, W) > 0: the image flows' three levels): m1 is
 * (B, cf_affine_ctx_blocked_floats) - only the 16 x 16 blocks on and below the diagonal of the per-sample matrix, block
 * (rt, g <= rt) at (rt (rt + 1) / 2 + g) * 256, row-major inside; the caller evaluates Conv1x1.CN with its rows permuted
 * to that order (layers/specialist.py::_cn_blocked): 10 / 16 of the bytes at C = 64, every fragment load contiguous.  */
int cf_affine_ctx_blocked_floats(int C, int H, int W);
int cf_affine_ctx_fwd(const float* x, const float* m1, const float* Wm, const float* m2, const float* t, const float* logs,
                      const float* lad, float cadd, float* z, float* ldj, int B, int C, int H, int W, int64_t x_bstride,
                      int in_squeeze, int accumulate, int m1_blocked, cf_stream_t stream);
/* ActNorm with a context net (actnorm.py:40-60): m (B, 2C) = CN(c) = [t_b | logs_b] (+ t, logs when non-NULL:
 * contextflow); z = (x - t_b) exp(-logs_b); ldj[b] = sum_c logs_b.                                             */
int cf_actnorm_ctx(const float* x, const float* m, const float* t, const float* logs, float* z, float* ldj, int B, int C,
                   int HW, int64_t x_bstride, cf_stream_t stream);
/* h[b,c,:] = act(h[b,c,:] + bias[b,c]) in place (the CN(c) term of the coupling net, coupling.py:44-47)        */
int cf_add_sample_bias(float* h, const float* bias, int B, int C, int HW, int relu, cf_stream_t stream);
/* GMM log_prob with per-sample shifts c (B,2,M,K,D) of means / pre-softplus scales (gaussian.py:142-158);
 * mG, sG (M,K,D,HW) raw parameters, logw = log_softmax(wG); out (B,M) assigned or accumulated.
 * lp_out (optional, (B, M*K)): the per-component log-joints, for cf_gmm_ctx_bwd of the same forward.            */
int cf_gmm_ctx_logprob(const float* x, const float* mG, const float* sG, const float* logw, const float* c, float* out,
                       float* lp_out, int B, int M, int K, int D, int HW, int64_t x_bstride, int accumulate,
                       cf_stream_t stream);
/* Table form for scale shifts that take only U distinct values (the priors of create_model look their shifts up in
 * embedding tables, model.py:157,162): cf_gmm_ctx_tables turns cs_tab (U, M*K, D) into inv_sig = 1/softplus(sG + cs_u),
 * dsig = softplus'(sG + cs_u) (optional; backward) - both (U, M*K, D*HW) - and lsum (U, M*K) = sum log softplus(..).
 * cf_gmm_ctx_logprob_tab = cf_gmm_ctx_logprob with key (B) int32 = u of every sample; c still supplies the per-sample
 * mean shifts (its scale half is not read) - or, with ckey (B) != NULL, c is itself a table (Um, M*K, D) of the distinct
 * mean shifts and ckey the row of every sample (no per-sample gather of embedding rows at all).  Same results, no
 * transcendental per term.                                                                                       */
int cf_gmm_ctx_tables(const float* sG, const float* cs_tab, float* inv_sig, float* dsig, float* lsum, int U, int MK, int D,
                      int HW, cf_stream_t stream);
int cf_gmm_ctx_logprob_tab(const float* x, const float* mG, const float* inv_sig, const float* lsum, const float* logw,
                           const float* c, const int* ckey, const int* key, float* out, float* lp_out, int B, int M, int K,
                           int D, int HW, int64_t x_bstride, int accumulate, cf_stream_t stream);

/* pieces of the variational context encoder (model.py:52-79, dequantize.py:104-118):
 * ConditionalGaussianDistribution.sample (gaussian.py:263-270): c (B,2D) = [mean|log_scale], eps (B,D) ->
 * x = mean + exp(log_scale) eps, logp (B); Sigmoid layer (activations.py:234-238): y, ldj (B).                 */
int cf_cond_gauss_sample(const float* c, const float* eps, float* x, float* logp, int B, int D, cf_stream_t stream);
int cf_sigmoid_ldj(const float* x, float* y, float* ldj, int B, int D, cf_stream_t stream);

/* ---- elementwise flow activations (layers/activations.py:34-118, 213-245) -------------------------------------------
 * x, y: (rows, D); forward: y = f(x), ldj[row] = sum_d log|f'(x)|; inverse: y = f^-1(x) (ldj unused, may be NULL).
 * mode 0 Identity | 1 LeakyRelu(a) | 2 SmoothLeakyRelu(a) | 3 SmoothTanh(a, b) | 4 Sigmoid(temperature a, eps b: the
 * clamp of the inverse) | 5 LearnableLeakyRelu (slope = sigmoid(*slope_logit) + 0.5, read on the device).  The smooth
 * activations invert by the reference's Newton iteration (100 steps, derivative clamped at 1e-2).                   */
int cf_activation(const float* x, float* y, float* ldj, int64_t rows, int D, int mode, float a, float b,
                  const float* slope_logit, int inverse, cf_stream_t stream);

/* ---- specialist training (contextflow: the CN nets and the priors' embedding tables train, the generalist's own
 * parameters stay frozen - coupling.py:36, conv1x1.py:27, actnorm.py:23, gaussian.py:134) -------------------------
 * backward of cf_flow_step_fwd_ctx mode 1: as cf_flow_step_bwd, the recompute adds sbias (B,C) to the conditioner
 * output; d/d sbias[b,c] = sum_p s_gh[b,c,p] (cf_sample_channel_sums).  Only gx and s_gh are written: the weights
 * are frozen in this mode, the other operand planes may be NULL.                                                 */
int cf_flow_step_bwd_ctx(const float* x, const float* gz, const float* gld, const void* ws, const void* wsb,
                         const float* sbias, float* gx, float* s_y0, float* s_h1, float* s_h2, float* s_gh, float* s_gh2,
                         float* s_gh1, float* s_gy, int B, int C, int H, int W, int64_t x_bstride, cf_stream_t stream);
/* Conv1x1 / ActNorm with a context net, backward: gx (B,C,HW) dense and gm = d/d CN(c) output ((B,C*C) / (B,2C)).
 * gld (B) = d/d of the layer's per-sample log-det.                                                              */
int cf_conv1x1_ctx_bwd(const float* x, const float* m, const float* Wm, const float* gz, const float* gld, float* gx,
                       float* gm, int B, int C, int HW, int64_t x_bstride, int64_t gz_bstride, cf_stream_t stream);
int cf_actnorm_ctx_bwd(const float* x, const float* m, const float* t, const float* logs, const float* gz, const float* gld,
                       float* gx, float* gm, int B, int C, int HW, int64_t x_bstride, int64_t gz_bstride, cf_stream_t stream);
/* out[b,c] = sum_p a[b,c,p] ; out = gy * (x > 0)                                                               */
int cf_sample_channel_sums(const float* a, float* out, int B, int C, int HW, cf_stream_t stream);
int cf_relu_bwd(const float* x, const float* gy, float* out, int64_t n, cf_stream_t stream);
/* context-shifted GMM, backward w.r.t. x (B,D,HW) and the per-sample shifts c (B,2,M,K,D); g (B,M) upstream;
 * lp (optional, (B, M*K)): lp_out of the forward - NULL = the log-joints are recomputed.                         */
int cf_gmm_ctx_bwd(const float* x, const float* mG, const float* sG, const float* logw, const float* c, const float* g,
                   const float* lp, float* gx, float* gc, int B, int M, int K, int D, int HW, int64_t x_bstride,
                   cf_stream_t stream);
int cf_gmm_ctx_bwd_tab(const float* x, const float* mG, const float* inv_sig, const float* dsig, const float* lsum,
                       const float* logw, const float* c, const int* key, const float* g, const float* lp, float* gx,
                       float* gc, int B, int M, int K, int D, int HW, int64_t x_bstride, cf_stream_t stream);
/* specialists trained WITHOUT contextflow keep the prior's own parameters trainable (gaussian.py:130-137): per-slab
 * partial sums pgm / pgs (ceil(B/slab), M*K, D*HW) of d/d mG and d/d sG; r (B, M*K) = upstream gradient x responsibility
 * (softmax_k of the kept log-joints); the caller sums the slabs in order.                                          */
int cf_gmm_ctx_pgrad_tab(const float* x, const float* mG, const float* inv_sig, const float* dsig, const float* c,
                          const int* key, const float* r, float* pgm, float* pgs, int B, int M, int K, int D, int HW,
                          int64_t x_bstride, int slab, cf_stream_t stream);

/* h[b,c2,:] += x[b, c2 % C, :] in place: identity branch of MaskedResidualBlock2d (`--coupling maf`,
 * layers/autoregressive/masked_conv_2d.py:93-98)                                                                */
int cf_add_repeat(float* h, const float* x, int B, int C2, int C, int HW, cf_stream_t stream);

/* ---- log-det bookkeeping (layers/flowsequential.py:18-27) --------------------------------------- */
/* out[b,m] = ldM[b,m] + ld1[b]                                                                      */
int cf_logdet_combine(const float* ldM, const float* ld1, float* out, int B, int M, cf_stream_t stream);
/* acc[0] += sum_b logsumexp_m logp[b,m] (fp64 accumulator; the scalar each rank all-reduces)         */
int cf_nll_sum(const float* logp, double* acc, int B, int M, cf_stream_t stream);

/* ---- optimizer update of the training step (model.py:289 optim.AdamW; experiment_cl.py:136, experiment_ad.py:213) ---------- */
/* torch.optim.AdamW's update (decoupled weight decay, lerp form of the first moment, bias-corrected; amsgrad off) of n tensors
 * in ceil(n / 72) launches: host arrays of n device pointers (p, g, exp_avg m, exp_avg_sq v) and element counts; `step` = device
 * scalar holding the update count t >= 1 of THIS update (the caller increments it: the call is capturable).  fp32 tensors;
 * the hyper-parameters arrive as doubles and 1 - beta, 1 - lr weight_decay are formed in double, as torch forms them.        */
int cf_adamw_step_batch(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel,
                        const float* step, double lr, double beta1, double beta2, double eps, double weight_decay, int maximize,
                        cf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CONTEXTFLOW_HIP_H */
