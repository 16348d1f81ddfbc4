// AdamW update of the training step (model.py:289: optim.AdamW(model.parameters(), lr); experiment_cl.py:136 /
// experiment_ad.py:213: optimizer.step()) over MANY tensors in one launch.
//
// The flow's 135 (cifar10) ... 571 (smap) parameter tensors are tiny (9 ... 147 K elements): torch's fused multi-tensor kernel
// takes 4 (16) launches of 43 (15) us for them - 172 (250) us of a 1.44 (1.62) ms training step at the reference's batch of
// 256 - because a launch takes at most 48 tensors in 64 K-element chunks.  Here the tensor table travels in the kernel
// arguments (kAdamBatch entries per launch), a workgroup owns 1024 consecutive elements of one tensor (found by a binary search
// over the table's cumulative workgroup counts), 16-byte accesses: HBM-bound (7 floats per element), a few microseconds.
// Arithmetic = torch.optim.AdamW's, term by term, in fp32 (adamw.py::_single_tensor_adamw): decoupled weight decay, lerp form
// of the first moment, bias corrections from the step count read on the device (capturable: the count is a device scalar
// the caller increments before the launch).
#include "cf_common.h"
#include <math.h>

namespace {

constexpr int kAdamBatch = 72;
struct AdamBatch {
    float* p[kAdamBatch];
    const float* g[kAdamBatch];
    float* m[kAdamBatch];
    float* v[kAdamBatch];
    int n[kAdamBatch];
    int first[kAdamBatch + 1];           // first workgroup of tensor i; first[count] = workgroups of the launch
    int count;
};

// w1 = 1 - beta1, w2 = 1 - beta2, decay = 1 - lr weight_decay are formed by the host in double, as torch forms them; the bias
// corrections 1 - beta^t in double on the device (t is only known there), from ln(beta): one exp per thread and moment
__global__ __launch_bounds__(256) void k_adamw(const AdamBatch tb, const float* __restrict__ step, double lr, double lnb1, double lnb2,
                                               float beta2, float w1, float w2, float eps, float decay, int maximize) {
    const int wg = blockIdx.x;
    int lo = 0, hi = tb.count;           // tensor i with first[i] <= wg < first[i + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tb.first[mid] <= wg) lo = mid; else hi = mid;
    }
    const int i = lo, n = tb.n[i];
    float* __restrict__ p = tb.p[i]; const float* __restrict__ g = tb.g[i];
    float* __restrict__ m = tb.m[i]; float* __restrict__ v = tb.v[i];
    const double t = (double)step[0];
    const double bc1 = 1.0 - exp(t * lnb1), bc2 = 1.0 - exp(t * lnb2);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    auto upd = [&](float& pe, float ge, float& me, float& ve) {
        if (maximize) ge = -ge;
        pe *= decay;
        me = me + w1 * (ge - me);                                   // exp_avg.lerp_(grad, 1 - beta1)
        ve = ve * beta2 + w2 * ge * ge;                             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(ve) / bc2_sqrt + eps;
        pe = pe - step_size * (me / denom);                         // param.addcdiv_(exp_avg, denom, value = -step_size)
    };
    const int e0 = ((wg - tb.first[i]) * 256 + threadIdx.x) * 4;
    if (e0 >= n) return;
    const bool vec = e0 + 4 <= n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                                      reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (vec) {
        float4 pe = *reinterpret_cast<float4*>(p + e0), me = *reinterpret_cast<float4*>(m + e0), ve = *reinterpret_cast<float4*>(v + e0);
        const float4 ge = *reinterpret_cast<const float4*>(g + e0);
        upd(pe.x, ge.x, me.x, ve.x); upd(pe.y, ge.y, me.y, ve.y); upd(pe.z, ge.z, me.z, ve.z); upd(pe.w, ge.w, me.w, ve.w);
        *reinterpret_cast<float4*>(p + e0) = pe; *reinterpret_cast<float4*>(m + e0) = me; *reinterpret_cast<float4*>(v + e0) = ve;
    } else {
        for (int e = e0; e < n && e < e0 + 4; ++e) {
            float pe = p[e], me = m[e], ve = v[e];
            upd(pe, g[e], me, ve);
            p[e] = pe; m[e] = me; v[e] = ve;
        }
    }
}

}  // namespace

extern "C" {

int cf_adamw_step_batch(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel,
                        const float* step, double lr, double beta1, double beta2, double eps, double weight_decay, int maximize,
                        cf_stream_t stream) {
    CF_REQUIRE(n >= 0 && p && g && m && v && numel && step && lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.);
    int i = 0;
    while (i < n) {
        AdamBatch tb{};
        int wgs = 0, c = 0;
        for (; i < n && c < kAdamBatch; ++i) {
            CF_REQUIRE(numel[i] >= 0 && numel[i] < (1ll << 31));
            if (numel[i] == 0) continue;
            CF_REQUIRE(p[i] && g[i] && m[i] && v[i]);
            tb.p[c] = p[i]; tb.g[c] = g[i]; tb.m[c] = m[i]; tb.v[c] = v[i]; tb.n[c] = (int)numel[i];
            tb.first[c] = wgs;
            wgs += (int)((numel[i] + 1023) / 1024);
            ++c;
        }
        if (c == 0) break;
        tb.first[c] = wgs; tb.count = c;
        k_adamw<<<dim3(wgs), dim3(256), 0, cf_s(stream)>>>(tb, step, lr, beta1 > 0. ? log(beta1) : -1e300, beta2 > 0. ? log(beta2) : -1e300, (float)beta2,
                                                           (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(1.0 - lr * weight_decay), maximize);
        CF_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
