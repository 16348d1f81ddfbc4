// Shared helpers for the gfx950 kernels of libcontextflow_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/contextflow_hip.h"

#define CF_WAVE 64

void cf_set_error(const char* fmt, ...);

#define CF_REQUIRE(cond)                                                        \
    do {                                                                        \
        if (!(cond)) {                                                          \
            cf_set_error("%s: requirement failed: %s", __func__, #cond);        \
            return CF_ERR_ARG;                                                  \
        }                                                                       \
    } while (0)

#define CF_LAUNCH_CHECK()                                                       \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            cf_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return (int)e_;                                                     \
        }                                                                       \
    } while (0)

static inline hipStream_t cf_s(cf_stream_t s) { return (hipStream_t)s; }

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel: the > 64 KiB dynamic-LDS opt-in is therefore
// taken once per (kernel, device), not once per process.  `done` is the caller's function-local bit set.
#include <atomic>
static inline int cf_raise_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done, const char* who) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { cf_set_error("%s: hipGetDevice: %s", who, hipGetErrorString(e)); return (int)e; }
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { cf_set_error("%s: cannot raise dynamic LDS to %d B: %s", who, bytes, hipGetErrorString(e)); return (int)e; }
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

// Hand-off of LDS data between the LANES OF ONE WAVE (a lane reads what another lane of its wave wrote, with no
// workgroup barrier in between).  The hardware keeps one wave's LDS instructions in order, but the COMPILER reasons per
// thread: a later ds_read whose address differs from this thread's own earlier ds_write may be hoisted above it (seen:
// hipcc moved phase-3 operand reads above the h2 plane stores of k_flow_step_small).  A wavefront-scope fence pins
// the order in the compiler; it emits no instruction of its own.
__device__ __forceinline__ void cf_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------------------
__device__ __forceinline__ float cf_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double cf_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float cf_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over a block of NW waves; result valid in every thread.  `scratch` holds >= NW floats.
template <int NW>
__device__ __forceinline__ float cf_block_sum(float v, float* scratch) {
    v = cf_wave_sum(v);
    if (NW == 1) return v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += scratch[i];
    return s;
}
