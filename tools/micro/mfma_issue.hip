// Microbenchmark: does one wave's VALU / SALU / LDS work overlap with its own MFMAs on gfx950?
// Loop body = 8 independent v_mfma_f32_32x32x2_f32 (64 cycles each) with NV v_add_f32, NS s_add_u32 and ND ds_read_b32
// slotted behind every MFMA.  Prints cycles per MFMA at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NS, int ND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    float d[4] = {0, 0, 0, 0};
    int sx = 0;
    const int addr = (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(t + j) & 7]) : "v"(b));
#pragma unroll
            for (int j = 0; j < NS; ++j) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
#pragma unroll
            for (int j = 0; j < ND; ++j) asm volatile("ds_read_b32 %0, %1" : "=v"(d[j & 3]) : "v"(addr));
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = sx;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += d[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NS, int ND>
void run(float* out, const char* name) {
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        dim3 grid(256 * wps), block(256);
        k<NV, NS, ND><<<grid, block>>>(out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<NV, NS, ND><<<grid, block>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves x iters x 8 MFMAs
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 8);
        printf("%-28s waves/SIMD=%d  %.1f cycles per MFMA (per-SIMD MFMA slot)\n", name, wps, cyc);
    }
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4 * sizeof(float));
    run<0, 0, 0>(out, "mfma only");
    run<1, 0, 0>(out, "+1 valu");
    run<2, 0, 0>(out, "+2 valu");
    run<4, 0, 0>(out, "+4 valu");
    run<8, 0, 0>(out, "+8 valu");
    run<0, 4, 0>(out, "+4 salu");
    run<0, 8, 0>(out, "+8 salu");
    run<0, 0, 1>(out, "+1 ds_read");
    run<0, 0, 2>(out, "+2 ds_read");
    run<0, 0, 4>(out, "+4 ds_read");
    run<4, 4, 1>(out, "+4 valu 4 salu 1 ds");
    return 0;
}
