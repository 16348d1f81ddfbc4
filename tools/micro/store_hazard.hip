// Reproducer: on gfx950, is it safe to overwrite the data VGPRs of `buffer_store_dwordx4 v[a:a+3], voff, s[rsrc], s_off offen`
// with a VALU instruction issued right behind it?  (hipcc assumes yes for a store whose soffset is a register; the
// plane stores of cf_step_common.h showed the opposite: DESIGN.md section 4.)
// Every lane stores {1.0, 2.0, 3.0, 4.0} to its own 16 bytes and - in the SAME asm block, NOPS wait states later - moves
// 0xDEADBEEF into the first two data registers.  The host counts stored words that are not 1.0 / 2.0.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/micro/store_hazard.hip -o /tmp/store_hazard && /tmp/store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NOPS, bool SREG>
__global__ __launch_bounds__(256) void k(float* out, int soff_arg, int rounds) {
    const uint64_t p = reinterpret_cast<uint64_t>(out);
    const i32x4 rs = {(int)(uint32_t)p, (int)((uint32_t)(p >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
    for (int r = 0; r < rounds; ++r) {
        const int voff = (((r * gridDim.x + blockIdx.x) * 256 + threadIdx.x) * 16);
#define PRE "v_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 2.0\n\tv_mov_b32 v22, 3.0\n\tv_mov_b32 v23, 4.0\n\ts_nop 7\n\t"
#define POST "v_mov_b32 v20, 0xdeadbeef\n\tv_mov_b32 v21, 0xdeadbeef\n"
#define CLOB "memory", "v20", "v21", "v22", "v23"
        if (SREG) {
            if (NOPS == 0) asm volatile(PRE "buffer_store_dwordx4 v[20:23], %0, %1, %2 offen\n\t" POST : : "v"(voff), "s"(rs), "s"(soff_arg) : CLOB);
            else asm volatile(PRE "buffer_store_dwordx4 v[20:23], %0, %1, %2 offen\n\ts_nop 1\n\t" POST : : "v"(voff), "s"(rs), "s"(soff_arg) : CLOB);
        } else {
            if (NOPS == 0) asm volatile(PRE "buffer_store_dwordx4 v[20:23], %0, %1, 0 offen\n\t" POST : : "v"(voff), "s"(rs) : CLOB);
            else asm volatile(PRE "buffer_store_dwordx4 v[20:23], %0, %1, 0 offen\n\ts_nop 1\n\t" POST : : "v"(voff), "s"(rs) : CLOB);
        }
    }
}

template <int NOPS, bool SREG>
static void run(const char* what, float* dev, size_t n, int blocks, int rounds) {
    hipMemset(dev, 0, n * 4);
    k<NOPS, SREG><<<blocks, 256>>>(dev, 0, rounds);
    hipDeviceSynchronize();
    std::vector<float> h(n);
    hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; i += 4) bad += (h[i] != 1.0f) + (h[i + 1] != 2.0f) + (h[i + 2] != 3.0f) + (h[i + 3] != 4.0f);
    printf("%-58s corrupted words: %zu of %zu\n", what, bad, n);
}

int main() {
    const int blocks = 4096, rounds = 16;
    const size_t n = (size_t)blocks * 256 * 4 * rounds;
    float* dev;
    hipMalloc(&dev, n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        run<0, true>("register soffset, VALU write right behind the store:", dev, n, blocks, rounds);
        run<1, true>("register soffset, s_nop 1 between:", dev, n, blocks, rounds);
        run<0, false>("literal soffset 0, VALU write right behind the store:", dev, n, blocks, rounds);
        run<1, false>("literal soffset 0, s_nop 1 between:", dev, n, blocks, rounds);
    }
    return 0;
}
