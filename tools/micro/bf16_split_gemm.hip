// Microbenchmark: exact-product fp32 GEMM on the bf16 matrix cores of gfx950.
//
// v_mfma_f32_*_f32 runs at the VECTOR rate (64 flop / clk / SIMD, 1/16 of the bf16 MFMA rate) and shares the FMA pipes with
// the vector instructions of its SIMD (tools/micro/mfma_issue.hip) - the step kernels of this repo are bound by exactly that
// sum.  An fp32 value is the exact sum of three bf16 values (8 + 8 + 8 significant bits, truncation split: and / sub / and /
// sub), and a product of two such pieces is exact in fp32, so  a * b = sum over the 9 pairs (i, j) of a_i * b_j  can be
// accumulated in fp32 by 9 bf16 MFMAs (NP = 9: every bit of every product, the error is the accumulation's alone, as for the
// f32 MFMA) or by the 6 pairs with i + j <= 2 (NP = 6: drops terms below 2^-24 |a b|).  At 16x the rate that is 9/16 (6/16)
// of the f32 MFMA time, on a pipe the vector instructions do NOT share.
//
// Shape of the test = the conditioner products of the flow steps: D (128 x N) = sum_r A_r (128 x 128) * B_r (128 x N), the
// B panel of a workgroup (128 x 128 columns) resident in LDS, R = 4 products per panel (B_r = the panel, re-split every
// time as a new activation plane would be), weights pre-split / pre-packed, as the prepare kernels would.
// Prints time, fp32-equivalent TFLOP/s and the error against an fp64 reference on sampled columns.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/bf16_split_gemm.hip -o /tmp/bf16_split_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int M = 128, K = 128, R = 4, PC = 128, LDB = PC;           // panel columns, LDS row stride (64 KB panel)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void stage_panel(float* lds, const float* __restrict__ Bm, int64_t N, int64_t n0) {
    for (int e = threadIdx.x; e < K * PC / 4; e += 256) {
        const int k = e / (PC / 4), c4 = (e % (PC / 4)) * 4;
        *reinterpret_cast<float4*>(&lds[k * LDB + c4]) = *reinterpret_cast<const float4*>(&Bm[(int64_t)k * N + n0 + c4]);
    }
    __syncthreads();
}

__device__ __forceinline__ void store_tiles(const f32x4 (&acc)[8][2], float* __restrict__ D, int64_t N, int64_t n0, int w, int col, int g) {
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) D[(int64_t)(16 * rt + 4 * g + i) * N + n0 + 32 * w + 16 * ct + col] = acc[rt][ct][i];
}

// ---- f32 MFMA: A packed [r][group of 4 k-steps][row tile][lane][4]
__global__ __launch_bounds__(256) void k_f32(const float* __restrict__ Ap, const float* __restrict__ Bm, float* __restrict__ D, int64_t N) {
    __shared__ __align__(16) float lds[K * LDB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * PC;
    stage_panel(lds, Bm, N, n0);
    f32x4 acc[8][2];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r)
#pragma unroll 2
        for (int sg = 0; sg < K / 16; ++sg) {
            float4 a[8];
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) a[rt] = *reinterpret_cast<const float4*>(&Ap[((((int64_t)r * (K / 16) + sg) * 8 + rt) * 64 + lane) * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float bv[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    bv[ct] = lds[(4 * (4 * sg + j) + g) * LDB + 32 * w + 16 * ct + col];
                    asm volatile("" : "+v"(bv[ct]));
                }
#pragma unroll
                for (int rt = 0; rt < 8; ++rt) {
                    const float av = j == 0 ? a[rt].x : j == 1 ? a[rt].y : j == 2 ? a[rt].z : a[rt].w;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[ct], acc[rt][ct], 0, 0, 0);
                }
            }
        }
    store_tiles(acc, D, N, n0, w, col, g);
}

// ---- bf16 split: A packed [r][k block of 32][row tile][piece][lane] 16 bytes = A_piece[16 rt + (lane & 15)][32 kb + 8 (lane >> 4) + 0..7]
// SPLITONLY: the splitting work without the MFMAs (what the vector pipes pay); NOSPLIT: the MFMAs on a fixed operand
template <int NP, int MODE>
__global__ __launch_bounds__(256) void k_bf16(const int4* __restrict__ Ap, const float* __restrict__ Bm, float* __restrict__ D, int64_t N) {
    __shared__ __align__(16) float lds[K * LDB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * PC;
    stage_panel(lds, Bm, N, n0);
    f32x4 acc[8][2];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    int4 bp[2][3] = {};
    for (int r = 0; r < R; ++r)
#pragma unroll 1
        for (int kb = 0; kb < K / 32; ++kb) {
            if (MODE < 2 || (r == 0 && kb == 0)) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    unsigned p0[8], p1[8], p2[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float x = lds[(32 * kb + 8 * g + j) * LDB + 32 * w + 16 * ct + col];
                        asm volatile("" : "+v"(x));
                        const unsigned u0 = __float_as_uint(x) & 0xffff0000u;
                        const float r1 = x - __uint_as_float(u0);
                        const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
                        const float r2 = r1 - __uint_as_float(u1);
                        p0[j] = u0; p1[j] = u1; p2[j] = __float_as_uint(r2);
                    }
                    // element j in bits 16 (j & 1) .. of dword j / 2: bytes [hi.3, hi.2, lo.3, lo.2]
                    auto pack = [](const unsigned (&p)[8]) {
                        return int4{(int)__builtin_amdgcn_perm(p[1], p[0], 0x07060302u), (int)__builtin_amdgcn_perm(p[3], p[2], 0x07060302u),
                                    (int)__builtin_amdgcn_perm(p[5], p[4], 0x07060302u), (int)__builtin_amdgcn_perm(p[7], p[6], 0x07060302u)};
                    };
                    bp[ct][0] = pack(p0); bp[ct][1] = pack(p1); bp[ct][2] = pack(p2);
                }
            }
            if (MODE == 1) {     // keep the split alive without MFMAs
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int pz = 0; pz < 3; ++pz) asm volatile("" :: "v"(bp[ct][pz].x), "v"(bp[ct][pz].y), "v"(bp[ct][pz].z), "v"(bp[ct][pz].w));
                continue;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {          // row tiles in two halves: 12 fragment registers x 4 live at a time
                int4 ap[4][3];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int pz = 0; pz < 3; ++pz) {       // MODE 3: the same (L1-resident) fragments every time - the matrix pipe alone
                        ap[q][pz] = Ap[((((int64_t)(MODE == 3 ? 0 : r) * (K / 32) + (MODE == 3 ? 0 : kb)) * 8 + 4 * h + q) * 3 + pz) * 64 + lane];
                        if (MODE == 3) asm volatile("" : "+v"(ap[q][pz].x));
                    }
                // small terms first: (2,2) (1,2) (2,1) | (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)
                constexpr int order[9][2] = {{2, 2}, {1, 2}, {2, 1}, {0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};
#pragma unroll
                for (int t = 9 - NP; t < 9; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[4 * h + q][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[q][order[t][0]]),
                                                                                         __builtin_bit_cast(bf16x8, bp[ct][order[t][1]]),
                                                                                         acc[4 * h + q][ct], 0, 0, 0);
            }
        }
    store_tiles(acc, D, N, n0, w, col, g);
}


// ---- bf16 split, the layout a step kernel would use: the producer of a plane (here: the staging pass) splits every element
// ONCE and leaves three bf16 planes [piece][column][k] in LDS (row stride 272 B: conflict-free 16-byte fragment reads); the four
// waves split the OUTPUT ROWS (2 row tiles each) and read every column's fragments with ds_read_b128 - no vector work in the loop
constexpr int LROW = K * 2 + 16;                                       // bytes per (piece, column) row
template <int NP>
__global__ __launch_bounds__(256) void k_bf16_lds(const int4* __restrict__ Ap, const float* __restrict__ Bm, float* __restrict__ D, int64_t N) {
    extern __shared__ __align__(16) unsigned char ldsb[];                // [3][PC][LROW]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, g = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * PC;
    f32x4 acc[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[q][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
        if (r) __syncthreads();
        // staging + split: thread -> (column, 8 consecutive k): 8 strided loads (k-major source), one 16-byte store per piece
        for (int e = threadIdx.x; e < PC * (K / 8); e += 256) {
            const int c = e % PC, k8 = e / PC;
            unsigned p0[8], p1[8], p2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = Bm[(int64_t)(8 * k8 + j) * N + n0 + c];
                const unsigned u0 = __float_as_uint(x) & 0xffff0000u;
                const float r1 = x - __uint_as_float(u0);
                const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(u1);
                p0[j] = u0; p1[j] = u1; p2[j] = __float_as_uint(r2);
            }
            auto pack = [](const unsigned (&p)[8]) {
                return int4{(int)__builtin_amdgcn_perm(p[1], p[0], 0x07060302u), (int)__builtin_amdgcn_perm(p[3], p[2], 0x07060302u),
                            (int)__builtin_amdgcn_perm(p[5], p[4], 0x07060302u), (int)__builtin_amdgcn_perm(p[7], p[6], 0x07060302u)};
            };
            *reinterpret_cast<int4*>(&ldsb[(0 * PC + c) * LROW + 16 * k8]) = pack(p0);
            *reinterpret_cast<int4*>(&ldsb[(1 * PC + c) * LROW + 16 * k8]) = pack(p1);
            *reinterpret_cast<int4*>(&ldsb[(2 * PC + c) * LROW + 16 * k8]) = pack(p2);
        }
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < K / 32; ++kb) {
            int4 ap[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int pz = 0; pz < 3; ++pz) ap[q][pz] = Ap[((((int64_t)r * (K / 32) + kb) * 8 + 2 * w + q) * 3 + pz) * 64 + lane];
            constexpr int order[9][2] = {{2, 2}, {1, 2}, {2, 1}, {0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                int4 bp[3];
#pragma unroll
                for (int pz = 0; pz < 3; ++pz) bp[pz] = *reinterpret_cast<const int4*>(&ldsb[(pz * PC + 16 * ct + col) * LROW + 64 * kb + 16 * g]);
#pragma unroll
                for (int t = 9 - NP; t < 9; ++t)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[q][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[q][order[t][0]]),
                                                                             __builtin_bit_cast(bf16x8, bp[order[t][1]]), acc[q][ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) D[(int64_t)(16 * (2 * w + q) + 4 * g + i) * N + n0 + 16 * ct + col] = acc[q][ct][i];
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : (1 << 19);
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A((size_t)R * M * K), B((size_t)K * N);
    for (auto& v : A) v = 0.1f * nd(rng);
    for (auto& v : B) v = nd(rng);
    // packed operands
    std::vector<float> Af((size_t)R * (K / 16) * 8 * 64 * 4);
    for (int r = 0; r < R; ++r) for (int sg = 0; sg < K / 16; ++sg) for (int rt = 0; rt < 8; ++rt) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j)
        Af[((((size_t)r * (K / 16) + sg) * 8 + rt) * 64 + l) * 4 + j] = A[((size_t)r * M + 16 * rt + (l & 15)) * K + 4 * (4 * sg + j) + (l >> 4)];
    std::vector<unsigned short> Ab((size_t)R * (K / 32) * 8 * 3 * 64 * 8);
    for (int r = 0; r < R; ++r) for (int kb = 0; kb < K / 32; ++kb) for (int rt = 0; rt < 8; ++rt) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
        const float x = A[((size_t)r * M + 16 * rt + (l & 15)) * K + 32 * kb + 8 * (l >> 4) + j];
        unsigned u; memcpy(&u, &x, 4);
        const unsigned u0 = u & 0xffff0000u; float f0; memcpy(&f0, &u0, 4);
        const float r1 = x - f0; unsigned v1; memcpy(&v1, &r1, 4);
        const unsigned u1 = v1 & 0xffff0000u; float f1; memcpy(&f1, &u1, 4);
        const float r2 = r1 - f1; unsigned u2; memcpy(&u2, &r2, 4);
        const unsigned pc[3] = {u0, u1, u2};
        for (int pz = 0; pz < 3; ++pz) Ab[((((((size_t)r * (K / 32) + kb) * 8 + rt) * 3 + pz) * 64 + l) * 8) + j] = (unsigned short)(pc[pz] >> 16);
    }
    float *dAf, *dB, *dD; int4* dAb;
    CK(hipMalloc(&dAf, Af.size() * 4)); CK(hipMalloc(&dAb, Ab.size() * 2)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, (size_t)M * N * 4));
    CK(hipMemcpy(dAf, Af.data(), Af.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dAb, Ab.data(), Ab.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    // fp64 reference on sampled columns
    const int NS = 64;
    std::vector<int64_t> cols(NS);
    for (int i = 0; i < NS; ++i) cols[i] = (int64_t)((double)i / NS * N) + (i % 7);
    std::vector<double> ref((size_t)NS * M), mag((size_t)NS * M);
    for (int i = 0; i < NS; ++i) for (int m = 0; m < M; ++m) {
        double s = 0, a = 0;
        for (int r = 0; r < R; ++r) for (int k = 0; k < K; ++k) { const double p = (double)A[((size_t)r * M + m) * K + k] * (double)B[(size_t)k * N + cols[i]]; s += p; a += fabs(p); }
        ref[(size_t)i * M + m] = s; mag[(size_t)i * M + m] = a;
    }
    std::vector<float> Dh((size_t)M * N);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, bool check) {
        CK(hipMemset(dD, 0xff, (size_t)M * N * 4));
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        double worst = 0, rms = 0;
        if (check) {
            CK(hipMemcpy(Dh.data(), dD, (size_t)M * N * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < NS; ++i) for (int m = 0; m < M; ++m) {
                const double e = fabs((double)Dh[(size_t)m * N + cols[i]] - ref[(size_t)i * M + m]) / mag[(size_t)i * M + m];
                worst = fmax(worst, e); rms += e * e;
            }
            rms = sqrt(rms / (NS * M));
        }
        printf("%-28s %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)", name, ms, 2.0 * M * K * R * (double)N / (ms * 1e-3) / 1e12);
        if (check) printf("   error / sum|a b|: max %.2e rms %.2e", worst, rms);
        printf("\n");
    };
    const dim3 grid((unsigned)(N / PC)), blk(256);
    run("f32 MFMA 16x16x4", [&] { k_f32<<<grid, blk>>>(dAf, dB, dD, N); }, true);
    run("bf16 split, 9 products", [&] { k_bf16<9, 0><<<grid, blk>>>(dAb, dB, dD, N); }, true);
    run("bf16 split, 6 products", [&] { k_bf16<6, 0><<<grid, blk>>>(dAb, dB, dD, N); }, true);
    run("bf16 split, 3 products", [&] { k_bf16<3, 0><<<grid, blk>>>(dAb, dB, dD, N); }, true);
    run("  split only (no MFMA)", [&] { k_bf16<9, 1><<<grid, blk>>>(dAb, dB, dD, N); }, false);
    run("  9 MFMAs, operand split once", [&] { k_bf16<9, 2><<<grid, blk>>>(dAb, dB, dD, N); }, false);
    run("  6 MFMAs, operand split once", [&] { k_bf16<6, 2><<<grid, blk>>>(dAb, dB, dD, N); }, false);
    run("  9 MFMAs, operands resident", [&] { k_bf16<9, 3><<<grid, blk>>>(dAb, dB, dD, N); }, false);
    run("  6 MFMAs, operands resident", [&] { k_bf16<6, 3><<<grid, blk>>>(dAb, dB, dD, N); }, false);
    const size_t lb = (size_t)3 * PC * LROW;
    CK(hipFuncSetAttribute((const void*)k_bf16_lds<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
    CK(hipFuncSetAttribute((const void*)k_bf16_lds<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
    run("pre-split LDS planes, 9", [&] { k_bf16_lds<9><<<grid, blk, lb>>>(dAb, dB, dD, N); }, true);
    run("pre-split LDS planes, 6", [&] { k_bf16_lds<6><<<grid, blk, lb>>>(dAb, dB, dD, N); }, true);
    return 0;
}
