// ubench_mfma_slot.hip -- what stops v_min3 from hiding under v_mfma_f32_32x32x16_bf16 in
// the NN filter's inner loop?  Variants of one "step" (4 chains x [M1, M2] + 32 v_min3):
//   V0  min3 on unrelated registers, MFMAs back to back            (tools/ubench_mfma_bf16.hip pattern)
//   V1  min3 on unrelated registers, interleaved slot order  8 min3 | M2(r-1) | M1(r)
//   V2  as V1 but min3 reads the accumulators (the real dependence)
//   V3  as V2 with one chain per slot reduced as 4 independent min chains
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_mfma_slot.hip -o tools/ubench_mfma_slot
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int ITERS = 1024;

template <int V>
__global__ __launch_bounds__(256) void k(float *out, int rnd)
{
    f32x16 acc[4];
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; i++) acc[i] = z;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 1e-3f + i;
    s16x8 as, bs;
    for (int i = 0; i < 8; i++) {
        // rnd: operands with random mantissas / signs / exponents (data-dependent power), else constants
        const unsigned hsh = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x * 97u) >> 7;
        as[i] = rnd ? (short)(((hsh & 0x80ff) | 0x3f00) ^ ((hsh >> 3) & 0x0080)) : (short)(0x3f80 + (threadIdx.x & 15));
        bs[i] = rnd ? (short)((((hsh >> 9) & 0x80ff) | 0x3e80)) : (short)0x3f80;
    }
    bf16x8 a8 = __builtin_bit_cast(bf16x8, as);
    bf16x8 bq[4];
    for (int r = 0; r < 4; r++) { bs[0] = (short)(bs[0] + r); bq[r] = __builtin_bit_cast(bf16x8, bs); }
    float m[4][4];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) m[r][c] = 1e30f;
    float a = 1.0f + threadIdx.x, b = 2.0f;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("" : "+v"(a8), "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));      // opaque: the MFMAs are not loop invariant
        if (V == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bq[r], z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bq[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 32; f++) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[f & 15]) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                __builtin_amdgcn_sched_barrier(0);
                if (V == 1) {
#pragma unroll
                    for (int f = 0; f < 8; f++) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[f & 7]) : "v"(a), "v"(b));
                } else if (V == 2) {
#pragma unroll
                    for (int e = 0; e < 16; e += 2) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m[r][0]) : "v"(acc[r][e]), "v"(acc[r][e + 1]));
                } else {
#pragma unroll
                    for (int e = 0; e < 16; e += 2) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m[r][(e >> 1) & 3]) : "v"(acc[r][e]), "v"(acc[r][e + 1]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (r > 0) acc[r - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bq[r - 1], acc[r - 1], 0, 0, 0);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bq[r], z, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bq[3], acc[3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    for (int i = 0; i < 16; i++) s += v[i];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) s += m[r][c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
static void run(int waves_per_simd, float *out, int rnd)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), 0, 0, out, rnd);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), 0, 0, out, rnd);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = ms * 1e-3 * 2.4e9 / ITERS;
    printf("V%d rnd=%d waves/SIMD=%d : %8.1f cycles per step per SIMD (8 MFMA + 32 min3 per wave-step; %.1f per wave-step)\n", V, rnd,
           waves_per_simd, cyc, cyc / waves_per_simd);
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    for (int rnd : {0, 1})
        for (int w : {1, 2, 4}) { run<0>(w, out, rnd); run<3>(w, out, rnd); }
    return 0;
}
