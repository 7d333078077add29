// ubench_mfma.hip -- does VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950?
// One iteration = 4 independent MFMAs (256 SIMD cycles at the documented 64
// cycles each) plus F filler VALU instructions of one class on unrelated
// registers.  If fillers hide under the MFMAs the time per iteration stays at the
// MFMA time until F x issue cost exceeds it; if the f32 MFMA occupies the VALU the
// times add.  Run with 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma && tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2048;

// KIND 0: v_min3_f32, 1: v_fma_f32, 2: v_add_u32, 3: v_min_i32 (wait: v_min_i32), 4: v_cndmask
template <int KIND, int F, int MF>
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float v[8];
    int w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = threadIdx.x * 1e-3f + i; w[i] = threadIdx.x + i; }
    float a = s0 + threadIdx.x, b = s1;
    for (int it = 0; it < ITERS; it++) {
        if (MF) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int f = 0; f < F; f++) {
            if (KIND == 0) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[f & 7]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f & 7]) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[f & 7]) : "v"(w[(f + 1) & 7]));
            if (KIND == 3) asm volatile("v_min_i32 %0, %0, %1" : "+v"(w[f & 7]) : "v"(w[(f + 1) & 7]));
            if (KIND == 4) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(w[f & 7]) : "v"(w[(f + 1) & 7]), "v"(w[(f + 2) & 7]));
        }
    }
    float s = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    for (int i = 0; i < 8; i++) s += v[i] + w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int F, int MF>
static void run(const char *name, int waves_per_simd, float *out)
{
    const int blocks = 256 * waves_per_simd;     // 256 CUs, a block = one wave per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND, F, MF>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, 2.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, F, MF>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, 2.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // cycles per iteration per SIMD at 2.4 GHz, all resident waves together
    const double cyc = ms * 1e-3 * 2.4e9 / ITERS;
    printf("%-10s mfma=%d F=%2d waves/SIMD=%d : %8.1f cycles/iter/SIMD  (%.1f per wave-iter)\n", name, MF * 4, F,
           waves_per_simd, cyc, cyc / waves_per_simd);
}

template <int KIND>
static void sweep(const char *name, float *out)
{
    for (int w : {1, 2, 4}) {
        run<KIND, 0, 1>(name, w, out);
        run<KIND, 16, 0>(name, w, out);
        run<KIND, 16, 1>(name, w, out);
        run<KIND, 32, 1>(name, w, out);
        run<KIND, 64, 1>(name, w, out);
    }
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    sweep<0>("min3_f32", out);
    sweep<1>("fma_f32", out);
    sweep<2>("add_u32", out);
    sweep<3>("min_i32", out);
    sweep<4>("min3_i32", out);
    return 0;
}
