// ubench_mfma_bf16.hip -- two questions about v_mfma_f32_32x32x16_bf16 on gfx950:
//  (A) do VALU instructions (v_min3_f32) issue under it (unlike under the f32 MFMA,
//      tools/ubench_mfma.hip)?
//  (B) how accurate is its K = 16 accumulation?  Products of bf16 are exact in f32;
//      the test feeds terms with heavy cancellation and compares D with the exactly
//      rounded sum (fp64; 17 terms of <= 16+8 significant bits are exact in fp64 when
//      their exponents span < 29 bits, which the generators guarantee).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_bf16.hip -o tools/ubench_mfma_bf16
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int ITERS = 2048;

template <int F, int MF>
__global__ __launch_bounds__(256) void thr(float *out, float s0, float s1)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i;
    s16x8 as, bs;
    for (int i = 0; i < 8; i++) { as[i] = (short)(0x3f80 + threadIdx.x); bs[i] = (short)0x3f80; }
    const bf16x8 a8 = __builtin_bit_cast(bf16x8, as), b8 = __builtin_bit_cast(bf16x8, bs);
    float a = s0 + threadIdx.x, b = s1;
    for (int it = 0; it < ITERS; it++) {
        if (MF) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int f = 0; f < F; f++) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[f & 7]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int F, int MF>
static void run(int waves_per_simd, float *out)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((thr<F, MF>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, 2.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((thr<F, MF>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, 2.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = ms * 1e-3 * 2.4e9 / ITERS;
    printf("bf16 mfma=%d min3 F=%2d waves/SIMD=%d : %8.1f cycles/iter/SIMD (%.1f per wave-iter)\n", MF * 4, F,
           waves_per_simd, cyc, cyc / waves_per_simd);
}

// ---- accuracy ----
// One wave: A[32][16], B[16][32] bf16, C[32][32] f32 -> D.  Lane l supplies
// A[l&31][8*(l>>5) .. +7] and B[8*(l>>5) .. +7][l&31]; D[i][j]: lane j + 32*((i>>2)&1), reg (i>>3)*4 + (i&3).
__global__ void acc_kernel(const uint16_t *A, const uint16_t *B, const float *C, float *D)
{
    const int l = threadIdx.x;
    s16x8 as, bs;
    for (int k = 0; k < 8; k++) {
        as[k] = (short)A[(l & 31) * 16 + 8 * (l >> 5) + k];
        bs[k] = (short)B[(8 * (l >> 5) + k) * 32 + (l & 31)];
    }
    f32x16 c;
    for (int r = 0; r < 16; r++) {
        const int i = (r >> 2) * 8 + (l >> 5) * 4 + (r & 3);
        c[r] = C[i * 32 + (l & 31)];
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, as), __builtin_bit_cast(bf16x8, bs), c, 0, 0, 0);
    for (int r = 0; r < 16; r++) {
        const int i = (r >> 2) * 8 + (l >> 5) * 4 + (r & 3);
        D[i * 32 + (l & 31)] = c[r];
    }
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main()
{
    float *out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 2, 4}) {
        run<0, 1>(w, out);
        run<8, 0>(w, out);
        run<8, 1>(w, out);
        run<16, 1>(w, out);
        run<32, 1>(w, out);
        run<64, 1>(w, out);
    }

    uint16_t *dA, *dB;
    float *dC, *dD;
    CHECK(hipMalloc(&dA, 32 * 16 * 2));
    CHECK(hipMalloc(&dB, 16 * 32 * 2));
    CHECK(hipMalloc(&dC, 32 * 32 * 4));
    CHECK(hipMalloc(&dD, 32 * 32 * 4));
    std::vector<uint16_t> A(32 * 16), B(16 * 32);
    std::vector<float> C(32 * 32), D(32 * 32);
    srand(1234);
    auto rnd = [] { return (double)rand() / RAND_MAX * 2.0 - 1.0; };
    // err / (u * sum|terms|), err / ulp-of-result-at-max-partial: worst over trials, per generator
    for (int gen = 0; gen < 4; gen++) {
        double worst_rel = 0, worst_vs_rn = 0;
        long mism_rn = 0, total = 0;
        for (int trial = 0; trial < 400; trial++) {
            for (int i = 0; i < 32; i++)
                for (int k = 0; k < 16; k++) {
                    double v = rnd();
                    if (gen == 1) v *= ldexp(1.0, -(k % 3) * 8);          // split-like magnitudes 1, 2^-8, 2^-16
                    if (gen == 2) v *= ldexp(1.0, -(rand() % 20));        // wide exponent spread
                    if (gen == 3) v = (k & 1) ? -fabs(v) : fabs(v);       // alternating signs: cancellation
                    A[i * 16 + k] = f2bf((float)v);
                }
            for (int k = 0; k < 16; k++)
                for (int j = 0; j < 32; j++) {
                    double v = rnd();
                    if (gen == 1) v *= ldexp(1.0, -((k / 3) % 3) * 8);
                    if (gen == 2) v *= ldexp(1.0, -(rand() % 20));
                    if (gen == 3) v = fabs(v);
                    B[k * 32 + j] = f2bf((float)v);
                }
            for (int i = 0; i < 32 * 32; i++) C[i] = (float)(rnd() * (gen == 3 ? 4.0 : 1.0));
            CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            CHECK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < 32; i++)
                for (int j = 0; j < 32; j++) {
                    double s = C[i * 32 + j], sabs = fabs(s);
                    for (int k = 0; k < 16; k++) {
                        const double p = (double)bf2f(A[i * 16 + k]) * (double)bf2f(B[k * 32 + j]);
                        s += p;
                        sabs += fabs(p);
                    }
                    const double err = fabs((double)D[i * 32 + j] - s);
                    const double rel = err / (5.9604644775390625e-8 * sabs);
                    if (rel > worst_rel) worst_rel = rel;
                    const float rn = (float)s;       // correctly rounded exact sum
                    if (rn != D[i * 32 + j]) mism_rn++;
                    const double ulp = fabs((double)nextafterf(rn, INFINITY) - (double)rn);
                    if (err / ulp > worst_vs_rn) worst_vs_rn = err / ulp;
                    total++;
                }
        }
        printf("accuracy gen %d: worst |err| = %.3f x 2^-24 x sum|terms|, worst %.3f ulp(result); %ld of %ld differ from RN(exact sum)\n",
               gen, worst_rel, worst_vs_rn, mism_rn, total);
    }
    return 0;
}
