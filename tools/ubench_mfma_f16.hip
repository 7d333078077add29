// ubench_mfma_f16.hip -- accuracy of v_mfma_f32_32x32x16_f16 (K = 16 products of f16 pairs
// accumulated in f32) for the two-piece NN filter: worst error against the exact sum
// (fp64; products of f16 pairs are exact in f32) over generators with cancellation,
// split-like magnitudes (1, 2^-11) and f16 SUBNORMAL inputs (are they flushed?).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f16.hip -o tools/ubench_mfma_f16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// Lane l supplies A[l&31][8*(l>>5) .. +7] and B[8*(l>>5) .. +7][l&31]; D[i][j]: lane j + 32*((i>>2)&1), reg (i>>3)*4 + (i&3).
__global__ void acc_kernel(const uint16_t *A, const uint16_t *B, const float *C, float *D)
{
    const int l = threadIdx.x;
    s16x8 as, bs;
    for (int k = 0; k < 8; k++) {
        as[k] = (short)A[(l & 31) * 16 + 8 * (l >> 5) + k];
        bs[k] = (short)B[(8 * (l >> 5) + k) * 32 + (l & 31)];
    }
    f32x16 c;
    for (int r = 0; r < 16; r++) c[r] = C[((r >> 2) * 8 + (l >> 5) * 4 + (r & 3)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, as), __builtin_bit_cast(h16x8, bs), c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[((r >> 2) * 8 + (l >> 5) * 4 + (r & 3)) * 32 + (l & 31)] = c[r];
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

int main()
{
    uint16_t *dA, *dB;
    float *dC, *dD;
    CHECK(hipMalloc(&dA, 32 * 16 * 2));
    CHECK(hipMalloc(&dB, 16 * 32 * 2));
    CHECK(hipMalloc(&dC, 32 * 32 * 4));
    CHECK(hipMalloc(&dD, 32 * 32 * 4));
    std::vector<uint16_t> A(32 * 16), B(16 * 32);
    std::vector<float> C(32 * 32), D(32 * 32);
    srand(4321);
    auto rnd = [] { return (double)rand() / RAND_MAX * 2.0 - 1.0; };
    for (int gen = 0; gen < 5; gen++) {
        double worst_rel = 0;
        long mism_rn = 0, total = 0, flushed = 0;
        for (int trial = 0; trial < 400; trial++) {
            for (int i = 0; i < 32; i++)
                for (int k = 0; k < 16; k++) {
                    double v = rnd() * 1024.0;
                    if (gen == 1) v *= ldexp(1.0, -((k >> 1) & 1) * 11);       // h, h, l, l pattern
                    if (gen == 2) v *= ldexp(1.0, -(rand() % 24));             // wide spread, reaches subnormals
                    if (gen == 3) v = (k & 1) ? -fabs(v) : fabs(v);            // alternating signs
                    if (gen == 4) v = rnd() * ldexp(1.0, -15 - (rand() % 9));  // all A subnormal in f16
                    A[i * 16 + k] = f2h((float)v);
                }
            for (int k = 0; k < 16; k++)
                for (int j = 0; j < 32; j++) {
                    double v = rnd() * 1024.0;
                    if (gen == 1) v *= ldexp(1.0, -(k & 1) * 11);
                    if (gen == 2) v *= ldexp(1.0, -(rand() % 24));
                    if (gen == 3) v = fabs(v);
                    B[k * 32 + j] = f2h((float)v);
                }
            for (int i = 0; i < 32 * 32; i++) C[i] = gen == 4 ? 0.0f : (float)(rnd() * 1024.0 * 1024.0);
            CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            CHECK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < 32; i++)
                for (int j = 0; j < 32; j++) {
                    double s = C[i * 32 + j], sabs = fabs(s);
                    for (int k = 0; k < 16; k++) {
                        const double p = (double)h2f(A[i * 16 + k]) * (double)h2f(B[k * 32 + j]);
                        s += p;
                        sabs += fabs(p);
                    }
                    const double err = fabs((double)D[i * 32 + j] - s);
                    const double rel = sabs > 0 ? err / (5.9604644775390625e-8 * sabs) : 0;
                    if (rel > worst_rel) worst_rel = rel;
                    if ((float)s != D[i * 32 + j]) mism_rn++;
                    if (D[i * 32 + j] == 0.0f && s != 0.0) flushed++;
                    total++;
                }
        }
        printf("f16 accuracy gen %d: worst |err| = %.3f x 2^-24 x sum|terms|; %ld of %ld differ from RN(exact); %ld exact zeros where the sum is not\n",
               gen, worst_rel, mism_rn, total, flushed);
    }
    return 0;
}
