// probe.hip -- hardware-premise probes the test-suite runs on the box it is on.
//
// The default NN path (nn_f16.hip) evaluates |t|^2 - 2 q.t for 32 x 32 pairs with ONE
// v_mfma_f32_32x32x16_f16 and proves its candidate lists complete from a bound on that instruction's K = 16
// summation error: |result - exact| <= 6.5 u sum|terms| (u = 2^-24; measured 3.1 u on the development box,
// tools/ubench_mfma_f16.hip).  The bound is a property of the matrix pipe's internal accumulation, which no
// document states: genpc_mfma_f16_probe runs the instruction on caller-supplied operands so that
// tests/test_gpu_mfma_premise.py can re-measure it wherever the suite runs -- a stepping that accumulates
// differently fails a test instead of silently corrupting nearest neighbours.
#include "common.h"
#include "../../include/genpc_hip.h"

#include <stdint.h>

namespace genpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// One wave per problem: D[32,32] = A[32,16] * B[16,32] + C[32,32] (A, B f16 bit patterns, row-major).
// Lane l supplies A[l&31][8*(l>>5) .. +7] and B[8*(l>>5) .. +7][l&31]; D[i][j]: lane j + 32*((i>>2)&1), register
// (i>>3)*4 + (i&3) -- the operand layout nn_f16_kernel uses.
__global__ __launch_bounds__(kWave) void mfma_f16_probe_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B,
                                                               const float *__restrict__ C, float *__restrict__ D)
{
    const int l = threadIdx.x;
    A += (size_t)blockIdx.x * 32 * 16;
    B += (size_t)blockIdx.x * 16 * 32;
    C += (size_t)blockIdx.x * 32 * 32;
    D += (size_t)blockIdx.x * 32 * 32;
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

}  // namespace genpc

GENPC_API int genpc_mfma_f16_probe(int problems, const unsigned short *a, const unsigned short *b, const float *c, float *d,
                                   void *stream)
{
    using namespace genpc;
    if (problems < 0) return -1;
    if (problems == 0) return 1;
    hipLaunchKernelGGL(mfma_f16_probe_kernel, dim3(problems), dim3(kWave), 0, (hipStream_t)stream, (const uint16_t *)a,
                       (const uint16_t *)b, c, d);
    return check(hipGetLastError(), "mfma probe launch") ? 1 : 0;
}
