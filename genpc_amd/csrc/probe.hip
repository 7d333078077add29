// probe.hip -- hardware-premise probes the test-suite runs on the box it is on:
// the f16 MFMA's summation error (below) and fastdiv.h's shared-reciprocal division against the compiler's.
//
// The default NN path (nn_f16.hip) evaluates |t|^2 - 2 q.t for 32 x 32 pairs with ONE
// v_mfma_f32_32x32x16_f16 and proves its candidate lists complete from a bound on that instruction's K = 16
// summation error: |result - exact| <= 6.5 u sum|terms| (u = 2^-24; measured 3.1 u on the development box,
// tools/ubench_mfma_f16.hip).  The bound is a property of the matrix pipe's internal accumulation, which no
// document states: genpc_mfma_f16_probe runs the instruction on caller-supplied operands so that
// tests/test_gpu_mfma_premise.py can re-measure it wherever the suite runs -- a stepping that accumulates
// differently fails a test instead of silently corrupting nearest neighbours.
#include "nn.h"
#include "fastdiv.h"
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

// fastdiv.h against the compiler's division, element by element: fast[i] = div_core(num[i], den[i], rcp_refined(den[i])),
// the packed form with the pair (i, i ^ 1), ieee[i] = num[i] / den[i], in_range[i] = what the callers' range test says.
__global__ __launch_bounds__(256) void fastdiv_probe_kernel(long long n, const float *__restrict__ num, const float *__restrict__ den,
                                                            float *__restrict__ fast, float *__restrict__ fast2,
                                                            float *__restrict__ ieee, unsigned char *__restrict__ in_range)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float a = num[i], d = den[i];
        const long long j = (i ^ 1) < n ? (i ^ 1) : i;
        const float a2 = num[j], d2 = den[j];
        fast[i] = div_core(a, d, rcp_refined(d));
        const v2f q = div_core2((v2f){a, a2}, (v2f){d, d2}, (v2f){rcp_refined(d), rcp_refined(d2)});
        fast2[i] = q.x;
        ieee[i] = a / d;
        DivRange r;
        r.add(a, d);
        in_range[i] = r.ok() ? 1 : 0;
    }
}

// the list hand-off's coded lower bounds (nn.h): out[i] = list_dec(base[i], list_enc(base[i], v[i]))
__global__ __launch_bounds__(256) void list_code_probe_kernel(long long n, const float *__restrict__ base, const float *__restrict__ v,
                                                              float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = list_dec(base[i], list_enc(base[i], v[i]));
}

}  // namespace genpc

GENPC_API int genpc_list_code_probe(long long n, const float *base, const float *v, float *out, void *stream)
{
    using namespace genpc;
    if (n < 0) return -1;
    if (n == 0) return 1;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(list_code_probe_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, n, base,
                       v, out);
    return check(hipGetLastError(), "list code probe launch") ? 1 : 0;
}

GENPC_API int genpc_fastdiv_probe(long long n, const float *num, const float *den, float *fast, float *fast_packed, float *ieee,
                                  unsigned char *in_range, void *stream)
{
    using namespace genpc;
    if (n < 0) return -1;
    if (n == 0) return 1;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(fastdiv_probe_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, n, num,
                       den, fast, fast_packed, ieee, in_range);
    return check(hipGetLastError(), "fastdiv probe launch") ? 1 : 0;
}

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
