// tools/burn.hip -- background load for tools/stress_concurrent.py: which ingredient of the f16 nearest-neighbour filter's main
// loop disturbs a farthest-point sampling on another stream?  kind bits: 1 MFMA f16 32x32x16, 2 ds_read_b128 storm, 4 VALU
// (v_min3) storm, 8 take 66 KB of LDS per block.  hipcc --offload-arch=gfx950 -shared -fPIC tools/burn.hip -o tools/_burn/libburn.so
#include <hip/hip_runtime.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BIG>
__global__ __launch_bounds__(512) void burn_kernel(int kind, int iters, float *out)
{
    __shared__ uint4 plane[BIG ? 4200 : 64];
    const int t = threadIdx.x;
    for (int i = t; i < (BIG ? 4200 : 64); i += 512) plane[i] = make_uint4(i, i * 3, i * 5, 0x3c003c00u);
    __syncthreads();
    f32x16 acc = {0};
    float m0 = 1e30f, m1 = 1e30f, m2 = 1e30f, m3 = 1e30f;
    uint4 a = plane[t & 63];
    for (int it = 0; it < iters; it++) {
        if (kind & 2) {
            const uint4 b = plane[(t * 7 + it * 13) % (BIG ? 4200 : 64)];
            a.x ^= b.x; a.y += b.y; a.z ^= b.z;
        }
        if (kind & 1) {
            const f32x16 z = {0};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), (kind & 4) ? z : acc, 0, 0, 0);
        }
        if (kind & 4) {
#pragma unroll
            for (int e = 0; e < 16; e += 8) {
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(acc[e]), "v"(acc[e + 1]));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(acc[e + 2]), "v"(acc[e + 3]));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m2) : "v"(acc[e + 4]), "v"(acc[e + 5]));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m3) : "v"(acc[e + 6]), "v"(acc[e + 7]));
            }
        }
    }
    float s = m0 + m1 + m2 + m3 + (float)a.x;
    for (int e = 0; e < 16; e++) s += acc[e];
    if (s == 12345.678f) out[0] = s;
}

extern "C" __attribute__((visibility("default"))) int burn(int kind, int iters, int blocks, float *out, void *stream)
{
    if (kind & 8) hipLaunchKernelGGL(burn_kernel<1>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, kind, iters, out);
    else hipLaunchKernelGGL(burn_kernel<0>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, kind, iters, out);
    return hipGetLastError() == hipSuccess;
}
