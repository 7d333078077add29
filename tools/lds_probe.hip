// tools/lds_probe.hip -- does an LDS broadcast read return stale data in some lanes while another kernel's f16 MFMAs share the SIMD?
// Block = 2 waves.  Wave 0 (lane 0) writes a record {i, i, i, i} to LDS, waits for it (lgkmcnt(0)), then publishes i in a flag word.
// Wave 1 polls the flag (first lane's view decides, like compiler-scalarised code), then EVERY lane reads the record with one
// ds_read_b32 / b64 / b96 / b128 and checks that all words are >= the flag it acted on.  Violations are counted per 16-lane row.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/_burn/lds_probe ;  tools/_burn/lds_probe [with_mfma_load]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void load_kernel(int iters, float *out)
{
    f32x16 acc = {0};
    uint4 a = make_uint4(threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    for (int it = 0; it < iters; it++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), acc, 0, 0, 0);
    float s = 0;
    for (int e = 0; e < 16; e++) s += acc[e];
    if (s == 12345.678f) out[0] = s;
}

template <int WORDS, int PACKED>
__global__ __launch_bounds__(128) void probe_kernel(int iters, unsigned long long *bad)
{
    __shared__ __attribute__((aligned(16))) unsigned rec[4];
    __shared__ unsigned flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { rec[0] = rec[1] = rec[2] = rec[3] = 0; flag = 0; }
    __syncthreads();
    if (wave == 0) {
        for (unsigned i = 1; i <= (unsigned)iters; i++) {
            if (lane == 0) {
                { typedef unsigned u4 __attribute__((ext_vector_type(4))); const u4 v = {i, i, i, i}; asm volatile("ds_write_b128 %0, %1" :: "v"(0u), "v"(v) : "memory"); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __hip_atomic_store(&flag, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_s_sleep(2);
        }
        return;
    }
    unsigned long long mine = 0;
    unsigned seen = 0;
    while (seen < (unsigned)iters) {
        const unsigned f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (f == seen) { __builtin_amdgcn_s_sleep(1); continue; }
        seen = f;
        unsigned w0, w1 = f, w2 = f, w3 = f;
        if (WORDS == 1) { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w0) : "v"(0u) : "memory"); }
        if (WORDS == 2) { unsigned long long v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(0u) : "memory"); w0 = (unsigned)v; w1 = (unsigned)(v >> 32); }
        if (WORDS == 3) { typedef unsigned u3 __attribute__((ext_vector_type(3))); u3 v; asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(0u) : "memory"); w0 = v.x; w1 = v.y; w2 = v.z; }
        if (WORDS == 4) { typedef unsigned u4 __attribute__((ext_vector_type(4))); u4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(0u) : "memory"); w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w; }
        // consume right away (a VALU op directly behind the wait, like the sampling kernel's update)
        const unsigned m = min(min(w0, w1), min(w2, w3));
        if (m < f) mine++;
        if (PACKED) {
            // the sampling kernel's own consumer: packed fp32 subtractions with one loaded word broadcast to both halves
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 c = {(float)(lane + 1), (float)(2 * lane + 3)};
            f2 wv = {__uint_as_float(w0), __uint_as_float(w1)}, r;
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(c), "v"(wv));
            const float e0 = c.x - __uint_as_float(w0), e1 = c.y - __uint_as_float(w0);
            if (r.x != e0 || r.y != e1) mine += 1ull << 32;
        }
    }
    if (mine) atomicAdd(&bad[lane >> 4], mine);
}

int main(int argc, char **argv)
{
    const bool load = argc > 1;
    unsigned long long *bad;
    float *out;
    hipMalloc(&bad, 4 * 4 * sizeof(unsigned long long));
    hipMalloc(&out, 16);
    hipMemset(bad, 0, 4 * 4 * sizeof(unsigned long long));
    hipStream_t s0, s1;
    hipStreamCreate(&s0); hipStreamCreate(&s1);
    for (int rep = 0; rep < 3; rep++) {
        if (load) for (int i = 0; i < 400; i++) hipLaunchKernelGGL(load_kernel, dim3(256), dim3(512), 0, s1, 400, out);
        hipLaunchKernelGGL((probe_kernel<1, 1>), dim3(512), dim3(128), 0, s0, 20000, bad + 0);
        hipLaunchKernelGGL((probe_kernel<2, 1>), dim3(512), dim3(128), 0, s0, 20000, bad + 4);
        hipLaunchKernelGGL((probe_kernel<3, 1>), dim3(512), dim3(128), 0, s0, 20000, bad + 8);
        hipLaunchKernelGGL((probe_kernel<4, 1>), dim3(512), dim3(128), 0, s0, 20000, bad + 12);
        hipDeviceSynchronize();
    }
    unsigned long long h[16];
    hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
    const char *nm[4] = {"ds_read_b32 ", "ds_read_b64 ", "ds_read_b96 ", "ds_read_b128"};
    printf("%s: stale reads per 16-lane row (of 3 x 512 blocks x 20000 reads per lane)\n", load ? "with f16 MFMA load on a second stream" : "alone");
    for (int k = 0; k < 4; k++) printf("  %s  stale: lanes 0-15 %llu, 16-31 %llu, 32-47 %llu, 48-63 %llu;  packed-add mismatches: %llu, %llu, %llu, %llu\n", nm[k],
                                       h[4 * k] & 0xffffffffull, h[4 * k + 1] & 0xffffffffull, h[4 * k + 2] & 0xffffffffull, h[4 * k + 3] & 0xffffffffull,
                                       h[4 * k] >> 32, h[4 * k + 1] >> 32, h[4 * k + 2] >> 32, h[4 * k + 3] >> 32);
    return 0;
}
