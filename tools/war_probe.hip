// tools/war_probe.hip -- the write-after-read pattern of the farthest-point sampling's pre-fix worker loop (csrc/fps.hip):
//     v_mov_b64  v[20:21], v[12:13]      ; the new running minima (a register PAIR once the update is packed) move to their home
//     ds_read_b96 v[10:12], ...          ; the next pivot comes back from LDS into the SAME registers
// Does the LDS data ever land before the 64-bit move has read its source, when another stream's kernel keeps the matrix
// pipe busy?  Every lane holds a known pair, copies it with a 64-bit VALU instruction and lets an LDS read overwrite the
// source at once; the copy must be the known pair.  kind 0: v_mov_b64, 1: v_pk_mov_b32, 2: two v_mov_b32 (32-bit control),
// 3: v_pk_add_f32 with zero.
//   hipcc --offload-arch=gfx950 -O3 tools/war_probe.hip -o /tmp/war_probe && /tmp/war_probe [seconds] [kind] [burner 0/1] [probe blocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Hit { unsigned it, lane, got_lo, got_hi; };

template <int KIND>
__global__ __launch_bounds__(256) void war_kernel(int iters, unsigned *nhit, Hit *hits, unsigned long long *done)
{
    __shared__ unsigned long long s_data[64];
    if (threadIdx.x < 64) s_data[threadIdx.x] = 0xdeadbeef00000000ull | 0xabad1deau;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const unsigned addr = (unsigned)(size_t)&s_data[(threadIdx.x >> 6) * 7 % 64];      // a wave reads ONE address: a broadcast, like the pivot
    for (int it = 0; it < iters; it++) {
        u32x2 src = {0x10000000u + (unsigned)it * 64u + lane, 0x20000000u + (unsigned)it * 64u + lane};
        u32x2 copy;
        if (KIND == 0)
            asm volatile("s_nop 1\n\tv_mov_b64 %0, %1\n\tds_read_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(copy), "+v"(src) : "v"(addr) : "memory");
        else if (KIND == 1)
            asm volatile("s_nop 1\n\tv_pk_mov_b32 %0, %1, %1 op_sel:[0,1]\n\tds_read_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(copy), "+v"(src) : "v"(addr) : "memory");
        else if (KIND == 2) {
            unsigned c0, c1, s0 = src.x, s1 = src.y;
            asm volatile("s_nop 1\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tds_read_b32 %2, %4\n\tds_read_b32 %3, %4 offset:4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(c0), "=&v"(c1), "+v"(s0), "+v"(s1) : "v"(addr) : "memory");
            copy.x = c0; copy.y = c1; src.x = s0; src.y = s1;
        }
        else
            asm volatile("s_nop 1\n\tv_pk_add_f32 %0, %1, 0\n\tds_read_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(copy), "+v"(src) : "v"(addr) : "memory");
        const unsigned want_lo = 0x10000000u + (unsigned)it * 64u + lane, want_hi = 0x20000000u + (unsigned)it * 64u + lane;
        if (copy.x != want_lo || copy.y != want_hi) {
            const unsigned k = atomicAdd(nhit, 1u);
            if (k < 4096) hits[k] = Hit{(unsigned)it, lane | (copy.x != want_lo ? 64u : 0u) | (copy.y != want_hi ? 128u : 0u), copy.x, copy.y};
        }
        if (src.x != 0xabad1deau || src.y != 0xdeadbeefu) atomicAdd(nhit + 1, 1u);      // (the LDS data itself)
    }
    if (threadIdx.x == 0) atomicAdd(done, (unsigned long long)iters);
}

__global__ __launch_bounds__(512) void burn_kernel(int iters, float *out)
{
    f32x16 acc = {0};
    uint4 a = make_uint4(threadIdx.x, threadIdx.x * 3, threadIdx.x * 5, 0x3c003c00u);
    for (int it = 0; it < iters; it++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), acc, 0, 0, 0);
        a.x += 1;
    }
    float s = 0; for (int e = 0; e < 16; e++) s += acc[e];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    const int burn = argc > 3 ? atoi(argv[3]) : 1;
    const int pblocks = argc > 4 ? atoi(argv[4]) : 256;
    hipStream_t sa, sb;
    (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    unsigned *nhit; Hit *hits; unsigned long long *done; float *out;
    (void)hipMalloc(&nhit, 8); (void)hipMalloc(&hits, sizeof(Hit) * 4096); (void)hipMalloc(&done, 8); (void)hipMalloc(&out, 4);
    (void)hipMemset(nhit, 0, 8); (void)hipMemset(done, 0, 8);
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        if (kind == 0) hipLaunchKernelGGL(war_kernel<0>, dim3(pblocks), dim3(256), 0, sa, 400000, nhit, hits, done);
        else if (kind == 1) hipLaunchKernelGGL(war_kernel<1>, dim3(pblocks), dim3(256), 0, sa, 400000, nhit, hits, done);
        else if (kind == 2) hipLaunchKernelGGL(war_kernel<2>, dim3(pblocks), dim3(256), 0, sa, 400000, nhit, hits, done);
        else hipLaunchKernelGGL(war_kernel<3>, dim3(pblocks), dim3(256), 0, sa, 400000, nhit, hits, done);
        if (burn) for (int q = 0; q < 8; q++) hipLaunchKernelGGL(burn_kernel, dim3(512), dim3(512), 0, sb, 20000, out);
        (void)hipStreamSynchronize(sa);
    }
    (void)hipDeviceSynchronize();
    unsigned n[2] = {0, 0}; unsigned long long d = 0;
    (void)hipMemcpy(n, nhit, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&d, done, 8, hipMemcpyDeviceToHost);
    Hit *h = (Hit *)malloc(sizeof(Hit) * 4096);
    (void)hipMemcpy(h, hits, sizeof(Hit) * 4096, hipMemcpyDeviceToHost);
    const char *names[4] = {"v_mov_b64", "v_pk_mov_b32", "two v_mov_b32", "v_pk_add_f32"};
    printf("%s, %d probe blocks, burner %s: %.3g wave-iterations, %u copies that are not their source (LDS reads that are not the LDS data: %u)\n",
           names[kind & 3], pblocks, burn ? "on" : "off", (double)d * 4.0, n[0], n[1]);
    int q[4] = {0, 0, 0, 0}, lo = 0, hi = 0, lds_lo = 0;
    for (unsigned k = 0; k < n[0] && k < 4096; k++) { q[(h[k].lane & 63) >> 4]++; lo += (h[k].lane >> 6) & 1; hi += (h[k].lane >> 7) & 1; lds_lo += h[k].got_lo == 0xabad1deau; }
    if (n[0]) {
        printf("  lanes 0-15: %d, 16-31: %d, 32-47: %d, 48-63: %d;  low register wrong: %d, high register wrong: %d;  wrong low register holding the LDS data: %d\n", q[0], q[1], q[2], q[3], lo, hi, lds_lo);
        for (unsigned k = 0; k < n[0] && k < 6; k++) printf("  iteration %u lane %u: copy = %08x %08x\n", h[k].it, h[k].lane & 63, h[k].got_lo, h[k].got_hi);
    }
    return 0;
}
