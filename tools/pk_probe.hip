// tools/pk_probe.hip -- does a packed fp32 instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) ever deliver a wrong
// half when another stream's kernel issues matrix instructions on the same SIMDs?  (csrc/fps.hip, DESIGN.md 6a: the
// farthest-point sampling's wrong samples went away when its update stopped using them.)
//
// probe kernel: every lane keeps two "points" and their running minima twice -- once updated with two-element vector
// arithmetic (packed instructions), once with scalar instructions behind opaque statements -- against a stream of pivots
// taken from scalar registers, as the sampling's workers do; after every pivot the two copies are compared bit for bit and a
// mismatch is recorded (iteration, lane, half, both values).  mode N > 0: the wave sleeps N x 64 clocks between pivots (s_sleep), like a
// worker waiting for its coordinator.
// burner kernel: v_mfma_f32_32x32x16_f16 in a loop.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/pk_probe.hip -o /tmp/pk_probe && /tmp/pk_probe [seconds] [mode]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Hit { unsigned it, lane_half, packed_bits, scalar_bits; };

__global__ __launch_bounds__(256) void probe_kernel(int iters, int mode_, unsigned *nhit, Hit *hits, unsigned long long *done)
{
    int mode = mode_;
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    unsigned s = t * 2654435761u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    const f32x2 px = {rnd(), rnd()}, py = {rnd(), rnd()}, pz = {rnd(), rnd()};
    f32x2 dp = {1e30f, 1e30f};
    float d0 = 1e30f, d1 = 1e30f;
    unsigned u = blockIdx.x * 747796405u + 2891336453u;      // (wave-uniform pivot stream)
    // lds != 0: the pivots come out of LDS as per-lane broadcast reads into the registers the arithmetic consumes at once (the
    // sampling's pre-fix form), a fresh read per iteration into the same registers
    __shared__ float s_piv[64][4];
    if (threadIdx.x < 64) { s_piv[threadIdx.x][0] = rnd(); s_piv[threadIdx.x][1] = rnd(); s_piv[threadIdx.x][2] = rnd(); s_piv[threadIdx.x][3] = 0.0f; }
    __syncthreads();
    __shared__ unsigned s_pair[64][2];
    if (threadIdx.x < 64) { s_pair[threadIdx.x][0] = __float_as_uint(rnd()); s_pair[threadIdx.x][1] = 0x00000041u + threadIdx.x * 64u; }
    __syncthreads();
    const int lds = (mode >> 8) & 1, opsel = (mode >> 9) & 1;
    mode &= 255;
    for (int it = 0; it < iters; it++) {
        u = u * 1664525u + 1013904223u;
        const unsigned u1 = u * 22695477u + 1u, u2 = u1 * 22695477u + 1u;
        float cx, cy, cz;
        if (lds) {
            const volatile float *pv = &s_piv[(it + (int)blockIdx.x) & 63][0];
            cx = pv[0]; cy = pv[1]; cz = pv[2];
        } else {
            cx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)(u >> 8) * (1.0f / 16777216.0f))));
            cy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)(u1 >> 8) * (1.0f / 16777216.0f))));
            cz = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)(u2 >> 8) * (1.0f / 16777216.0f))));
        }
        // the wave idles between pivots like a worker waiting for its coordinator: mode = 64-clock units (1, 8, 32 or 127)
        if (mode == 1) __builtin_amdgcn_s_sleep(1);
        else if (mode == 8) __builtin_amdgcn_s_sleep(8);
        else if (mode == 32) __builtin_amdgcn_s_sleep(32);
        else if (mode == 127) __builtin_amdgcn_s_sleep(127);
        // opsel != 0: the sampling's failing form in isolation -- an 8-byte LDS read returns (value, unrelated word) into a register
        // pair, and a packed subtract right behind the wait takes the LOW half for both of its lanes (op_sel_hi:[1,0])
        if (opsel) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 pv;
            f32x2 dxp;
            const unsigned a2 = (unsigned)(size_t)&s_pair[(it + (int)blockIdx.x) & 63][0];
            asm volatile("ds_read_b64 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %1, %3, %0 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1"
                         : "=&v"(pv), "=&v"(dxp) : "v"(a2), "v"(px) : "memory");
            const float c = __uint_as_float(pv.x);
            float r0 = px.x - c, r1 = px.y - c;
            asm volatile("" : "+v"(r0), "+v"(r1));
            if (__float_as_uint(dxp.x) != __float_as_uint(r0) || __float_as_uint(dxp.y) != __float_as_uint(r1)) {
                const unsigned k = atomicAdd(nhit, 1u);
                const bool b0 = __float_as_uint(dxp.x) != __float_as_uint(r0);
                if (k < 4096) hits[k] = Hit{(unsigned)it, (threadIdx.x & 63u) * 2u + (b0 ? 0u : 1u), __float_as_uint(b0 ? dxp.x : dxp.y), __float_as_uint(b0 ? r0 : r1)};
            }
            continue;
        }
        // packed
        const f32x2 dx = px - cx, dy = py - cy, dz = pz - cz;
        const f32x2 dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
        dp.x = dp.x < dd.x ? dp.x : dd.x;
        dp.y = dp.y < dd.y ? dp.y : dd.y;
        // scalar
        float ax = px.x - cx, ay = py.x - cy, az = pz.x - cz, bx = px.y - cx, by = py.y - cy, bz = pz.y - cz;
        asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(bx), "+v"(by), "+v"(bz));
        float ta = __fmul_rn(ay, ay), tb = __fmul_rn(by, by);
        asm volatile("" : "+v"(ta), "+v"(tb));
        ta = __fmaf_rn(ax, ax, ta); tb = __fmaf_rn(bx, bx, tb);
        asm volatile("" : "+v"(ta), "+v"(tb));
        ta = __fmaf_rn(az, az, ta); tb = __fmaf_rn(bz, bz, tb);
        asm volatile("" : "+v"(ta), "+v"(tb));
        d0 = d0 < ta ? d0 : ta;
        d1 = d1 < tb ? d1 : tb;
        const bool bad0 = __float_as_uint(dp.x) != __float_as_uint(d0), bad1 = __float_as_uint(dp.y) != __float_as_uint(d1);
        if (bad0 || bad1) {
            const unsigned k = atomicAdd(nhit, 1u);
            if (k < 4096) hits[k] = Hit{(unsigned)it, (threadIdx.x & 63u) * 2u + (bad0 ? 0u : 1u), __float_as_uint(bad0 ? dp.x : dp.y), __float_as_uint(bad0 ? d0 : d1)};
            dp.x = d0; dp.y = d1;          // resynchronise
        }
        if ((it & 255) == 0) { d0 = dp.x = 1e30f; d1 = dp.y = 1e30f; }      // (keep the minima moving)
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
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int burn = argc > 3 ? atoi(argv[3]) : 1;
    const int pblocks = argc > 4 ? atoi(argv[4]) : 512;      // 256: ONE probe wave per SIMD (a worker of the sampling has its SIMD to itself)
    hipStream_t sa, sb;
    hipStreamCreate(&sa); hipStreamCreate(&sb);
    unsigned *nhit; Hit *hits; unsigned long long *done; float *out;
    hipMalloc(&nhit, 4); hipMalloc(&hits, sizeof(Hit) * 4096); hipMalloc(&done, 8); hipMalloc(&out, 4);
    hipMemset(nhit, 0, 4); hipMemset(done, 0, 8);
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        // one probe wave and one burner block per SIMD pair: 512 probe blocks of four waves beside 512 burner blocks of eight
        hipLaunchKernelGGL(probe_kernel, dim3(pblocks), dim3(256), 0, sa, (mode & 255) >= 32 ? 4000 : ((mode & 255) >= 8 ? 20000 : 200000), mode, nhit, hits, done);
        if (burn) for (int q = 0; q < 8; q++) hipLaunchKernelGGL(burn_kernel, dim3(512), dim3(512), 0, sb, 20000, out);
        hipStreamSynchronize(sa);
        launches++;
    }
    hipDeviceSynchronize();
    unsigned n = 0; unsigned long long d = 0;
    hipMemcpy(&n, nhit, 4, hipMemcpyDeviceToHost); hipMemcpy(&d, done, 8, hipMemcpyDeviceToHost);
    Hit *h = (Hit *)malloc(sizeof(Hit) * 4096);
    hipMemcpy(h, hits, sizeof(Hit) * 4096, hipMemcpyDeviceToHost);
    printf("%d probe blocks, mode %d, burner %s: %d launches, %.3g wave-iterations (8 packed instructions each), %u mismatches between the packed and the scalar copy\n",
           pblocks, mode, burn ? "on" : "off", launches, (double)d * 4.0, n);
    int lanes[64] = {0}, halves[2] = {0, 0};
    for (unsigned k = 0; k < n && k < 4096; k++) { lanes[(h[k].lane_half >> 1) & 63]++; halves[h[k].lane_half & 1]++; }
    if (n) {
        printf("  by half: low %d, high %d;  by lane quarter: 0-15 %d, 16-31 %d, 32-47 %d, 48-63 %d\n", halves[0], halves[1],
               [&] { int s = 0; for (int i = 0; i < 16; i++) s += lanes[i]; return s; }(), [&] { int s = 0; for (int i = 16; i < 32; i++) s += lanes[i]; return s; }(),
               [&] { int s = 0; for (int i = 32; i < 48; i++) s += lanes[i]; return s; }(), [&] { int s = 0; for (int i = 48; i < 64; i++) s += lanes[i]; return s; }());
        for (unsigned k = 0; k < n && k < 8; k++)
            printf("  iteration %u lane %u %s half: packed %08x scalar %08x\n", h[k].it, h[k].lane_half >> 1, (h[k].lane_half & 1) ? "high" : "low", h[k].packed_bits, h[k].scalar_bits);
    }
    return 0;
}
