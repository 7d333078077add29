// tools/opsel_probe.hip -- the form that fails inside the farthest-point sampling (csrc/fps.hip), in isolation: a packed
// fp32 instruction whose LOW lane takes the HIGH half of a source pair (op_sel:[0,1]) or whose HIGH lane takes the LOW half
// (op_sel_hi:[1,0]), on known data, beside another stream's v_mfma_f32_32x32x16_f16 loop.
//   hipcc --offload-arch=gfx950 -O3 tools/opsel_probe.hip -o /tmp/opsel_probe && /tmp/opsel_probe [seconds] [kind 0|1|2] [burner] [probe blocks] [sleep]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <atomic>
#include <thread>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Hit { unsigned it, lane, got_lo, got_hi; };
// tools/burn.hip's kernel itself (burner selector 99 = its kind 1): the neighbour beside which the probe fails when both are
// launched from Python (tools/opsel_beside_filter.py)
#define burn_kernel burn_kernel_ref
#define h16x8 h16x8_ref
#define f32x16 f32x16_ref
#include "burn.hip"
#undef burn_kernel
#undef h16x8
#undef f32x16

template <int KIND, int SLEEP>
__global__ __launch_bounds__(256) void opsel_kernel(int iters, unsigned *nhit, Hit *hits, unsigned long long *done)
{
    const unsigned lane = threadIdx.x & 63u;
    for (int it = 0; it < iters; it++) {
        const float A = (float)(it & 1023) + 0.25f, B = (float)(it & 1023) + 4096.5f + (float)lane;
        f32x2 pair = {A, B};
        const f32x2 p = {1.0f + (float)lane, 2.0f + (float)lane};
        f32x2 r;
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        // three in a row like the sampling's subtracts; the last one's result is checked
        if (KIND == 0)        // low lane takes the high half: both lanes p - B
            asm volatile("s_nop 3\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3" : "=&v"(r) : "v"(p), "v"(pair));
        else if (KIND == 1)   // high lane takes the low half: both lanes p - A
            asm volatile("s_nop 3\n\tv_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3" : "=&v"(r) : "v"(p), "v"(pair));
        else if (KIND == 2)   // plain pair
            asm volatile("s_nop 3\n\tv_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3" : "=&v"(r) : "v"(p), "v"(pair));
        else if (KIND == 3)   // other instructions of the class, the same half selection: a packed multiply ...
            asm volatile("s_nop 3\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 3" : "=&v"(r) : "v"(p), "v"(pair));
        else if (KIND == 4)   // ... a packed move (the LOW lane takes the high half of its one source)
            asm volatile("s_nop 3\n\tv_pk_mov_b32 %0, %2, %2 op_sel:[1,1]\n\ts_nop 3" : "=&v"(r) : "v"(p), "v"(pair));
        else if (KIND == 5)   // ... the selection on src0 instead of src1
            asm volatile("s_nop 3\n\tv_pk_add_f32 %0, %2, %1 op_sel:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\ts_nop 3" : "=&v"(r) : "v"(p), "v"(pair));
        else {                // ... and a packed f16 add: the low result takes the HIGH f16 of src1 (one 32-bit register holds the pair)
            unsigned ph = 0x4400u | (0x4500u << 16), qh = (0x3c00u + (lane & 15u)) | ((0x4000u + (lane & 15u)) << 16), rh;      // p = (4, 5), q = (1 + e, 2 + e')
            asm volatile("s_nop 3\n\tv_pk_add_f16 %0, %1, %2 op_sel:[0,1]\n\ts_nop 3" : "=&v"(rh) : "v"(ph), "v"(qh));
            unsigned ref_lo, ref_hi;      // both lanes add q.hi: computed without packed instructions
            {
                _Float16 plo = __builtin_bit_cast(_Float16, (unsigned short)(ph & 0xffffu)), phi = __builtin_bit_cast(_Float16, (unsigned short)(ph >> 16));
                _Float16 qhi = __builtin_bit_cast(_Float16, (unsigned short)(qh >> 16));
                ref_lo = __builtin_bit_cast(unsigned short, (_Float16)(plo + qhi));
                ref_hi = __builtin_bit_cast(unsigned short, (_Float16)(phi + qhi));
            }
            r.x = __uint_as_float(rh & 0xffffu); r.y = __uint_as_float(rh >> 16);
            const float w0h = __uint_as_float(ref_lo), w1h = __uint_as_float(ref_hi);
            if (__float_as_uint(r.x) != __float_as_uint(w0h) || __float_as_uint(r.y) != __float_as_uint(w1h)) {
                const unsigned k = atomicAdd(nhit, 1u);
                if (k < 4096) hits[k] = Hit{(unsigned)it, lane | (__float_as_uint(r.x) != __float_as_uint(w0h) ? 64u : 0u) | (__float_as_uint(r.y) != __float_as_uint(w1h) ? 128u : 0u), rh & 0xffffu, rh >> 16};
            }
            continue;
        }
        const float w0 = KIND == 3 ? p.x * B : (KIND == 4 ? B : p.x - (KIND == 0 || KIND == 5 ? B : A));
        const float w1 = KIND == 3 ? p.y * B : (KIND == 4 ? B : p.y - (KIND == 1 ? A : B));
        if (__float_as_uint(r.x) != __float_as_uint(w0) || __float_as_uint(r.y) != __float_as_uint(w1)) {
            const unsigned k = atomicAdd(nhit, 1u);
            if (k < 4096) hits[k] = Hit{(unsigned)it, lane | (__float_as_uint(r.x) != __float_as_uint(w0) ? 64u : 0u) | (__float_as_uint(r.y) != __float_as_uint(w1) ? 128u : 0u),
                                        __float_as_uint(r.x), __float_as_uint(r.y)};
        }
    }
    if (threadIdx.x == 0) atomicAdd(done, (unsigned long long)iters);
}

// sel (a run-time 0): the accumulator passes through sixteen v_cndmask_b32_e64 (a VOP3 VALU instruction reading the matrix
// instruction's result registers under an SGPR mask) in front of every MFMA -- tools/burn.hip kind 1, the neighbour beside which
// the probe fails; sel < 0: a bare chain of dependent MFMAs, beside which it does not
__global__ __launch_bounds__(512) void burn_kernel(int iters, float *out, int sel)
{
    f32x16 acc = {0};
    uint4 a = make_uint4(threadIdx.x, threadIdx.x * 3, threadIdx.x * 5, 0x3c003c00u);
    for (int it = 0; it < iters; it++) {
        if (sel >= 0) {
            const f32x16 z = {0};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), (sel & 4) ? z : acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), acc, 0, 0, 0);
            a.x += 1;
        }
    }
    float s = 0; for (int e = 0; e < 16; e++) s += acc[e];
    if (s == 12345.678f) out[0] = s;
}

// tools/burn.hip's kernel again, with one ingredient removed per variant (burner selectors 100 + V), to see which one the
// failure needs: V 0 the copy; 1 no LDS / barrier (the operand from registers); 2 no branches on `kind` in the loop (the select
// stays); 3 the select replaced by a plain dependent chain; 4 the v_cndmask kept but on a register that is NOT the MFMA's result
template <int V>
__global__ __launch_bounds__(512) void burn_variant(int kind, int iters, float *out)
{
    __shared__ uint4 plane[64];
    const int t = threadIdx.x;
    uint4 a;
    if (V != 1) {
        for (int i = t; i < 64; i += 512) plane[i] = make_uint4(i, i * 3, i * 5, 0x3c003c00u);
        __syncthreads();
        a = plane[t & 63];
    } else {
        a = make_uint4(t & 63, (t & 63) * 3, (t & 63) * 5, 0x3c003c00u);
    }
    f32x16 acc = {0};
    float side = (float)t;
    for (int it = 0; it < iters; it++) {
        if (V < 2) {
            if (kind & 2) { const uint4 b = plane[(t * 7 + it * 13) % 64]; a.x ^= b.x; a.y += b.y; a.z ^= b.z; }
            if (kind & 1) { const f32x16 z = {0}; acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), (kind & 4) ? z : acc, 0, 0, 0); }
        } else if (V == 2) {
            const f32x16 z = {0};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), (kind & 4) ? z : acc, 0, 0, 0);
        } else if (V == 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), acc, 0, 0, 0);
        } else if (V == 4) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 16; q++) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(side) : "s"((unsigned long long)(kind & 4 ? 0ull : ~0ull)));
        } else if (V == 5) {          // no matrix instruction at all; the select's constant operand is 1.0
#pragma unroll
            for (int q = 0; q < 16; q++) asm volatile("v_cndmask_b32_e64 %0, 1.0, %0, %1" : "+v"(side) : "s"((unsigned long long)(kind & 4 ? 0ull : ~0ull)));
        } else if (V == 6) {          // ... the constant in src1 instead of src0
#pragma unroll
            for (int q = 0; q < 16; q++) asm volatile("v_cndmask_b32_e64 %0, %0, 1.0, %1" : "+v"(side) : "s"((unsigned long long)(kind & 4 ? ~0ull : 0ull)));
        } else if (V == 7) {          // ... a VOP3 add with the constant 1.0 in src0 (no mask operand)
#pragma unroll
            for (int q = 0; q < 16; q++) asm volatile("v_add_f32_e64 %0, 1.0, %0" : "+v"(side));
        } else if (V == 8) {          // ... the same add in its VOP2 encoding
#pragma unroll
            for (int q = 0; q < 16; q++) asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(side));
        } else {                      // 9 .. 12: the MFMA chain with sixteen VALU instructions on an unrelated register behind each
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, a), acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if (V == 9) asm volatile("v_cndmask_b32_e64 %0, 1.0, %0, %1" : "+v"(side) : "s"((unsigned long long)(kind & 4 ? 0ull : ~0ull)));      // the constant 1.0
                else if (V == 10) asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(side));                                                          // a VOP2 add
                else if (V == 11) asm volatile("v_mov_b32_e32 %0, %0" : "+v"(side));                                                                 // a move, no constant
                else asm volatile("v_cndmask_b32_e64 %0, %0, %0, %1" : "+v"(side) : "s"((unsigned long long)(kind & 4 ? 0ull : ~0ull)));           // the select without a constant
            }
        }
    }
    float s = side + (float)a.x;
    for (int e = 0; e < 16; e++) s += acc[e];
    if (s == 12345.678f) out[0] = s;
}

static void launch_variant(int v, int iters, int blocks, float *out, hipStream_t st)
{
    switch (v) {
    case 0: hipLaunchKernelGGL(burn_variant<0>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    case 1: hipLaunchKernelGGL(burn_variant<1>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    case 2: hipLaunchKernelGGL(burn_variant<2>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    case 3: hipLaunchKernelGGL(burn_variant<3>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    case 4: hipLaunchKernelGGL(burn_variant<4>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    case 5: hipLaunchKernelGGL(burn_variant<5>, dim3(blocks), dim3(512), 0, st, 1, iters * 8, out); break;
    case 6: hipLaunchKernelGGL(burn_variant<6>, dim3(blocks), dim3(512), 0, st, 1, iters * 8, out); break;
    case 7: hipLaunchKernelGGL(burn_variant<7>, dim3(blocks), dim3(512), 0, st, 1, iters * 8, out); break;
    case 8: hipLaunchKernelGGL(burn_variant<8>, dim3(blocks), dim3(512), 0, st, 1, iters * 8, out); break;
    case 9: hipLaunchKernelGGL(burn_variant<9>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    case 10: hipLaunchKernelGGL(burn_variant<10>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    case 11: hipLaunchKernelGGL(burn_variant<11>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    default: hipLaunchKernelGGL(burn_variant<12>, dim3(blocks), dim3(512), 0, st, 1, iters, out); break;
    }
}

template <int KIND> static void launch(int sleep, int pblocks, hipStream_t st, unsigned *nhit, Hit *hits, unsigned long long *done)
{
    if (sleep) hipLaunchKernelGGL((opsel_kernel<KIND, 8>), dim3(pblocks), dim3(256), 0, st, 40000, nhit, hits, done);
    else hipLaunchKernelGGL((opsel_kernel<KIND, 0>), dim3(pblocks), dim3(256), 0, st, 400000, nhit, hits, done);
}

// for tools/opsel_beside_filter.py: one launch on `stream`; counters (two unsigned) and hits (4096 records) are device memory
extern "C" __attribute__((visibility("default"))) int opsel_launch(int kind, int pblocks, int sleep, unsigned *nhit, void *hits, unsigned long long *done, void *stream)
{
    if (kind == 0) launch<0>(sleep, pblocks, (hipStream_t)stream, nhit, (Hit *)hits, done);
    else if (kind == 1) launch<1>(sleep, pblocks, (hipStream_t)stream, nhit, (Hit *)hits, done);
    else launch<2>(sleep, pblocks, (hipStream_t)stream, nhit, (Hit *)hits, done);
    return hipGetLastError() == hipSuccess;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    const int burn = argc > 3 ? atoi(argv[3]) : 1;
    const int pblocks = argc > 4 ? atoi(argv[4]) : 256;
    const int sleep = argc > 5 ? atoi(argv[5]) : 0;
    const int biters = argc > 6 ? atoi(argv[6]) : 20000;      // MFMAs per burner wave: 400 = a burner kernel of ~5 us, relaunched all the time
    const int bblocks = argc > 7 ? atoi(argv[7]) : 512;
    const int bsel = argc > 8 ? atoi(argv[8]) : 0;            // 0: MFMA + v_cndmask on its results (fails), -1: bare MFMA chain (does not)
    hipStream_t sa, sb;
    (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    unsigned *nhit; Hit *hits; unsigned long long *done; float *out;
    (void)hipMalloc(&nhit, 8); (void)hipMalloc(&hits, sizeof(Hit) * 4096); (void)hipMalloc(&done, 8); (void)hipMalloc(&out, 4);
    (void)hipMemset(nhit, 0, 8); (void)hipMemset(done, 0, 8);
    const auto t0 = std::chrono::steady_clock::now();
    // burn >= 2: that many burner streams, each driven by a host thread of its own (as when the library's filter runs on other
    // streams of other threads), 20 launches per synchronisation
    std::atomic<bool> stop{false};
    std::vector<std::thread> bt;
    if (burn >= 2)
        for (int q = 0; q < burn; q++)
            bt.emplace_back([&, q] {
                hipStream_t s; (void)hipStreamCreate(&s);
                float *o; (void)hipMalloc(&o, 4);
                while (!stop.load()) {
                    for (int j = 0; j < 20; j++) if (bsel >= 100) launch_variant(bsel - 100, biters, bblocks, o, s); else if (bsel == 99) ::burn(1, biters, bblocks, o, s); else hipLaunchKernelGGL(burn_kernel, dim3(bblocks), dim3(512), 0, s, biters, o, bsel);
                    (void)hipStreamSynchronize(s);
                }
            });
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        if (kind == 0) launch<0>(sleep, pblocks, sa, nhit, hits, done);
        else if (kind == 1) launch<1>(sleep, pblocks, sa, nhit, hits, done);
        else if (kind == 2) launch<2>(sleep, pblocks, sa, nhit, hits, done);
        else if (kind == 3) launch<3>(sleep, pblocks, sa, nhit, hits, done);
        else if (kind == 4) launch<4>(sleep, pblocks, sa, nhit, hits, done);
        else if (kind == 5) launch<5>(sleep, pblocks, sa, nhit, hits, done);
        else launch<6>(sleep, pblocks, sa, nhit, hits, done);
        if (burn == 1) for (int q = 0; q < (biters >= 20000 ? 8 : 2000); q++) if (bsel >= 100) launch_variant(bsel - 100, biters, bblocks, out, sb); else if (bsel == 99) ::burn(1, biters, bblocks, out, sb); else hipLaunchKernelGGL(burn_kernel, dim3(bblocks), dim3(512), 0, sb, biters, out, bsel);
        (void)hipStreamSynchronize(sa);
    }
    stop.store(true);
    for (auto &t : bt) t.join();
    (void)hipDeviceSynchronize();
    unsigned n = 0; unsigned long long d = 0;
    (void)hipMemcpy(&n, nhit, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&d, done, 8, hipMemcpyDeviceToHost);
    Hit *h = (Hit *)malloc(sizeof(Hit) * 4096);
    (void)hipMemcpy(h, hits, sizeof(Hit) * 4096, hipMemcpyDeviceToHost);
    const char *names[7] = {"v_pk_add_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel_hi:[1,0]", "v_pk_add_f32 plain pair", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_mov_b32 op_sel:[1,1]",
                            "v_pk_add_f32 op_sel:[1,0] (src0)", "v_pk_add_f16 op_sel:[0,1]"};
    printf("%s, %d probe blocks, sleep %d, burner %s (%d blocks x %d MFMAs per launch, %s): %.3g wave-iterations, %u wrong results\n", names[kind % 7], pblocks, sleep,
           burn ? "on" : "off", bblocks, biters, bsel >= 100 ? "burn.hip's kernel minus one ingredient" : (bsel == 99 ? "tools/burn.hip's kernel" : (bsel >= 0 ? "v_cndmask on the results in front of each" : "bare chain")), (double)d * 4.0, n);
    int q[4] = {0, 0, 0, 0}, lo = 0, hi = 0;
    for (unsigned k = 0; k < n && k < 4096; k++) { q[(h[k].lane & 63) >> 4]++; lo += (h[k].lane >> 6) & 1; hi += (h[k].lane >> 7) & 1; }
    if (n) {
        printf("  lanes 0-15: %d, 16-31: %d, 32-47: %d, 48-63: %d;  low result wrong: %d, high result wrong: %d\n", q[0], q[1], q[2], q[3], lo, hi);
        for (unsigned k = 0; k < n && k < 4; k++) {
            float lo, hi; memcpy(&lo, &h[k].got_lo, 4); memcpy(&hi, &h[k].got_hi, 4);
            const unsigned ln = h[k].lane & 63;
            printf("  iteration %u lane %u: result = (%.2f, %.2f); p = (%.0f, %.0f): the low lane subtracted %.2f\n", h[k].it, ln, lo, hi, 1.0f + ln, 2.0f + ln, 1.0f + ln - lo);
        }
    }
    return 0;
}
