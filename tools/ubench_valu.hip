// ubench_valu.hip -- VALU issue-rate microbenchmarks for gfx950 (MI355X).
// Establishes the slot model DESIGN.md uses for the VALU-bound NN kernels:
// how many lane-operations per clock per CU each instruction class sustains.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
typedef float float2_ __attribute__((ext_vector_type(2)));

#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1)
{
    float v[8];
    float2_ p[8];
    double dv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = threadIdx.x * 1e-3f + i; p[i] = float2_{v[i], v[i] + 1}; dv[i] = v[i]; }
    float a = s0, b = s1;
    float2_ pa = {s0, s1}, pb = {s1, s0};
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) {            // v_fma_f32, VGPR operands
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 1) {     // v_pk_fma_f32
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 2) {     // v_sub_f32 with SGPR operand
#define OP(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[i]) : "s"(s0));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 3) {     // v_min3_f32
#define OP(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 4) {     // v_pk_add_f32
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 5) {     // v_pk_mul_f32
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 6) {     // v_sqrt_f32 (transcendental)
#define OP(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 7) {     // v_add_f64
#define OP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dv[i]) : "v"((double)1.0));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 8) {     // v_cvt_f64_f32 + v_cvt_f32_f64 pair
#define OP(i) asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0" : "+v"(dv[i]), "+v"(v[i]));
            BODY8(OP)
#undef OP
        } else if (KIND == 9) {     // v_cmp_lt_f32 + v_cndmask_b32 pair
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(v[i]) : "v"(a), "v"(b) : "vcc");
            BODY8(OP)
#undef OP
        } else if (KIND == 10) {    // dependent chain: v_fma_f32 on ONE register
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(v[0]) : "v"(a), "v"(b));
        } else if (KIND == 11) {    // v_med3_f32
#define OP(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 12) {    // v_fmac_f32 e32 (2-operand encoding)
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP

        } else if (KIND == 13) {    // v_sub_f32 vgpr,vgpr
#define OP(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 14) {    // v_add_f32 vgpr,vgpr
#define OP(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 15) {    // v_mul_f32 vgpr,vgpr
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 16) {    // v_fma_f32 with an SGPR operand
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(s0), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 17) {    // v_min_f32
#define OP(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 18) {    // v_cmp_lt_f32 only
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(a) : "vcc");
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 19) {    // v_cndmask_b32 only
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : );
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 20) {    // v_mov_b32 v, s
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "s"(s0));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 21) {    // v_sub_f32 e64 with SGPR
#define OP(i) asm volatile("v_sub_f32_e64 %0, %1, %0" : "+v"(v[i]) : "s"(s0));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 22) {    // v_fmac_f32 with SGPR src0
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "s"(s0), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 23) {    // v_fma_f32 as subtraction: fma(q, -1.0, s)
#define OP(i) asm volatile("v_fma_f32 %0, %0, -1.0, %1" : "+v"(v[i]) : "s"(s0));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 24) {    // v_sub_f32 with inline constant
#define OP(i) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(v[i]));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 25) {    // v_mov_b32 v, v
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 26) {    // v_sub_f32 with SGPR, 16 independent destinations (no RAW within 8)
#define OP(i) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[i]) : "s"(s0), "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 27) {    // v_min3_f32 independent dest
#define OP(i) asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(a), "v"(b), "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 28) {    // v_max3_f32 / v_min_f32 mix: v_min_f32 e64
#define OP(i) asm volatile("v_min_f32_e64 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 29) {    // v_minimum3_f32 (gfx950, IEEE-754-2019 minimum)
#define OP(i) asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 30) {    // v_pk_min_f16
#define OP(i) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 31) {    // v_pk_minimum3_f16 (gfx950)
#define OP(i) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 32) {    // v_min3_i32
#define OP(i) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 33) {    // v_pk_min_i16
#define OP(i) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 34) {    // v_cvt_pkrtz_f16_f32
#define OP(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 35) {    // v_min_u32
#define OP(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 36) {    // v_perm_b32
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 37) {    // v_and_or_b32
#define OP(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 38) {    // v_min3_u32
#define OP(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 39) {    // v_min_i32
#define OP(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 40) {    // v_max_f32
#define OP(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 41) {    // v_max3_f32
#define OP(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 42) {    // v_min_u16
#define OP(i) asm volatile("v_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 43) {    // v_pk_min_u16
#define OP(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 44) {    // v_min_f32 dpp row_shr:1
#define OP(i) asm volatile("v_min_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 45) {    // v_min_f16
#define OP(i) asm volatile("v_min_f16 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 46) {    // v_add_f32_e64 clamp
#define OP(i) asm volatile("v_add_f32_e64 %0, %0, %1 clamp" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 47) {    // v_min_f64
#define OP(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(dv[i]) : "v"((double)1.0));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 48) {    // v_min3_f16
#define OP(i) asm volatile("v_min3_f16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 49) {    // v_min_u32 dpp quad_perm
#define OP(i) asm volatile("v_min_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 50) {    // v_max_u32
#define OP(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            BODY8(OP) BODY8(OP)
#undef OP
        } else if (KIND == 51) {    // v_min3_i16
#define OP(i) asm volatile("v_min3_i16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            BODY8(OP) BODY8(OP)
#undef OP
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += v[i] + p[i].x + p[i].y + (float)dv[i];
    if (acc == 12345.678f) out[0] = acc;
}

struct Case { const char *name; void (*fn)(float *, float, float); int ops_per_iter; int results_per_op; };

int main()
{
    float *d;
    CHECK(hipMalloc(&d, 1024));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    Case cases[] = {
        {"v_fma_f32 (vgpr)", k<0>, 16, 1}, {"v_pk_fma_f32", k<1>, 16, 2}, {"v_sub_f32 (sgpr src)", k<2>, 16, 1},
        {"v_min3_f32", k<3>, 16, 1}, {"v_pk_add_f32", k<4>, 16, 2}, {"v_pk_mul_f32", k<5>, 16, 2},
        {"v_sqrt_f32", k<6>, 16, 1}, {"v_add_f64", k<7>, 16, 1}, {"cvt f32<->f64 pair", k<8>, 16, 1},
        {"v_cmp+v_cndmask pair", k<9>, 16, 1}, {"v_fma_f32 dependent chain", k<10>, 16, 1},
        {"v_med3_f32", k<11>, 16, 1}, {"v_fmac_f32 e32", k<12>, 16, 1},
        {"v_sub_f32 vgpr", k<13>, 16, 1}, {"v_add_f32 vgpr", k<14>, 16, 1}, {"v_mul_f32 vgpr", k<15>, 16, 1},
        {"v_fma_f32 (sgpr src1)", k<16>, 16, 1}, {"v_min_f32", k<17>, 16, 1}, {"v_cmp_lt_f32", k<18>, 16, 1},
        {"v_cndmask_b32", k<19>, 16, 1}, {"v_mov_b32 v,s", k<20>, 16, 1}, {"v_sub_f32_e64 sgpr", k<21>, 16, 1},
        {"v_fmac_f32 sgpr src0", k<22>, 16, 1}, {"v_fma(q,-1,s)", k<23>, 16, 1}, {"v_sub_f32 1.0 const", k<24>, 16, 1},
        {"v_mov_b32 v,v", k<25>, 16, 1}, {"v_sub sgpr indep dst", k<26>, 16, 1}, {"v_min3 indep dst", k<27>, 16, 1},
        {"v_min_f32_e64", k<28>, 16, 1}, {"v_minimum3_f32", k<29>, 16, 1}, {"v_pk_min_f16", k<30>, 16, 2},
        {"v_pk_minimum3_f16", k<31>, 16, 2}, {"v_min3_i32", k<32>, 16, 1}, {"v_pk_min_i16", k<33>, 16, 2},
        {"v_cvt_pkrtz_f16_f32", k<34>, 16, 1}, {"v_min_u32", k<35>, 16, 1}, {"v_perm_b32", k<36>, 16, 1},
        {"v_and_or_b32", k<37>, 16, 1},
        // round 5: every remaining way to take a minimum
        {"v_min3_u32", k<38>, 16, 1},
        {"v_min_i32", k<39>, 16, 1},
        {"v_max_f32", k<40>, 16, 1},
        {"v_max3_f32", k<41>, 16, 1},
        {"v_min_u16", k<42>, 16, 1},
        {"v_pk_min_u16", k<43>, 16, 2},
        {"v_min_f32 dpp row_shr:1", k<44>, 16, 1},
        {"v_min_f16", k<45>, 16, 1},
        {"v_add_f32_e64 clamp", k<46>, 16, 1},
        {"v_min_f64", k<47>, 16, 1},
        {"v_min3_f16", k<48>, 16, 1},
        {"v_min_u32 dpp quad_perm", k<49>, 16, 1},
        {"v_max_u32", k<50>, 16, 1},
        {"v_min3_i16", k<51>, 16, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int waves_per_simd : {2, 8}) {
        int blocks = prop.multiProcessorCount * waves_per_simd;   // 256 thr = 4 waves = 1 per SIMD
        printf("--- %d wave(s) per SIMD (%d blocks x 256)\n", waves_per_simd, blocks);
        for (auto &c : cases) {
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double insts = (double)blocks * 4 /*waves*/ * ITERS * c.ops_per_iter;      // wave-instructions
            double lane_ops = insts * 64;
            double per_s = lane_ops / (best * 1e-3);
            printf("%-28s %8.3f ms  %7.2f T lane-inst/s  %6.2f lane-inst/clk/CU @2.4GHz  (x%d results)\n", c.name, best,
                   per_s * 1e-12, per_s / (prop.multiProcessorCount * 2.4e9), c.results_per_op);
        }
    }
    return 0;
}
