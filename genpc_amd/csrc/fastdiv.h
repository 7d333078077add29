// fastdiv.h -- correctly rounded fp32 division with a reciprocal shared between quotients.
//
// hipcc lowers `n / d` (fp32, denormals on) to
//     ds = v_div_scale(d), ns = v_div_scale(n)            -- exact power-of-two rescue of extreme exponents
//     r0 = v_rcp(ds);  e0 = fma(-ds, r0, 1);  r1 = fma(e0, r0, r0)
//     q0 = ns * r1;    e1 = fma(-ds, q0, ns); q1 = fma(e1, r1, q0)
//     e2 = fma(-ds, q1, ns);  q = v_div_fmas(e2, r1, q1)  -- fma, then the inverse power of two
//     v_div_fixup(q, d, n)                                -- zeros, infinities, NaNs, overflow
// When v_div_scale leaves both operands alone and v_div_fixup has nothing to fix, that is the five-operation
// core below on the refined reciprocal r1 -- which depends on d only, so quotients with a common denominator
// share it, and every step is a plain mul / fma that packs two quotients per instruction.  div_in_range() is a
// sufficient condition for "leaves alone / nothing to fix" (V_DIV_SCALE_F32 scales when an operand is zero or
// denormal, when 1/d or n/d would be denormal, when exponent(n) - exponent(d) >= 96 or exponent(n) <= 23): all
// magnitudes in [2^-50, 2^50].  Callers test it per wave and take the compiler's division otherwise.
// tests/test_gpu_fastdiv.py compares both over random and edge-case operands through genpc_fastdiv_probe.
#pragma once

namespace genpc {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float kDivLo = 8.8817841970012523e-16f;    // 2^-50
constexpr float kDivHi = 1125899906842624.0f;        // 2^50

__device__ __forceinline__ float rcp_refined(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}

__device__ __forceinline__ float div_core(float n, float d, float r1)
{
    const float q0 = n * r1;
    const float e1 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e2, r1, q1);
}

__device__ __forceinline__ v2f div_core2(v2f n, v2f d, v2f r1)
{
    const v2f q0 = n * r1;
    const v2f e1 = __builtin_elementwise_fma(-d, q0, n);
    const v2f q1 = __builtin_elementwise_fma(e1, r1, q0);
    const v2f e2 = __builtin_elementwise_fma(-d, q1, n);
    return __builtin_elementwise_fma(e2, r1, q1);
}

// running minimum / maximum of magnitudes for the range test (NaNs pass through min/max unnoticed: a NaN
// operand gives a NaN quotient on either path)
struct DivRange {
    float lo = kDivHi, hi = kDivLo;
    __device__ __forceinline__ void add(float a, float b) { lo = fminf(fminf(lo, fabsf(a)), fabsf(b)); hi = fmaxf(fmaxf(hi, fabsf(a)), fabsf(b)); }
    __device__ __forceinline__ void add(float a) { lo = fminf(lo, fabsf(a)); hi = fmaxf(hi, fabsf(a)); }
    __device__ __forceinline__ bool ok() const { return lo >= kDivLo && hi <= kDivHi; }
};

}  // namespace genpc
