// nn_f16.hip -- nearest-neighbour filter with ONE v_mfma_f32_32x32x16_f16 per 32x32 tile.
//
// Filter-and-prove scheme (the finish half is nn_finish.hip).  A split-bf16 variant would
// need two chained instructions per tile (three 8-bit pieces per operand, 27 products) and
// was retired in round 3.  An f16 piece carries 11 bits, so two pieces v = h + l (h = RN16(v), l = RN16(v - h), residual
// <= 2^-22 |v|) and all four products per coordinate fit the K = 16 of a single
// instruction:
//     lanes  0..31 (k 0..7):   A = [xh,xh,xl,xl, yh,yh,yl,yl]   B = [qxh,qxl,qxh,qxl, qyh,qyl,qyh,qyl]
//     lanes 32..63 (k 8..15):  A = [zh,zh,zl,zl, Th,Tl,0,0]     B = [qzh,qzl,qzh,qzl, 2^8,2^8,0,0]
// with x.. the pieces of -2 s x', q.. of s q', T of s^2 |t'|^2 2^-8.  f16 has a narrow
// exponent range, so each block scales by a power of two s (exact) that puts the largest
// centred coordinate of ITS queries and ITS target slice in [2^10, 2^11): products stay
// below 2^23, |t'|^2 s^2 2^-8 below 3 x 2^14, small coordinates may reach f16 subnormals,
// which the instruction does not flush (tools/ubench_mfma_f16.hip) -- their absolute
// error, <= 2^-25 in scaled units, is far below the bound.  The list values s^2 (|t'|^2 -
// 2 q'.t') are unscaled (two exact multiplications) when the lists are published.
//
// Error bound, u = 2^-24, T = max |t'|: residuals 4u per operand -> 16u |q'|T on
// -2 q'.t' and 4u T^2 on |t'|^2; the instruction's summation, measured at <= 3.1 u
// sum|terms| (adversarial cancellation, profiles/r01_ubench_mfma_f16.txt), budgeted at
// 6.5u (2|q'|T + T^2); three fp32 roundings each in |t'|^2 and |q'|^2; one unit on T^2
// for subnormal pieces:  E1 = u (30 |q'| T + 15 T^2 + 3 |q'|^2).
#include "nn.h"
#include <hip/hip_ext.h>
#include "../../include/genpc_hip.h"

#include <stdlib.h>

namespace genpc {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// tools/nn_timeline.py builds a private copy of the library with -DGENPC_NN_TIMELINE: thread 0 of every block stamps the
// shader clock at the phase boundaries of nn_f16_kernel (nothing of this exists in the shipped library)
#ifdef GENPC_NN_TIMELINE
__device__ unsigned long long g_timeline[4096 * 8];
#define GENPC_TL(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) { g_timeline[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); \
    if ((k) == 0) g_timeline[blockIdx.x * 8 + 6] = wall_clock64(); if ((k) == 5) g_timeline[blockIdx.x * 8 + 7] = wall_clock64(); } } while (0)
#else
#define GENPC_TL(k) do {} while (0)
#endif
constexpr int kHTile = 1024;           // targets per LDS tile: 2 planes x 16 B = 32 KiB (template parameter HT: 1024 or 2048)
constexpr double kQTh = 30.0, kTTh = 15.0;

__device__ __forceinline__ unsigned f16_bits(float v) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v); }

// v = h + l in f16 (v already scaled into range); returns h | l << 16
__device__ __forceinline__ unsigned split2(float v)
{
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

__device__ __forceinline__ unsigned dup_lo(unsigned v) { return __builtin_amdgcn_perm(v, v, 0x01000100u); }
__device__ __forceinline__ unsigned dup_hi(unsigned v) { return __builtin_amdgcn_perm(v, v, 0x03020302u); }

// largest power of two s with s * m < 2^11 (m > 0 finite), 1 otherwise
__device__ __forceinline__ float scale_for(float m)
{
    if (!(m > 0.0f) || !(m < __builtin_inff())) return 1.0f;
    const int e = __builtin_amdgcn_frexp_expf(m);       // m = f * 2^e, f in [0.5, 1)
    return __builtin_ldexpf(1.0f, 11 - e);
}

// One block = one (direction, target slice, batch, 128*Q-query block) unit; wave w owns Q
// tiles of 32 queries.  U = target tiles per bookkeeping unit, NL = candidate lists per lane.
// W = waves per SIMD the register budget is held to (Q = 4: 182-202 VGPRs natural -> 2 waves; W = 3
// caps at 168 with ~50-150 B of scratch: +1 us on a single-round launch, -4 % over many rounds)
// The lists go to nn_finish_kernel (nn_finish.hip), a second launch.  (Round 2 also carried a fused form -- the last
// slice block to arrive for a query block ran the finish step itself: 13x16384^2 313 -> 296 us but 1x16384^2
// 36 -> 59, and a cliff on scan-like clouds whose 512-query blocks overflow the block's work list -- and a
// Morton-sorted mode that culled whole (query block, slice) pairs: 336 -> 532 us on the 13 scans.  Both lost on
// the inputs that matter and were removed in round 3; DESIGN.md section 4.1 keeps the measurements.)
template <int Q, int U, int NL, int W, int HT, int WV>
__global__ __launch_bounds__(WV * kWave, W) void nn_f16_kernel(NNArgs a)
{
    // WV waves per block (4, or 8 for the wide single-round form: 1024 queries share one staged slice)
    constexpr int kThreads = WV * kWave;
    // Hazard: the accumulators are consumed by inline-asm v_min3, which the compiler's hazard
    // recognizer does not pad (an 8-pass MFMA result needs 11 wait states before a VALU
    // read).  The pipeline supplies them by construction when Q >= 2: between the MFMA that
    // writes acc[r] and the first read of acc[r] lie the other query tiles' slots, each 8
    // v_min3 + one MFMA (Q = 2: plus an explicit s_nop).  With Q = 1 the read follows at
    // once -- measured: 25 % wrong minima -- so that case is not instantiated.
    static_assert(Q >= 2, "one accumulator chain per wave violates the MFMA -> VALU wait states");
    constexpr int kC = 32 * U;
    constexpr int kRows = HT + 64;              // + two spare tiles: the pipeline fetches two tiles ahead
    __shared__ uint4 plane[2][kRows];
    __shared__ float s_red[WV];
    GENPC_TL(0);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31;
    int bid = blockIdx.x;
    const int d = (a.ndir > 1 && bid >= a.dir[1].block_begin) ? 1 : 0;
    const NNDir &D = a.dir[d];
    bid -= D.block_begin;
    const int qb = bid % D.qblocks;
    const int rest = bid / D.qblocks;
    const int batch = rest % a.b;
    const int slice = rest / a.b;

    const int nq = D.nq, nt = D.nt;
    const float *__restrict__ Qp = D.q + (size_t)batch * nq * 3;
    const float *__restrict__ T = D.t + (size_t)batch * nt * 3;
    // both clouds are centred on the first point of the direction-0 target cloud
    const float *cptr = a.dir[0].t + (size_t)batch * a.dir[0].nt * 3;
    const float cx = cptr[0], cy = cptr[1], cz = cptr[2];

    const int k_begin = slice * a.slice_len;
    int k_end = k_begin + a.slice_len;
    if (k_end > nt) k_end = nt;

    // centred queries of this lane; the block's largest |coordinate| over queries and slice
    float qc0[Q], qc1[Q];
    float mx = 0.0f;
    const int q0 = (qb * WV + wave) * (32 * Q) + col;
#pragma unroll
    for (int r = 0; r < Q; r++) {
        int j = q0 + r * 32;
        if (j >= nq) j = nq - 1;
        // lanes < 32 carry x and y of the query, lanes >= 32 z
        qc0[r] = Qp[(size_t)j * 3 + (half ? 2 : 0)] - (half ? cz : cx);
        qc1[r] = half ? 0.0f : Qp[(size_t)j * 3 + 1] - cy;
        // a query at infinity is answered by the exhaustive pass; it must not set the block's scale
        const float m0 = fabsf(qc0[r]), m1 = fabsf(qc1[r]);
        mx = fmaxf(mx, fmaxf(m0 < __builtin_inff() ? m0 : 0.0f, m1 < __builtin_inff() ? m1 : 0.0f));
    }
    // (A per-launch pre-pass that hands every block its batch element's scale -- so that a slice longer than an LDS tile is
    // not read twice -- was measured: 13 x 16384^2 280.5 -> 278.9 us, 4 x 16384 x 8192 54.0 -> 58.2: the pre-scan of a
    // multi-round launch hides behind the other resident blocks, the extra launch does not.)
    // raw coordinates of an LDS tile, HT / kBlock targets per thread.  A slice that fits one LDS tile (HT = 2048 at
    // 1 x 16384^2) is read from memory ONCE: the registers that feed the scale's maximum also feed the staging,
    // and the whole slice is staged behind a single barrier pair (with 1024-target tiles a block spent 43 % of its
    // time outside the MFMA loop: 3 us on this pass, 2 us re-reading tile 0, 2 us staging tile 1 between the loops
    // with every wave waiting for the slowest -- tools/nn_timeline.py).
    constexpr int kPer = HT / kThreads;
    float pre[kPer][3];
    // later copies of a bit-identical target (nn_dedupe.hip; bit i: this thread's i-th target of the tile) are staged like
    // padding: the first copy answers for them, and three equal minima in one list would send the query to the exhaustive pass
    const unsigned *__restrict__ dupm = D.dupmask ? D.dupmask + (D.dup_shared ? (size_t)0 : (size_t)batch * ((nt + 31) >> 5)) : nullptr;
    unsigned predup = 0u;
    auto prefetch = [&](int t0) {
        predup = 0u;
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            int t = t0 + i * kThreads + threadIdx.x;
            t = t < nt ? t : nt - 1;
#pragma unroll
            for (int k = 0; k < 3; k++) pre[i][k] = T[(size_t)t * 3 + k];
            if (dupm) predup |= ((dupm[t >> 5] >> (t & 31)) & 1u) << i;
        }
    };
    const bool resident = k_end - k_begin <= HT;        // block-uniform
    if (resident) {
        if (k_begin < k_end) prefetch(k_begin);
#pragma unroll
        for (int i = 0; i < kPer; i++)
            if (k_begin + i * kThreads + (int)threadIdx.x < k_end)
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(pre[i][0] - cx), fabsf(pre[i][1] - cy)), fabsf(pre[i][2] - cz)));
    } else {
        for (int t = k_begin + threadIdx.x; t < k_end; t += kThreads) {
            const float *tp = T + (size_t)t * 3;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(tp[0] - cx), fabsf(tp[1] - cy)), fabsf(tp[2] - cz)));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
#pragma unroll
    for (int w2 = 1; w2 < WV; w2++) mx = fmaxf(mx, s_red[w2]);
    GENPC_TL(1);
    const float sc = scale_for(mx);             // NaN input: comparisons fail -> 1
    const float isc = 1.0f / sc;                 // exact (power of two)
    const float tsc = -2.0f * sc;

    h16x8 bq[Q];
    Top3 lst[Q][NL];
#pragma unroll
    for (int r = 0; r < Q; r++) {
        const unsigned p0 = split2(qc0[r] * sc), p1 = split2(qc1[r] * sc);
        // [h,l,h,l | h,l,h,l]; lanes >= 32: z pieces, then 2^8 twice (f16 0x5c00), zeros
        const uint4 v = half ? make_uint4(p0, p0, 0x5c005c00u, 0u) : make_uint4(p0, p0, p1, p1);
        bq[r] = __builtin_bit_cast(h16x8, v);
#pragma unroll
        for (int n = 0; n < NL; n++) top3_init(lst[r][n]);
    }

    // One accumulator per query tile, all on the same 32-target tile.  The rows of the next
    // two tiles are in registers (set p = tile parity) while the current tile's
    // accumulators are reduced.
    f32x16 acc[Q];
    uint4 A[2];
    auto fetch = [&](int p, int row) { A[p] = plane[half][row + col]; };
    auto mm = [&](int p, int r) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, A[p]), bq[r], z, 0, 0, 0);
    };
    // One target tile (index parity p): fold the accumulators of tile `row` into the running
    // unit minima (four independent chains per query tile) and start tile row + 32 (set
    // p^1): slot r = 8 v_min3(acc[r]) | MFMA(r).  No branches: the last tile of an LDS tile
    // uses `last` (reduce only); fetches past the end read the spare rows.
    auto step = [&](auto p_tag, auto last_tag, int row, float (&m)[Q][4]) {
        constexpr int p = decltype(p_tag)::value;
        constexpr bool last = decltype(last_tag)::value;
        if (!last) fetch(p, row + 64);
        if (Q == 2) asm volatile("s_nop 3");      // 8 v_min3 + MFMA + ds_read = 10 wait states, 11 needed
#pragma unroll
        for (int r = 0; r < Q; r++) {
            const f32x16 &c = acc[r];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 16; e += 2)
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m[r][(e >> 1) & 3]) : "v"(m[r][(e >> 1) & 3]), "v"(c[e]), "v"(c[e + 1]));
            __builtin_amdgcn_sched_barrier(0);
            if (!last) mm(p ^ 1, r);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float tmax2 = 0.0f;
    float nf = 0.0f;      // NaN once a target of the slice had a non-finite coordinate
    if (!resident && k_begin < k_end) prefetch(k_begin);
    for (int t0 = k_begin; t0 < k_end && !(a.debug & 4); t0 += HT) {
        const int tn = min(HT, k_end - t0);
        const int tn_pad = (tn + kC - 1) / kC * kC;
        __syncthreads();                 // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const int t = i * kThreads + threadIdx.x;
            if (t < tn_pad) {
                // centre, scale, two f16 pieces of -2s x', -2s y', -2s z', s^2 |t'|^2 2^-8
                const float x = pre[i][0] - cx, y = pre[i][1] - cy, z = pre[i][2] - cz;
                const float xs = x * sc, ys = y * sc, zs = z * sc;
                const float tts = __fmaf_rn(zs, zs, __fmaf_rn(ys, ys, __fmul_rn(xs, xs)));
                const unsigned px = split2(x * tsc), py = split2(y * tsc), pz = split2(z * tsc);
                const unsigned pt = split2(tts * 0.00390625f);
                uint4 V0 = make_uint4(dup_lo(px), dup_hi(px), dup_lo(py), dup_hi(py));
                uint4 V1 = make_uint4(dup_lo(pz), dup_hi(pz), pt, 0u);
                if (t >= tn || ((predup >> i) & 1u)) {
                    // padding: |t'|^2 = +inf (f16 0x7c00) x 2^8 never wins, the other terms are 0
                    V0 = make_uint4(0u, 0u, 0u, 0u);
                    V1 = make_uint4(0u, 0u, 0x7c00u, 0u);
                } else {
                    const float tt = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
                    tmax2 = fmaxf(tmax2, tt);
                    nf = __fmaf_rn(tt, 0.0f, nf);        // inf x 0 = NaN, NaN sticks (v_max drops NaNs)
                }
                const int row = (t & ~31) | tile_row(t & 31);
                plane[0][row] = V0;
                plane[1][row] = V1;
            }
        }
        if (t0 + HT < k_end) prefetch(t0 + HT);
        __syncthreads();
        if (t0 == k_begin) GENPC_TL(2); else GENPC_TL(3);
        // prologue of the LDS tile: target tile 0, rows of tiles 0 and 1
        fetch(0, 0);
        fetch(1, 32);
#pragma unroll
        for (int r = 0; r < Q; r++) mm(0, r);
        asm volatile("s_nop 15");           // wait states before the first inline-asm read of acc[0]
        for (int rb0 = 0; rb0 < tn_pad; rb0 += NL * kC) {
#pragma unroll
            for (int n = 0; n < NL; n++) {
                const int rb = rb0 + n * kC;
                if (rb < tn_pad) {
                    float m[Q][4];
#pragma unroll
                    for (int r = 0; r < Q; r++) m[r][0] = m[r][1] = m[r][2] = m[r][3] = __builtin_inff();
#pragma unroll
                    for (int g = 0; g < U - 1; g++) {
                        if (g & 1) step(std::integral_constant<int, 1>{}, std::false_type{}, rb + 32 * g, m);
                        else step(std::integral_constant<int, 0>{}, std::false_type{}, rb + 32 * g, m);
                    }
                    // U is even: the last tile of a unit has parity 1
                    if (rb + kC < tn_pad) step(std::integral_constant<int, 1>{}, std::false_type{}, rb + kC - 32, m);
                    else step(std::integral_constant<int, 1>{}, std::true_type{}, rb + kC - 32, m);
#pragma unroll
                    for (int r = 0; r < Q; r++) {
                        float v;
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(v) : "v"(m[r][0]), "v"(m[r][1]), "v"(m[r][2]));
                        asm("v_min_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(m[r][3]));
                        top3_insert(lst[r][n], v, t0 + rb + half);      // scaled value; bit 0: which 16 rows of each tile
                    }
                }
            }
        }
    }

    GENPC_TL(4);
    // max |t'|^2 of the slice, for the bound of the finish step: every query block sees the same targets, the
    // first one publishes
    if (qb == 0) {
        // non-finite targets: +inf tells the finish step to answer every query of this cloud
        // exhaustively (the reference's tile semantics for NaNs, nn_exhaustive in nn.h)
        if (nf != nf) tmax2 = __builtin_inff();
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tmax2 = fmaxf(tmax2, __shfl_xor(tmax2, o));
        __syncthreads();
        if (lane == 0) s_red[wave] = tmax2;
        __syncthreads();
        float tm = s_red[0];
#pragma unroll
        for (int w2 = 1; w2 < WV; w2++) tm = fmaxf(tm, s_red[w2]);
        if (threadIdx.x == 0) D.tmaxp[(size_t)batch * D.slices + slice] = tm;
    }

    // Publish the lists: the two lane halves folded, NL lists of two 8-byte words per query and slice
    // (a1, c1) (codes of a2 and a3, c2): list_enc in nn.h.
    const size_t bnq = (size_t)a.b * nq;
    unsigned long long *P = D.part + (size_t)batch * nq;
#pragma unroll
    for (int r = 0; r < Q; r++) {
#pragma unroll
        for (int n = 0; n < NL; n++) {
            Top3 &f = lst[r][n];
            // unscale (the block's scale is one power of two, so the order was not affected):
            // s^2 (|t'|^2 - 2 q'.t') -> |t'|^2 - 2 q'.t', two exact multiplications
            f.a1 = (f.a1 * isc) * isc;
            f.a2 = (f.a2 * isc) * isc;
            f.a3 = (f.a3 * isc) * isc;
            const float o1 = __shfl_xor(f.a1, 32), o2 = __shfl_xor(f.a2, 32), o3 = __shfl_xor(f.a3, 32);
            const int oc1 = __shfl_xor(f.c1, 32), oc2 = __shfl_xor(f.c2, 32);
            top3_insert(f, o1, oc1);
            top3_insert(f, o2, oc2);
            top3_insert(f, o3, -1);
            const int j = q0 + r * 32;
            if (!half && j < nq) {
                unsigned long long *p = P + (size_t)(slice * NL + n) * 2 * bnq + j;
                const unsigned code2 = list_enc(f.a1, f.a2);
                const unsigned code3 = list_enc(list_dec(f.a1, code2), f.a3);
                p[0] = ((unsigned long long)__float_as_uint(f.a1) << 32) | (unsigned)f.c1;
                p[bnq] = ((unsigned long long)((code2 << 16) | code3) << 32) | (unsigned)f.c2;
            }
        }
    }
    GENPC_TL(5);
}

template <int Q, int NL>
static void launch_main(const NNArgs &a, int blocks, int u, int tight, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    // (e0 / e1: genpc_nn_profile -- the events ride on the kernel's own dispatch, hipExtLaunchKernelGGL: they take the dispatch's
    //  start and end timestamps, what a rocprofv3 kernel trace reports, not the interval between two marker packets around it)
#define NN_F16_LAUNCH(KERNEL, GRID, BLOCK)                                                                              \
    do {                                                                                                                \
        if (e0) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, e0, e1, 0, a);                                        \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, a);                                                         \
    } while (0)
    // tight 1: three waves per SIMD for the 512-query blocks (planner: launches of several rounds)
    // tight 2 (planner: single round, Q = 4, slice_len > 1024): 8-wave blocks of 1024 queries, one per CU
    // HT = 2048 (66 KiB of LDS per block; a slice of up to 2048 targets is resident) where two waves per SIMD are all the
    // registers allow anyway: Q = 4, not tight, slice_len > 1024 (4 x 16384 x 8192, slices of 4096: 57.9 -> 54.7 us)
    static const int env_ht = tune_env("GENPC_NN_HT", 0, "f16 filter: 1024 = never stage 2048-target LDS tiles");
    const bool big = Q == 4 && tight != 1 && a.slice_len > kHTile && env_ht != 1024;
    if (Q == 4 && tight == 2) {
        if (u == 2) NN_F16_LAUNCH((nn_f16_kernel<4, 2, NL, 1, 2 * kHTile, 8>), dim3(blocks), dim3(8 * kWave));
        else NN_F16_LAUNCH((nn_f16_kernel<4, 4, NL, 1, 2 * kHTile, 8>), dim3(blocks), dim3(8 * kWave));
        return;
    }
    if (big) {
        if (u == 2) NN_F16_LAUNCH((nn_f16_kernel<4, 2, NL, 2, 2 * kHTile, 4>), dim3(blocks), dim3(kBlock));
        else NN_F16_LAUNCH((nn_f16_kernel<4, 4, NL, 2, 2 * kHTile, 4>), dim3(blocks), dim3(kBlock));
        return;
    }
    if (u == 2) {
        if (tight) NN_F16_LAUNCH((nn_f16_kernel<Q, 2, NL, (Q == 4 ? 3 : 4), kHTile, 4>), dim3(blocks), dim3(kBlock));
        else NN_F16_LAUNCH((nn_f16_kernel<Q, 2, NL, (Q == 4 ? 2 : 4), kHTile, 4>), dim3(blocks), dim3(kBlock));
    } else {
        if (tight) NN_F16_LAUNCH((nn_f16_kernel<Q, 4, NL, (Q == 4 ? 3 : 4), kHTile, 4>), dim3(blocks), dim3(kBlock));
        else NN_F16_LAUNCH((nn_f16_kernel<Q, 4, NL, (Q == 4 ? 2 : 4), kHTile, 4>), dim3(blocks), dim3(kBlock));
    }
#undef NN_F16_LAUNCH
}

// genpc_nn_profile(): HIP events ON the filter kernel's dispatch (bench.py's roofline line).  Per calling host thread (round 4
// kept this state process-wide: a lane that profiled raced every other lane's launches -- VERDICT r4 weak #13): only the
// launches of the thread that asked are bracketed, with that thread's own pair of events.
static thread_local bool g_prof_on = false;
static thread_local hipEvent_t g_prof_e0 = nullptr, g_prof_e1 = nullptr;

// Launches the filter and the finish kernel.  q / u / nl as chosen by the planner in chamfer.hip.
int launch_nn_f16(NNArgs &a, int q, int u, int nl, int tight, long long total_blocks, hipStream_t st)
{
    size_t bytes = 0;
    size_t off_t[2];
    for (int d = 0; d < a.ndir; d++) {
        a.dir[d].ntmax = a.dir[d].slices;
        off_t[d] = bytes;
        bytes += ((size_t)a.b * a.dir[d].ntmax * sizeof(float) + 255) & ~(size_t)255;
    }
    char *ws = (char *)workspace(9, bytes, st);
    if (!ws) return 0;
    for (int d = 0; d < a.ndir; d++) a.dir[d].tmaxp = (float *)(ws + off_t[d]);
    const int blocks = (int)total_blocks;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (g_prof_on) {
        if (!g_prof_e0) { (void)hipEventCreate(&g_prof_e0); (void)hipEventCreate(&g_prof_e1); }
        pe0 = g_prof_e0;
        pe1 = g_prof_e1;
    }
    // (256-target bookkeeping units, U = 8, in the wide form: filter -0.4 us, finish +1.7 us at 1 x 16384^2 -- not kept)
    if (q == 4) {
        if (nl == 2) launch_main<4, 2>(a, blocks, u, tight, st, pe0, pe1);
        else launch_main<4, 1>(a, blocks, u, tight, st, pe0, pe1);
    } else {
        if (nl == 2) launch_main<2, 2>(a, blocks, u, 0, st, pe0, pe1);
        else launch_main<2, 1>(a, blocks, u, 0, st, pe0, pe1);
    }
    if (!check(hipGetLastError(), "nn_f16_kernel launch")) return 0;
    return launch_nn_finish(a, nl, u, (float)kQTh, (float)kTTh, 0.0f, st);
}

}  // namespace genpc

#ifdef GENPC_NN_TIMELINE
extern "C" __attribute__((visibility("default"))) int genpc_nn_timeline_read(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(genpc::g_timeline), sizeof(unsigned long long) * 4096 * 8) == hipSuccess;
}
#endif

GENPC_API float genpc_nn_profile(int enable)
{
    using namespace genpc;
    float ms = -1.0f;
    if (g_prof_on && g_prof_e1 && hipEventSynchronize(g_prof_e1) == hipSuccess) (void)hipEventElapsedTime(&ms, g_prof_e0, g_prof_e1);
    g_prof_on = enable != 0;
    return ms;
}
