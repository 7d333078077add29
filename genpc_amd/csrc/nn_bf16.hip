// nn_bf16.hip -- nearest-neighbour filter on the bf16 matrix pipe of gfx950.
//
// Same scheme as the fp32-MFMA path in chamfer.hip (approximate |t'|^2 - 2 q'.t' for
// every pair, keep the smallest unit minima per query, prove with an error bound that
// the reference's nearest neighbour lies in the listed tiles, evaluate those with the
// reference's exact arithmetic), but the approximation runs on
// v_mfma_f32_32x32x16_bf16, 16x the fp32 matrix rate, and -- measured,
// tools/ubench_mfma_bf16.hip -- VALU instructions issue underneath it, which they do
// not under the fp32 MFMA (tools/ubench_mfma.hip: times add).
//
// Precision comes from splitting every fp32 operand into three bf16 pieces
// v = h + m + l (h = RN16(v), m = RN16(v - h), l = RN16(v - h - m); residual
// <= 2^-27 |v|) and keeping all products but l x l:
//     t.q ~ th(qh+qm+ql) + tm(qh+qm+ql) + tl(qh+qm)        8 terms per coordinate
// 3 coordinates x 8 + (tth, ttm, ttl) x 1 = 27 of the K = 32 slots of two chained
// instructions.  Products of bf16 pairs are exact in fp32; the instruction's
// internal summation was measured at <= 3.2 u sum|terms| per K = 16 step
// (u = 2^-24, adversarial cancellation; profiles/r01_ubench_mfma_bf16.txt); the bound
// below budgets 6.5 u per step, i.e. 13 u (2|q'||t'| + |t'|^2) for the chain, plus
// the three roundings of |t'|^2 and of |q'|^2:  E1 = u (27 |q'| T + 17 T^2 + 3 |q'|^2).
//
// Operands: per target four 16-byte vectors X|Y|Z|T, X = [h,h,h,m,m,m,l,l] of -2x'
// etc., T = [tth,ttm,ttl,0...]; per query X|Y|Z = [h,m,l,h,m,l,h,m] of x' etc. (the
// fourth B vector is the constant [1,1,1,0...]).  Lanes 0..31 feed k = 0..7 of an
// instruction, lanes 32..63 k = 8..15: instruction 0 gets X (lanes < 32) and Y,
// instruction 1 Z and T.  Both clouds are centred on the first point of the
// direction-0 target cloud.  nn_split_kernel (one launch for both clouds) writes the
// query vectors as they are used and the targets as the 12 pieces only (24 B, three
// 8-byte planes): every block re-streams its target slice from L2, and at 64 B per
// target that traffic, not the matrix pipe, bounded the first version (1.7 GB for
// 13 x 16384^2).  The four vectors are rebuilt with 12 v_perm_b32 per target while a
// tile is staged into LDS; the next tile's pieces are prefetched into registers.
#include "nn.h"

#include <stdlib.h>
#include <type_traits>

namespace genpc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef GENPC_BTILE
#define GENPC_BTILE 512
#endif
constexpr int kBTile = GENPC_BTILE;    // targets per LDS tile: 4 planes x 16 B = 32 KiB at 512
constexpr double kQT16 = 27.0, kTT16 = 17.0;

__device__ __forceinline__ unsigned bf16_rn(float v)      // v_cvt_pk_bf16_f32 (round to nearest even)
{
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
}

__device__ __forceinline__ void split3(float v, unsigned &h, unsigned &m, unsigned &l)
{
    h = bf16_rn(v);
    const float r1 = v - __uint_as_float(h << 16);
    m = bf16_rn(r1);
    const float r2 = r1 - __uint_as_float(m << 16);
    l = bf16_rn(r2);
}

__device__ __forceinline__ uint4 a_vec(float v)    // [h,h,h,m,m,m,l,l]
{
    unsigned h, m, l;
    split3(v, h, m, l);
    return make_uint4(h | (h << 16), h | (m << 16), m | (m << 16), l | (l << 16));
}

__device__ __forceinline__ uint4 b_vec(float v)    // [h,m,l,h,m,l,h,m]
{
    unsigned h, m, l;
    split3(v, h, m, l);
    return make_uint4(h | (m << 16), l | (h << 16), m | (l << 16), h | (m << 16));
}

// ---------------------------------------------------------------------------
// Pre-split path (large launches): the four operand vectors of every target are written
// once (64 B per target, planes X|Y|Z|T of [B][ntp] rows, ntp = nt rounded up to 128 with
// padding rows that can never win) and the filter kernel streams them into LDS with
// global_load_lds_dwordx4, no VGPRs or VALU on the way.  Without it every query block
// re-derives the vectors of the targets it stages (~50 VALU ops per target).
struct SplitJob {
    const float *src;     // targets [B, n, 3]
    uint4 *out;           // [B][4][np]
    float *tmax;          // per-block max |t'|^2 [B][bpb]
    int n, np, bpb, block_begin;
};

struct SplitArgs {
    SplitJob job[2];
    int njobs;
    const float *centre;  // direction-0 targets [B, cn, 3]
    int cn;
};

__global__ __launch_bounds__(kBlock) void nn_split_kernel(SplitArgs a)
{
    __shared__ float s_red[kWavesPerBlock];
    const int j = (a.njobs > 1 && (int)blockIdx.x >= a.job[1].block_begin) ? 1 : 0;
    const SplitJob &J = a.job[j];
    const int local = blockIdx.x - J.block_begin;
    const int batch = local / J.bpb, blk = local % J.bpb;
    const int i = blk * kBlock + threadIdx.x;
    const float *c = a.centre + (size_t)batch * a.cn * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    float tt = 0.0f;
    if (i < J.np) {
        uint4 X = make_uint4(0u, 0u, 0u, 0u), Y = X, Z = X, W = make_uint4(0x7f80u, 0u, 0u, 0u);   // padding row
        if (i < J.n) {
            const float *p = J.src + ((size_t)batch * J.n + i) * 3;
            const float x = p[0] - cx, y = p[1] - cy, z = p[2] - cz;
            tt = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
            unsigned th, tm, tl;
            split3(tt, th, tm, tl);
            X = a_vec(-2.0f * x);
            Y = a_vec(-2.0f * y);
            Z = a_vec(-2.0f * z);
            W = make_uint4(th | (tm << 16), tl, 0u, 0u);
        }
        uint4 *o = J.out + (size_t)batch * 4 * J.np + ((i & ~31) | tile_row(i & 31));
        o[0] = X;
        o[(size_t)J.np] = Y;
        o[(size_t)2 * J.np] = Z;
        o[(size_t)3 * J.np] = W;
    }
    float m = (tt * 0.0f != 0.0f) ? __builtin_inff() : tt;      // non-finite target: +inf (see nn_finish_kernel)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & (kWave - 1)) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) J.tmax[(size_t)batch * J.bpb + blk] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ float min16(float m, const f32x16 &c)
{
    // raw v_min3: fminf() would first canonicalise every MFMA output (v_max x,x)
#pragma unroll
    for (int i = 0; i < 16; i += 2) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(c[i]), "v"(c[i + 1]));
    return m;
}

// One block = one (direction, target slice, batch, 128*Q-query block) unit; wave w
// owns Q tiles of 32 queries (Q = 4: a wave reads 2 KiB of LDS per target tile, so the
// LDS port -- 128 B/clk/CU -- would cap Q = 1 at the MFMA rate of ONE query tile).
// U = target tiles per bookkeeping unit, NL = candidate lists per lane (units are
// dealt round-robin to the lists).
template <int Q, int U, int NL, int FMA, int PRE>
__global__ __launch_bounds__(kBlock) void nn_bf16_kernel(NNArgs a)
{
    static_assert(Q >= 2, "one accumulator chain per wave violates the MFMA -> VALU wait states (nn_f16.hip)");
    constexpr int kC = 32 * U;
    constexpr int kRows = kBTile + 64;          // + two spare tiles: the pipeline fetches two tiles ahead
    __shared__ uint4 plane[PRE ? 2 : 1][4][kRows];   // pre-split path: double buffered, filled by LDS-DMA
    const uint4 *pb = &plane[0][0][0];          // buffer holding the current LDS tile
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

    bf16x8 b0[Q], b1[Q];
    Top3 lst[Q][NL];
    const int q0 = (qb * kWavesPerBlock + wave) * (32 * Q) + col;
#pragma unroll
    for (int r = 0; r < Q; r++) {
        int j = q0 + r * 32;
        if (j >= nq) j = nq - 1;
        // lanes < 32 carry the X and Z vectors of the query, lanes >= 32 Y and the ones
        const float c0 = Qp[(size_t)j * 3 + half] - (half ? cy : cx);
        const float c1 = Qp[(size_t)j * 3 + 2] - cz;
        const uint4 v0 = b_vec(c0);
        const uint4 v1 = half ? make_uint4(0x3f803f80u, 0x00003f80u, 0u, 0u) : b_vec(c1);
        b0[r] = __builtin_bit_cast(bf16x8, v0);
        b1[r] = __builtin_bit_cast(bf16x8, v1);
#pragma unroll
        for (int n = 0; n < NL; n++) top3_init(lst[r][n]);
    }

    // One accumulator chain per query tile, all on the same 32-target tile.  The rows of
    // the next two tiles are in registers (set p = tile parity) while the current
    // tile's accumulators are reduced.
    f32x16 acc[Q];
    uint4 A0[2], A1[2];
    auto fetch = [&](int p, int row) {
        A0[p] = pb[half * kRows + row + col];
        A1[p] = pb[(2 + half) * kRows + row + col];
    };
    auto m1 = [&](int p, int r) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A0[p]), b0[r], z, 0, 0, 0);
    };
    auto m2 = [&](int p, int r) {
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[p]), b1[r], acc[r], 0, 0, 0);
    };
    // One target tile (index parity p): the accumulators hold tile `row`; while they are
    // folded into the running unit minima m[] the chains of tile row + 32 (set p^1) are
    // started, skewed by one slot so that no MFMA waits on its predecessor:
    //   slot r:  8 v_min3(acc[r]) | M2(r-1) | M1(r)
    // and the rows of tile row + 64 are fetched into set p.  No branches: the last tile
    // of an LDS tile uses `last` (reduce only); fetches past the end read the spare rows.
    auto step = [&](auto p_tag, auto last_tag, int row, float (&m)[Q][4]) {
        constexpr int p = decltype(p_tag)::value;
        constexpr bool last = decltype(last_tag)::value;
        if (!last) fetch(p, row + 64);
        if (Q == 2) asm volatile("s_nop 3");      // MFMA -> inline-asm VALU read needs 11 wait states (nn_f16.hip)
#pragma unroll
        for (int r = 0; r < Q; r++) {
            const f32x16 &c = acc[r];
            __builtin_amdgcn_sched_barrier(0);
            // four independent minimum chains: back-to-back dependent VALU ops of one wave
            // do not issue at rate
#pragma unroll
            for (int e = 0; e < 16; e += 2)
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m[r][(e >> 1) & 3]) : "v"(m[r][(e >> 1) & 3]), "v"(c[e]), "v"(c[e + 1]));
            __builtin_amdgcn_sched_barrier(0);
            if (!last) {
                if (r > 0) m2(p ^ 1, r - 1);
                m1(p ^ 1, r);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!last) m2(p ^ 1, Q - 1);
    };

    // raw coordinates of the next LDS tile, kBTile / kBlock targets per thread
    constexpr int kPer = kBTile / kBlock;
    float pre[kPer][3];
    auto prefetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            int t = t0 + i * kBlock + threadIdx.x;
            t = t < nt ? t : nt - 1;
#pragma unroll
            for (int k = 0; k < 3; k++) pre[i][k] = T[(size_t)t * 3 + k];
        }
    };
    float tmax2 = 0.0f;
    float nf = 0.0f;      // NaN once a target of the slice had a non-finite coordinate
    // pre-split path: wave w copies plane w of a tile, 64 rows (1 KiB) per instruction
    const uint4 *__restrict__ AR = D.arec + ((size_t)batch * 4 + wave) * D.ntp;
    auto dma = [&](int buf, int t0, int rows) {
#pragma unroll
        for (int i = 0; i < kBTile / 64; i++) {
            if (i * 64 < rows)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(AR + t0 + i * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)&plane[buf][wave][i * 64], 16, 0, 0);
        }
    };
    auto rows_of = [&](int t0) {
        const int tn = min(kBTile, k_end - t0);
        return (tn + kC - 1) / kC * kC;
    };
    if (PRE) {
        if (k_begin < k_end) dma(0, k_begin, rows_of(k_begin));
    } else if (k_begin < k_end) {
        prefetch(k_begin);
    }
    int tile_i = 0;
    for (int t0 = k_begin; t0 < k_end && !(a.debug & 4); t0 += kBTile) {
        const int tn = min(kBTile, k_end - t0);
        const int tn_pad = (tn + kC - 1) / kC * kC;
        if (PRE) {
            // this tile's DMAs (issued one tile ago) have landed; everyone is done with the other buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            pb = &plane[tile_i & 1][0][0];
            if (t0 + kBTile < k_end) dma((tile_i & 1) ^ 1, t0 + kBTile, rows_of(t0 + kBTile));
            tile_i++;
        } else if (!(a.debug & 64)) {
            __syncthreads();                 // every wave is done reading the previous tile
        }
#pragma unroll
        for (int i = 0; i < (PRE ? 0 : kPer); i++) {
            const int t = i * kBlock + threadIdx.x;
            if (t < tn_pad && !(a.debug & 64)) {
                // centre, |t'|^2, three bf16 pieces of -2x', -2y', -2z', |t'|^2, the four operand vectors
                const float x = pre[i][0] - cx, y = pre[i][1] - cy, z = pre[i][2] - cz;
                const float tt = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
                unsigned th, tm, tl;
                split3(tt, th, tm, tl);
                uint4 X = a_vec(-2.0f * x), Y = a_vec(-2.0f * y), Z = a_vec(-2.0f * z);
                uint4 W = make_uint4(th | (tm << 16), tl, 0u, 0u);
                if (t >= tn) {
                    // padding: |t'|^2 = +inf (bf16 0x7f80) x 1 never wins, the other terms are 0
                    X = Y = Z = make_uint4(0u, 0u, 0u, 0u);
                    W = make_uint4(0x7f80u, 0u, 0u, 0u);
                } else {
                    tmax2 = fmaxf(tmax2, tt);
                    nf = __fmaf_rn(tt, 0.0f, nf);
                }
                const int row = (t & ~31) | tile_row(t & 31);
                plane[0][0][row] = X;
                plane[0][1][row] = Y;
                plane[0][2][row] = Z;
                plane[0][3][row] = W;
            }
        }
        if (!PRE) {
            if (t0 + kBTile < k_end && !(a.debug & 64)) prefetch(t0 + kBTile);
            if (!(a.debug & 64)) __syncthreads();
        }
        // prologue of the LDS tile: chains of target tile 0, rows of tiles 0 and 1
        fetch(0, 0);
        fetch(1, 32);
#pragma unroll
        for (int r = 0; r < Q; r++) m1(0, r);
#pragma unroll
        for (int r = 0; r < Q; r++) m2(0, r);
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
                        float mm;
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(mm) : "v"(m[r][0]), "v"(m[r][1]), "v"(m[r][2]));
                        asm("v_min_f32 %0, %1, %2" : "=v"(mm) : "v"(mm), "v"(m[r][3]));
                        if (!(a.debug & 128)) top3_insert(lst[r][n], mm, t0 + rb + half);      // bit 0: which 16 rows of each tile
                        else lst[r][n].a1 = mm;
                    }
                }
            }
        }
    }

    // max |t'|^2 of the slice, for the bound in nn_finish_kernel (every query block sees
    // the same targets: the first one publishes)
    if (!PRE && qb == 0) {
        __shared__ float s_red[kWavesPerBlock];
        if (nf != nf) tmax2 = __builtin_inff();      // non-finite targets: every query of the cloud goes exhaustive
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tmax2 = fmaxf(tmax2, __shfl_xor(tmax2, o));
        if (lane == 0) s_red[wave] = tmax2;
        __syncthreads();
        if (threadIdx.x == 0)
            D.tmaxp[(size_t)batch * D.slices + slice] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    }

    // Publish the lists: the two lane halves folded, NL lists of three 8-byte words
    // (a1,c1) (a2,c2) (a3,-) per query and slice, for nn_finish_kernel.
    const size_t bnq = (size_t)a.b * nq;
    unsigned long long *P = D.part + (size_t)batch * nq;
#pragma unroll
    for (int r = 0; r < Q; r++) {
#pragma unroll
        for (int n = 0; n < NL; n++) {
            Top3 &f = lst[r][n];
            const float o1 = __shfl_xor(f.a1, 32), o2 = __shfl_xor(f.a2, 32), o3 = __shfl_xor(f.a3, 32);
            const int oc1 = __shfl_xor(f.c1, 32), oc2 = __shfl_xor(f.c2, 32);
            top3_insert(f, o1, oc1);
            top3_insert(f, o2, oc2);
            top3_insert(f, o3, -1);
            const int j = q0 + r * 32;
            if (!half && j < nq) {
                unsigned long long *p = P + (size_t)(slice * NL + n) * 3 * bnq + j;
                p[0] = ((unsigned long long)__float_as_uint(f.a1) << 32) | (unsigned)f.c1;
                p[bnq] = ((unsigned long long)__float_as_uint(f.a2) << 32) | (unsigned)f.c2;
                p[2 * bnq] = (unsigned long long)__float_as_uint(f.a3) << 32;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Second launch: one block per 64 queries (4 threads per query).  Gathers the query's
// lists, derives the acceptance threshold tau from the smallest approximate value,
// evaluates every listed tile that is not provably out with the reference's exact
// arithmetic (work items of 32 targets, spread over the block), and writes
// (distance, first index).  A query with a list whose THIRD minimum is not provably out
// (or with non-finite values) is re-done exhaustively by the block.
constexpr int kFQ = 64;                // queries per finish block
constexpr int kFWork = 2048;           // work-item capacity (64 queries x 32 pieces)

template <int FMA>
__global__ __launch_bounds__(kBlock) void nn_finish_kernel(NNArgs a, int nl, int upieces, float kqt, float ktt, float t2min)
{
    __shared__ unsigned long long s_best[kFQ];   // (distance bits << 32 | index): atomic min == (distance, first index)
    __shared__ float4 s_q[kFQ];
    __shared__ float s_a[4][kFQ];
    __shared__ int s_qflag[kFQ];
    __shared__ float s_tau[kFQ], s_qq[kFQ];
    __shared__ int s_flagged[kFQ];
    __shared__ unsigned s_work[kFWork];          // query slot << 22 | tile (first target / 32) << 1 | lane half
    __shared__ float s_red[kWavesPerBlock];
    __shared__ int s_fi[kWavesPerBlock];
    __shared__ int s_misc[2];                    // work items, flagged queries
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    const int d = (a.ndir > 1 && bid >= a.dir[1].fin_begin) ? 1 : 0;
    const NNDir &D = a.dir[d];
    bid -= D.fin_begin;
    const int nq = D.nq, nt = D.nt;
    const int fblocks = (nq + kFQ - 1) / kFQ;
    const int batch = bid / fblocks, fb = bid % fblocks;
    const float *__restrict__ Qp = D.q + (size_t)batch * nq * 3;
    const float *__restrict__ T = D.t + (size_t)batch * nt * 3;
    float *__restrict__ od = D.out_d + (size_t)batch * nq;
    int *__restrict__ oi = D.out_i + (size_t)batch * nq;
    const size_t bnq = (size_t)a.b * nq;
    const unsigned long long *P = D.part + (size_t)batch * nq;
    const int nlists = D.slices * nl;

    // (a finish block's 64 queries lie in one 512-query filter block)
    const NNSortDev *S = a.srt;
    const unsigned needmask = S ? S->need[d][(size_t)batch * D.qblocks + (fb * kFQ) / 512] : 0xffffffffu;
    const int *__restrict__ permT = S ? S->perm_t[d] + (size_t)batch * nt : nullptr;
    const int ql = threadIdx.x & (kFQ - 1), part = threadIdx.x >> 6;     // part == wave
    int j = fb * kFQ + ql;
    const bool live = j < nq;
    j = live ? j : nq - 1;

    const float *cptr = a.dir[0].t + (size_t)batch * a.dir[0].nt * 3;      // common centre of the filter
    const float ccx = cptr[0], ccy = cptr[1], ccz = cptr[2];
    // this thread's lists: li = part + 4k
    unsigned long long w0[kMaxLists / 4], w1[kMaxLists / 4], w2[kMaxLists / 4];
    float amin = __builtin_inff();
#pragma unroll
    for (int k = 0; k < kMaxLists / 4; k++) {
        const int li = part + 4 * k;
        w0[k] = w1[k] = w2[k] = 0x7f800000ull << 32;      // (+inf, 0)
        // sorted mode: a slice whose filter block never ran has no lists (its targets are provably too far)
        if (li < nlists && ((needmask >> (li / nl)) & 1u)) {
            const unsigned long long *p = P + (size_t)li * 3 * bnq + j;
            w0[k] = p[0];
            w1[k] = p[bnq];
            w2[k] = p[2 * bnq];
        }
    }
#pragma unroll
    for (int k = 0; k < kMaxLists / 4; k++) amin = fminf(amin, __uint_as_float((unsigned)(w0[k] >> 32)));
    s_a[part][ql] = amin;
    // max |t'|^2 over the whole target cloud (per-slice maxima of the filter kernel)
    float tmax2 = 0.0f;
    {
        const float *tp = D.tmaxp + (size_t)batch * D.ntmax;
        for (int i = threadIdx.x; i < D.ntmax; i += kBlock) tmax2 = fmaxf(tmax2, tp[i]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tmax2 = fmaxf(tmax2, __shfl_xor(tmax2, o));
        if (lane == 0) s_red[wave] = tmax2;
    }
    if (part == 0) {
        s_q[ql] = make_float4(Qp[(size_t)j * 3 + 0], Qp[(size_t)j * 3 + 1], Qp[(size_t)j * 3 + 2], 0.0f);
        s_best[ql] = ~0ull;
        s_qflag[ql] = 0;
    }
    if (threadIdx.x == 0) { s_misc[0] = 0; s_misc[1] = 0; }
    __syncthreads();
    tmax2 = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float abest = fminf(fminf(s_a[0][ql], s_a[1][ql]), fminf(s_a[2][ql], s_a[3][ql]));
    // the threshold is fp64 arithmetic (three square roots): once per query, shared through LDS
    float qq = 0.0f;
    if (part == 0) {
        const float4 qv = s_q[ql];
        const float x = qv.x - ccx, y = qv.y - ccy, z = qv.z - ccz;
        qq = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
        float t = nn_tau(abest, qq, tmax2, (double)kqt, (double)ktt);
        if (!(tmax2 >= t2min)) t = __builtin_nanf("");      // below the magnitudes the filter's bound covers: exhaustive
        // non-finite targets anywhere in the cloud (the filter publishes +inf) or a non-finite query:
        // the reference's result depends on its 512-target tiling, only nn_exhaustive reproduces it
        if (!(tmax2 < __builtin_inff()) || !(qq < __builtin_inff())) t = __builtin_nanf("");
        if (a.debug & 16) t = __builtin_inff();          // test hook: every listed tile is evaluated
        s_tau[ql] = t;
        s_qq[ql] = qq;
    }
    __syncthreads();
    const float tau = s_tau[ql];
    qq = s_qq[ql];
    bool flag = (a.debug & 8) != 0 || !(tau == tau);   // test hook / non-finite input: exhaustive pass
    int ncand = 0;
    // a listed tile whose minimum is not provably out becomes work items of 32 targets
    auto consider = [&](float av, int c) {
        if (av <= tau) {
            ncand++;
            if (c < 0) {
                flag = true;
            } else if (live) {
                // c = first target of the unit | lane half: rows 8i + 4h + (0..3) of each tile
                const int h = c & 1, c0 = c & ~1;
                const int left = (nt - c0 + 31) >> 5;
                const int n2 = left < upieces ? left : upieces;
                const int w = atomicAdd(&s_misc[0], n2);
                if (w + n2 <= kFWork) {
                    for (int k = 0; k < n2; k++) s_work[w + k] = ((unsigned)ql << 22) | (unsigned)((((c0 >> 5) + k) << 1) | h);
                } else {
                    flag = true;
                }
            }
        }
    };
#pragma unroll
    for (int k = 0; k < kMaxLists / 4; k++) {
        if (!(__uint_as_float((unsigned)(w2[k] >> 32)) > tau)) flag = true;
        consider(__uint_as_float((unsigned)(w0[k] >> 32)), (int)(unsigned)w0[k]);
        consider(__uint_as_float((unsigned)(w1[k] >> 32)), (int)(unsigned)w1[k]);
    }
    if (flag && live) s_qflag[ql] = 1;
    if (a.debug & 32) {      // diagnostics: approximate minimum and candidate count instead of the result
        __syncthreads();
        if (part == 0) s_a[0][ql] = 0.0f;
        __syncthreads();
        atomicAdd(&s_a[0][ql], (float)ncand);
        __syncthreads();
        if (part == 0 && live) { od[j] = abest + qq; oi[j] = (int)s_a[0][ql] | (s_qflag[ql] ? 1 << 16 : 0); }
        return;
    }
    __syncthreads();
    const int nwork = min(s_misc[0], kFWork);
    for (int w = threadIdx.x; w < nwork && !(a.debug & 1); w += kBlock) {
        const unsigned it = s_work[w];
        const int slot = (int)(it >> 22);
        const float4 qv = s_q[slot];
        if (permT) {
            atomicMin(&s_best[slot], rescan_half_perm<FMA>(T, permT, nt, (int)((it & 0x3fffffu) >> 1) << 5, (int)(it & 1u), qv.x, qv.y, qv.z));
        } else {
            float dd;
            int ii;
            rescan_half<FMA>(T, nt, (int)((it & 0x3fffffu) >> 1) << 5, (int)(it & 1u), qv.x, qv.y, qv.z, dd, ii);
            atomicMin(&s_best[slot], ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)ii);
        }
    }
    __syncthreads();
    if (part == 0 && live) {
        // sorted mode: the result belongs to the query's original position
        const int jo = S ? S->perm_q[d][(size_t)batch * nq + j] : j;
        if (s_qflag[ql]) {
            s_flagged[atomicAdd(&s_misc[1], 1)] = jo;
        } else {
            const unsigned long long v = s_best[ql];
            od[jo] = __uint_as_float((unsigned)(v >> 32));
            oi[jo] = (int)(unsigned)v;
        }
    }
    __syncthreads();
    const int nflag = s_misc[1];
    if (a.stats && threadIdx.x == 0) {
        atomicAdd(&a.stats[0], (unsigned long long)min(kFQ, nq - fb * kFQ));
        atomicAdd(&a.stats[1], (unsigned long long)nflag);
        atomicAdd(&a.stats[2], (unsigned long long)nwork);
    }
    // (sorted mode: the exhaustive pass runs on the caller's arrays -- the reference's tile order matters there)
    const float *Qx = S ? S->q_orig[d] + (size_t)batch * nq * 3 : Qp;
    const float *Tx = S ? S->t_orig[d] + (size_t)batch * nt * 3 : T;
    for (int fidx = 0; fidx < nflag; fidx++) nn_exhaustive<FMA>(Qx, Tx, nt, s_flagged[fidx], od, oi, s_red, s_fi);
}

template <int Q, int U, int NL, int PRE>
static void launch_main2(const NNArgs &a, int blocks, hipStream_t st)
{
    if (a.fma)
        hipLaunchKernelGGL((nn_bf16_kernel<Q, U, NL, 1, PRE>), dim3(blocks), dim3(kBlock), 0, st, a);
    else
        hipLaunchKernelGGL((nn_bf16_kernel<Q, U, NL, 0, PRE>), dim3(blocks), dim3(kBlock), 0, st, a);
}

template <int Q, int U, int NL>
static void launch_main(const NNArgs &a, int blocks, int pre, hipStream_t st)
{
    if (pre) launch_main2<Q, U, NL, 1>(a, blocks, st);
    else launch_main2<Q, U, NL, 0>(a, blocks, st);
}

// Launches (the target split when `pre`,) the filter and the finish kernel.  q / nl / pre as
// chosen by the planner in chamfer.hip.
int launch_nn_bf16(NNArgs &a, int q, int pre, int nl, long long total_blocks, hipStream_t st)
{
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t bytes = 0;
    size_t off_t[2], off_a[2] = {0, 0};
    for (int d = 0; d < a.ndir; d++) {
        NNDir &D = a.dir[d];
        D.ntp = (D.nt + 127) / 128 * 128;
        D.ntmax = pre ? ceil_div(D.ntp, kBlock) : D.slices;
        off_t[d] = bytes;
        bytes = align(bytes + (size_t)a.b * D.ntmax * sizeof(float));
        if (pre) {
            off_a[d] = bytes;
            bytes = align(bytes + (size_t)a.b * 4 * D.ntp * sizeof(uint4));
        }
    }
    char *ws = (char *)workspace(9, bytes, st);
    if (!ws) return 0;
    for (int d = 0; d < a.ndir; d++) {
        a.dir[d].tmaxp = (float *)(ws + off_t[d]);
        a.dir[d].arec = (const uint4 *)(ws + off_a[d]);
    }
    if (pre) {
        SplitArgs sa{};
        sa.centre = a.dir[0].t;
        sa.cn = a.dir[0].nt;
        long long sb = 0;
        for (int d = 0; d < a.ndir; d++) {
            SplitJob &J = sa.job[sa.njobs++];
            J.src = a.dir[d].t; J.out = (uint4 *)(ws + off_a[d]); J.tmax = a.dir[d].tmaxp;
            J.n = a.dir[d].nt; J.np = a.dir[d].ntp; J.bpb = a.dir[d].ntmax; J.block_begin = (int)sb;
            sb += (long long)a.b * J.bpb;
        }
        if (sb > 0x7fffffffLL) {
            set_error("chamfer: problem too large for one launch");
            return 0;
        }
        hipLaunchKernelGGL(nn_split_kernel, dim3((unsigned)sb), dim3(kBlock), 0, st, sa);
    }
    const int blocks = (int)total_blocks;
    if (q == 4) {
        if (nl == 2) launch_main<4, 4, 2>(a, blocks, pre, st);
        else launch_main<4, 4, 1>(a, blocks, pre, st);
    } else {
        if (nl == 2) launch_main<2, 4, 2>(a, blocks, pre, st);
        else launch_main<2, 4, 1>(a, blocks, pre, st);
    }
    if (!check(hipGetLastError(), "nn_bf16_kernel launch")) return 0;
    // the bf16 MFMA may flush subnormal products (<= 1.2e-38 each): negligible against u T^2 only for T^2 >= 2^-60
    return launch_nn_finish(a, nl, 4, (float)kQT16, (float)kTT16, 8.673617379884035e-19f, st);
}

// Second launch of the filtered paths.  kqt / ktt: coefficients of the filter's error bound.
int launch_nn_finish(NNArgs &a, int nl, int upieces, float kqt, float ktt, float t2min, hipStream_t st)
{
    long long fb = 0;
    for (int d = 0; d < a.ndir; d++) {
        a.dir[d].fin_begin = (int)fb;
        fb += (long long)a.b * ceil_div(a.dir[d].nq, kFQ);
    }
    if (fb > 0x7fffffffLL) {
        set_error("chamfer: problem too large for one launch");
        return 0;
    }
    if (a.fma)
        hipLaunchKernelGGL((nn_finish_kernel<1>), dim3((unsigned)fb), dim3(kBlock), 0, st, a, nl, upieces, kqt, ktt, t2min);
    else
        hipLaunchKernelGGL((nn_finish_kernel<0>), dim3((unsigned)fb), dim3(kBlock), 0, st, a, nl, upieces, kqt, ktt, t2min);
    return check(hipGetLastError(), "nn_finish_kernel launch") ? 1 : 0;
}

}  // namespace genpc
