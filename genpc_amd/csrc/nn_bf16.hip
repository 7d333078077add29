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
constexpr int kBTile = 512;            // targets per LDS tile: 4 planes x 16 B = 32 KiB
constexpr double kQT16 = 27.0, kTT16 = 17.0;
constexpr int kMaxLists = 16;          // slices x lists per lane, when sliced (planner: chamfer.hip)

// ---------------------------------------------------------------------------
struct SplitJob {
    const float *src;     // [B, n, 3]
    void *out;            // kind 0: uint2 [B][3][n] (12 bf16 pieces), kind 1: uint4 [B][3][n]
    float *aux;           // kind 0: per-block max |t'|^2 [B][bpb], kind 1: |q'|^2 [B][n]
    int n, kind, bpb, block_begin;
};

struct SplitArgs {
    SplitJob job[4];
    int njobs;
    const float *centre;  // direction-0 targets [B, cn, 3]
    int cn;
};

__device__ __forceinline__ unsigned bf16_rn(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ void split3(float v, unsigned &h, unsigned &m, unsigned &l)
{
    h = bf16_rn(v);
    const float r1 = v - __uint_as_float(h << 16);
    m = bf16_rn(r1);
    const float r2 = r1 - __uint_as_float(m << 16);
    l = bf16_rn(r2);
}

__device__ __forceinline__ uint4 b_vec(float v)    // [h,m,l,h,m,l,h,m]
{
    unsigned h, m, l;
    split3(v, h, m, l);
    return make_uint4(h | (m << 16), l | (h << 16), m | (l << 16), h | (m << 16));
}

__global__ __launch_bounds__(kBlock) void nn_split_kernel(SplitArgs a)
{
    __shared__ float s_red[kWavesPerBlock];
    int j = 0;
    while (j + 1 < a.njobs && (int)blockIdx.x >= a.job[j + 1].block_begin) j++;
    const SplitJob &J = a.job[j];
    const int local = blockIdx.x - J.block_begin;
    const int batch = local / J.bpb, blk = local % J.bpb;
    const int i = blk * kBlock + threadIdx.x;
    const float *c = a.centre + (size_t)batch * a.cn * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    float tt = 0.0f;
    if (i < J.n) {
        const float *p = J.src + ((size_t)batch * J.n + i) * 3;
        const float x = p[0] - cx, y = p[1] - cy, z = p[2] - cz;
        tt = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
        if (J.kind == 0) {
            uint2 *o = (uint2 *)J.out + (size_t)batch * 3 * J.n + i;
            unsigned xh, xm, xl, yh, ym, yl, zh, zm, zl, th, tm, tl;
            split3(-2.0f * x, xh, xm, xl);
            split3(-2.0f * y, yh, ym, yl);
            split3(-2.0f * z, zh, zm, zl);
            split3(tt, th, tm, tl);
            o[0] = make_uint2(xh | (xm << 16), xl | (yh << 16));
            o[(size_t)J.n] = make_uint2(ym | (yl << 16), zh | (zm << 16));
            o[(size_t)2 * J.n] = make_uint2(zl | (th << 16), tm | (tl << 16));
        } else {
            uint4 *o = (uint4 *)J.out + (size_t)batch * 3 * J.n + i;
            o[0] = b_vec(x);
            o[(size_t)J.n] = b_vec(y);
            o[(size_t)2 * J.n] = b_vec(z);
            J.aux[(size_t)batch * J.n + i] = tt;
        }
    }
    if (J.kind == 0) {
        float m = tt;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & (kWave - 1)) == 0) s_red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0)
            J.aux[(size_t)batch * J.bpb + blk] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    }
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
template <int Q, int U, int NL, int FMA>
__global__ __launch_bounds__(kBlock) void nn_bf16_kernel(NNArgs a)
{
    constexpr int kC = 32 * U;
    __shared__ uint4 plane[4][kBTile];
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
    const uint2 *__restrict__ AR = (const uint2 *)D.arec + (size_t)batch * 3 * nt;
    const uint4 *__restrict__ BR = D.brec + (size_t)batch * 3 * nq;

    const int k_begin = slice * a.slice_len;
    int k_end = k_begin + a.slice_len;
    if (k_end > nt) k_end = nt;

    float qx[Q], qy[Q], qz[Q], qq[Q];
    bf16x8 b0[Q], b1[Q];
    Top3 lst[Q][NL];
    const int q0 = (qb * kWavesPerBlock + wave) * (32 * Q) + col;
#pragma unroll
    for (int r = 0; r < Q; r++) {
        int j = q0 + r * 32;
        if (j >= nq) j = nq - 1;
        qx[r] = Qp[(size_t)j * 3 + 0];
        qy[r] = Qp[(size_t)j * 3 + 1];
        qz[r] = Qp[(size_t)j * 3 + 2];
        qq[r] = D.qqv[(size_t)batch * nq + j];
        const uint4 v0 = BR[(size_t)half * nq + j];                       // X (lanes < 32) | Y
        const uint4 v1 = half ? make_uint4(0x3f803f80u, 0x00003f80u, 0u, 0u) : BR[(size_t)2 * nq + j];   // Z | ones
        b0[r] = __builtin_bit_cast(bf16x8, v0);
        b1[r] = __builtin_bit_cast(bf16x8, v1);
#pragma unroll
        for (int n = 0; n < NL; n++) top3_init(lst[r][n]);
    }

    // One accumulator chain per query tile, all on the same 32-target tile; the rows of
    // the next tile are in A0/A1 while the current tile's accumulators are reduced.
    f32x16 acc[Q];
    uint4 A0, A1;
    auto fetch = [&](int row) {
        A0 = plane[half][row + col];
        A1 = plane[2 + half][row + col];
    };
    auto m1 = [&](int r) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A0), b0[r], z, 0, 0, 0);
    };
    auto m2 = [&](int r) {
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1), b1[r], acc[r], 0, 0, 0);
    };
    // One target tile: the accumulators hold tile `row`; while they are folded into the
    // running unit minima m[] the chains of tile row + 32 are started, skewed by one slot
    // so that the second instruction of a chain issues one slot (8 v_min3 + one MFMA)
    // after the first and no MFMA waits on its predecessor:
    //   slot r:  4 min3(acc[r]) | M2(r-1) | 4 min3(acc[r]) | M1(r)
    auto step = [&](int row, int tn_pad, float (&m)[Q]) {
        const bool more = row + 32 < tn_pad;
#pragma unroll
        for (int r = 0; r < Q; r++) {
            const f32x16 &c = acc[r];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; e += 2) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m[r]) : "v"(m[r]), "v"(c[e]), "v"(c[e + 1]));
            __builtin_amdgcn_sched_barrier(0);
            if (more && r > 0) m2(r - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 8; e < 16; e += 2) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m[r]) : "v"(m[r]), "v"(c[e]), "v"(c[e + 1]));
            __builtin_amdgcn_sched_barrier(0);
            if (more) m1(r);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            m2(Q - 1);
            if (row + 64 < tn_pad) fetch(row + 64);      // rows of the tile after next
        }
    };

    // pieces of the next tile, kBTile / kBlock targets per thread
    constexpr int kPer = kBTile / kBlock;
    uint2 pre[kPer][3];
    auto prefetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            int t = t0 + i * kBlock + threadIdx.x;
            t = t < nt ? t : nt - 1;
#pragma unroll
            for (int pl = 0; pl < 3; pl++) pre[i][pl] = AR[(size_t)pl * nt + t];
        }
    };
    if (k_begin < k_end) prefetch(k_begin);
    for (int t0 = k_begin; t0 < k_end && !(a.debug & 4); t0 += kBTile) {
        const int tn = min(kBTile, k_end - t0);
        const int tn_pad = (tn + kC - 1) / kC * kC;
        __syncthreads();                 // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const int t = i * kBlock + threadIdx.x;
            if (t < tn_pad) {
                const uint2 p0 = pre[i][0], p1 = pre[i][1], p2 = pre[i][2];
                uint4 X, Y, Z, W;
                X.x = __builtin_amdgcn_perm(p0.x, p0.x, 0x01000100u);     // xh xh
                X.y = p0.x;                                               // xh xm
                X.z = __builtin_amdgcn_perm(p0.x, p0.x, 0x03020302u);     // xm xm
                X.w = __builtin_amdgcn_perm(p0.y, p0.y, 0x01000100u);     // xl xl
                Y.x = __builtin_amdgcn_perm(p0.y, p0.y, 0x03020302u);     // yh yh
                Y.y = __builtin_amdgcn_perm(p0.y, p1.x, 0x01000706u);     // yh ym
                Y.z = __builtin_amdgcn_perm(p1.x, p1.x, 0x01000100u);     // ym ym
                Y.w = __builtin_amdgcn_perm(p1.x, p1.x, 0x03020302u);     // yl yl
                Z.x = __builtin_amdgcn_perm(p1.y, p1.y, 0x01000100u);     // zh zh
                Z.y = p1.y;                                               // zh zm
                Z.z = __builtin_amdgcn_perm(p1.y, p1.y, 0x03020302u);     // zm zm
                Z.w = __builtin_amdgcn_perm(p2.x, p2.x, 0x01000100u);     // zl zl
                W.x = __builtin_amdgcn_perm(p2.x, p2.y, 0x01000706u);     // th tm
                W.y = p2.y >> 16;                                         // tl 0
                W.z = 0u;
                W.w = 0u;
                if (t >= tn) {
                    // padding: |t'|^2 = +inf (bf16 0x7f80) x 1 never wins, the other terms are 0
                    X = Y = Z = make_uint4(0u, 0u, 0u, 0u);
                    W = make_uint4(0x7f80u, 0u, 0u, 0u);
                }
                plane[0][t] = X;
                plane[1][t] = Y;
                plane[2][t] = Z;
                plane[3][t] = W;
            }
        }
        if (t0 + kBTile < k_end) prefetch(t0 + kBTile);
        __syncthreads();
        // prologue of the LDS tile: chains of target tile 0, rows of tile 1
        fetch(0);
#pragma unroll
        for (int r = 0; r < Q; r++) m1(r);
#pragma unroll
        for (int r = 0; r < Q; r++) m2(r);
        if (32 < tn_pad) fetch(32);
        for (int rb0 = 0; rb0 < tn_pad; rb0 += NL * kC) {
#pragma unroll
            for (int n = 0; n < NL; n++) {
                const int rb = rb0 + n * kC;
                if (rb < tn_pad) {
                    float m[Q];
#pragma unroll
                    for (int r = 0; r < Q; r++) m[r] = __builtin_inff();
#pragma unroll
                    for (int g = 0; g < U; g++) step(rb + 32 * g, tn_pad, m);
#pragma unroll
                    for (int r = 0; r < Q; r++) top3_insert(lst[r][n], m[r], t0 + rb);
                }
            }
        }
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
__global__ __launch_bounds__(kBlock) void nn_finish_kernel(NNArgs a, int nl, int upieces)
{
    __shared__ unsigned long long s_best[kFQ];   // (distance bits << 32 | index): atomic min == (distance, first index)
    __shared__ float4 s_q[kFQ];
    __shared__ float s_a[4][kFQ];
    __shared__ int s_qflag[kFQ];
    __shared__ int s_flagged[kFQ];
    __shared__ unsigned s_work[kFWork];          // query slot << 22 | first target / 32
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

    const int ql = threadIdx.x & (kFQ - 1), part = threadIdx.x >> 6;     // part == wave
    int j = fb * kFQ + ql;
    const bool live = j < nq;
    j = live ? j : nq - 1;

    // this thread's lists: li = part + 4k
    unsigned long long w0[kMaxLists / 4], w1[kMaxLists / 4], w2[kMaxLists / 4];
    float amin = __builtin_inff();
#pragma unroll
    for (int k = 0; k < kMaxLists / 4; k++) {
        const int li = part + 4 * k;
        w0[k] = w1[k] = w2[k] = 0x7f800000ull << 32;      // (+inf, 0)
        if (li < nlists) {
            const unsigned long long *p = P + (size_t)li * 3 * bnq + j;
            w0[k] = p[0];
            w1[k] = p[bnq];
            w2[k] = p[2 * bnq];
        }
    }
#pragma unroll
    for (int k = 0; k < kMaxLists / 4; k++) amin = fminf(amin, __uint_as_float((unsigned)(w0[k] >> 32)));
    s_a[part][ql] = amin;
    // max |t'|^2 over the whole target cloud (per-block maxima of the split kernel)
    float tmax2 = 0.0f;
    {
        const int nblk = (nt + kBlock - 1) / kBlock;
        const float *tp = D.tmaxp + (size_t)batch * nblk;
        for (int i = threadIdx.x; i < nblk; i += kBlock) tmax2 = fmaxf(tmax2, tp[i]);
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
    const float qq = D.qqv[(size_t)batch * nq + j];
    float tau = nn_tau(abest, qq, tmax2, kQT16, kTT16);
    if (a.debug & 16) tau = __builtin_inff();          // test hook: every listed tile is evaluated
    bool flag = (a.debug & 8) != 0 || !(tau == tau);   // test hook / non-finite input: exhaustive pass
    int ncand = 0;
    // a listed tile whose minimum is not provably out becomes work items of 32 targets
    auto consider = [&](float av, int c) {
        if (av <= tau) {
            ncand++;
            if (c < 0) {
                flag = true;
            } else if (live) {
                const int left = (nt - c + 31) >> 5;
                const int n2 = left < upieces ? left : upieces;
                const int w = atomicAdd(&s_misc[0], n2);
                if (w + n2 <= kFWork) {
                    for (int k = 0; k < n2; k++) s_work[w + k] = ((unsigned)ql << 22) | (unsigned)((c >> 5) + k);
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
        float dd;
        int ii;
        rescan_chunk<FMA, 32>(T, nt, (int)(it & 0x3fffffu) << 5, qv.x, qv.y, qv.z, dd, ii);
        atomicMin(&s_best[slot], ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)ii);
    }
    __syncthreads();
    if (part == 0 && live) {
        if (s_qflag[ql]) {
            s_flagged[atomicAdd(&s_misc[1], 1)] = j;
        } else {
            const unsigned long long v = s_best[ql];
            od[j] = __uint_as_float((unsigned)(v >> 32));
            oi[j] = (int)(unsigned)v;
        }
    }
    __syncthreads();
    const int nflag = s_misc[1];
    for (int fidx = 0; fidx < nflag; fidx++) nn_exhaustive<FMA>(Qp, T, nt, s_flagged[fidx], od, oi, s_red, s_fi);
}

template <int Q, int U, int NL>
static void launch_main(const NNArgs &a, int blocks, hipStream_t st)
{
    if (arith_mode() != 0)
        hipLaunchKernelGGL((nn_bf16_kernel<Q, U, NL, 1>), dim3(blocks), dim3(kBlock), 0, st, a);
    else
        hipLaunchKernelGGL((nn_bf16_kernel<Q, U, NL, 0>), dim3(blocks), dim3(kBlock), 0, st, a);
}

// Writes the split records of every direction's queries and targets (workspace slot 9),
// then launches the filter.  q / nl as chosen by the planner in chamfer.hip.
int launch_nn_bf16(NNArgs &a, int q, int u, int nl, long long total_blocks, hipStream_t st)
{
    SplitArgs sa{};
    sa.centre = a.dir[0].t;
    sa.cn = a.dir[0].nt;
    size_t bytes = 0;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off_a[2], off_t[2], off_b[2], off_q[2];
    for (int d = 0; d < a.ndir; d++) {
        const NNDir &D = a.dir[d];
        off_a[d] = bytes; bytes = align(bytes + (size_t)a.b * 3 * D.nt * sizeof(uint2));
        off_t[d] = bytes; bytes = align(bytes + (size_t)a.b * ceil_div(D.nt, kBlock) * sizeof(float));
        off_b[d] = bytes; bytes = align(bytes + (size_t)a.b * 3 * D.nq * sizeof(uint4));
        off_q[d] = bytes; bytes = align(bytes + (size_t)a.b * D.nq * sizeof(float));
    }
    char *ws = (char *)workspace(9, bytes, st);
    if (!ws) return 0;
    long long sb = 0;
    for (int d = 0; d < a.ndir; d++) {
        NNDir &D = a.dir[d];
        D.arec = (const uint4 *)(ws + off_a[d]);
        D.tmaxp = (const float *)(ws + off_t[d]);
        D.brec = (const uint4 *)(ws + off_b[d]);
        D.qqv = (const float *)(ws + off_q[d]);
        SplitJob &ja = sa.job[sa.njobs++];
        ja.src = D.t; ja.out = ws + off_a[d]; ja.aux = (float *)(ws + off_t[d]);
        ja.n = D.nt; ja.kind = 0; ja.bpb = ceil_div(D.nt, kBlock); ja.block_begin = (int)sb;
        sb += (long long)a.b * ja.bpb;
        SplitJob &jb = sa.job[sa.njobs++];
        jb.src = D.q; jb.out = ws + off_b[d]; jb.aux = (float *)(ws + off_q[d]);
        jb.n = D.nq; jb.kind = 1; jb.bpb = ceil_div(D.nq, kBlock); jb.block_begin = (int)sb;
        sb += (long long)a.b * jb.bpb;
    }
    if (sb > 0x7fffffffLL) {
        set_error("chamfer: problem too large for one launch");
        return 0;
    }
    hipLaunchKernelGGL(nn_split_kernel, dim3((unsigned)sb), dim3(kBlock), 0, st, sa);
    const int blocks = (int)total_blocks;
    if (u == 2) {
        if (q == 4) {
            if (nl == 2) launch_main<4, 2, 2>(a, blocks, st);
            else launch_main<4, 2, 1>(a, blocks, st);
        } else if (q == 2) {
            if (nl == 4) launch_main<2, 2, 4>(a, blocks, st);
            else if (nl == 2) launch_main<2, 2, 2>(a, blocks, st);
            else launch_main<2, 2, 1>(a, blocks, st);
        } else {
            if (nl == 4) launch_main<1, 2, 4>(a, blocks, st);
            else if (nl == 2) launch_main<1, 2, 2>(a, blocks, st);
            else launch_main<1, 2, 1>(a, blocks, st);
        }
    } else if (q == 4) {
        if (nl == 2) launch_main<4, 4, 2>(a, blocks, st);
        else launch_main<4, 4, 1>(a, blocks, st);
    } else if (q == 2) {
        if (nl == 4) launch_main<2, 4, 4>(a, blocks, st);
        else if (nl == 2) launch_main<2, 4, 2>(a, blocks, st);
        else launch_main<2, 4, 1>(a, blocks, st);
    } else {
        if (nl == 4) launch_main<1, 4, 4>(a, blocks, st);
        else if (nl == 2) launch_main<1, 4, 2>(a, blocks, st);
        else launch_main<1, 4, 1>(a, blocks, st);
    }
    if (!check(hipGetLastError(), "nn_bf16_kernel launch")) return 0;
    long long fb = 0;
    for (int d = 0; d < a.ndir; d++) {
        a.dir[d].fin_begin = (int)fb;
        fb += (long long)a.b * ceil_div(a.dir[d].nq, kFQ);
    }
    if (fb > 0x7fffffffLL) {
        set_error("chamfer: problem too large for one launch");
        return 0;
    }
    if (arith_mode() != 0)
        hipLaunchKernelGGL((nn_finish_kernel<1>), dim3((unsigned)fb), dim3(kBlock), 0, st, a, nl, u);
    else
        hipLaunchKernelGGL((nn_finish_kernel<0>), dim3((unsigned)fb), dim3(kBlock), 0, st, a, nl, u);
    return check(hipGetLastError(), "nn_finish_kernel launch") ? 1 : 0;
}

}  // namespace genpc
