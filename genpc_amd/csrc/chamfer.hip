// chamfer.hip -- brute-force nearest neighbour (Chamfer3D forward/backward) for
// gfx950.  Replaces loss_functions/Chamfer3D/chamfer3D.cu of the reference.
//
// Forward design.  The kernel is VALU-bound (6.5 kflop per HBM byte).
// Measured issue model on MI355X (tools/ubench_valu.hip): v_fma/v_mul/v_add/v_sub
// f32 with VGPR operands issue at ~105 lane-ops/clk/CU; every instruction with an
// SGPR operand, v_min/v_min3/v_cmp/v_cndmask, packed-f32 and f64 ops issue at half
// of that.  Hence:
//   * a lane owns R queries in VGPRs; targets are staged per block in LDS (groups
//     of four as x0..x3|y0..y3|z0..z3) and read with uniform-address (broadcast)
//     ds_read_b128 -- three reads deliver four targets -- so that the
//     three subtractions per pair keep VGPR operands and stay full rate (fetching
//     the wave-uniform targets through the scalar cache as SGPR operands costs 10
//     issue units per pair instead of 7 -- measured 1.4x slower at large batch);
//   * the inner loop tracks only the minimum VALUE of a chunk of kChunk targets
//     (one v_min3 per two pairs); per tile the chunk that produced a new minimum
//     is re-scanned in LDS to recover the FIRST index attaining it, which is the
//     reference's strict-'<' tie-break (chamfer3D.cu:36,46,56,66,119,126);
//     per pair: 3 sub + mul + 2 fma (full rate) + 1/2 min3 (half rate) = 7 units;
//   * both directions (A->B, B->A), all batch elements, all query blocks and all
//     target slices are ONE launch sized from the problem (the reference's fixed
//     32x16 grid leaves 94 % of an MI355X idle at B=1); the running minimum never
//     leaves registers (the reference read-modify-writes result[] in global
//     memory every 512 targets);
//   * when targets are split into S>1 slices for occupancy, per-slice
//     (min,argmin) go to scratch and the LAST block to finish a query block folds
//     them in slice order with strict '<' (earlier slice == lower index wins), so
//     results do not depend on S and no second launch is needed.  The hand-off is
//     the agent-scope release -> ticket -> acquire recipe (per-XCD L2s are not
//     coherent with each other).
// Arithmetic is written with explicit __fmaf_rn/__fmul_rn/__fadd_rn and the file
// is compiled with -ffp-contract=off, so the compiler cannot re-associate or
// contract differently from oracle/genpc_oracle.c.
#include "nn.h"
#include "../../include/genpc_hip.h"

#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <mutex>

namespace genpc {

constexpr int kTile = 2048;      // targets per LDS tile (24 KiB: 12 B per target)




// One block = one (direction, target slice, batch, 256*R-query block) unit; wave w
// of the block owns queries [w*64*R, (w+1)*64*R) of that block.
template <int R, int FMA>
__global__ __launch_bounds__(kBlock) void nn_forward_kernel(NNArgs a)
{
    __shared__ float4 tile[kTile / 4 * 3 + 3];      // + one spare group for the pipeline's last prefetch
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    const int d = (a.ndir > 1 && bid >= a.dir[1].block_begin) ? 1 : 0;
    const NNDir &D = a.dir[d];
    bid -= D.block_begin;
    // bid = (slice * b + batch) * qblocks + qblock
    const int qb = bid % D.qblocks;
    const int rest = bid / D.qblocks;
    const int batch = rest % a.b;
    const int slice = rest / a.b;

    const int nq = D.nq, nt = D.nt;
    const float *__restrict__ Q = D.q + (size_t)batch * nq * 3;
    const float *__restrict__ T = D.t + (size_t)batch * nt * 3;

    const int k_begin = slice * a.slice_len;
    int k_end = k_begin + a.slice_len;
    if (k_end > nt) k_end = nt;

    float qx[R], qy[R], qz[R], best[R];
    int bchunk[R];
    const int q0 = (qb * kWavesPerBlock + wave) * (kWave * R) + lane;
#pragma unroll
    for (int r = 0; r < R; r++) {
        int j = q0 + r * kWave;
        if (j >= nq) j = nq - 1;
        qx[r] = Q[(size_t)j * 3 + 0];
        qy[r] = Q[(size_t)j * 3 + 1];
        qz[r] = Q[(size_t)j * 3 + 2];
        best[r] = __builtin_inff();
        bchunk[r] = k_begin;
    }

    // LDS tile layout: groups of 4 targets as three float4s (x0..x3 | y0..y3 | z0..z3),
    // so that every byte a ds_read_b128 fetches is used (12 B per target).
    float *tile_f = (float *)tile;
    float nf = 0.0f;      // NaN once a target of the slice had a non-finite coordinate
    for (int t0 = k_begin; t0 < k_end && !(a.debug & 4); t0 += kTile) {
        const int tn = min(kTile, k_end - t0);
        const int tn_pad = (tn + kChunk - 1) / kChunk * kChunk;
        __syncthreads();                       // readers of the previous tile are done
        for (int t = threadIdx.x; t < tn_pad; t += kBlock) {
            float x, y, z;
            if (t < tn) {
                const float *tp = T + (size_t)(t0 + t) * 3;
                x = tp[0]; y = tp[1]; z = tp[2];
                nf = __fmaf_rn((fabsf(x) + fabsf(y)) + fabsf(z), 0.0f, nf);      // inf x 0 = NaN, NaN sticks
            } else {
                // pad the ragged tail of the last chunk: a +inf distance never wins a strict '<'
                x = y = z = __builtin_inff();
            }
            float *g = tile_f + (t >> 2) * 12 + (t & 3);
            g[0] = x; g[4] = y; g[8] = z;
        }
        __syncthreads();

        // Software pipeline: the three reads of group g+1 are issued before the
        // VALU work of group g (the scheduler otherwise sinks each read next to its
        // first use and every group eats a full LDS round trip).
        float4 X = tile[0], Y = tile[1], Z = tile[2];
        for (int c0 = 0; c0 < tn_pad; c0 += kChunk) {
            const float4 *tl = tile + (c0 >> 2) * 3;
            float m[R];
#pragma unroll
            for (int r = 0; r < R; r++) m[r] = __builtin_inff();
#pragma unroll
            for (int g = 0; g < kChunk / 4; g++) {
                // one group past the end of the last chunk is read and dropped (the
                // array has a spare group)
                const float4 Xn = tl[g * 3 + 3], Yn = tl[g * 3 + 4], Zn = tl[g * 3 + 5];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float d0 = sqdist<FMA>(X.x - qx[r], Y.x - qy[r], Z.x - qz[r]);
                    const float d1 = sqdist<FMA>(X.y - qx[r], Y.y - qy[r], Z.y - qz[r]);
                    const float d2 = sqdist<FMA>(X.z - qx[r], Y.z - qy[r], Z.z - qz[r]);
                    const float d3 = sqdist<FMA>(X.w - qx[r], Y.w - qy[r], Z.w - qz[r]);
                    m[r] = fminf(fminf(m[r], d0), d1);
                    m[r] = fminf(fminf(m[r], d2), d3);
                }
                __builtin_amdgcn_sched_barrier(0);
                X = Xn; Y = Yn; Z = Zn;
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool lt = m[r] < best[r];
                best[r] = lt ? m[r] : best[r];
                bchunk[r] = lt ? t0 + c0 : bchunk[r];
            }
        }
    }

    // Non-finite input: the minimum-of-chunks scan silently skips NaN distances, the reference's
    // tiled scan does not always (nn_exhaustive, nn.h).  A slice that saw a non-finite target marks
    // its partials; queries of such a cloud, and non-finite queries, are answered exhaustively.
    int bad = __syncthreads_or(nf != nf);
    if (D.slices > 1) {
        // Publish this slice's (minimum, chunk) per query; the last block to arrive for
        // this (direction, batch, query block) folds all S slices in slice order with
        // strict '<' (earlier slice == lower index wins) and carries on alone.
        // The partial (minimum, chunk) of a query is ONE 8-byte word, stored and loaded
        // with agent-scope atomics (write-through `sc1` stores, L1-bypassing loads): with
        // 8-byte agent atomics on both sides no release / acquire fence is needed -- a
        // release would write back the whole XCD's dirty L2 lines.
        const size_t bnq = (size_t)a.b * nq;
        unsigned long long *P = D.part + (size_t)batch * nq;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int j = q0 + r * kWave;
            if (j < nq) {
                const unsigned long long v = bad ? ~0ull : ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)bchunk[r];
                __hip_atomic_store(P + (size_t)slice * bnq + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (a.debug & 2) return;
        int *cnt = a.arrive + D.unit_begin + batch * D.qblocks + qb;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *s_ticket = (int *)tile;      // the tile is dead after the barrier above (one LDS object)
        if (threadIdx.x == 0)
            *s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*s_ticket != D.slices - 1) return;
        if (threadIdx.x == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#pragma unroll
        for (int r = 0; r < R; r++) {
            int j = q0 + r * kWave;
            j = j < nq ? j : nq - 1;
            unsigned long long v = __hip_atomic_load(P + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad |= v == ~0ull;
            float bv = __uint_as_float((unsigned)(v >> 32));
            int bc = (int)(unsigned)v;
            if (v == ~0ull) { bv = __builtin_inff(); bc = 0; }      // marked slice: no chunk to recover (the query goes exhaustive)
            for (int s2 = 1; s2 < D.slices; s2++) {
                v = __hip_atomic_load(P + (size_t)s2 * bnq + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad |= v == ~0ull;
                const float vv = __uint_as_float((unsigned)(v >> 32));
                const bool lt = vv < bv;
                bv = lt ? vv : bv;
                bc = lt ? (int)(unsigned)v : bc;
            }
            best[r] = bv;
            bchunk[r] = bc;
        }
    }

    // Recover the FIRST index attaining best[r] inside the remembered chunk: same
    // operands, same operations => bitwise-equal distance.  Done once per query (by
    // the merging block when the targets were sliced), from global memory, as 24
    // 16-byte loads per query.  Positions past the end of the cloud are clamped to
    // its last target; positions past the end of the chunk's slice belong to the
    // next slice.  Either can only match at a HIGHER position than the true first
    // index, which lies inside the slice, and the lowest match wins.
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    float *__restrict__ od = D.out_d + (size_t)batch * nq;
    int *__restrict__ oi = D.out_i + (size_t)batch * nq;
    // the target tile is dead: its LDS holds the list of queries for the exhaustive pass
    __syncthreads();
    int *s_nflag = (int *)tile;
    int *s_xfi = s_nflag + 4;
    float *s_xred = (float *)(s_nflag + 8);
    int *s_flag = s_nflag + 16;                  // up to kBlock * R entries
    if (threadIdx.x == 0) *s_nflag = 0;
    bad = __syncthreads_or(bad);                 // merged partials: every lane read the same marks
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int base = bchunk[r];
        int first = 0;
        if (!(a.debug & 1)) {
            if (base + kChunk <= nt) {
                const f4u *tp = (const f4u *)(T + (size_t)base * 3);
#pragma unroll
                for (int c8 = kChunk - 8; c8 >= 0; c8 -= 8) {
                    f4u v[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) v[i] = tp[(c8 >> 2) * 3 + i];
                    const float f[24] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w,
                                         v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w,
                                         v[4].x, v[4].y, v[4].z, v[4].w, v[5].x, v[5].y, v[5].z, v[5].w};
#pragma unroll
                    for (int c = 7; c >= 0; c--) {
                        const float dd = sqdist<FMA>(f[c * 3 + 0] - qx[r], f[c * 3 + 1] - qy[r], f[c * 3 + 2] - qz[r]);
                        first = (dd == best[r]) ? c8 + c : first;
                    }
                }
            } else {
                for (int c = kChunk - 1; c >= 0; c--) {
                    int kk = base + c;
                    kk = kk < nt ? kk : nt - 1;
                    const float *tp = T + (size_t)kk * 3;
                    const float dd = sqdist<FMA>(tp[0] - qx[r], tp[1] - qy[r], tp[2] - qz[r]);
                    first = (dd == best[r]) ? c : first;
                }
            }
        }
        const int j = q0 + r * kWave;
        const bool qbad = !((fabsf(qx[r]) + fabsf(qy[r])) + fabsf(qz[r]) < __builtin_inff());
        if (j < nq) {
            if (bad || qbad) {
                s_flag[atomicAdd(s_nflag, 1)] = j;
            } else {
                od[j] = best[r];
                oi[j] = base + first;
            }
        }
    }
    __syncthreads();
    const int nflag = *s_nflag;
    for (int fidx = 0; fidx < nflag; fidx++) nn_exhaustive<FMA>(Q, T, nt, s_flag[fidx], od, oi, s_xred, s_xfi);
}

// ---------------------------------------------------------------------------
// MFMA path.  |q - t|^2 = |q|^2 + (|t|^2 - 2 q.t): the bracket is a K = 4 product of
// (qx, qy, qz, 1) with (-2tx, -2ty, -2tz, |t|^2), which v_mfma_f32_32x32x2_f32
// evaluates for 32 targets x 32 queries in two instructions (bitwise a k-ordered
// fmaf chain, MI355X guide) at the fp32 matrix rate -- 1024 pairs per 128 SIMD
// cycles, against ~7 issue units per pair on the VALU path.  The value is only an
// APPROXIMATION of the reference's fl((x2-x1)^2 + ...) (different roundings), so it
// is used as a filter with a rigorous error bound, never as the result:
//   * coordinates are centred on the target cloud's first point c (q' = fl(q - c),
//     t' = fl(t - c)), so magnitudes -- and the bound -- scale with the cloud's
//     extent, not with its offset from the origin;
//   * lane l of a wave holds, per MFMA pair, 16 approximate values of query l&31
//     against rows {8i + 4(l>>5) + 0..3} of a 32-target tile; it keeps the three
//     smallest minima over its (tile, half) units, with the tile index for two of them
//     (Top3); lanes l and l^32, then the target slices, fold their lists;
//   * with a1 <= a2 <= a3 the folded minima: every unit outside the first (first
//     two) listed tile(s) has approximate value >= a2 (a3).  nn_safe() proves from
//     the bound that the reference's minimum lies inside the listed tile(s); those
//     32 (64) targets are then evaluated with the reference's exact arithmetic, in
//     index order -> bitwise the reference's (distance, first index).  Queries for
//     which the proof fails (three tiles within the bound: exact ties on gridded
//     data, non-finite input) are re-done exhaustively by the merging block.
// Error bound, u = 2^-24, T = max |t'|, per target: the chain rounds four times on
// partial sums <= 2|q'||t'| + |t'|^2 and |t'|^2 itself carries 3 roundings, |q'|^2
// three: |a + |q'|^2 - |q' - t'|^2| <= E1 = u (8|q'|T + 7T^2 + 3|q'|^2); centring
// moves each point by <= u|p'|: |sqrt(D') - sqrt(D)| <= eta = u (|q'| + T); the
// reference's value is D (1 + theta), |theta| <= 6u (three subtractions, squared,
// three to five roundings of non-negative terms).  All evaluated in fp64 with u
// inflated by 1 %.
constexpr int kMTile = 2048;     // targets per LDS tile on the MFMA path (2 planes x 8 B)


// True when every target whose approximate value is >= a_rest provably has a
// reference distance above that of the target that produced a_best.
__device__ __forceinline__ bool nn_safe(float a_best, float a_rest, float qq, float tmax2)
{
    const double u = 1.01 * 5.9604644775390625e-8;
    const double T = sqrt((double)tmax2), qn = sqrt((double)qq);
    const double kSub = 64.0 * 1.401298464324817e-45;      // subnormal results round absolutely (see nn_tau)
    const double E1 = u * (8.0 * qn * T + 7.0 * T * T + 3.0 * (double)qq) + kSub;
    const double eta = u * (qn + T);
    double up = (double)a_best + (double)qq + E1;
    up = sqrt(up > 0.0 ? up : 0.0) + eta;
    up = up * up * (1.0 + 6.0 * u) + kSub;
    double lo = (double)a_rest + (double)qq - E1;
    lo = sqrt(lo > 0.0 ? lo : 0.0) - eta;
    lo = lo > 0.0 ? lo : 0.0;
    lo = lo * lo * (1.0 - 6.0 * u);
    return lo > up;      // false for NaN
}

// Exact (reference arithmetic) minimum and FIRST index over targets [base, base+len)
// of a cloud of nt targets, positions past the end clamped to the last target.

// One block = one (direction, target slice, batch, 128*Q-query block) unit; wave w
// owns Q tiles of 32 queries.  U = MFMA tiles per bookkeeping unit (re-scan
// granularity 32*U targets).
template <int Q, int U, int FMA>
__global__ __launch_bounds__(kBlock) void nn_mfma_kernel(NNArgs a)
{
    constexpr int kC = 32 * U;
    __shared__ float2 plane[2][kMTile + 2 * kC];  // + two spare units for the pipeline's last fetches
    __shared__ float s_red[kWavesPerBlock];
    __shared__ int s_fi[kWavesPerBlock];
    __shared__ int s_misc[2];                 // [0] arrival ticket, [1] number of flagged queries
    __shared__ int s_flag[kWavesPerBlock * Q * 32];
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
    const float cx = T[0], cy = T[1], cz = T[2];

    const int k_begin = slice * a.slice_len;
    int k_end = k_begin + a.slice_len;
    if (k_end > nt) k_end = nt;

    float qx[Q], qy[Q], qz[Q], b0[Q], b1[Q], qq[Q];
    Top3 st[Q];
    const int q0 = (qb * kWavesPerBlock + wave) * (32 * Q) + col;
#pragma unroll
    for (int r = 0; r < Q; r++) {
        int j = q0 + r * 32;
        if (j >= nq) j = nq - 1;
        qx[r] = Qp[(size_t)j * 3 + 0];
        qy[r] = Qp[(size_t)j * 3 + 1];
        qz[r] = Qp[(size_t)j * 3 + 2];
        const float px = qx[r] - cx, py = qy[r] - cy, pz = qz[r] - cz;
        qq[r] = __fmaf_rn(pz, pz, __fmaf_rn(py, py, __fmul_rn(px, px)));
        b0[r] = half ? py : px;          // k = 0 (x) on lanes 0..31, k = 1 (y) on lanes 32..63
        b1[r] = half ? 1.0f : pz;        // k = 2 (z),               k = 3 (|t|^2 x 1)
        top3_init(st[r]);
    }

    f32x16 acc0[U * Q], acc1[U * Q];
    float2 A[U];
    auto fetch = [&](int rb) {
#pragma unroll
        for (int s = 0; s < U; s++) A[s] = plane[half][rb + s * 32 + col];
    };
    // MFMAs of the unit whose rows are in A, then the rows of unit `next` into A
    auto issue = [&](int next, f32x16 (&dst)[U * Q]) {
#pragma unroll
        for (int s = 0; s < U; s++) {
#pragma unroll
            for (int r = 0; r < Q; r++) {
                f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s].x, b0[r], c, 0, 0, 0);
                dst[s * Q + r] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s].y, b1[r], c, 0, 0, 0);
            }
        }
        fetch(next);
    };
    auto reduce = [&](const f32x16 (&src)[U * Q], int id) {
#pragma unroll
        for (int r = 0; r < Q; r++) {
            float m;
#pragma unroll
            for (int s = 0; s < U; s++) {
                const f32x16 &c = src[s * Q + r];
                // raw v_min3: fminf() would first canonicalise every MFMA output (v_max x,x)
                if (s == 0) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(c[0]), "v"(c[1]), "v"(c[2]));
                else        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(c[0]), "v"(c[1]));
#pragma unroll
                for (int i = (s == 0 ? 3 : 2); i + 1 < 16; i += 2)
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(c[i]), "v"(c[i + 1]));
                if (s == 0) asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(c[15]));
            }
            top3_insert(st[r], m, id);
        }
    };

    float tmax2 = 0.0f;
    float nf = 0.0f;      // NaN once a target of the slice had a non-finite coordinate
    for (int t0 = k_begin; t0 < k_end && !(a.debug & 4); t0 += kMTile) {
        const int tn = min(kMTile, k_end - t0);
        const int tn_pad = (tn + 2 * kC - 1) / (2 * kC) * (2 * kC);
        __syncthreads();
        for (int t = threadIdx.x; t < tn_pad; t += kBlock) {
            float ax = 0.0f, ay = 0.0f, az = 0.0f, tt = __builtin_inff();   // padding: +inf never wins
            if (t < tn) {
                const float *tp = T + (size_t)(t0 + t) * 3;
                const float x = tp[0] - cx, y = tp[1] - cy, z = tp[2] - cz;
                tt = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
                tmax2 = fmaxf(tmax2, tt);
                nf = __fmaf_rn(tt, 0.0f, nf);        // inf x 0 = NaN, NaN sticks (v_max drops NaNs)
                ax = -2.0f * x; ay = -2.0f * y; az = -2.0f * z;
            }
            plane[0][t] = make_float2(ax, az);
            plane[1][t] = make_float2(ay, tt);
        }
        __syncthreads();
        // Two-stage software pipeline over bookkeeping units: the MFMAs of unit i+1 are
        // issued before the VALU minima of unit i, into the other accumulator set.
        fetch(0);
        issue(kC, acc0);
        for (int rb = 0; rb < tn_pad; rb += 2 * kC) {
            issue(rb + 2 * kC, acc1);          // past the end: spare rows, results dropped
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7");           // a 16-pass MFMA result needs 19 wait states before the inline-asm
            reduce(acc0, t0 + rb);             // v_min3 (not padded by the compiler): 8 here + the instructions between
            __builtin_amdgcn_sched_barrier(0);
            issue(rb + 3 * kC, acc0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7");
            reduce(acc1, t0 + rb + kC);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // lanes l and l^32 hold the two halves of the same queries: fold (both end up equal)
#pragma unroll
    for (int r = 0; r < Q; r++) {
        const float o1 = __shfl_xor(st[r].a1, 32), o2 = __shfl_xor(st[r].a2, 32), o3 = __shfl_xor(st[r].a3, 32);
        const int oc1 = __shfl_xor(st[r].c1, 32), oc2 = __shfl_xor(st[r].c2, 32);
        Top3 lo = st[r], hi = st[r];
        if (half) { lo.a1 = o1; lo.a2 = o2; lo.a3 = o3; lo.c1 = oc1; lo.c2 = oc2; }
        else      { hi.a1 = o1; hi.a2 = o2; hi.a3 = o3; hi.c1 = oc1; hi.c2 = oc2; }
        top3_insert(lo, hi.a1, hi.c1);
        top3_insert(lo, hi.a2, hi.c2);
        top3_insert(lo, hi.a3, -1);
        st[r] = lo;
    }
    // slice maximum of |t'|^2; +inf when a target is not finite: every query is then answered by
    // nn_exhaustive, which reproduces the reference's tile semantics for NaNs
    if (nf != nf) tmax2 = __builtin_inff();
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tmax2 = fmaxf(tmax2, __shfl_xor(tmax2, o));
    __syncthreads();
    if (lane == 0) s_red[wave] = tmax2;
    __syncthreads();
    tmax2 = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));

    if (D.slices > 1) {
        // Per query and slice three 8-byte words (a1,c1) (a2,c2) (a3,tmax2), agent-scope
        // atomic stores / loads on both sides (see the VALU kernel); the last block to
        // arrive folds the slices in slice order.
        const size_t bnq = (size_t)a.b * nq;
        unsigned long long *P = D.part + (size_t)batch * nq;
        if (!half) {
#pragma unroll
            for (int r = 0; r < Q; r++) {
                const int j = q0 + r * 32;
                if (j < nq) {
                    unsigned long long *p = P + (size_t)slice * 3 * bnq + j;
                    __hip_atomic_store(p, ((unsigned long long)__float_as_uint(st[r].a1) << 32) | (unsigned)st[r].c1,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p + bnq, ((unsigned long long)__float_as_uint(st[r].a2) << 32) | (unsigned)st[r].c2,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p + 2 * bnq,
                                       ((unsigned long long)__float_as_uint(st[r].a3) << 32) | __float_as_uint(tmax2),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (a.debug & 2) return;
        int *cnt = a.arrive + D.unit_begin + batch * D.qblocks + qb;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            s_misc[0] = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_misc[0] != D.slices - 1) return;
        if (threadIdx.x == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int r = 0; r < Q; r++) {
            int j = q0 + r * 32;
            j = j < nq ? j : nq - 1;
            Top3 f;
            top3_init(f);
            float tm = 0.0f;
            for (int s2 = 0; s2 < D.slices; s2++) {
                const unsigned long long *p = P + (size_t)s2 * 3 * bnq + j;
                const unsigned long long w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long w1 = __hip_atomic_load(p + bnq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long w2 = __hip_atomic_load(p + 2 * bnq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                top3_insert(f, __uint_as_float((unsigned)(w0 >> 32)), (int)(unsigned)w0);
                top3_insert(f, __uint_as_float((unsigned)(w1 >> 32)), (int)(unsigned)w1);
                top3_insert(f, __uint_as_float((unsigned)(w2 >> 32)), -1);
                tm = fmaxf(tm, __uint_as_float((unsigned)w2));
            }
            st[r] = f;
            tmax2 = tm;      // the same for every query of the block
        }
    }

    // Decide, re-scan the listed tile(s) exactly, or flag the query for the exhaustive pass.
    float *__restrict__ od = D.out_d + (size_t)batch * nq;
    int *__restrict__ oi = D.out_i + (size_t)batch * nq;
    __syncthreads();
    if (threadIdx.x == 0) s_misc[1] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < Q; r++) {
        const Top3 &f = st[r];
        bool two = false, flag = false;
        if (!nn_safe(f.a1, f.a2, qq[r], tmax2) || (a.debug & 16)) {
            two = true;
            if (!nn_safe(f.a1, f.a3, qq[r], tmax2) || f.c2 < 0) flag = true;
        }
        if (f.c1 < 0 || (a.debug & 8)) flag = true;
        if (!(tmax2 < __builtin_inff()) || !(qq[r] < __builtin_inff())) flag = true;      // non-finite input
        const int j = q0 + r * 32;
        if (flag) {
            if (!half && j < nq) s_flag[atomicAdd(&s_misc[1], 1)] = j;
            continue;
        }
        // lanes 0..31 take the first tile, lanes 32..63 the second (when needed and different)
        const int base = half ? f.c2 : f.c1;
        float bd = __builtin_inff();
        int bi = 0;
        if ((!half || (two && f.c2 != f.c1)) && !(a.debug & 1)) rescan_chunk<FMA, kC>(T, nt, base, qx[r], qy[r], qz[r], bd, bi);
        const float obd = __shfl_xor(bd, 32);
        const int obi = __shfl_xor(bi, 32);
        if (!half && j < nq) {
            const bool other = obd < bd || (obd == bd && obi < bi);
            od[j] = other ? obd : bd;
            oi[j] = other ? obi : bi;
        }
    }
    __syncthreads();
    const int nflag = s_misc[1];
    if (a.stats && threadIdx.x == 0) {
        atomicAdd(&a.stats[0], (unsigned long long)min(128 * Q, nq - qb * 128 * Q));
        atomicAdd(&a.stats[1], (unsigned long long)nflag);
    }
    for (int fidx = 0; fidx < nflag; fidx++) nn_exhaustive<FMA>(Qp, T, nt, s_flag[fidx], od, oi, s_red, s_fi);
}

// Chamfer backward, both directions in one launch (chamfer3D.cu:155-195).
// Thread j < B*N: direction 1 term of point j of cloud 1; B*N <= j < B*(N+M):
// direction 2 term of point j-B*N of cloud 2.  Accumulates with fp32 atomics
// into the caller-zeroed gradients, exactly like the reference.
// PHASE 0: both halves with atomics in one launch (small calls: one launch is what a call costs).  PHASE 1 then PHASE 2 (two
// launches, large calls): a point's OWN row first, as a plain read-modify-write (coalesced; nobody else touches the buffers
// during that launch), then the scattered halves with atomics -- half the atomics, which are what bounds this kernel (six
// per point at ~14 per clock chip-wide: 0.73 ms for 64 x 32768 x 2 points, 0.04 of the HBM roofline on its 56 B per point).
template <int PHASE>
__global__ __launch_bounds__(kBlock) void chamfer_grad_kernel(int b, int n, const float *__restrict__ xyz1, int m,
                                                              const float *__restrict__ xyz2,
                                                              const float *__restrict__ gd1, const int *__restrict__ idx1,
                                                              const float *__restrict__ gd2, const int *__restrict__ idx2,
                                                              float *__restrict__ gx1, float *__restrict__ gx2)
{
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long bn = (long long)b * n, bm = (long long)b * m;
    const float *P1, *P2, *G;
    const int *I;
    float *O1, *O2;
    int np1, np2;
    if (t < bn) {
        P1 = xyz1; P2 = xyz2; G = gd1; I = idx1; O1 = gx1; O2 = gx2; np1 = n; np2 = m;
    } else if (t < bn + bm) {
        t -= bn;
        P1 = xyz2; P2 = xyz1; G = gd2; I = idx2; O1 = gx2; O2 = gx1; np1 = m; np2 = n;
    } else {
        return;
    }
    const long long i = t / np1;
    const float x1 = P1[t * 3 + 0], y1 = P1[t * 3 + 1], z1 = P1[t * 3 + 2];
    const long long j2 = i * np2 + I[t];
    const float x2 = P2[j2 * 3 + 0], y2 = P2[j2 * 3 + 1], z2 = P2[j2 * 3 + 2];
    const float g = __fmul_rn(G[t], 2.0f);
    const float vx = __fmul_rn(g, x1 - x2), vy = __fmul_rn(g, y1 - y2), vz = __fmul_rn(g, z1 - z2);
    if (PHASE == 0) {
        atomicAdd(&O1[t * 3 + 0], vx);
        atomicAdd(&O1[t * 3 + 1], vy);
        atomicAdd(&O1[t * 3 + 2], vz);
    } else if (PHASE == 1) {
        O1[t * 3 + 0] = __fadd_rn(O1[t * 3 + 0], vx);
        O1[t * 3 + 1] = __fadd_rn(O1[t * 3 + 1], vy);
        O1[t * 3 + 2] = __fadd_rn(O1[t * 3 + 2], vz);
    }
    if (PHASE != 1) {
        atomicAdd(&O2[j2 * 3 + 0], -vx);
        atomicAdd(&O2[j2 * 3 + 1], -vy);
        atomicAdd(&O2[j2 * 3 + 2], -vz);
    }
}

// The scattered halves of large calls without global atomics (round 5).  The float atomics of PHASE 2 each cost a 64-byte
// memory transaction (profiles/r05_streaming_64x32768.json: 790 MB of HBM traffic for 4 M points' 24-byte updates, 612 us).
// Here a block OWNS a tile of kGradTile target rows of one cloud and direction: it walks the index array of the other cloud,
// accumulates the terms that land in its tile in LDS (ds_add_f32), and adds the tile to the gradient with plain coalesced
// read-modify-writes -- every output row has exactly one owner, and the own-row launch (PHASE 1) has finished before.
// grid (tiles, b, 2).
constexpr int kGradTile = 4096;
constexpr int kGradBlock = 1024;      // (64 x 32768 both ways: 122 us; 512 threads 144, 256 threads 228 before the loads were batched; capped at 64 VGPRs for two blocks per CU 159)
__global__ __launch_bounds__(kGradBlock) void chamfer_grad_scatter_tiled_kernel(int n, const float *__restrict__ xyz1, int m,
                                                                            const float *__restrict__ xyz2,
                                                                            const float *__restrict__ gd1, const int *__restrict__ idx1,
                                                                            const float *__restrict__ gd2, const int *__restrict__ idx2,
                                                                            float *__restrict__ gx1, float *__restrict__ gx2)
{
    __shared__ float acc[kGradTile * 3];
    const int dir = blockIdx.z, e = blockIdx.y;
    // direction 0: queries = cloud 1 (n points), targets = cloud 2 (m rows of gx2); direction 1 the converse
    const int nq = dir ? m : n, nt = dir ? n : m;
    const int t0 = blockIdx.x * kGradTile;
    if (t0 >= nt) return;
    const float *__restrict__ Q = (dir ? xyz2 : xyz1) + (size_t)e * nq * 3;
    const float *__restrict__ T = (dir ? xyz1 : xyz2) + (size_t)e * nt * 3;
    const float *__restrict__ G = (dir ? gd2 : gd1) + (size_t)e * nq;
    const int *__restrict__ I = (dir ? idx2 : idx1) + (size_t)e * nq;
    float *__restrict__ O = (dir ? gx1 : gx2) + (size_t)e * nt * 3;
    const int rows = min(kGradTile, nt - t0);
    for (int i = threadIdx.x; i < rows * 3; i += kGradBlock) acc[i] = 0.0f;
    __syncthreads();
    // Eight entries per thread and trip, every load of a trip issued before anything waits: the indices, then -- for all eight,
    // hit or not, at clamped addresses -- the query row, its weight and the target row (L2 hits: a cloud is half a MiB), then
    // the LDS adds of the hits.  (One entry in eight lands in the tile; fetched only for the hits, behind a branch per entry,
    // the seven dependent loads of a hit were a round trip each, 128 times per thread: 228 us for 64 x 32768 both ways.)
    for (int q0 = threadIdx.x; q0 < nq; q0 += 8 * kGradBlock) {
        int kk[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = q0 + u * kGradBlock;
            kk[u] = q < nq ? I[q] - t0 : -1;
        }
        float qx[8], qy[8], qz[8], gg[8], tx[8], ty[8], tz[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool ok = (unsigned)kk[u] < (unsigned)rows;
            const int q = ok ? q0 + u * kGradBlock : 0, k = ok ? kk[u] : 0;
            qx[u] = Q[(size_t)q * 3 + 0]; qy[u] = Q[(size_t)q * 3 + 1]; qz[u] = Q[(size_t)q * 3 + 2];
            gg[u] = G[q];
            tx[u] = T[(size_t)(t0 + k) * 3 + 0]; ty[u] = T[(size_t)(t0 + k) * 3 + 1]; tz[u] = T[(size_t)(t0 + k) * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = kk[u];
            if ((unsigned)k >= (unsigned)rows) continue;
            const float g = __fmul_rn(gg[u], 2.0f);
            atomicAdd(&acc[k * 3 + 0], -__fmul_rn(g, qx[u] - tx[u]));
            atomicAdd(&acc[k * 3 + 1], -__fmul_rn(g, qy[u] - ty[u]));
            atomicAdd(&acc[k * 3 + 2], -__fmul_rn(g, qz[u] - tz[u]));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * 3; i += kGradBlock) O[(size_t)t0 * 3 + i] = __fadd_rn(O[(size_t)t0 * 3 + i], acc[i]);
}

// Large calls, round 6: ONE pass per direction computes every term once and puts it in both places (chamfer3D.cu:157-171) --
// a block owns a tile of kGradTile target rows of one cloud; it walks the other cloud's index array, and for the entries that
// land in its tile (exactly one block per entry) it evaluates the term, adds it to the QUERY's own gradient row (a plain
// read-modify-write: that row has no other writer in this launch) and accumulates its negative in the tile's LDS copy, which
// is added to the gradient at the end (a plain coalesced read-modify-write: the tile has one owner).  Direction 0 and
// direction 1 are two launches (each cloud's gradient is "own rows" in one and "tiles" in the other).  Against round 5's
// own-row launch + tile launch: the rows, weights and indices are read once, not twice; the hits of a wave's 512 entries are
// compacted so that only they issue loads (round 5 loaded for all eight entries of a thread at clamped addresses: seven of
// eight load instructions fetched nothing); the blocks of a batch element share an XCD, so the tiles' walks of the same index
// array and the partial lines of the own rows meet in one L2.
constexpr int kGradList = 128;     // compacted hits a wave keeps per trip (more: the hits are dense, the lanes take them in place)
template <int TILE>
__global__ __launch_bounds__(kGradBlock) void chamfer_grad_dir_kernel(int b, int tiles, int nq, const float *__restrict__ Qc, int nt,
                                                                      const float *__restrict__ Tc, const float *__restrict__ Gd,
                                                                      const int *__restrict__ Ix, float *__restrict__ Oq,
                                                                      float *__restrict__ Ot)
{
    __shared__ float acc[TILE * 3];
    __shared__ unsigned long long lst[kGradBlock / kWave][kGradList];      // query << 32 | row of the tile
    int e, tile;
    {
        const int lin = blockIdx.x;
        if ((b & 7) == 0) { const int xcd = lin & 7, k = lin >> 3; e = 8 * (k / tiles) + xcd; tile = k % tiles; }
        else { e = lin / tiles; tile = lin % tiles; }
    }
    if (e >= b) return;
    const int t0 = tile * TILE;
    const float *__restrict__ Q = Qc + (size_t)e * nq * 3;
    const float *__restrict__ T = Tc + (size_t)e * nt * 3;
    const float *__restrict__ G = Gd + (size_t)e * nq;
    const int *__restrict__ I = Ix + (size_t)e * nq;
    float *__restrict__ OQ = Oq + (size_t)e * nq * 3;
    float *__restrict__ OT = Ot + (size_t)e * nt * 3;
    const int rows = min(TILE, nt - t0);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < rows * 3; i += kGradBlock) acc[i] = 0.0f;
    __syncthreads();
    auto term = [&](int q, int k) {
        const float qx = Q[(size_t)q * 3 + 0], qy = Q[(size_t)q * 3 + 1], qz = Q[(size_t)q * 3 + 2];
        const float g = __fmul_rn(G[q], 2.0f);
        const float tx = T[(size_t)(t0 + k) * 3 + 0], ty = T[(size_t)(t0 + k) * 3 + 1], tz = T[(size_t)(t0 + k) * 3 + 2];
        const float o0 = OQ[(size_t)q * 3 + 0], o1 = OQ[(size_t)q * 3 + 1], o2 = OQ[(size_t)q * 3 + 2];
        const float vx = __fmul_rn(g, qx - tx), vy = __fmul_rn(g, qy - ty), vz = __fmul_rn(g, qz - tz);
        OQ[(size_t)q * 3 + 0] = __fadd_rn(o0, vx);
        OQ[(size_t)q * 3 + 1] = __fadd_rn(o1, vy);
        OQ[(size_t)q * 3 + 2] = __fadd_rn(o2, vz);
        atomicAdd(&acc[k * 3 + 0], -vx);
        atomicAdd(&acc[k * 3 + 1], -vy);
        atomicAdd(&acc[k * 3 + 2], -vz);
    };
    for (int q0 = threadIdx.x; q0 - (int)threadIdx.x < nq; q0 += 8 * kGradBlock) {      // (wave-uniform trip count)
        int kk[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = q0 + u * kGradBlock;
            kk[u] = q < nq ? I[q] - t0 : -1;
        }
        // the wave's hits among its 512 entries
        int pos[8], total = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool hit = (unsigned)kk[u] < (unsigned)rows;
            const unsigned long long m = __ballot(hit);
            pos[u] = hit ? total + (int)__popcll(m & ((1ull << lane) - 1ull)) : -1;
            total += (int)__popcll(m);
        }
        if (total == 0) continue;
        if (total <= kGradList) {
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (pos[u] >= 0) lst[wave][pos[u]] = ((unsigned long long)(unsigned)(q0 + u * kGradBlock) << 32) | (unsigned)kk[u];
            // (the list is this wave's own: LDS operations of a wave are performed in order)
            for (int i = lane; i < total; i += kWave) {
                const unsigned long long w = lst[wave][i];
                term((int)(w >> 32), (int)(unsigned)w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (pos[u] >= 0) term(q0 + u * kGradBlock, kk[u]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * 3; i += kGradBlock) OT[(size_t)t0 * 3 + i] = __fadd_rn(OT[(size_t)t0 * 3 + i], acc[i]);
}

struct NNConfig {
    int r;                // VALU path, queries per lane: 0 = pick, else 2 or 4
    int blocks_per_cu;    // occupancy target used to pick the slice count
    int mfma;             // 4: cell-sorted pruned search (opt-in: wins when most queries have a near target), 3: two-piece f16 MFMA filter (default), 1: fp32 MFMA filter (small problems), 0: VALU path (cross-check).  (2 was the bf16 filter, removed in round 3.)
    int q;                // MFMA path, 32-query tiles per wave: 0 = pick, else 1 or 2
    int u;                // MFMA path, tiles per bookkeeping unit: 0 = pick, else 1 or 2
    bool env_path, env_wps;   // GENPC_NN_PATH / GENPC_NN_WPS were given
    int dbg;              // GENPC_NN_DEBUG
};

// genpc_nn_tune() overrides (-1: use the environment / default).  Thread-local: a test that
// switches kernel families does not change what other host threads launch.
thread_local int t_tune_path = -1, t_tune_hooks = -1;      // (declared in common.h: genpc_thread_state_export / _import)

// Tunables; the environment is read ONCE, at first use (C++11 static initialisation is
// thread-safe), never per launch.
static const NNConfig &nn_config()
{
    static const NNConfig c = [] {
        NNConfig k{0, 4, 3, 0, 0, false, false, 0};
        k.r = tune_env("GENPC_NN_R", 0, "VALU nearest-neighbour path: queries per lane (2 | 4, 0 = pick)");
        if (const char *e = tune_env_str("GENPC_NN_PATH", "nearest-neighbour kernel family: valu | mfma32 | f16 (default) | grid (opt-in)")) {
            // (anything else used to fall through to the opt-in cell search -- e.g. 'bf16', a family removed in round 3)
            if (e[0] == 'v') k.mfma = 0;
            else if (e[0] == 'm') k.mfma = 1;
            else if (e[0] == 'f') k.mfma = 3;
            else if (e[0] == 'g') k.mfma = 4;
            else fprintf(stderr, "genpc_hip: GENPC_NN_PATH=%s is not one of valu | mfma32 | f16 | grid: keeping the default (f16)\n", e);
            k.env_path = e[0] == 'v' || e[0] == 'm' || e[0] == 'f' || e[0] == 'g';
        }
        k.q = tune_env("GENPC_NN_Q", 0, "MFMA filter: 32-query tiles per wave (2 | 4, 0 = pick)");
        k.u = tune_env("GENPC_NN_U", 0, "MFMA filter: target tiles per bookkeeping unit (2 | 4, 0 = pick)");
        if (k.q != 1 && k.q != 2 && k.q != 4) k.q = 0;
        if (k.u != 1 && k.u != 2 && k.u != 4) k.u = 0;
        {
            const int w = tune_env("GENPC_NN_WPS", 0, "resident blocks per CU the nearest-neighbour planner assumes (0 = per kernel family)");
            if (w > 0) { k.blocks_per_cu = w; k.env_wps = true; }
        }
        k.dbg = tune_env("GENPC_NN_DEBUG", 0, "test hooks of the filtered nearest-neighbour paths (bit mask, see genpc_nn_tune)");
        if (k.r != 2 && k.r != 4) k.r = 0;
        if (k.blocks_per_cu < 1) k.blocks_per_cu = 1;
        return k;
    }();
    return c;
}

template <int Q, int U>
static void launch_mfma(const NNArgs &a, int blocks, hipStream_t st)
{
    if (a.fma)
        hipLaunchKernelGGL((nn_mfma_kernel<Q, U, 1>), dim3(blocks), dim3(kBlock), 0, st, a);
    else
        hipLaunchKernelGGL((nn_mfma_kernel<Q, U, 0>), dim3(blocks), dim3(kBlock), 0, st, a);
}

template <int R>
static void launch_r(const NNArgs &a, int blocks, hipStream_t st)
{
    if (a.fma)
        hipLaunchKernelGGL((nn_forward_kernel<R, 1>), dim3(blocks), dim3(kBlock), 0, st, a);
    else
        hipLaunchKernelGGL((nn_forward_kernel<R, 0>), dim3(blocks), dim3(kBlock), 0, st, a);
}

// Duplicate pre-pass policy of the f16 filter (nn_dedupe.hip).  The pre-pass is two launches with three dependent
// memory round trips each: 26 us on an idle GPU whatever the size, 59 us at 8 x 32768 x 2 points -- against 33 us for
// the whole headline step.  So it runs
//   * when the caller brings the masks (the alignment loop, ICP and the scale search make them once per call: their
//     clouds' duplicates are the same at every step / for every candidate), or
//   * while the input asks for it.  The finish kernel adds (a sample of) the queries it had to re-do exhaustively to
//     a host-visible word, the pre-pass adds (a sample of) the copies it found to another; the NEXT call reads both:
//     many re-dos switch the pre-pass on, a pre-pass that finds next to nothing switches it off again.  The words are
//     read without any synchronisation -- a stale value delays the switch by a call; results do not depend on it (the
//     masks only remove targets that can never be reported).
// GENPC_NN_DEDUPE=0 / 1 forces it off / on (A/B; tests: hooks 2048 / 4096 of genpc_nn_tune).
struct DedupeHint {
    unsigned *host = nullptr;          // cumulative, written by the GPU: [0] exhaustive re-dos, [1] copies found, [2] pre-passes completed (hipHostMalloc, mapped)
    std::mutex mu;
    unsigned seen[3] = {0, 0, 0};      // what the host has accounted for
    bool on = false;
};
static DedupeHint *dedupe_hint()
{
    static DedupeHint hints[16];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    DedupeHint &h = hints[dev];
    if (!h.host) {
        std::lock_guard<std::mutex> l(mu);
        if (!h.host) {
            void *p = nullptr;
            if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
            for (int i = 0; i < 16; i++) ((volatile unsigned *)p)[i] = 0;
            h.host = (unsigned *)p;
        }
    }
    return &h;
}

int nn_forward(int b, int ndir, const float *q0, int n0, const float *t0, int m0, float *d0, int *i0,
               const float *q1, int n1, const float *t1, int m1, float *d1, int *i1, hipStream_t st,
               float radius2, const unsigned *dup0, const unsigned *dup1, int dup_shared)
{
    // A direction with no queries or no targets does nothing (the reference's
    // loops do not execute, outputs keep the caller's zeros).
    NNConfig cfg = nn_config();
    if (t_tune_path >= 0) cfg.mfma = t_tune_path;
    NNArgs a{};
    a.b = b;
    a.fma = arith_mode() != 0 ? 1 : 0;
    a.radius2 = radius2;
    a.debug = t_tune_hooks >= 0 ? t_tune_hooks : cfg.dbg;
    if (a.debug & 512) {
        a.stats = (unsigned long long *)workspace(12, 256, nullptr, nullptr, 256);      // one block per device, shared by all streams
        if (!a.stats) return 0;
    }
    const float *qs[2] = {q0, q1};
    const float *ts[2] = {t0, t1};
    float *ds[2] = {d0, d1};
    int *is[2] = {i0, i1};
    const unsigned *dups[2] = {dup0, dup1};
    int nqs[2] = {n0, n1}, nts[2] = {m0, m1};
    int nd = 0;
    int nt_max = 0;
    for (int d = 0; d < ndir; d++) {
        if (b <= 0 || nqs[d] <= 0 || nts[d] <= 0) continue;
        NNDir &D = a.dir[nd++];
        D.q = qs[d]; D.t = ts[d]; D.out_d = ds[d]; D.out_i = is[d];
        D.nq = nqs[d]; D.nt = nts[d];
        D.dupmask = dups[d];
        D.dup_shared = dups[d] ? dup_shared : 0;
        if (D.nt > nt_max) nt_max = D.nt;
    }
    a.ndir = nd;
    if (nd == 0) return 1;
    int path = cfg.mfma;
    double pairs = 0.0;
    for (int d = 0; d < nd; d++) pairs += (double)b * a.dir[d].nq * a.dir[d].nt;
    // measured on MI355X (tools/nn_sweep.py): below ~6M pairs the single-launch fp32-MFMA
    // kernel wins (1x1024^2 11.0 vs 13.5 us), from 2048^2 on the two-launch f16 filter
    if (path == 3 && pairs < 6e6 && t_tune_path < 0 && !cfg.env_path) path = 1;
    if (radius2 < __builtin_inff()) path = 4;      // only the cell search knows how to stop at a distance
    if (path == 4) {
        // three launches, O(N + M) work (nn_grid.hip); needs both directions to be each other's swap
        if (nd == 1 || (a.dir[1].q == a.dir[0].t && a.dir[1].t == a.dir[0].q)) return launch_nn_grid(a, st);
        path = 3;
    }
    if (path == 2) path = 3;                             // (the bf16 filter's number: gone)
    const bool f16 = path == 3;
    if (f16) path = 2;                                   // below, 2 = "filter + finish kernel" planning
    if (path == 2 && nt_max >= (1 << 25)) path = 1;      // finish kernel packs tile indices in 21 bits
    // R = 4 (fewer LDS reads per pair, more independent chains per lane) when the query
    // blocks alone fill the chip; R = 2 otherwise: twice the blocks, half the per-wave
    // epilogue (measured on MI355X: 1x16384^2 87 us vs 97 us, 13x16384^2 870 us vs 835 us).
    long long want_blocks = (long long)num_cus() * cfg.blocks_per_cu;
    int r = cfg.r;
    if (!r) {
        long long unsplit4 = 0;
        for (int d = 0; d < nd; d++) unsplit4 += (long long)b * ceil_div(a.dir[d].nq, kBlock * 4);
        r = unsplit4 >= num_cus() ? 4 : 2;
    }
    // MFMA path: a block covers 128*Q queries; Q = 2 halves the LDS reads and the blocks.
    int q = cfg.q;
    if (path && !q) {
        long long unsplit2 = 0;
        for (int d = 0; d < nd; d++) unsplit2 += (long long)b * ceil_div(a.dir[d].nq, 256);
        // fp32 MFMA: Q = 2 measured slower at every size from 1x2048^2 to 13x16384^2
        // f16 filter: 512-query blocks (Q = 4) once there is enough work to fill the chip with
        // them (1x16384^2 37.8 vs 49.7 us), 256-query blocks below (1x8192^2 20.4 vs 21.6 us)
        q = path == 2 ? (pairs >= 2e8 ? 4 : 2) : 1;
        (void)unsplit2;
    }
    // f16 filter: at least two accumulator chains per wave (see the hazard note in nn_f16.hip)
    if (path == 2 && q < 2) q = 2;
    if (path == 2 && !cfg.env_wps) want_blocks = (long long)num_cus() * (q == 4 ? 2 : (q == 2 ? 3 : 4));   // resident blocks per CU (VGPRs)
    const int qper = path ? 128 * q : kBlock * r;       // queries per block
    const int gran = path == 2 ? 128 : (path ? 64 : kChunk);   // slice granularity: one bookkeeping unit
    int pwords = path ? 3 : 1;                          // 8-byte words per (slice, query)
    // One slice length for the whole launch, so that every block does the same amount
    // of work, and per-direction slice counts S_d = ceil(nt_d / slice_len) (no empty
    // slices when the two clouds differ in size).  slice_len is what makes the launch
    // ~want_blocks blocks: sum_d b * qblocks_d * nt_d / want_blocks, but never below 8
    // chunks, and not below 16 chunks (512 targets: below that the per-block staging /
    // merge latency outweighs the parallelism; measured 1x8192^2 44 us vs 51 us) as long
    // as two blocks per CU remain.
    long long work = 0;          // blocks x targets if every block took one target
    for (int d = 0; d < nd; d++) {
        a.dir[d].qblocks = ceil_div(a.dir[d].nq, qper);
        work += (long long)b * a.dir[d].qblocks * a.dir[d].nt;
    }
    auto blocks_at = [&](long long len) {
        long long t = 0;
        for (int d = 0; d < nd; d++) t += (long long)b * a.dir[d].qblocks * ceil_div64(a.dir[d].nt, len);
        return t;
    };
    long long len = ceil_div64(ceil_div64(work, want_blocks), gran) * gran;
    if (len < kChunk * 8) len = kChunk * 8;
    if (len < kChunk * 16 && blocks_at(kChunk * 16) >= 2 * num_cus()) len = kChunk * 16;
    if (len > nt_max) len = ceil_div64(nt_max, gran) * gran;
    a.slice_len = (int)len;
    // bookkeeping per 64 targets once a block has enough of them to amortise the coarser
    // re-scan (measured: 1x2048^2 14.5 vs 16.0 us, 1x16384^2 63.6 vs 61.2, 13x16384^2 644 vs 590)
    const int u = (cfg.u == 1 || cfg.u == 2) ? cfg.u : (len >= 2048 ? 2 : 1);
    // filter path: the launch runs in rounds of `want_blocks` resident blocks, so the
    // slice count is chosen to minimise rounds x (targets per block + a fixed per-block
    // cost worth ~192 targets), over the slice counts that keep a query's candidate lists
    // (slices x NL; NL = 2 lists per lane below three slices: a query is flagged only when THREE
    // candidate units fall into one list) within kMaxLists = 16.
    int nl = 1;
    bool tight = false, wide = false;
    if (path == 2) {
        auto lists_of = [&](long long l, int &nl_out) {
            int smin = 1 << 30, smax = 0;
            for (int d = 0; d < nd; d++) {
                const int sd = (int)ceil_div64(a.dir[d].nt, l);
                smin = std::min(smin, sd);
                smax = std::max(smax, sd);
            }
            nl_out = smin >= 3 ? 1 : 2;
            return smax * nl_out;
        };
        // The f16 kernel's 512-query blocks (Q = 4) exist in two register budgets: 2 resident
        // blocks per CU, or 3 (168 VGPRs, a little scratch: ~4 % faster per pair over many rounds,
        // ~1 us slower when the launch is a single round).  All resident blocks share the VALU, so
        // a round costs (resident blocks) x (targets per block + fixed part).
        long long best_len = 0;
        double best_cost = 0.0;
        int best_res = 0;
        const bool env_wps = cfg.env_wps;
        for (int res = (f16 && q == 4 && !env_wps) ? 3 : 0; res != 1 && res >= 0; res = (res == 3 ? 2 : -1)) {
            const long long slots = res ? (long long)num_cus() * res : want_blocks;
            const double per_block = res == 3 ? 3.0 * 0.96 : (res == 2 ? 2.0 : 1.0);
            for (int sc = 1; sc <= 16; sc++) {
                const long long l = ceil_div64(ceil_div64(nt_max, sc), gran) * gran;
                if (sc > 1 && l < 256) break;
                int nl_c;
                if (lists_of(l, nl_c) > 16) continue;
                const long long rounds = ceil_div64(blocks_at(l), slots);
                if (res == 3 && rounds < 3) continue;         // few rounds: the leaner kernel wins
                const double cost = (double)rounds * (double)(std::min<long long>(l, nt_max) + 192) * per_block;
                if (!best_len || cost < best_cost * 0.98) {      // prefer fewer slices unless clearly better
                    best_len = l;
                    best_cost = cost;
                    best_res = res;
                }
            }
        }
        tight = best_res == 3;
        len = best_len;
        (void)lists_of(len, nl);
        a.slice_len = (int)len;
        // Single-round launches whose slice fits the kernel's 2048-target LDS tile: blocks of 8 waves / 1024 queries,
        // one per CU -- the same two waves per SIMD, but a slice is read, split into f16 pieces and staged once per
        // 1024 queries instead of once per 512 (the prologue was a third of a block's time: tools/nn_timeline.py).
        static const bool no_wide = tune_env("GENPC_NN_NOWIDE", 0, "f16 filter: 1 = no 8-wave blocks for single-round launches") != 0;
        if (f16 && q == 4 && !tight && !no_wide && len > 1024 && blocks_at(len) <= 2 * (long long)num_cus()) {
            wide = true;
            for (int d = 0; d < nd; d++) a.dir[d].qblocks = ceil_div(a.dir[d].nq, 2 * qper);
        }
        pwords = 2 * nl;        // (a1, c1) (codes of a2 | a3, c2) per list: list_enc in nn.h
    }
    long long tb = 0;
    int units = 0;
    size_t part = 0;
    bool any_split = false;
    for (int d = 0; d < nd; d++) {
        NNDir &D = a.dir[d];
        D.slices = ceil_div(D.nt, a.slice_len);
        D.block_begin = (int)tb;
        D.unit_begin = units;
        tb += (long long)D.slices * b * D.qblocks;
        units += b * D.qblocks;
        if (D.slices > 1 || path == 2) {      // the filter always hands its lists to a second launch
            part += (size_t)D.slices * b * D.nq * pwords;
            any_split = true;
        }
    }
    if (tb > 0x7fffffffLL) {
        set_error("chamfer: problem too large for one launch");
        return 0;
    }
    if (any_split) {
        // [arrival counters | per-slice partials]; counters are zeroed when the block is
        // (re)allocated and restored to zero by the merging block, so steady-state calls
        // need no memset.
        // Usually the counters fit the 64 KiB prefix that is zero between calls.  A launch with more
        // query blocks than that (1000 scale-search candidates x 128 blocks) takes a larger area,
        // which earlier calls used for partials: it is cleared on the stream first (the kernels hand
        // every counter back as zero, so the 64 KiB invariant survives such a call).
        const size_t cnt_min = (size_t)1 << 16;
        size_t cnt_bytes = cnt_min;
        if ((size_t)units * sizeof(int) > cnt_bytes) cnt_bytes = ((size_t)units * sizeof(int) + 255) & ~(size_t)255;
        char *ws = (char *)workspace(0, cnt_bytes + part * 8, st, nullptr, cnt_min);
        if (!ws) return 0;
        if (cnt_bytes > cnt_min && !check(hipMemsetAsync(ws, 0, cnt_bytes, st), "hipMemsetAsync(arrival counters)")) return 0;
        a.arrive = (int *)ws;
        unsigned long long *wp = (unsigned long long *)(ws + cnt_bytes);
        size_t off = 0;
        for (int d = 0; d < nd; d++) {
            if (a.dir[d].slices > 1 || path == 2) {
                a.dir[d].part = wp + off;
                off += (size_t)a.dir[d].slices * b * a.dir[d].nq * pwords;
            }
        }
    }
    if (path == 2) {
        // exact duplicates among the targets (policy above nn_forward)
        static const int env_dd = tune_env("GENPC_NN_DEDUPE", -1, "f16 filter: exact-duplicate pre-pass 0 off / 1 on (-1: callers' masks + adaptive policy)");
        const int force = (a.debug & 2048) ? 0 : ((a.debug & 4096) ? 1 : env_dd);
        DedupeHint *H = dedupe_hint();
        bool own_masks = true;
        for (int d = 0; d < nd; d++) own_masks = own_masks && a.dir[d].dupmask != nullptr;
        if (force == 0) {
            for (int d = 0; d < nd; d++) a.dir[d].dupmask = nullptr;
        } else if (!own_masks) {
            long long queries = 0, targets = 0;
            for (int d = 0; d < nd; d++) { queries += (long long)b * a.dir[d].nq; targets += (long long)b * a.dir[d].nt; }
            bool run = force == 1;
            if (!run && H) {
                // Evidence the GPU has delivered since the last look (cumulative words: a host that runs ahead of the
                // GPU simply sees no news and keeps its state): re-dos of calls that ran without the pre-pass switch
                // it on, completed pre-passes that found next to nothing switch it off.
                std::lock_guard<std::mutex> l(H->mu);
                volatile unsigned *hv = H->host;
                const unsigned redo = hv[0] - H->seen[0], found = hv[1] - H->seen[1], done = hv[2] - H->seen[2];
                H->seen[0] += redo; H->seen[1] += found; H->seen[2] += done;
                if (!H->on) {
                    if ((long long)redo * 32 > queries) H->on = true;
                } else if (done > 0 && (long long)(found / done) * 32 < targets) {
                    H->on = false;
                }
                run = H->on;
            }
            if (run) {
                // one pre-pass per distinct target cloud (the two directions' targets are each other's queries)
                const float *pp[2];
                int pn[2];
                unsigned *pm[2];
                size_t words[2];
                int nc = 0;
                for (int d = 0; d < nd; d++) {
                    pp[nc] = a.dir[d].t;
                    pn[nc] = a.dir[d].nt;
                    words[nc] = (nn_dedupe_mask_words(b, pn[nc]) + 63) & ~(size_t)63;
                    nc++;
                }
                unsigned *mw = (unsigned *)workspace(25, (words[0] + (nc > 1 ? words[1] : 0)) * sizeof(unsigned), st);
                if (!mw) return 0;
                pm[0] = mw;
                pm[1] = mw + words[0];
                if (!launch_nn_dedupe(b, nc, pp, pn, pm, H ? H->host + 1 : nullptr, st)) return 0;
                for (int d = 0; d < nd; d++) { a.dir[d].dupmask = pm[d]; a.dir[d].dup_shared = 0; }
            }
        }
        a.hint = H ? H->host : nullptr;
        // bookkeeping unit: the finish kernel re-reads 16 targets per tile of every candidate unit --
        // per QUERY, while the filter's work is per PAIR: short target clouds (many queries per
        // pair) take 64-target units (half the re-read, +10 % filter VALU), long ones 128
        const int fu = cfg.u == 2 || cfg.u == 4 ? cfg.u : (nt_max <= 8192 ? 2 : 4);
        return launch_nn_f16(a, q, fu, nl, wide ? 2 : (tight ? 1 : 0), tb, st);
    } else if (path) {
        if (q == 2) { if (u == 2) launch_mfma<2, 2>(a, (int)tb, st); else launch_mfma<2, 1>(a, (int)tb, st); }
        else        { if (u == 2) launch_mfma<1, 2>(a, (int)tb, st); else launch_mfma<1, 1>(a, (int)tb, st); }
    } else if (r == 4) {
        launch_r<4>(a, (int)tb, st);
    } else {
        launch_r<2>(a, (int)tb, st);
    }
    return check(hipGetLastError(), "nn_forward_kernel launch") ? 1 : 0;
}

}  // namespace genpc

GENPC_API int genpc_nn_tune(int path, int hooks)
{
    const int prev = genpc::t_tune_path >= 0 ? genpc::t_tune_path : genpc::nn_config().mfma;
    if (path >= 0 && path <= 4 && path != 2) genpc::t_tune_path = path;
    if (hooks >= 0) genpc::t_tune_hooks = hooks;
    return prev;
}

GENPC_API int genpc_nn_stats(unsigned long long out[3], int reset, void *stream)
{
    using namespace genpc;
    unsigned long long *dev = (unsigned long long *)workspace(12, 256, nullptr, nullptr, 256);
    if (!dev) return 0;
    if (!check(hipStreamSynchronize((hipStream_t)stream), "genpc_nn_stats sync")) return 0;
    if (!check(hipMemcpy(out, dev, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost), "genpc_nn_stats copy")) return 0;
    if (reset && !check(hipMemset(dev, 0, 256), "genpc_nn_stats reset")) return 0;
    return 1;
}

GENPC_API int genpc_chamfer_forward(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                                    int *idx1, float *dist2, int *idx2, void *stream)
{
    return genpc::nn_forward(b, 2, xyz1, n, xyz2, m, dist1, idx1, xyz2, m, xyz1, n, dist2, idx2,
                             (hipStream_t)stream);
}

GENPC_API int genpc_nm_distance(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                                int *result_i, void *stream)
{
    return genpc::nn_forward(b, 1, xyz, n, xyz2, m, result, result_i, nullptr, 0, nullptr, 0, nullptr, nullptr,
                             (hipStream_t)stream);
}

GENPC_API int genpc_nm_distance_within(int b, int n, const float *xyz, int m, const float *xyz2, float radius2,
                                       float *result, int *result_i, void *stream)
{
    if (!(radius2 >= 0.0f)) {
        genpc::set_error("genpc_nm_distance_within: radius2 must be >= 0");
        return -1;
    }
    return genpc::nn_forward(b, 1, xyz, n, xyz2, m, result, result_i, nullptr, 0, nullptr, 0, nullptr, nullptr,
                             (hipStream_t)stream, radius2);
}

GENPC_API int genpc_chamfer_backward(int b, int n, const float *xyz1, int m, const float *xyz2,
                                     const float *graddist1, const int *idx1, const float *graddist2,
                                     const int *idx2, float *gradxyz1, float *gradxyz2, void *stream)
{
    using namespace genpc;
    // The reference indexes xyz2 with idx1 unconditionally; with an empty cloud
    // there is nothing to differentiate.
    if (b <= 0 || n <= 0 || m <= 0) return 1;
    long long total = (long long)b * n + (long long)b * m;
    long long blocks = ceil_div64(total, kBlock);
    if (blocks > 0x7fffffffLL) {
        set_error("chamfer backward: problem too large for one launch");
        return 0;
    }
    static const int env_split = tune_env("GENPC_CHAMFER_GRAD_SPLIT", 262144, "chamfer backward: points (both clouds) from which the gradient is accumulated through LDS tiles that own their output rows instead of global atomics (0 = never)");
    if (env_split > 0 && (long long)b * ((long long)n + m) >= env_split) {
        static const int env_tiled = tune_env("GENPC_CHAMFER_GRAD_TILED", 2, "chamfer backward, large calls: 2 = one pass per direction (own rows + tiles fused), 1 = own rows and tiles as two launches (round 5), 0 = global atomics");
        static const int env_tile = tune_env("GENPC_CHAMFER_GRAD_TILE", 4096, "chamfer backward, large calls: target rows a block owns (2048 | 4096 | 8192)");
        const int tile_rows = env_tile == 2048 || env_tile == 8192 ? env_tile : 4096;
        const int tiles_n = ceil_div(n, tile_rows), tiles_m = ceil_div(m, tile_rows);
        const long long bb = (b & 7) == 0 ? b : b;      // (blocks per tile column: the batch; a multiple of eight is dealt XCD by XCD)
        if (env_tiled == 2 && bb * (tiles_n > tiles_m ? tiles_n : tiles_m) <= 0x7fffffffLL) {
            // direction 0: queries = cloud 1, tiles of cloud 2's gradient; direction 1 the converse
#define GENPC_GRAD_DIR(TILE)                                                                                                          \
            do {                                                                                                                      \
                hipLaunchKernelGGL(chamfer_grad_dir_kernel<TILE>, dim3((unsigned)(b * tiles_m)), dim3(kGradBlock), 0, (hipStream_t)stream, b, tiles_m, \
                                   n, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);                                            \
                hipLaunchKernelGGL(chamfer_grad_dir_kernel<TILE>, dim3((unsigned)(b * tiles_n)), dim3(kGradBlock), 0, (hipStream_t)stream, b, tiles_n, \
                                   m, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);                                            \
            } while (0)
            if (tile_rows == 2048) GENPC_GRAD_DIR(2048);
            else if (tile_rows == 8192) GENPC_GRAD_DIR(8192);
            else GENPC_GRAD_DIR(4096);
#undef GENPC_GRAD_DIR
            return check(hipGetLastError(), "chamfer_grad_dir_kernel launch") ? 1 : 0;
        }
        hipLaunchKernelGGL(chamfer_grad_kernel<1>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, b, n, xyz1, m,
                           xyz2, graddist1, idx1, graddist2, idx2, gradxyz1, gradxyz2);
        const int tiles = ceil_div(n > m ? n : m, kGradTile);
        if (env_tiled && tiles <= 65535 && b <= 65535)
            hipLaunchKernelGGL(chamfer_grad_scatter_tiled_kernel, dim3(tiles, b, 2), dim3(kGradBlock), 0, (hipStream_t)stream, n, xyz1, m, xyz2,
                               graddist1, idx1, graddist2, idx2, gradxyz1, gradxyz2);
        else
            hipLaunchKernelGGL(chamfer_grad_kernel<2>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, b, n, xyz1, m,
                               xyz2, graddist1, idx1, graddist2, idx2, gradxyz1, gradxyz2);
    } else {
        hipLaunchKernelGGL(chamfer_grad_kernel<0>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, b, n, xyz1, m,
                           xyz2, graddist1, idx1, graddist2, idx2, gradxyz1, gradxyz2);
    }
    return check(hipGetLastError(), "chamfer_grad_kernel launch") ? 1 : 0;
}
