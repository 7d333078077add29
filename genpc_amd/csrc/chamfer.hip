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
#include "common.h"
#include "../../include/genpc_hip.h"

#include <stdlib.h>

namespace genpc {

constexpr int kChunk = 32;       // targets per min-only chunk (re-scan granularity)
constexpr int kBlock = 256;      // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kTile = 2048;      // targets per LDS tile (24 KiB: 12 B per target)

template <int FMA>
__device__ __forceinline__ float sqdist(float dx, float dy, float dz)
{
    if (FMA) {
        float t = __fmul_rn(dy, dy);
        t = __fmaf_rn(dx, dx, t);
        return __fmaf_rn(dz, dz, t);
    } else {
        float a = __fmul_rn(dx, dx);
        float b = __fmul_rn(dy, dy);
        float c = __fmul_rn(dz, dz);
        return __fadd_rn(__fadd_rn(a, b), c);
    }
}

struct NNDir {
    const float *q;    // queries  [B, nq, 3]
    const float *t;    // targets  [B, nt, 3]
    float *out_d;      // final    [B, nq]
    int *out_i;
    unsigned long long *part;   // per-slice partials [S, B*nq] (S > 1): distance bits << 32 | chunk
    int nq, nt;
    int qblocks;       // ceil(nq / (256*R))
    int slices;        // S for this direction: ceil(nt / slice_len)
    int block_begin;   // first block id of this direction
    int unit_begin;    // first arrival counter of this direction
};

struct NNArgs {
    NNDir dir[2];
    int ndir;
    int b;
    int slice_len;     // targets per slice (all blocks of a launch do equal work), multiple of kChunk
    int *arrive;       // [sum_d b*qblocks] arrival counters, zero between launches
    int debug;         // experiment switches (GENPC_NN_DEBUG): 1 skip index recovery, 2 skip merge, 4 skip main loop
};

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
    for (int t0 = k_begin; t0 < k_end && !(a.debug & 4); t0 += kTile) {
        const int tn = min(kTile, k_end - t0);
        const int tn_pad = (tn + kChunk - 1) / kChunk * kChunk;
        __syncthreads();                       // readers of the previous tile are done
        for (int t = threadIdx.x; t < tn_pad; t += kBlock) {
            float x, y, z;
            if (t < tn) {
                const float *tp = T + (size_t)(t0 + t) * 3;
                x = tp[0]; y = tp[1]; z = tp[2];
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
                const unsigned long long v = ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)bchunk[r];
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
            float bv = __uint_as_float((unsigned)(v >> 32));
            int bc = (int)(unsigned)v;
            for (int s2 = 1; s2 < D.slices; s2++) {
                v = __hip_atomic_load(P + (size_t)s2 * bnq + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        if (j < nq) {
            od[j] = best[r];
            oi[j] = base + first;
        }
    }
}

// Chamfer backward, both directions in one launch (chamfer3D.cu:155-195).
// Thread j < B*N: direction 1 term of point j of cloud 1; B*N <= j < B*(N+M):
// direction 2 term of point j-B*N of cloud 2.  Accumulates with fp32 atomics
// into the caller-zeroed gradients, exactly like the reference.
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
    atomicAdd(&O1[t * 3 + 0], vx);
    atomicAdd(&O1[t * 3 + 1], vy);
    atomicAdd(&O1[t * 3 + 2], vz);
    atomicAdd(&O2[j2 * 3 + 0], -vx);
    atomicAdd(&O2[j2 * 3 + 1], -vy);
    atomicAdd(&O2[j2 * 3 + 2], -vz);
}

struct NNConfig {
    int r;                // queries per lane: 0 = pick, else 2 or 4
    int blocks_per_cu;    // occupancy target used to pick the slice count
};

// Tunables; GENPC_NN_R / GENPC_NN_WPS override for experiments.
static NNConfig nn_config()
{
    static NNConfig c = [] {
        NNConfig k{0, 4};
        if (const char *e = getenv("GENPC_NN_R")) k.r = atoi(e);
        if (const char *e = getenv("GENPC_NN_WPS")) k.blocks_per_cu = atoi(e);
        if (k.r != 2 && k.r != 4) k.r = 0;
        if (k.blocks_per_cu < 1) k.blocks_per_cu = 1;
        return k;
    }();
    return c;
}

template <int R>
static void launch_r(const NNArgs &a, int blocks, hipStream_t st)
{
    if (arith_mode() != 0)
        hipLaunchKernelGGL((nn_forward_kernel<R, 1>), dim3(blocks), dim3(kBlock), 0, st, a);
    else
        hipLaunchKernelGGL((nn_forward_kernel<R, 0>), dim3(blocks), dim3(kBlock), 0, st, a);
}

static int nn_forward(int b, int ndir, const float *q0, int n0, const float *t0, int m0, float *d0, int *i0,
                      const float *q1, int n1, const float *t1, int m1, float *d1, int *i1, hipStream_t st)
{
    // A direction with no queries or no targets does nothing (the reference's
    // loops do not execute, outputs keep the caller's zeros).
    const NNConfig cfg = nn_config();
    NNArgs a{};
    a.b = b;
    static const int dbg = getenv("GENPC_NN_DEBUG") ? atoi(getenv("GENPC_NN_DEBUG")) : 0;
    a.debug = dbg;
    const float *qs[2] = {q0, q1};
    const float *ts[2] = {t0, t1};
    float *ds[2] = {d0, d1};
    int *is[2] = {i0, i1};
    int nqs[2] = {n0, n1}, nts[2] = {m0, m1};
    int nd = 0;
    int nt_max = 0;
    for (int d = 0; d < ndir; d++) {
        if (b <= 0 || nqs[d] <= 0 || nts[d] <= 0) continue;
        NNDir &D = a.dir[nd++];
        D.q = qs[d]; D.t = ts[d]; D.out_d = ds[d]; D.out_i = is[d];
        D.nq = nqs[d]; D.nt = nts[d];
        if (D.nt > nt_max) nt_max = D.nt;
    }
    a.ndir = nd;
    if (nd == 0) return 1;
    // R = 4 (fewer LDS reads per pair, more independent chains per lane) when the query
    // blocks alone fill the chip; R = 2 otherwise: twice the blocks, half the per-wave
    // epilogue (measured on MI355X: 1x16384^2 87 us vs 97 us, 13x16384^2 870 us vs 835 us).
    const long long want_blocks = (long long)kNumCU * cfg.blocks_per_cu;
    int r = cfg.r;
    if (!r) {
        long long unsplit4 = 0;
        for (int d = 0; d < nd; d++) unsplit4 += (long long)b * ceil_div(a.dir[d].nq, kBlock * 4);
        r = unsplit4 >= kNumCU ? 4 : 2;
    }
    // One slice length for the whole launch, so that every block does the same amount
    // of work, and per-direction slice counts S_d = ceil(nt_d / slice_len) (no empty
    // slices when the two clouds differ in size).  slice_len is what makes the launch
    // ~want_blocks blocks: sum_d b * qblocks_d * nt_d / want_blocks, but never below 8
    // chunks, and not below 16 chunks (512 targets: below that the per-block staging /
    // merge latency outweighs the parallelism; measured 1x8192^2 44 us vs 51 us) as long
    // as two blocks per CU remain.
    long long work = 0;          // blocks x targets if every block took one target
    for (int d = 0; d < nd; d++) {
        a.dir[d].qblocks = ceil_div(a.dir[d].nq, kBlock * r);
        work += (long long)b * a.dir[d].qblocks * a.dir[d].nt;
    }
    auto blocks_at = [&](long long len) {
        long long t = 0;
        for (int d = 0; d < nd; d++) t += (long long)b * a.dir[d].qblocks * ceil_div64(a.dir[d].nt, len);
        return t;
    };
    long long len = ceil_div64(ceil_div64(work, want_blocks), kChunk) * kChunk;
    if (len < kChunk * 8) len = kChunk * 8;
    if (len < kChunk * 16 && blocks_at(kChunk * 16) >= 2 * kNumCU) len = kChunk * 16;
    if (len > nt_max) len = ceil_div64(nt_max, kChunk) * kChunk;
    a.slice_len = (int)len;
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
        if (D.slices > 1) {
            part += (size_t)D.slices * b * D.nq;
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
        const size_t cnt_bytes = (size_t)1 << 16;
        if ((size_t)units * sizeof(int) > cnt_bytes) {
            set_error("chamfer: too many query blocks for the arrival-counter area");
            return 0;
        }
        char *ws = (char *)workspace(0, cnt_bytes + part * 8, st, nullptr, cnt_bytes);
        if (!ws) return 0;
        a.arrive = (int *)ws;
        unsigned long long *wp = (unsigned long long *)(ws + cnt_bytes);
        size_t off = 0;
        for (int d = 0; d < nd; d++) {
            if (a.dir[d].slices > 1) {
                a.dir[d].part = wp + off;
                off += (size_t)a.dir[d].slices * b * a.dir[d].nq;
            }
        }
    }
    if (r == 4) launch_r<4>(a, (int)tb, st); else launch_r<2>(a, (int)tb, st);
    return check(hipGetLastError(), "nn_forward_kernel launch") ? 1 : 0;
}

}  // namespace genpc

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
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, b, n, xyz1, m,
                       xyz2, graddist1, idx1, graddist2, idx2, gradxyz1, gradxyz2);
    return check(hipGetLastError(), "chamfer_grad_kernel launch") ? 1 : 0;
}
