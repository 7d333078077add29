// nn_grid.hip -- exact nearest neighbour by a cell-sorted, pruned search.
//
// The reference (chamfer3D.cu:12-134) evaluates all N x M pairs.  Its RESULT -- the smallest
// fl((x2-x1)^2 + ...) and the lowest index attaining it -- only depends on the few targets near
// each query, so this family evaluates only those, with the reference's arithmetic, and proves
// the rest away with a bound that accounts for every rounding:
//
//   launch 1  grid_build_kernel    one 1024-thread block per (batch element, cloud): a uniform grid
//                                  over BOTH clouds (box from a strided sample: it only has to be
//                                  reasonable, out-of-box points clamp to the border cells, which
//                                  are treated as unbounded), then a counting sort of the block's
//                                  cloud entirely in LDS (histogram with LDS atomics, scan, scatter)
//                                  to cell order as (x, y, z, original index).  Global atomics were
//                                  measured 5-10x slower here (7 + 21 us at 1 x 16384, 55 + 66 us at
//                                  13 x 16384 for a two-launch histogram / scatter).
//   launch 2  grid_query_kernel    LPQ lanes per query (queries taken in cell order: a wave's
//                                  lanes walk the same few cells, loads coalesce).  Near phase: the
//                                  27 cells around the query's cell, dealt to the LPQ lanes; done
//                                  when everything outside them is provably farther than the best.
//                                  Far phase (a partial scan against the complete shape): cells are
//                                  numbered coarse-major (4 x 4 x 4 blocks), so a coarse cell is one
//                                  contiguous run of the sorted cloud; shells of COARSE cells at
//                                  Chebyshev distance 0, 1, 2, ... are walked, a coarse cell is
//                                  evaluated (its points strided over the LPQ lanes) unless its box
//                                  is provably farther than the best so far, until the next shell
//                                  and everything beyond is provably farther.  The running best is
//                                  the 64-bit key (distance bits << 32 | original index): its
//                                  minimum is the reference's (distance, lowest index), whatever
//                                  order the atomics of launch 1 left inside a cell.
//
// Bound.  cell(p) = clamp(floor(fl(fl(p - lo) * inv)), 0, G - 1) is monotone in p, and a point of
// cell c satisfies  lo + c h (1 - 3u) <= p < lo + (c + 1) h (1 + 3u)  (u = 2^-24, h = 1 / inv;
// border cells unbounded outwards).  With walls evaluated in fp32 and a slack of
// 16u (|lo| + G h + |q|) per axis, gap_a = max(0, wall_lo - s - q, q - wall_hi - s) <= |p_a - q_a|
// for every point p of the cell; the reference's distance is >= (sum gap_a^2)(1 - 6u); a cell is
// skipped iff  (sum gap_a^2)(1 - 2^-20) > best  (strictly: ties with lower indices are still
// found).  Underflow only weakens the bound; an overflowing bound equals +inf and is only used
// against a finite best.  Non-finite input (flags raised by launch 1) takes the per-query
// exhaustive scan with the reference's 512-target tile semantics (see nn_exhaustive in nn.h).
//
// Cost.  O(N) work for clouds of bounded density; a cloud crammed into one cell degenerates to
// the brute force (one lane per query): correct, slow.  Distances to far-away targets (a partial
// scan against the complete shape) cost shells, not the whole cloud.
#include "nn.h"

#include <stdlib.h>
#include <algorithm>

namespace genpc {

constexpr int kGridMaxCells = 16000;     // LDS scan / offsets: < 64 KiB of int per block
constexpr int kGridBlock = 1024;         // build kernel: one block per (batch element, cloud)
constexpr int kGridWaves = kGridBlock / kWave;
constexpr float kU16 = 9.5367431640625e-7f;      // 16 u

struct GridHdr {            // one per batch element, written by launch 1
    float lo[3];
    float inv;              // cells per unit length (cubic cells of side h)
    float h;
    float slack[3];         // 16u (|lo| + (G + 1) h) per axis; the query adds 16u |q|
    int g[3];               // fine cells per axis
    int gc[3];              // coarse cells per axis: (g + 3) / 4
    int cells;              // gc[0] gc[1] gc[2] * 64 (fine cells incl. the padding of partial coarse cells)
    int pad;
};

struct GridArgs {
    const float *cloud[2];  // [B, n, 3]
    int n[2];
    int b;
    int cells_max;          // per cloud and batch element (host: from the cloud sizes)
    int cells_target;
    GridHdr *hdr;           // [B]
    int *bad;               // [B][2]: cloud c of the batch element holds a non-finite coordinate
    int *start;             // [B][2][cells_max + 1]     first sorted position of every fine cell
    int *cstart;            // [B][2][cells_max / 64 + 2] first sorted position of every coarse cell
    int slab_max;           // fine cells per build block (LDS counters)
    float4 *sorted;         // [2][B][n_c]
    size_t sorted_off[2];   // element offset of cloud c in `sorted`
    // queries
    int ndir;
    int qcloud[2];          // direction d: queries = cloud qcloud[d], targets = the other
    float *out_d[2];
    int *out_i[2];
    int qblocks[2];         // blocks per batch element of direction d
    int block_begin[2];
    int fma;
    float radius2;          // search limit (squared distance), +inf = none
    int near_only;          // experiment: stop after the 27 cells, count the unresolved queries in stats[1]
    int sort_cloud[2];      // cloud c is sorted (it is a target cloud, or its queries are wanted in cell order)
    unsigned long long *stats;
};

__device__ __forceinline__ int grid_cell1(float p, float lo, float inv, int g)
{
    const float t = __fmul_rn(__fsub_rn(p, lo), inv);
    int c = (int)floorf(t);          // NaN -> 0 on gfx950 (v_cvt_i32_f32); such clouds take the exhaustive path anyway
    c = c < 0 ? 0 : c;
    return c > g - 1 ? g - 1 : c;
}

// coarse-major numbering: 4 x 4 x 4 fine cells of a coarse cell are consecutive
__device__ __forceinline__ int grid_index(const GridHdr &H, int cx, int cy, int cz)
{
    const int C = ((cz >> 2) * H.gc[1] + (cy >> 2)) * H.gc[0] + (cx >> 2);
    return (C << 6) | ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3);
}

__device__ __forceinline__ int grid_cell(const GridHdr &H, float x, float y, float z, int &cx, int &cy, int &cz)
{
    cx = grid_cell1(x, H.lo[0], H.inv, H.g[0]);
    cy = grid_cell1(y, H.lo[1], H.inv, H.g[1]);
    cz = grid_cell1(z, H.lo[2], H.inv, H.g[2]);
    return grid_index(H, cx, cy, cz);
}

// Box of a strided sample of both clouds (1024 points each) -> grid of about cells_target cubic
// cells.  Both blocks of a batch element compute the same header (min / max are order independent).
__device__ void grid_setup(const GridArgs &a, int batch, GridHdr &H, float *s_red)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    float v[2][3];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int n = a.n[c];
        const float *P = a.cloud[c] + (size_t)batch * n * 3;
        const int stride = n / kGridBlock > 0 ? n / kGridBlock : 1;
        int j = threadIdx.x * stride;
        j = j < n ? j : n - 1;
#pragma unroll
        for (int k = 0; k < 3; k++) v[c][k] = P[(size_t)j * 3 + k];
    }
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float w = v[c][k];
            if (fabsf(w) < __builtin_inff()) {      // non-finite samples do not shape the grid
                mn[k] = fminf(mn[k], w);
                mx[k] = fmaxf(mx[k], w);
            }
        }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], o));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { s_red[wave * 6 + k] = mn[k]; s_red[wave * 6 + 3 + k] = mx[k]; }
    }
    __syncthreads();
    for (int w = 0; w < kGridWaves; w++) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(mn[k], s_red[w * 6 + k]);
            mx[k] = fmaxf(mx[k], s_red[w * 6 + 3 + k]);
        }
    }
    float ext[3];
    bool act[3];
    int nact = 0;
    float emax = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (!(mn[k] <= mx[k])) { mn[k] = 0.0f; mx[k] = 0.0f; }      // no finite sample on this axis
        ext[k] = mx[k] - mn[k];
        if (!(ext[k] < __builtin_inff())) ext[k] = 0.0f;             // overflowing extent: one cell on this axis
        act[k] = ext[k] > 0.0f;
        nact += act[k] ? 1 : 0;
        emax = fmaxf(emax, ext[k]);
    }
    // cubic cells of side h with about cells_target cells over the axes that are wider than h
    // (extents relative to the largest one: no overflow for clouds 1e-18 or 1e+18 across)
    float h = 0.0f;
    for (int it = 0; it < 3 && nact > 0; it++) {
        float vol = 1.0f;
        for (int k = 0; k < 3; k++) if (act[k]) vol *= ext[k] / emax;
        const float r = vol / (float)a.cells_target;
        h = emax * (nact == 3 ? cbrtf(r) : (nact == 2 ? sqrtf(r) : r));
        bool dropped = false;
        for (int k = 0; k < 3; k++) {
            if (act[k] && !(ext[k] > h)) { act[k] = false; nact--; dropped = true; }
        }
        if (!dropped) break;
    }
    if (!(h > 0.0f) || !(h < __builtin_inff()) || nact == 0) {
        h = 1.0f;
        for (int k = 0; k < 3; k++) act[k] = false;
    }
    int g[3], gc[3];
    for (int rep = 0; rep < 12; rep++) {
        long long cells = 64;
        for (int k = 0; k < 3; k++) {
            float q = act[k] ? ceilf(ext[k] / h) : 1.0f;
            if (!(q >= 1.0f)) q = 1.0f;
            if (q > 1024.0f) q = 1024.0f;
            g[k] = (int)q;
            gc[k] = (g[k] + 3) >> 2;
            cells *= gc[k];
        }
        if (cells <= a.cells_max) break;
        h *= 1.26f;
        if (rep == 11) { act[0] = act[1] = act[2] = false; }
    }
    float inv = 1.0f / h;
    if (!(inv > 0.0f) || !(inv < __builtin_inff())) {
        inv = 1.0f; h = 1.0f;
        for (int k = 0; k < 3; k++) { g[k] = 1; gc[k] = 1; }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (!act[k]) { g[k] = 1; gc[k] = 1; }
        H.lo[k] = mn[k];
        H.g[k] = g[k];
        H.gc[k] = gc[k];
        H.slack[k] = kU16 * (fabsf(mn[k]) + (float)(g[k] + 1) * h);
    }
    H.inv = inv;
    H.h = h;
    H.cells = gc[0] * gc[1] * gc[2] * 64;
    H.pad = 0;
}

// launch 1: block (batch, cloud, slab k of K).  A slab is a range of coarse-cell ROWS (Cz, Cy): the
// cell numbering is coarse-major with Cz, Cy most significant, so a slab is a contiguous range of
// the cell index space and of the sorted output.  Every block reads ALL points of its cloud
// (coalesced, from L2) but classifies them by two coordinates only (12 VALU ops per point); the
// points of its slab go through the histogram (LDS atomics), scan and scatter.  The slab's first
// output position is the number of points in lower rows, which the block counts itself while
// reading: no communication between blocks, no global atomics.
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));

__global__ __launch_bounds__(kGridBlock) void grid_build_kernel(GridArgs a, int K)
{
    extern __shared__ int s_cnt[];            // the slab's counters, then kGridWaves wave totals (x2), then the box reduction
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    const int k = bid % K;
    bid /= K;
    const int c = bid & 1, batch = bid >> 1;
    const int n = a.n[c];
    const float *__restrict__ P = a.cloud[c] + (size_t)batch * n * 3;      // sizeof(f3u) is 16: index in floats
    int *s_w = s_cnt + a.slab_max;
    float *s_red = (float *)(s_w + 2 * kGridWaves);
    if (!a.sort_cloud[c]) {
        // queries taken in their original order: nothing to sort; its non-finite flag is per query
        if (k == 0 && threadIdx.x == 0) a.bad[batch * 2 + c] = 0;
        return;
    }
    GridHdr H;
    grid_setup(a, batch, H, s_red);
    if (c == (a.sort_cloud[0] ? 0 : 1) && k == 0 && threadIdx.x == 0) a.hdr[batch] = H;
    const int cells = H.cells, ncoarse = cells >> 6;
    const int rows = H.gc[1] * H.gc[2], rowlen = H.gc[0] << 6;      // fine cells per coarse row
    const int R0 = (int)(((long long)k * rows) / K), R1 = (int)(((long long)(k + 1) * rows) / K);
    const int lo = R0 * rowlen, width = (R1 - R0) * rowlen;
    for (int i = threadIdx.x; i < width; i += kGridBlock) s_cnt[i] = 0;
    __syncthreads();
    // pass 1: histogram of the slab, count of the points below it; four points per thread in flight
    int bad = 0, below = 0;
    for (int j0 = threadIdx.x; j0 < n; j0 += 4 * kGridBlock) {
        f3u px[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int j = j0 + i * kGridBlock;
            j = j < n ? j : n - 1;
            px[i] = *(const f3u *)(P + (size_t)j * 3);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (j0 + i * kGridBlock < n) {
                bad |= !((fabsf(px[i].x) + fabsf(px[i].y)) + fabsf(px[i].z) < __builtin_inff());
                const int cy = grid_cell1(px[i].y, H.lo[1], H.inv, H.g[1]), cz = grid_cell1(px[i].z, H.lo[2], H.inv, H.g[2]);
                const int row = (cz >> 2) * H.gc[1] + (cy >> 2);
                below += row < R0 ? 1 : 0;
                if (row >= R0 && row < R1) {
                    const int cx = grid_cell1(px[i].x, H.lo[0], H.inv, H.g[0]);
                    atomicAdd(&s_cnt[grid_index(H, cx, cy, cz) - lo], 1);
                }
            }
        }
    }
    bad = __syncthreads_or(bad);
    if (k == 0 && threadIdx.x == 0) a.bad[batch * 2 + c] = bad;
    // exclusive scan of the slab: thread t owns the segment [t per, (t + 1) per); per is odd (LDS banks)
    const int per = ((width + kGridBlock - 1) / kGridBlock) | 1;
    int sum = 0;
    for (int i = 0; i < per; i++) {
        const int q = threadIdx.x * per + i;
        if (q < width) { const int w = s_cnt[q]; s_cnt[q] = sum; sum += w; }
    }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) below += __shfl_xor(below, o);
    if (lane == kWave - 1) s_w[wave] = inc;
    if (lane == 0) s_w[kGridWaves + wave] = below;
    __syncthreads();
    int base = inc - sum;
    for (int w = 0; w < kGridWaves; w++) {
        base += w < wave ? s_w[w] : 0;
        base += s_w[kGridWaves + w];
    }
    for (int i = 0; i < per; i++) {
        const int q = threadIdx.x * per + i;
        if (q < width) s_cnt[q] += base;
    }
    __syncthreads();
    int *start = a.start + ((size_t)batch * 2 + c) * (a.cells_max + 1);
    int *cstart = a.cstart + ((size_t)batch * 2 + c) * (a.cells_max / 64 + 2);
    for (int i = threadIdx.x; i < width; i += kGridBlock) {
        start[lo + i] = s_cnt[i];
        if ((i & 63) == 0) cstart[(lo + i) >> 6] = s_cnt[i];
    }
    if (k == K - 1 && threadIdx.x == 0) { start[cells] = n; cstart[ncoarse] = n; }
    __syncthreads();
    // pass 2: scatter (the offsets in LDS become the cursors)
    float4 *out = a.sorted + a.sorted_off[c] + (size_t)batch * n;
    for (int j0 = threadIdx.x; j0 < n; j0 += 4 * kGridBlock) {
        f3u px[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int j = j0 + i * kGridBlock;
            j = j < n ? j : n - 1;
            px[i] = *(const f3u *)(P + (size_t)j * 3);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = j0 + i * kGridBlock;
            if (j < n) {
                const int cy = grid_cell1(px[i].y, H.lo[1], H.inv, H.g[1]), cz = grid_cell1(px[i].z, H.lo[2], H.inv, H.g[2]);
                const int row = (cz >> 2) * H.gc[1] + (cy >> 2);
                if (row >= R0 && row < R1) {
                    const int cx = grid_cell1(px[i].x, H.lo[0], H.inv, H.g[0]);
                    const int pos = atomicAdd(&s_cnt[grid_index(H, cx, cy, cz) - lo], 1);
                    out[pos] = make_float4(px[i].x, px[i].y, px[i].z, __int_as_float(j));
                }
            }
        }
    }
}

__device__ __forceinline__ unsigned long long key_min(unsigned long long a, unsigned long long b) { return b < a ? b : a; }

// One query against every target, the reference's scan order and tile semantics (non-finite input).
template <int FMA>
__device__ unsigned long long grid_exhaustive_lane(const float *__restrict__ T, int nt, float x, float y, float z)
{
    float res = 0.0f;
    int res_i = 0;
    for (int k2 = 0; k2 < nt; k2 += kRefTile) {
        const int end = min(nt, k2 + kRefTile);
        float best = 0.0f;
        int best_i = 0;
        for (int k = k2; k < end; k++) {
            const float *tp = T + (size_t)k * 3;
            const float d = sqdist<FMA>(tp[0] - x, tp[1] - y, tp[2] - z);
            if (k == k2 || d < best) { best = d; best_i = k; }
        }
        if (k2 == 0 || res > best) { res = best; res_i = best_i; }
    }
    return ((unsigned long long)__float_as_uint(res) << 32) | (unsigned)res_i;
}

template <int LPQ>
__device__ __forceinline__ unsigned long long group_min(unsigned long long best)
{
#pragma unroll
    for (int o = 1; o < LPQ; o <<= 1) {
        const unsigned lo32 = __shfl_xor((unsigned)best, o), hi32 = __shfl_xor((unsigned)(best >> 32), o);
        best = key_min(best, ((unsigned long long)hi32 << 32) | lo32);
    }
    return best;
}

// exact evaluation of sorted targets [p0, p1) taken with stride `step` from p0 + first: four loads in
// flight (positions past the end re-read the last one: the key minimum is idempotent)
template <int FMA>
__device__ __forceinline__ void grid_eval(const float4 *__restrict__ ST, int p0, int p1, int first, int step, float x, float y,
                                          float z, unsigned long long &best)
{
    for (int p = p0 + first; p < p1; p += 4 * step) {
        float4 t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int pp = p + i * step;
            t[i] = ST[pp < p1 ? pp : p];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float dd = sqdist<FMA>(t[i].x - x, t[i].y - y, t[i].z - z);
            best = key_min(best, ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(t[i].w));
        }
    }
}

// launch 3.  LPQ lanes share a query.
template <int FMA, int LPQ>
__global__ __launch_bounds__(kBlock) void grid_query_kernel(GridArgs a)
{
    int bid = blockIdx.x;
    const int d = (a.ndir > 1 && bid >= a.block_begin[1]) ? 1 : 0;
    bid -= a.block_begin[d];
    const int batch = bid / a.qblocks[d], qb = bid % a.qblocks[d];
    const int qc = a.qcloud[d], tc = qc ^ 1;
    const int nq = a.n[qc], nt = a.n[tc];
    const int qi = (qb * kBlock + threadIdx.x) / LPQ;
    const int sub = threadIdx.x & (LPQ - 1);
    const bool live = qi < nq;
    const float4 *SQ = a.sorted + a.sorted_off[qc] + (size_t)batch * nq;
    const float4 *__restrict__ ST = a.sorted + a.sorted_off[tc] + (size_t)batch * nt;
    float4 qv;
    if (a.sort_cloud[qc]) {
        qv = SQ[live ? qi : nq - 1];
    } else {
        const int j = live ? qi : nq - 1;
        const float *qp = a.cloud[qc] + ((size_t)batch * nq + j) * 3;
        qv = make_float4(qp[0], qp[1], qp[2], __int_as_float(j));
    }
    const GridHdr H = a.hdr[batch];
    // offsets of the target cloud's cells: read where needed (a block touches a few neighbouring
    // cells; staging the table in LDS cost 32 KiB of traffic per block)
    const int *__restrict__ s_start = a.start + ((size_t)batch * 2 + tc) * (a.cells_max + 1);
    const int bad = a.bad[batch * 2] | a.bad[batch * 2 + 1] | (int)!((fabsf(qv.x) + fabsf(qv.y)) + fabsf(qv.z) < __builtin_inff());
    const float x = qv.x, y = qv.y, z = qv.z;
    const int orig = __float_as_int(qv.w);
    // (NaN bits, -1) is above every real key; with a search limit the key (limit, -1) stands in for
    // "nothing within the limit": cells farther than it are culled like cells farther than a real find
    unsigned long long best = a.radius2 < __builtin_inff() ? (((unsigned long long)__float_as_uint(a.radius2) << 32) | 0xffffffffull) : ~0ull;
    unsigned long long evals = 0;
    if (bad) {
        if (sub == 0 && live) best = grid_exhaustive_lane<FMA>(a.cloud[tc] + (size_t)batch * nt * 3, nt, x, y, z);
        best = group_min<LPQ>(best);
    } else {
        int cq[3];
        grid_cell(H, x, y, z, cq[0], cq[1], cq[2]);
        const int gx = H.g[0], gy = H.g[1], gz = H.g[2];
        const int cells = H.cells;
        const float h = H.h;
        const float sx = H.slack[0] + kU16 * fabsf(x);
        const float sy = H.slack[1] + kU16 * fabsf(y);
        const float sz = H.slack[2] + kU16 * fabsf(z);
        const float smax = fmaxf(sx, fmaxf(sy, sz));
        const float kShrink = 0.99999905f;      // 1 - 2^-20
        const float inf = __builtin_inff();
        // lower bound of |p_a - q_a| over the points p of cells [c, c + w) of axis a (g cells in all)
        auto gap1 = [&](int c, int w, int g, float lo, float q, float s) {
            const float wl = c > 0 ? __fadd_rn(lo, __fmul_rn((float)c, h)) : -inf;
            const float wh = c + w < g ? __fadd_rn(lo, __fmul_rn((float)(c + w), h)) : inf;
            return fmaxf(0.0f, fmaxf((wl - s) - q, (q - s) - wh));
        };
        // near phase: the 3 x 3 x 3 fine cells around the query's cell
        for (int i = sub; i < 27; i += LPQ) {
            const int dz = i / 9, dy = (i / 3) % 3, dx = i % 3;
            const int cx = cq[0] + dx - 1, cy = cq[1] + dy - 1, cz = cq[2] + dz - 1;
            if (cx < 0 || cx >= gx || cy < 0 || cy >= gy || cz < 0 || cz >= gz) continue;
            const int cell = grid_index(H, cx, cy, cz);
            const int p0 = s_start[cell], p1 = s_start[cell + 1];
            if (p0 == p1) continue;
            if (LPQ < 27) {
                const float gxv = gap1(cx, 1, gx, H.lo[0], x, sx), gyv = gap1(cy, 1, gy, H.lo[1], y, sy);
                const float gzv = gap1(cz, 1, gz, H.lo[2], z, sz);
                const float lb = __fmaf_rn(gxv, gxv, __fmaf_rn(gyv, gyv, __fmul_rn(gzv, gzv))) * kShrink;
                if (lb > __uint_as_float((unsigned)(best >> 32))) continue;      // false while nothing is found (NaN key)
            }
            grid_eval<FMA>(ST, p0, p1, 0, 1, x, y, z, best);
            evals += (unsigned)(p1 - p0);
        }
        best = group_min<LPQ>(best);
        // everything outside the 27 cells differs by >= 2 cells on some axis: distance >= h - 2 slack
        float reach = fmaxf(0.0f, h - 2.0f * smax);
        bool done = __fmul_rn(__fmul_rn(reach, reach), kShrink) > __uint_as_float((unsigned)(best >> 32));
        if (!done && a.near_only) {
            if (a.stats && sub == 0 && live) atomicAdd(&a.stats[1], 1ull);
        } else if (!done) {
            // Far phase.  Coarse cells (4 x 4 x 4 fine cells, one contiguous run of the sorted cloud
            // each) are tested against the best so far, the lanes of the group taking LPQ of them at
            // a time; inside a surviving coarse cell the 64 fine cells are tested the same way and
            // the survivors evaluated.  The bound comes from a spread of sampled targets first, then
            // from the coarse cell nearest to the query, before the sweep proper.
            const int Gx = H.gc[0], Gy = H.gc[1];
            const int nC = cells >> 6;
            const int *__restrict__ cst = a.cstart + ((size_t)batch * 2 + tc) * (a.cells_max / 64 + 2);
            auto coarse_lb = [&](int C) {
                const int Cx = C % Gx, Cy = (C / Gx) % Gy, Cz = C / (Gx * Gy);
                const float gxv = gap1(Cx * 4, 4, gx, H.lo[0], x, sx), gyv = gap1(Cy * 4, 4, gy, H.lo[1], y, sy);
                const float gzv = gap1(Cz * 4, 4, gz, H.lo[2], z, sz);
                return __fmaf_rn(gxv, gxv, __fmaf_rn(gyv, gyv, __fmul_rn(gzv, gzv))) * kShrink;
            };
            // the fine cells of coarse cell C against the bound, survivors evaluated
            auto coarse_eval = [&](int C) {
                const int Cx = C % Gx, Cy = (C / Gx) % Gy, Cz = C / (Gx * Gy);
                float g2[3][4];
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    const float vx = gap1(Cx * 4 + f, 1, gx, H.lo[0], x, sx), vy = gap1(Cy * 4 + f, 1, gy, H.lo[1], y, sy);
                    const float vz = gap1(Cz * 4 + f, 1, gz, H.lo[2], z, sz);
                    g2[0][f] = __fmul_rn(vx, vx); g2[1][f] = __fmul_rn(vy, vy); g2[2][f] = __fmul_rn(vz, vz);
                }
                const float bd = __uint_as_float((unsigned)(best >> 32));
                for (int f = sub; f < 64; f += LPQ) {
                    const int p0 = s_start[(C << 6) + f], p1 = s_start[(C << 6) + f + 1];
                    if (p0 == p1) continue;
                    // select without dynamic register indexing
                    const int fx = f & 3, fy = (f >> 2) & 3, fz = f >> 4;
                    const float ax = fx == 0 ? g2[0][0] : (fx == 1 ? g2[0][1] : (fx == 2 ? g2[0][2] : g2[0][3]));
                    const float ay = fy == 0 ? g2[1][0] : (fy == 1 ? g2[1][1] : (fy == 2 ? g2[1][2] : g2[1][3]));
                    const float az = fz == 0 ? g2[2][0] : (fz == 1 ? g2[2][1] : (fz == 2 ? g2[2][2] : g2[2][3]));
                    const float lb = __fadd_rn(__fadd_rn(ax, ay), az) * kShrink;
                    if (lb > bd) continue;
                    grid_eval<FMA>(ST, p0, p1, 0, 1, x, y, z, best);
                    evals += (unsigned)(p1 - p0);
                }
                best = group_min<LPQ>(best);
            };
            {
                const int ns = LPQ >= 8 ? LPQ : 8;
                for (int i = sub; i < ns; i += LPQ) {
                    const int p = (int)(((long long)(2 * i + 1) * nt) / (2 * ns));
                    const float4 t = ST[p];
                    const float dd = sqdist<FMA>(t.x - x, t.y - y, t.z - z);
                    best = key_min(best, ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(t.w));
                }
                best = group_min<LPQ>(best);
            }
            // nearest non-empty coarse cell (smallest bound; ties: lowest number)
            {
                unsigned long long near = ~0ull;
                for (int C = sub; C < nC; C += LPQ) {
                    if (cst[C + 1] > cst[C]) near = key_min(near, ((unsigned long long)__float_as_uint(coarse_lb(C)) << 32) | (unsigned)C);
                }
                near = group_min<LPQ>(near);
                if (near != ~0ull) coarse_eval((int)(unsigned)near);
            }
            const int glane = (threadIdx.x & (kWave - 1)) & ~(LPQ - 1);      // first lane of the group in the wave
            for (int base = 0; base < nC; base += LPQ) {
                const int C = base + sub;
                bool pass = false;
                if (C < nC && cst[C + 1] > cst[C]) pass = !(coarse_lb(C) > __uint_as_float((unsigned)(best >> 32)));
                unsigned long long m = __ballot(pass);
                m = LPQ == 64 ? m : ((m >> glane) & ((1ull << LPQ) - 1ull));
                while (m) {
                    const int kbit = __builtin_ctzll(m);
                    m &= m - 1;
                    // the bound may have tightened since the ballot
                    if (!(coarse_lb(base + kbit) > __uint_as_float((unsigned)(best >> 32)))) coarse_eval(base + kbit);
                }
            }
        }
    }
    if (live && sub == 0) {
        float *od = a.out_d[d] + (size_t)batch * nq;
        int *oi = a.out_i[d] + (size_t)batch * nq;
        const bool none = (unsigned)best == 0xffffffffu;      // only possible with a search limit
        od[orig] = none ? __builtin_inff() : __uint_as_float((unsigned)(best >> 32));
        oi[orig] = (int)(unsigned)best;
    }
    if (a.stats) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) evals += __shfl_xor(evals, o);
        if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(&a.stats[2], evals);
        if (threadIdx.x == 0) atomicAdd(&a.stats[0], (unsigned long long)min(kBlock / LPQ, max(0, nq - qb * (kBlock / LPQ))));
    }
}

// Host side.  Returns 1 ok, 0 error.  Directions as in NNArgs (dir[0]: queries cloud 0 / targets
// cloud 1 when called from genpc_chamfer_forward).
int launch_nn_grid(const NNArgs &na, hipStream_t st)
{
    GridArgs a{};
    a.b = na.b;
    a.fma = na.fma;
    a.stats = na.stats;
    a.radius2 = na.radius2;
    static const int near_env = (tune_env("GENPC_GRID_NEARONLY", 0, "cell-sorted nearest-neighbour search: experiment, stop after the 27 near cells") != 0 ? 1 : 0);
    a.near_only = near_env;
    // one direction: only the target cloud is sorted (the queries lose some locality, a whole cloud less to sort)
    a.sort_cloud[0] = na.ndir > 1 ? 1 : 0;
    a.sort_cloud[1] = 1;
    // clouds: direction 0 queries = cloud 0, targets = cloud 1; a second direction must be the swap
    a.cloud[0] = na.dir[0].q;
    a.n[0] = na.dir[0].nq;
    a.cloud[1] = na.dir[0].t;
    a.n[1] = na.dir[0].nt;
    a.ndir = na.ndir;
    a.qcloud[0] = 0;
    a.out_d[0] = na.dir[0].out_d;
    a.out_i[0] = na.dir[0].out_i;
    if (na.ndir > 1) {
        if (na.dir[1].q != na.dir[0].t || na.dir[1].t != na.dir[0].q) {
            set_error("nn grid: second direction must swap the clouds of the first");
            return 0;
        }
        a.qcloud[1] = 1;
        a.out_d[1] = na.dir[1].out_d;
        a.out_i[1] = na.dir[1].out_i;
    }
    const int nmax = std::max(a.sort_cloud[0] ? a.n[0] : 0, a.n[1]);      // the grid follows the density of the sorted (target) clouds
    // about three points per cell of the denser cloud if it filled the box; surfaces fill ~ G^2 of G^3 cells
    static const int env_ppc = tune_env("GENPC_GRID_PPC_X10", 30, "cell-sorted nearest-neighbour search: target points per cell x 10");      // points per cell x 10
    int target = (int)((long long)nmax * 10 / (env_ppc > 0 ? env_ppc : 30));
    target = std::max(8, std::min(target, kGridMaxCells / 2));
    a.cells_target = target;
    a.cells_max = std::min(kGridMaxCells - 256, 2 * target + 512);      // coarse cells are padded to 4 x 4 x 4
    const int stride = a.cells_max + 1;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t hdr_bytes = al((size_t)a.b * sizeof(GridHdr));
    const size_t bad_bytes = al((size_t)a.b * 2 * sizeof(int));
    const size_t start_bytes = al((size_t)a.b * 2 * stride * sizeof(int));
    const size_t cstart_bytes = al((size_t)a.b * 2 * (a.cells_max / 64 + 2) * sizeof(int));
    const size_t sorted_bytes = al((size_t)a.b * ((size_t)a.n[0] + a.n[1]) * sizeof(float4));
    char *ws = (char *)workspace(13, hdr_bytes + bad_bytes + start_bytes + cstart_bytes + sorted_bytes, st);
    if (!ws) return 0;
    a.hdr = (GridHdr *)ws;
    a.bad = (int *)(ws + hdr_bytes);
    a.start = (int *)(ws + hdr_bytes + bad_bytes);
    a.cstart = (int *)(ws + hdr_bytes + bad_bytes + start_bytes);
    a.sorted = (float4 *)(ws + hdr_bytes + bad_bytes + start_bytes + cstart_bytes);
    a.sorted_off[0] = 0;
    a.sorted_off[1] = (size_t)a.b * a.n[0];
    // slabs per cloud: enough blocks to occupy a good part of the chip when the batch is small
    static const int env_k = tune_env("GENPC_GRID_K", 0, "cell-sorted nearest-neighbour search: slab blocks per cloud of the build (0 = pick)");
    int K = std::max(1, std::min(8, 64 / (2 * a.b)));
    K = std::min(K, std::max(1, nmax / 2048));
    if (env_k > 0) K = std::min(env_k, 64);
    if ((long long)a.b * 2 * K > 0x7fffffffLL) {
        set_error("nn grid: problem too large for one launch");
        return 0;
    }
    a.slab_max = a.cells_max;      // a slab is whole coarse rows: a line-like cloud has a single row
    const size_t lds_build = ((size_t)a.slab_max + 2 * kGridWaves + kGridWaves * 6 + 8) * sizeof(int);
    hipLaunchKernelGGL(grid_build_kernel, dim3((unsigned)(a.b * 2 * K)), dim3(kGridBlock), lds_build, st, a, K);
    // lanes per query: as many as it takes to put ~8 waves on every SIMD (latency, not arithmetic,
    // bounds a small launch), one when there are that many queries anyway
    long long queries = 0;
    for (int d = 0; d < a.ndir; d++) queries += (long long)a.b * a.n[a.qcloud[d]];
    const long long want = 8LL * (4 * num_cus()) * kWave;
    static const int env_lpq = tune_env("GENPC_GRID_LPQ", 0, "cell-sorted nearest-neighbour search: lanes per query (1 | 4 | 16 | 32, 0 = pick)");
    int lpq = queries * 16 <= want ? 32 : (queries * 4 <= want ? 16 : (queries <= want ? 4 : 1));
    if (env_lpq == 1 || env_lpq == 4 || env_lpq == 16 || env_lpq == 32) lpq = env_lpq;
    long long tb = 0;
    for (int d = 0; d < a.ndir; d++) {
        a.qblocks[d] = ceil_div(a.n[a.qcloud[d]] * lpq, kBlock);
        a.block_begin[d] = (int)tb;
        tb += (long long)a.b * a.qblocks[d];
    }
    if (tb > 0x7fffffffLL) {
        set_error("nn grid: problem too large for one launch");
        return 0;
    }
    const size_t lds_query = 0;
#define GENPC_GRID_LAUNCH(L)                                                                                              \
    do {                                                                                                                  \
        if (a.fma) hipLaunchKernelGGL((grid_query_kernel<1, L>), dim3((unsigned)tb), dim3(kBlock), lds_query, st, a);     \
        else hipLaunchKernelGGL((grid_query_kernel<0, L>), dim3((unsigned)tb), dim3(kBlock), lds_query, st, a);           \
    } while (0)
    if (lpq == 32) GENPC_GRID_LAUNCH(32);
    else if (lpq == 16) GENPC_GRID_LAUNCH(16);
    else if (lpq == 4) GENPC_GRID_LAUNCH(4);
    else GENPC_GRID_LAUNCH(1);
#undef GENPC_GRID_LAUNCH
    return check(hipGetLastError(), "nn grid launch") ? 1 : 0;
}

}  // namespace genpc
