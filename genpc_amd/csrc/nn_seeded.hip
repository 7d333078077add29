// nn_seeded.hip -- the alignment loop's nearest neighbours from the second Adam step on.
//
// diff_obj_pose.py:529-556 evaluates two partial-matching Chamfer terms per step, i.e. a full bidirectional nearest-neighbour
// query between the posed complete cloud and the partial cloud, 4 x 201 times per scan -- and the pose moves by at most
// its learning rate per step.  The brute-force filter (nn_f16.hip) re-solves the query from nothing every time: 61 of the
// ~170 us of a step.  Here every query starts from the index it was answered with at the previous step:
// the reference's distance to THAT target is an upper bound on the answer, and only the targets inside that ball can
// change it.  Both clouds are sorted ONCE per call into uniform grids (emd_grid_build_kernel, cells numbered x-fastest):
//   * the partial cloud does not move: the posed points search its grid directly;
//   * the complete cloud moves rigidly (similarity): its grid is built in the REST frame and a partial point searches it
//     from where the inverse pose puts it, q' = c + R^T (q - c - t) / s.  That frame is used for culling only, with a margin
//     (1e-5 relative + 1e-5 of the scene's size absolute: a hundred times the rounding of pose_point and of q') -- every
//     candidate that survives is valued with the reference's arithmetic on the POSED coordinates of this step.
// Results are the reference's (distance, first index): the running best is the 64-bit key distance bits << 32 | index, a row
// of cells is skipped only if its bound is STRICTLY above the best distance (ties with lower indices are still found), the
// bound itself is nn_grid.hip's (every rounding accounted for).  The loop's histories are bit-identical to the brute-force
// path (tests/test_gpu_pipeline.py, GENPC_POSE_SEEDED=0 for A/B).
#include "nn.h"
#include "emd.h"
#include "../../include/genpc_hip.h"

#include <stdlib.h>

namespace genpc {

constexpr int kSLPQ = 4;            // lanes per query
constexpr float kSU16 = 9.5367431640625e-7f;

struct SeededDir {
    const float *q;            // queries [b][nq][3], world frame
    const float *tpos;         // the targets' coordinates of THIS step, world frame [b][nt][3] (what distances are taken to)
    const float4 *sorted;      // the targets in cell order of their grid's frame: (x, y, z, index)
    const int *start;          // [b][cells_max + 1]
    const EGridHdr *hdr;       // [b]
    const int *seed;           // [b][nq] last step's answers
    float *out_d;
    int *out_i;
    int nq, nt, moving;        // moving: the grid is in the rest frame of the posed cloud
    int block_begin;
};
struct SeededArgs {
    SeededDir d[2];
    int b, cells_max;
    int sample;                // > 1: only every sample-th block runs (a timing probe of the alignment loop)
    const float *center; int cstride;
    const float *params; int pstride;
};

__device__ __forceinline__ int sgrid_cell1(float p, float lo, float inv, int g)
{
    const float t = __fmul_rn(__fsub_rn(p, lo), inv);
    int c = (int)floorf(t);
    c = c < 0 ? 0 : c;
    return c > g - 1 ? g - 1 : c;
}

template <int FMA>
__global__ __launch_bounds__(kBlock) void nn_seeded_kernel(SeededArgs a)
{
    int bid = blockIdx.x * a.sample;
    const int di = bid >= a.d[1].block_begin ? 1 : 0;
    const SeededDir &D = a.d[di];
    bid -= D.block_begin;
    const int qblocks = (D.nq * kSLPQ + kBlock - 1) / kBlock;
    const int batch = bid / qblocks, qb = bid - batch * qblocks;
    const int j = (qb * kBlock + threadIdx.x) / kSLPQ, sub = threadIdx.x & (kSLPQ - 1);
    const bool live = j < D.nq;
    const int jj = live ? j : D.nq - 1;
    const float *qp = D.q + ((size_t)batch * D.nq + jj) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float *__restrict__ TP = D.tpos + (size_t)batch * D.nt * 3;
    const float4 *__restrict__ S = D.sorted + (size_t)batch * D.nt;
    const int *__restrict__ ST = D.start + (size_t)batch * (a.cells_max + 1);
    const EGridHdr H = D.hdr[batch];
    const float inf = __builtin_inff();
    // the query in the grid's frame; what a unit of grid distance is worth in the world, and the culling margins
    float gx_ = qx, gy_ = qy, gz_ = qz, scale = 1.0f, rel = 1.0f, dg = 0.0f;
    bool cull = !H.bad && (fabsf(qx) + fabsf(qy)) + fabsf(qz) < inf;
    if (D.moving) {
        const float *pr = a.params + (size_t)batch * a.pstride, *cc = a.center + (size_t)batch * a.cstride;
        // pytorch3d.transforms.rotation_6d_to_matrix (rows b1, b2, b1 x b2), as the transform kernel computes it
        const float a1x = pr[0], a1y = pr[1], a1z = pr[2], a2x = pr[3], a2y = pr[4], a2z = pr[5];
        float n1 = sqrtf(a1x * a1x + a1y * a1y + a1z * a1z);
        n1 = n1 > 1e-12f ? n1 : 1e-12f;
        const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
        const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
        float b2x = a2x - dt * b1x, b2y = a2y - dt * b1y, b2z = a2z - dt * b1z;
        float n2 = sqrtf(b2x * b2x + b2y * b2y + b2z * b2z);
        n2 = n2 > 1e-12f ? n2 : 1e-12f;
        b2x /= n2; b2y /= n2; b2z /= n2;
        const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
        const float s = expf(pr[9]);
        const float ux = qx - cc[0] - pr[6], uy = qy - cc[1] - pr[7], uz = qz - cc[2] - pr[8];
        // R^T u: the columns of R are (b1x, b2x, b3x), ...
        const float is = 1.0f / s;
        gx_ = cc[0] + (b1x * ux + b2x * uy + b3x * uz) * is;
        gy_ = cc[1] + (b1y * ux + b2y * uy + b3y * uz) * is;
        gz_ = cc[2] + (b1z * ux + b2z * uy + b3z * uz) * is;
        scale = s;
        rel = 1.00001f;
        const float ext = H.h * (float)(H.g[0] + H.g[1] + H.g[2]);
        const float mag = fabsf(qx) + fabsf(qy) + fabsf(qz) + fabsf(cc[0]) + fabsf(cc[1]) + fabsf(cc[2]) + fabsf(pr[6]) + fabsf(pr[7]) +
                          fabsf(pr[8]) + s * ext;
        dg = 1e-5f * mag * is;
        // an orthonormal R is what makes rest-frame distances x s world distances: anything else (a diverged pose) sweeps everything
        const float o12 = b1x * b2x + b1y * b2y + b1z * b2z, l1 = b1x * b1x + b1y * b1y + b1z * b1z, l2 = b2x * b2x + b2y * b2y + b2z * b2z;
        cull = cull && s > 1e-20f && s < 1e20f && fabsf(o12) < 1e-5f && fabsf(l1 - 1.0f) < 1e-5f && fabsf(l2 - 1.0f) < 1e-5f &&
               (fabsf(gx_) + fabsf(gy_)) + fabsf(gz_) < inf;
    }
    // the seed: last step's answer, valued at THIS step's coordinates
    unsigned long long best = ~0ull;
    if (D.seed) {
        const int si = D.seed[(size_t)batch * D.nq + jj];
        if ((unsigned)si < (unsigned)D.nt) {
            const float dd = sqdist<FMA>(TP[(size_t)si * 3 + 0] - qx, TP[(size_t)si * 3 + 1] - qy, TP[(size_t)si * 3 + 2] - qz);
            best = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)si;
        }
    }
    const int gx = H.g[0], gy = H.g[1], gz = H.g[2];
    const float h = H.h;
    const float sx = H.slack[0] + kSU16 * fabsf(gx_), sy = H.slack[1] + kSU16 * fabsf(gy_), sz = H.slack[2] + kSU16 * fabsf(gz_);
    const float kShrink = 0.99999905f;
    auto gap1 = [&](int c, int g, float lo, float q, float s) {
        const float wl = c > 0 ? __fadd_rn(lo, __fmul_rn((float)c, h)) : -inf;
        const float wh = c + 1 < g ? __fadd_rn(lo, __fmul_rn((float)(c + 1), h)) : inf;
        return fmaxf(0.0f, fmaxf((wl - s) - q, (q - s) - wh));
    };
    // squared grid-frame radius inside which a target can still beat (or tie) the best: static grid: the best distance
    // itself (the bound is exact there); moving grid: with the margins.  NaN (no seed yet / non-finite input): nothing is culled
    auto reach2 = [&](unsigned long long k) {
        const float bd = __uint_as_float((unsigned)(k >> 32));
        if (!D.moving) return bd;
        const float r = sqrtf(bd) / scale * rel + dg;
        return r * r * rel;
    };
    float rg2 = cull ? reach2(best) : __builtin_nanf("");
    int bx0 = 0, bx1 = gx - 1, by0 = 0, by1 = gy - 1, bz0 = 0, bz1 = gz - 1;
    if (rg2 == rg2 && rg2 < inf) {
        const float R = sqrtf(rg2) * 1.000001f;
        bx0 = sgrid_cell1((gx_ - R) - sx, H.lo[0], H.inv, gx); bx1 = sgrid_cell1((gx_ + R) + sx, H.lo[0], H.inv, gx);
        by0 = sgrid_cell1((gy_ - R) - sy, H.lo[1], H.inv, gy); by1 = sgrid_cell1((gy_ + R) + sy, H.lo[1], H.inv, gy);
        bz0 = sgrid_cell1((gz_ - R) - sz, H.lo[2], H.inv, gz); bz1 = sgrid_cell1((gz_ + R) + sz, H.lo[2], H.inv, gz);
    }
    // Rows of the box, kSLPQ lanes striding over them four at a time: the extent of a row inside the box's x range is
    // fetched FIRST, for four rows at once -- a query with no target nearby (the back of the complete shape against a partial
    // scan) has a box of a hundred and more rows almost all of which are empty, and a row used to cost ~100 instructions
    // (bounds, a square root, two cell lookups) before its two loads said so: 4.8 ms per step at 32 x (32768 + 32768).
    const int wy = by1 - by0 + 1, nrows = wy * (bz1 - bz0 + 1);
    int iy = sub % wy, iz = sub / wy;                    // row r = iz * wy + iy, advanced incrementally
    for (int r0 = sub; r0 < nrows; r0 += 4 * kSLPQ) {
        int rowb[4], q0[4], q1[4], cyv[4], czv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool in = r0 + k * kSLPQ < nrows;
            cyv[k] = by0 + iy; czv[k] = bz0 + iz;
            rowb[k] = in ? (czv[k] * gy + cyv[k]) * gx : 0;
            q0[k] = in ? ST[rowb[k] + bx0] : 0;
            q1[k] = in ? ST[rowb[k] + bx1 + 1] : 0;
            iy += kSLPQ;
            while (iy >= wy) { iy -= wy; iz++; }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (q1[k] <= q0[k]) continue;                // nothing of this row inside the box (or past the last row)
            int p0 = q0[k], p1 = q1[k];
            if (rg2 == rg2) {
                const float gyv = gap1(cyv[k], gy, H.lo[1], gy_, sy), gzv = gap1(czv[k], gz, H.lo[2], gz_, sz);
                const float lb = __fmaf_rn(gyv, gyv, __fmul_rn(gzv, gzv)) * kShrink;
                if (lb > rg2) continue;                  // strictly farther than the best: not even a tie
                const float W = sqrtf(fmaxf(0.0f, __fmul_rn(rg2, 1.000001f) - lb)) * 1.000001f;
                const int cx0 = max(bx0, sgrid_cell1((gx_ - W) - sx, H.lo[0], H.inv, gx));
                const int cx1 = min(bx1, sgrid_cell1((gx_ + W) + sx, H.lo[0], H.inv, gx));
                if (cx0 > cx1) continue;
                if (cx0 > bx0) p0 = ST[rowb[k] + cx0];
                if (cx1 < bx1) p1 = ST[rowb[k] + cx1 + 1];
            }
            for (int p = p0; p < p1; p += 4) {
                float4 e[4];
#pragma unroll
                for (int i = 0; i < 4; i++) e[i] = S[p + i < p1 ? p + i : p];
                float tx[4], ty[4], tz[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (D.moving) {
                        const int kk = __float_as_int(e[i].w);
                        tx[i] = TP[(size_t)kk * 3 + 0]; ty[i] = TP[(size_t)kk * 3 + 1]; tz[i] = TP[(size_t)kk * 3 + 2];
                    } else {
                        tx[i] = e[i].x; ty[i] = e[i].y; tz[i] = e[i].z;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float dd = sqdist<FMA>(tx[i] - qx, ty[i] - qy, tz[i] - qz);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(e[i].w);
                    best = key < best ? key : best;      // (positions past the run re-read its first entry: idempotent)
                }
                if (rg2 == rg2) rg2 = reach2(best);
            }
        }
    }
#pragma unroll
    for (int o = 1; o < kSLPQ; o <<= 1) {
        const unsigned lo32 = __shfl_xor((unsigned)best, o), hi32 = __shfl_xor((unsigned)(best >> 32), o);
        const unsigned long long ob = ((unsigned long long)hi32 << 32) | lo32;
        best = ob < best ? ob : best;
    }
    if (live && sub == 0) {
        D.out_d[(size_t)batch * D.nq + j] = __uint_as_float((unsigned)(best >> 32));
        D.out_i[(size_t)batch * D.nq + j] = (int)(unsigned)best;
    }
}

// One step's bidirectional query.  dir 0: queries `moving_pts` (the posed cloud, nm points) against the static cloud;
// dir 1: queries the static cloud (ns points) against the posed cloud through its rest-frame grid.
int launch_nn_seeded(int b, int nm, const float *moving_pts, int ns, const float *static_pts, const SeededGrids &g, const float *center,
                     int cstride, const float *params, int pstride, float *d1, int *i1, float *d2, int *i2, int fma, hipStream_t st, int sample)
{
    SeededArgs a{};
    a.b = b; a.cells_max = kEGMaxCells;
    a.sample = sample > 1 ? sample : 1;
    a.center = center; a.cstride = cstride; a.params = params; a.pstride = pstride;
    SeededDir &A = a.d[0], &B = a.d[1];
    A.q = moving_pts; A.tpos = static_pts; A.sorted = g.sorted_static; A.start = g.start_static; A.hdr = g.hdr_static; A.seed = i1;
    A.out_d = d1; A.out_i = i1; A.nq = nm; A.nt = ns; A.moving = 0; A.block_begin = 0;
    B.q = static_pts; B.tpos = moving_pts; B.sorted = g.sorted_rest; B.start = g.start_rest; B.hdr = g.hdr_rest; B.seed = i2;
    B.out_d = d2; B.out_i = i2; B.nq = ns; B.nt = nm; B.moving = 1;
    const long long blocks0 = (long long)b * ceil_div(nm * kSLPQ, kBlock), blocks1 = (long long)b * ceil_div(ns * kSLPQ, kBlock);
    if (blocks0 + blocks1 > 0x7fffffffLL) { set_error("nn seeded: problem too large for one launch"); return 0; }
    B.block_begin = (int)blocks0;
    const unsigned grid = (unsigned)((blocks0 + blocks1 + a.sample - 1) / a.sample);
    if (fma) hipLaunchKernelGGL((nn_seeded_kernel<1>), dim3(grid), dim3(kBlock), 0, st, a);
    else hipLaunchKernelGGL((nn_seeded_kernel<0>), dim3(grid), dim3(kBlock), 0, st, a);
    return check(hipGetLastError(), "nn_seeded_kernel launch") ? 1 : 0;
}

size_t seeded_grids_bytes(int b, int nm, int ns)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return 2 * al((size_t)b * sizeof(EGridHdr)) + 2 * al((size_t)b * (kEGMaxCells + 1) * sizeof(int)) + al((size_t)b * nm * sizeof(float4)) +
           al((size_t)b * ns * sizeof(float4));
}

// Both grids of a call: the static cloud in the world frame, the moving cloud in its rest frame.
int build_seeded_grids(int b, int nm, const float *rest_pts, int ns, const float *static_pts, void *ws, SeededGrids &g, hipStream_t st)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char *p = (char *)ws;
    g.hdr_static = (EGridHdr *)p; p += al((size_t)b * sizeof(EGridHdr));
    g.hdr_rest = (EGridHdr *)p; p += al((size_t)b * sizeof(EGridHdr));
    g.start_static = (int *)p; p += al((size_t)b * (kEGMaxCells + 1) * sizeof(int));
    g.start_rest = (int *)p; p += al((size_t)b * (kEGMaxCells + 1) * sizeof(int));
    g.sorted_static = (float4 *)p; p += al((size_t)b * ns * sizeof(float4));
    g.sorted_rest = (float4 *)p;
    auto target = [](int n) { int t = n / 2; return t < 8 ? 8 : (t > kEGMaxCells * 3 / 4 ? kEGMaxCells * 3 / 4 : t); };
    if (!launch_emd_grid_build(b, ns, static_pts, nullptr, (EGridHdr *)g.hdr_static, (int *)g.start_static, (float4 *)g.sorted_static, nullptr,
                               nullptr, target(ns), kEGMaxCells, st))
        return 0;
    return launch_emd_grid_build(b, nm, rest_pts, nullptr, (EGridHdr *)g.hdr_rest, (int *)g.start_rest, (float4 *)g.sorted_rest, nullptr, nullptr,
                                 target(nm), kEGMaxCells, st);
}

}  // namespace genpc
