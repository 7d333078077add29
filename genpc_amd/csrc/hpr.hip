// hpr.hip -- exact hidden-point removal: the operator the reference applies per
// viewpoint through open3d's PointCloud.hidden_point_removal(camera, radius) (DepthPrompting.py:273-290;
// Katz, Tal, Basri 2007): spherical flipping p' = v + 2 (radius - |v|) v / |v| with v = p - camera, then
// the convex hull of the flipped points and the origin; visible = the hull's vertices.
//
// No hull is built here.  A flipped point p'_i is a hull vertex iff some plane through it has every other
// flipped point and the origin strictly on one side, i.e. iff there is a normal n with n.p'_i > 0 and
// n.p'_j < n.p'_i for all j.  n.p'_i > 0 lets n be scaled to n = u + a e1 + b e2 (u = p'_i / |p'_i|;
// e1, e2 an orthonormal basis of u's normal plane); then n.p'_i = |p'_i| and every other point is ONE
// LINEAR constraint on (a, b):
//       a (e1.p'_j) + b (e2.p'_j) <= |p'_i| - u.p'_j
// The feasible (a, b) are a convex polygon -- the cross-section of the vertex's normal cone; for the
// reference's large radii it is the power cell of point i among the cloud's directions, weighted by depth --
// and the point is visible iff the polygon survives clipping by every other point.
//
// Pipeline (one call = all viewpoints):
//   hpr_bounds / hpr_key / rocprim sort  per view, the points in 2-D Morton order of their DIRECTION from the
//                         eye: a surface and what it hides become neighbours, so a hidden point meets its
//                         occluders at once.  The order never changes a result, only when it is reached.
//   hpr_flip_kernel       p' per (view, position), fp64
//   hpr_tile_kernel       per tile of 128 consecutive positions: axis, half-angle, largest |p'| of its directions
//   hpr_accept_kernel     u itself separates (u.p'_j < |p'_i| for all j: the origin of the (a, b) plane strictly
//                         feasible): visible, no polygon -- one dot product per candidate
//   hpr_compact_kernel    the left-over positions per view, in order
//   hpr_kernel            one thread per left-over (view, point), polygon in LDS ([vertex][thread], <= kHprMaxV):
//                         (1) the point's home tile, the next, the previous; (2) one interior point of the
//                         polygon tried as THE normal against everything (strictly feasible: visible);
//                         (3) a few batches of the other tiles outward from the group's tile (tiles whose cone
//                         cannot reach the polygon skipped whole, candidates filtered by two reach bounds, then
//                         clipped); whatever is undecided then -- and, for clouds under 256 tiles, everything
//                         undecided after (2), with its polygon saved -- goes to
//   hpr_overflow_kernel   one WAVE per point (also: polygons that outgrow kHprMaxV): 64 candidates tested against
//                         the polygon at a time, clips by all lanes; 128-vertex polygons first, larger ones in a
//                         second launch
// Every skip and every early decision is conservative (margins 1e-7 .. 1e-10 against fp64 roundoff of 1e-16),
// so the mask is that of clipping every polygon by every point.
//
// All arithmetic is double with contraction OFF and the operation order of oracle/genpc_oracle_hpr.c, which
// restates the ordering, the accept test, the grouping and the candidate order: the two produce the same
// polygons bit for bit, hence the same mask.  That restatement is pinned against qhull (scipy) on random clouds,
// real scans and lattices.  Deviations from the hull definition: normals tilted from u by more than atan(1e4)
// are not considered; of exact duplicates only the lowest-index copy takes part (qhull keeps one copy too).
//
// Measured numbers and the steps that led here: DESIGN.md section 4.6.
#include "common.h"
#include "../../include/genpc_hip.h"

#include <limits.h>
#include <stdlib.h>
#include <algorithm>

#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#pragma clang fp contract(off)

namespace genpc {

constexpr int kHprThreads = 128;
constexpr int kHprMaxV = 10;            // polygon vertices per thread in LDS (20 KB per 128-thread block).  Round 2: 24 -> 16 (24.7 -> 20.9 ms at 64 x 10000; 12 then lost: too many points fell to a second pass that held five waves per CU).  With the second pass's 4 KiB tier the balance moved: 1024 x 10000 blob / scan 83.7 / 51.4 ms at 16, 68.5 / 43.1 at 12, 66.2 / 41.9 at 10, 75.4 / 43.3 at 9, 99 / 52 at 8
constexpr int kHprOverCap = 1024;       // vertices per polygon in the second pass (2 x 16 KB of LDS per wave)
constexpr double kHprBox = 1.0e4;

// p' for every (view, point), stored in Morton order (row pos = point perm[pos]); open3d: |v| = 0 -> 1e-4
__global__ __launch_bounds__(256) void hpr_flip_kernel(int n, const float *__restrict__ pts, const int *__restrict__ perm,
                                                      const double *__restrict__ eyes, double radius, double *__restrict__ fl,
                                                      const unsigned char *__restrict__ dup)
{
    const int pos = blockIdx.x * 256 + threadIdx.x, view = blockIdx.y;
    if (pos >= n) return;
    const int i = perm[(size_t)view * n + pos];
    if (dup[i]) {
        // a later copy of an exact duplicate: NaN -- hidden, and it cuts nothing (its lowest-index twin stands for both)
        double *o = fl + ((size_t)view * n + pos) * 3;
        o[0] = o[1] = o[2] = __builtin_nan("");
        return;
    }
    const double vx = (double)pts[(size_t)i * 3 + 0] - eyes[view * 3 + 0];
    const double vy = (double)pts[(size_t)i * 3 + 1] - eyes[view * 3 + 1];
    const double vz = (double)pts[(size_t)i * 3 + 2] - eyes[view * 3 + 2];
    double r = sqrt(vx * vx + vy * vy + vz * vz);
    if (r == 0.0) r = 0.0001;
    const double k = 2.0 * (radius - r) / r;
    double *o = fl + ((size_t)view * n + pos) * 3;
    o[0] = vx + k * vx;
    o[1] = vy + k * vy;
    o[2] = vz + k * vz;
}

__device__ __forceinline__ unsigned hpr_ord(float f)      // order-preserving float -> uint
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float hpr_unord(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Exact duplicates.  qhull (open3d's hidden_point_removal) reports ONE copy of a group of coincident points as a hull
// vertex -- which one is its own business -- so only the copy with the lowest index takes part here: three stable
// radix sorts (z, y, x keys; -0 counts as +0) order the points lexicographically with the index as the last key, and a
// point equal to its predecessor is a later copy.
__global__ __launch_bounds__(256) void hpr_dupkey_kernel(int n, const float *__restrict__ pts, int axis, const int *__restrict__ order,
                                                        unsigned *__restrict__ keys, int *__restrict__ idx)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int i = order ? order[p] : p;
    keys[p] = hpr_ord(pts[(size_t)i * 3 + axis] + 0.0f);
    if (!order) idx[p] = p;
}
__global__ __launch_bounds__(256) void hpr_dupmark_kernel(int n, const float *__restrict__ pts, const int *__restrict__ order,
                                                         unsigned char *__restrict__ dup)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int i = order[p];
    bool d = false;
    if (p > 0) {
        const int j = order[p - 1];
        d = pts[(size_t)i * 3 + 0] == pts[(size_t)j * 3 + 0] && pts[(size_t)i * 3 + 1] == pts[(size_t)j * 3 + 1] &&
            pts[(size_t)i * 3 + 2] == pts[(size_t)j * 3 + 2];
    }
    dup[i] = d ? 1 : 0;
}

// bounds[0..2] = min, [3..5] = max of the finite coordinates, as ordered uints (0xffffffff / 0 initially)
__global__ __launch_bounds__(256) void hpr_bounds_kernel(int n, const float *__restrict__ pts, unsigned *bounds)
{
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = pts[(size_t)i * 3 + k];
            if (!(fabsf(v) < __builtin_inff())) continue;
            const unsigned o = hpr_ord(v);
            mn[k] = o < mn[k] ? o : mn[k];
            mx[k] = o > mx[k] ? o : mx[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned a = (unsigned)__shfl_xor((int)mn[k], off, kWave), b = (unsigned)__shfl_xor((int)mx[k], off, kWave);
            mn[k] = a < mn[k] ? a : mn[k];
            mx[k] = b > mx[k] ? b : mx[k];
        }
        if ((threadIdx.x & (kWave - 1)) == 0) {
            atomicMin(&bounds[k], mn[k]);
            atomicMax(&bounds[3 + k], mx[k]);
        }
    }
}

__device__ __forceinline__ unsigned hpr_spread10(unsigned v)      // 10 bits -> every second bit
{
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// The frame in which a view's directions are ordered: w = towards the centre of the cloud's bounding box,
// (E1, E2) the completion hpr_frame uses, s = the largest |x|, |y| a direction to a point of the box can have
// (sine of the bounding sphere's angular radius, 5 % slack; 1 when the eye is inside that sphere).
struct HprViewFrame {
    double e1x, e1y, e1z, e2x, e2y, e2z, s;
};

__device__ __forceinline__ HprViewFrame hpr_view_frame(const unsigned *bounds, const double *eye)
{
    const double lx = (double)hpr_unord(bounds[0]), ly = (double)hpr_unord(bounds[1]), lz = (double)hpr_unord(bounds[2]);
    const double hx = (double)hpr_unord(bounds[3]), hy = (double)hpr_unord(bounds[4]), hz = (double)hpr_unord(bounds[5]);
    double wx = (lx + hx) * 0.5 - eye[0], wy = (ly + hy) * 0.5 - eye[1], wz = (lz + hz) * 0.5 - eye[2];
    const double dist = sqrt(wx * wx + wy * wy + wz * wz);
    const double hd = 0.5 * sqrt((hx - lx) * (hx - lx) + (hy - ly) * (hy - ly) + (hz - lz) * (hz - lz));
    HprViewFrame v;
    if (dist > 0.0 && dist < __builtin_inf()) { wx /= dist; wy /= dist; wz /= dist; }
    else { wx = 0.0; wy = 0.0; wz = 1.0; }
    const double ax = fabs(wx), ay = fabs(wy), az = fabs(wz);
    double x, y, z;
    if (ax <= ay && ax <= az) { x = 0.0; y = wz; z = -wy; }
    else if (ay <= az)        { x = -wz; y = 0.0; z = wx; }
    else                      { x = wy; y = -wx; z = 0.0; }
    const double l = sqrt(x * x + y * y + z * z);
    v.e1x = x / l; v.e1y = y / l; v.e1z = z / l;
    v.e2x = wy * v.e1z - wz * v.e1y;
    v.e2y = wz * v.e1x - wx * v.e1z;
    v.e2z = wx * v.e1y - wy * v.e1x;
    double s = 1.0;
    if (dist > hd) {
        s = 1.05 * hd / dist;
        s = s < 1.0 ? s : 1.0;
    }
    v.s = s > 0.0 ? s : 1.0;
    return v;
}

// Sort key of (view, point): view << 20 | 2-D Morton code of the point's DIRECTION from the eye, 10 bits per
// axis over [-s, s]^2 (non-finite points last within their view).  Points that lie in the same direction --
// a front surface and what it hides -- become neighbours; the order has no influence on the result.
__global__ __launch_bounds__(256) void hpr_key_kernel(int n, const float *__restrict__ pts, const unsigned *__restrict__ bounds,
                                                     const double *__restrict__ eyes, unsigned *__restrict__ keys, int *__restrict__ idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x, view = blockIdx.y;
    if (i >= n) return;
    const HprViewFrame V = hpr_view_frame(bounds, eyes + view * 3);
    const double vx = (double)pts[(size_t)i * 3 + 0] - eyes[view * 3 + 0];
    const double vy = (double)pts[(size_t)i * 3 + 1] - eyes[view * 3 + 1];
    const double vz = (double)pts[(size_t)i * 3 + 2] - eyes[view * 3 + 2];
    const double r = sqrt(vx * vx + vy * vy + vz * vz);
    unsigned key = 0xfffffu;
    if (r > 0.0 && r < __builtin_inf()) {
        const double dx = vx / r, dy = vy / r, dz = vz / r;
        const double x = dx * V.e1x + dy * V.e1y + dz * V.e1z;
        const double y = dx * V.e2x + dy * V.e2y + dz * V.e2z;
        double qx = floor((x / V.s + 1.0) * 512.0), qy = floor((y / V.s + 1.0) * 512.0);
        qx = qx < 0.0 ? 0.0 : (qx > 1023.0 ? 1023.0 : qx);
        qy = qy < 0.0 ? 0.0 : (qy > 1023.0 ? 1023.0 : qy);
        key = hpr_spread10((unsigned)qx) | (hpr_spread10((unsigned)qy) << 1);
    }
    keys[(size_t)view * n + i] = ((unsigned)view << 20) | key;
    idx[(size_t)view * n + i] = i;
}

__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }

struct HprFrame {
    double px, py, pz, rho, ux, uy, uz, e1x, e1y, e1z, e2x, e2y, e2z;
};

// u, e1, e2 of a flipped point; false when it coincides with the origin or is not finite
__device__ __forceinline__ bool hpr_frame(const double *p, HprFrame &f)
{
    f.px = p[0]; f.py = p[1]; f.pz = p[2];
    f.rho = sqrt(f.px * f.px + f.py * f.py + f.pz * f.pz);
    if (!(f.rho > 0.0) || !(f.rho < __builtin_inf())) return false;
    f.ux = f.px / f.rho; f.uy = f.py / f.rho; f.uz = f.pz / f.rho;
    const double ax = fabs(f.ux), ay = fabs(f.uy), az = fabs(f.uz);
    double x, y, z;
    if (ax <= ay && ax <= az) { x = 0.0; y = f.uz; z = -f.uy; }
    else if (ay <= az)        { x = -f.uz; y = 0.0; z = f.ux; }
    else                      { x = f.uy; y = -f.ux; z = 0.0; }
    const double l = sqrt(x * x + y * y + z * z);
    f.e1x = x / l; f.e1y = y / l; f.e1z = z / l;
    f.e2x = f.uy * f.e1z - f.uz * f.e1y;
    f.e2y = f.uz * f.e1x - f.ux * f.e1z;
    f.e2z = f.ux * f.e1y - f.uy * f.e1x;
    return true;
}

// What a thread needs to know about a tile of kHprThreads consecutive (Morton-neighbour) candidates to skip
// it whole: their directions lie within angle phi of the axis w, their lengths are <= rho_max.
struct HprTile {
    double wx, wy, wz, cos_phi, sin_phi, cos2_phi, sin2_phi, inv_rho_max;
};

__global__ __launch_bounds__(kHprThreads) void hpr_tile_kernel(int n, const double *__restrict__ fl_all, HprTile *__restrict__ tiles)
{
    __shared__ double red[4][kHprThreads];
    const int view = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int pos = tile * kHprThreads + tid;
    double ux = 0.0, uy = 0.0, uz = 0.0, rho = 0.0;
    bool ok = false;
    if (pos < n) {
        const double *p = fl_all + ((size_t)view * n + pos) * 3;
        rho = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        ok = rho > 0.0 && rho < __builtin_inf();
        if (ok) { ux = p[0] / rho; uy = p[1] / rho; uz = p[2] / rho; }
        else rho = 0.0;
    }
    red[0][tid] = ux; red[1][tid] = uy; red[2][tid] = uz; red[3][tid] = rho;
    __syncthreads();
    for (int h = kHprThreads / 2; h > 0; h >>= 1) {
        if (tid < h) {
            red[0][tid] += red[0][tid + h];
            red[1][tid] += red[1][tid + h];
            red[2][tid] += red[2][tid + h];
            red[3][tid] = red[3][tid] > red[3][tid + h] ? red[3][tid] : red[3][tid + h];
        }
        __syncthreads();
    }
    const double sx = red[0][0], sy = red[1][0], sz = red[2][0], rho_max = red[3][0];
    const double l = sqrt(sx * sx + sy * sy + sz * sz);
    __syncthreads();
    const bool axis = l > 0.0 && rho_max > 0.0;
    const double wx = axis ? sx / l : 0.0, wy = axis ? sy / l : 0.0, wz = axis ? sz / l : 0.0;
    red[0][tid] = ok ? ux * wx + uy * wy + uz * wz : 1.0;
    __syncthreads();
    for (int h = kHprThreads / 2; h > 0; h >>= 1) {
        if (tid < h) red[0][tid] = red[0][tid] < red[0][tid + h] ? red[0][tid] : red[0][tid + h];
        __syncthreads();
    }
    if (tid == 0) {
        HprTile t;
        double c = red[0][0] - 1e-12;          // widen the cone past roundoff
        if (!axis) c = -1.0;                   // no usable axis: never skipped
        t.wx = wx; t.wy = wy; t.wz = wz;
        t.cos_phi = c;
        t.cos2_phi = c * c;
        t.sin2_phi = 1.0 - c * c + 1e-15;
        t.sin_phi = sqrt(t.sin2_phi) * (1.0 + 1e-15);
        t.inv_rho_max = rho_max > 0.0 ? 1.0 / (rho_max * (1.0 + 1e-12)) : 0.0;
        tiles[(size_t)view * gridDim.x + tile] = t;
    }
}


// Sutherland-Hodgman against a A + b B <= C: src (nv vertices, element k at src[k * ss]) -> dst (stride ds);
// returns the new count.  A vertex exactly on the line is kept and spawns no intersection point.
__device__ __forceinline__ int hpr_clip(const double2 *src, int ss, int nv, double A, double B, double C, double2 *dst, int ds)
{
    int m = 0;
    double2 cur = src[0];
    double s0 = cur.x * A + cur.y * B - C;
    for (int k = 0; k < nv; k++) {
        const int k2 = k + 1 < nv ? k + 1 : 0;
        const double2 nxt = src[(size_t)k2 * ss];
        const double s1 = nxt.x * A + nxt.y * B - C;
        if (!(s0 > 0.0)) dst[(size_t)(m++) * ds] = cur;
        if ((s0 > 0.0) != (s1 > 0.0) && s0 != 0.0 && s1 != 0.0) {
            const double t = s0 / (s0 - s1);
            double2 x;
            x.x = cur.x + t * (nxt.x - cur.x);
            x.y = cur.y + t * (nxt.y - cur.y);
            dst[(size_t)(m++) * ds] = x;
        }
        cur = nxt;
        s0 = s1;
    }
    return m;
}

// The same clip IN PLACE (no scratch): the outside vertices of a convex polygon are one cyclic run [f, f+L);
// it is replaced by the two edge crossings.  Each crossing is computed exactly as hpr_clip does (from the
// edge's first vertex, in the polygon's orientation), so the new polygon is hpr_clip's up to a rotation of the
// array -- which changes nothing downstream.  `out` = number of outside vertices (0 < out < nv), f = the first
// of them (the one whose predecessor is inside).
__device__ __forceinline__ int hpr_clip_inplace(double2 *p, int ss, int nv, int out, int f, double A, double B, double C)
{
    const int ia = f ? f - 1 : nv - 1, ib = f;
    int ic = f + out - 1, id = f + out;
    ic -= ic >= nv ? nv : 0;
    id -= id >= nv ? nv : 0;
    const double2 va = p[(size_t)ia * ss], vb = p[(size_t)ib * ss], vc = p[(size_t)ic * ss], vd = p[(size_t)id * ss];
    const double sa = va.x * A + va.y * B - C, sb = vb.x * A + vb.y * B - C;
    const double sc = vc.x * A + vc.y * B - C, sd = vd.x * A + vd.y * B - C;
    double2 x1, x2;
    const bool h1 = sa != 0.0, h2 = sd != 0.0;          // (sb > 0 and sc > 0 always)
    {
        const double t = sa / (sa - sb);
        x1.x = va.x + t * (vb.x - va.x);
        x1.y = va.y + t * (vb.y - va.y);
    }
    {
        const double t = sc / (sc - sd);
        x2.x = vc.x + t * (vd.x - vc.x);
        x2.y = vc.y + t * (vd.y - vc.y);
    }
    const int e = (h1 ? 1 : 0) + (h2 ? 1 : 0);
    const double2 y0 = h1 ? x1 : x2;          // the crossings in order: y0 [, x2]
    const int m = nv - out + e;
    if (f + out <= nv) {
        // no wrap: [0, f) stays, the crossings, then the tail [f + out, nv) moved to f + e
        const int from = f + out, to = f + e, cnt = nv - from;
        if (to < from) {
            for (int k = 0; k < cnt; k++) p[(size_t)(to + k) * ss] = p[(size_t)(from + k) * ss];
        } else if (to > from) {
            for (int k = cnt - 1; k >= 0; k--) p[(size_t)(to + k) * ss] = p[(size_t)(from + k) * ss];
        }
        if (e > 0) p[(size_t)f * ss] = y0;
        if (e > 1) p[(size_t)(f + 1) * ss] = x2;
    } else {
        // the run wraps: the inside vertices [id, f) move to the front, the crossings follow
        const int cnt = f - id;
        if (id > 0)
            for (int k = 0; k < cnt; k++) p[(size_t)k * ss] = p[(size_t)(id + k) * ss];
        if (e > 0) p[(size_t)cnt * ss] = y0;
        if (e > 1) p[(size_t)(cnt + 1) * ss] = x2;
    }
    return m;
}

// What a polygon can reach: its largest squared vertex norm D^2 and its support max_v (v . d) in the eight
// directions d = (+-1, 0), (0, +-1), (+-1, +-1).  UPPER bounds suffice (they only feed conservative rejections),
// and a clip only shrinks the polygon, so these are refreshed now and then, not after every clip.
constexpr int kHprBatchA = 64;     // tiles per round of the accept pass

struct HprReach {
    double d2, xp, xn, yp, yn, pp, pn, np, nn;
};

__device__ __forceinline__ HprReach hpr_reach(const double2 *p, int ss, int nv)
{
    HprReach r;
    double d2 = 0.0, xp = -__builtin_inf(), xn = xp, yp = xp, yn = xp, pp = xp, pn = xp, np = xp, nn = xp;
    for (int k = 0; k < nv; k++) {
        const double2 v = p[(size_t)k * ss];
        const double r2 = v.x * v.x + v.y * v.y, s = v.x + v.y, t = v.x - v.y;
        d2 = r2 > d2 ? r2 : d2;
        xp = v.x > xp ? v.x : xp;   xn = -v.x > xn ? -v.x : xn;
        yp = v.y > yp ? v.y : yp;   yn = -v.y > yn ? -v.y : yn;
        pp = s > pp ? s : pp;       nn = -s > nn ? -s : nn;
        pn = t > pn ? t : pn;       np = -t > np ? -t : np;
    }
    r.d2 = d2;
    r.xp = xp; r.xn = xn; r.yp = yp; r.yn = yn; r.pp = pp; r.pn = pn; r.np = np; r.nn = nn;
    return r;
}

// True when the candidate's line a A + b B = C provably misses the polygon (then every vertex has
// a A + b B - C < 0 in floating point as well: the margins are 1e-9, the roundoff 1e-15).  Two bounds on
// max_v (a A + b B): Cauchy-Schwarz with the largest vertex norm, and -- for polygons that run out to the
// box in some directions (points on the silhouette) -- (A, B) split into its two neighbouring support
// directions: with a = |A| >= b = |B|, (A, B) = (a - b)(sx, 0) + b (sx, sy), so v.(A, B) <= (a-b) h(sx,0) + b h(sx,sy).
__device__ __forceinline__ bool hpr_far(const HprReach r, double A, double B, double C)
{
    if (C > 0.0 && r.d2 * (A * A + B * B) * 1.000000001 < C * C) return true;
    const double a = fabs(A), b = fabs(B);
    const double hx = A >= 0.0 ? r.xp : r.xn, hy = B >= 0.0 ? r.yp : r.yn;
    const double hd = A >= 0.0 ? (B >= 0.0 ? r.pp : r.pn) : (B >= 0.0 ? r.np : r.nn);
    const double t1 = a >= b ? (a - b) * hx : (b - a) * hy;
    const double t2 = (a >= b ? b : a) * hd;
    return t1 + t2 + 1e-9 * (fabs(t1) + fabs(t2) + fabs(C)) < C;
}

// Can any candidate of tile T cut the polygon?  Every candidate q = rho_q u_q of the tile has u_q within phi of
// the axis w and rho_q <= rho_max, so for a normal n:  n.q <= rho_max |n| cos(max(0, angle(n, w) - phi)).
// First for all n = u + d with |d| <= D at once (|n| <= L, angle(n, u) <= psi), then vertex by vertex.  The
// tile is skipped when the bound stays below |p'_i| (margins 1e-9 >> roundoff; squares instead of sqrt).
__device__ __forceinline__ bool hpr_tile_needed(const HprFrame &f, double cpsi, double spsi, const double2 *p, int ss, int nv,
                                                const HprTile &T)
{
    if (!(T.cos_phi > 0.0)) return true;
    const double uw = f.ux * T.wx + f.uy * T.wy + f.uz * T.wz;
    const double rr = f.rho * (1.0 - 1e-9) * T.inv_rho_max;
    {
        const double cc = T.cos_phi * cpsi - T.sin_phi * spsi;       // cos(phi + psi), psi = atan D
        const double sc = T.sin_phi * cpsi + T.cos_phi * spsi;       // sin(phi + psi)
        if (cc > 0.0 && uw < cc * (1.0 - 1e-9)) {                        // angle(u, w) > phi + psi
            const double rhs = rr * cpsi - 1e-9 - cc * uw;                 // cpsi = 1 / L, L = sqrt(1 + D^2) >= |n|
            const double s2 = sc * sc * ((1.0 - uw * uw) * (1.0 + 1e-9) + 1e-12);
            if (rhs > 0.0 && s2 < rhs * rhs) return false;
        }
    }
    const double e1w = f.e1x * T.wx + f.e1y * T.wy + f.e1z * T.wz;
    const double e2w = f.e2x * T.wx + f.e2y * T.wy + f.e2z * T.wz;
    for (int k = 0; k < nv; k++) {
        const double2 v = p[(size_t)k * ss];
        const double nn = 1.0 + v.x * v.x + v.y * v.y;
        const double dw = uw + v.x * e1w + v.y * e2w;
        if (dw > 0.0 && dw * dw >= T.cos2_phi * nn * (1.0 - 1e-9)) {
            if (!(nn < rr * rr)) return true;                            // n inside the cone: bound rho_max |n|
        } else {
            const double rhs = rr - 1e-9 * nn - T.cos_phi * dw;
            const double s2 = T.sin2_phi * ((nn - dw * dw) * (1.0 + 1e-9) + 1e-12 * nn);
            if (!(rhs > 0.0 && s2 < rhs * rhs)) return true;
        }
    }
    return false;
}

// tiles outward from the own one: step 0 = own, 1 = own + 1, 2 = own - 1, 3 = own + 2, ...
__device__ __forceinline__ int hpr_tile_of(int step, int own) { return (step & 1) ? own + (step + 1) / 2 : own - step / 2; }

// the tile a group of 128 consecutive listed points starts from: that of its middle point
__device__ __forceinline__ int hpr_base_tile(const int *hl, int nhard, int rank)
{
    int mid = (rank / kHprThreads) * kHprThreads + kHprThreads / 2;
    mid = mid < nhard ? mid : nhard - 1;
    return hl[mid] / kHprThreads;
}

// Early accept.  If the point's own direction is already a separating normal -- u.p'_j < |p'_i| for every other
// point, i.e. the origin of the (a, b) plane satisfies every constraint strictly -- the polygon contains a disc
// around the origin and the point is visible; no polygon has to be built.  That is the case for most visible
// points (on a smooth patch nothing projects farther along the point's own ray direction).  The test is one dot
// product per candidate, tiles are skipped with the cone bound for the single normal n = u (D = 0), and the
// margin (1e-8 of the larger |p'|) is far above anything roundoff does to the clipped polygon, so the answer
// agrees with the full computation.  Points that fail go through hpr_kernel as before.
// Blocks go to the 8 XCDs round-robin by linear id, each XCD with its own 4 MB L2.  1-D launches of ntiles blocks for each
// of c views, mapped so that the blocks of a view run on ONE XCD (c a multiple of 8; otherwise view-major order): they all
// read the view's flipped points and tile records, and with a view's blocks on eight XCDs every L2 held every view in flight
// (1 GB fetched from HBM per launch for 246 MB of input at 1024 x 10000).
__device__ __forceinline__ void hpr_block(int ntiles, int c, int &view, int &tile)
{
    const int lin = blockIdx.x;
    if ((c & 7) == 0) {
        const int xcd = lin & 7, k = lin >> 3;
        view = 8 * (k / ntiles) + xcd;
        tile = k % ntiles;
    } else {
        view = lin / ntiles;
        tile = lin % ntiles;
    }
}

__global__ __launch_bounds__(kHprThreads) void hpr_accept_kernel(int n, const double *__restrict__ fl_all, const int *__restrict__ perm,
                                                                const HprTile *__restrict__ tiles_all, unsigned char *__restrict__ hard,
                                                                unsigned char *__restrict__ vis, int *__restrict__ cnt, int accept_none,
                                                                int nviews)
{
    __shared__ double4 s_stage[kHprThreads];
    __shared__ unsigned long long s_mask;
    const int tid = threadIdx.x;
    const int ntiles = ceil_div_dev(n, kHprThreads);
    int view, own;
    hpr_block(ntiles, nviews, view, own);
    const double *fl = fl_all + (size_t)view * n * 3;
    const int pos = own * kHprThreads + tid;
    const HprTile *tiles = tiles_all + (size_t)view * ntiles;
    HprFrame f;
    bool active = false;
    if (pos < n) active = hpr_frame(fl + (size_t)pos * 3, f);
    const bool valid = active;
    if (accept_none) active = false;           // measurement knob: every point goes through the polygon kernel
    bool ok = active;
    HprTile *s_rec = (HprTile *)s_stage;
    for (int step0 = 0; step0 < 2 * ntiles; step0 += kHprBatchA) {
        if (__syncthreads_count(active) == 0) break;
        if (tid < kHprBatchA) {
            const int tile = hpr_tile_of(step0 + tid, own);
            if (tile >= 0 && tile < ntiles) s_rec[tid] = tiles[tile];
        }
        if (tid == 0) s_mask = 0ull;
        __syncthreads();
        unsigned long long mine = 0ull;
        if (active) {
            for (int b = 0; b < kHprBatchA; b++) {
                const int tile = hpr_tile_of(step0 + b, own);
                if (tile < 0 || tile >= ntiles) continue;
                const HprTile &T = s_rec[b];
                bool need = !(T.cos_phi > 0.0);
                if (!need) {
                    // max over the tile of u.q <= rho_max cos(max(0, angle(u, w) - phi)); needed unless that is
                    // below |p'_i| minus the margin of the candidate test
                    const double uw = f.ux * T.wx + f.uy * T.wy + f.uz * T.wz;
                    const double rr = f.rho * (1.0 - 1e-7) * T.inv_rho_max;
                    if (uw >= T.cos_phi * (1.0 - 1e-9)) {
                        need = !(rr > 1.0);
                    } else {
                        const double rhs = rr - 1e-9 - T.cos_phi * uw;
                        const double s2 = T.sin2_phi * ((1.0 - uw * uw) * (1.0 + 1e-9) + 1e-12);
                        need = !(rhs > 0.0 && s2 < rhs * rhs);
                    }
                }
                if (need) mine |= 1ull << b;
            }
        }
        {
            unsigned long long w = mine;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) w |= (unsigned long long)__shfl_xor((long long)w, o, kWave);
            if ((tid & (kWave - 1)) == 0 && w) atomicOr(&s_mask, w);
        }
        __syncthreads();
        unsigned long long todo = s_mask;
        while (todo) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int tile = hpr_tile_of(step0 + b, own);
            const int tile0 = tile * kHprThreads;
            const int tn = min(kHprThreads, n - tile0);
            {
                double4 q = make_double4(__builtin_nan(""), 0.0, 0.0, 0.0);
                if (tid < tn) {
                    const double *g = fl + (size_t)(tile0 + tid) * 3;
                    q = make_double4(g[0], g[1], g[2], 0.0);
                }
                s_stage[tid] = q;
            }
            __syncthreads();
            if (active && ((mine >> b) & 1ull)) {
                // The test has a margin of 1e-8 |p'| and only ever ACCEPTS (a point it turns down takes the exact path), so
                // its arithmetic need not be the oracle's: fused multiply-adds, and the point itself recognised by its
                // position instead of by value (14 -> 8 instructions per candidate; the kernel is VALU-bound).
                const double lim = f.rho - 1e-8 * f.rho;          // C < thr  <=>  u.q > lim
                const int self_t = tile == own ? tid : -1;
                bool bad = false;
#pragma unroll 8
                for (int t = 0; t < kHprThreads; t++) {
                    const double4 q = s_stage[t];          // rows past the end are NaN: the comparison below is false
                    const double d = __builtin_fma(f.uz, q.z, __builtin_fma(f.uy, q.y, f.ux * q.x));
                    bad |= t != self_t && d > lim;
                }
                if (bad) { ok = false; active = false; }
            }
            __syncthreads();
        }
    }
    if (pos < n) {
        hard[(size_t)view * n + pos] = (valid && !ok) ? 1 : 0;
        vis[(size_t)view * n + perm[(size_t)view * n + pos]] = ok ? 1 : 0;          // (the polygon kernel overwrites its own points)
    }
    const int c = __syncthreads_count(ok);
    if (tid == 0 && c) atomicAdd(&cnt[view], c);
}

// The points the accept pass left over, per view in Morton order: hardlist[view][rank] = position, hardcnt[view].
// One block per view; an order-preserving scan (the order defines who shares a block of hpr_kernel).
__global__ __launch_bounds__(1024) void hpr_compact_kernel(int n, const unsigned char *__restrict__ hard, int *__restrict__ hardlist,
                                                          int *__restrict__ hardcnt)
{
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int view = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int p0 = 0; p0 < n; p0 += 1024) {
        const int pos = p0 + tid;
        const bool h = pos < n && hard[(size_t)view * n + pos];
        const unsigned long long bal = __ballot(h);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int wbase = 0, total = 0;
        for (int w = 0; w < 16; w++) {
            wbase += w < wave ? s_w[w] : 0;
            total += s_w[w];
        }
        const int base = s_base;
        if (h) hardlist[(size_t)view * n + base + wbase + before] = pos;
        __syncthreads();
        if (tid == 0) s_base = base + total;
        __syncthreads();
    }
    if (tid == 0) hardcnt[view] = s_base;
}

// tools/hpr_phases.py builds a private copy with -DGENPC_HPR_PROF: per block, shader-clock ticks of hpr_kernel's phases and
// the trip counts of its clip loop, summed into g_hpr_prof (compiled out of the shipped library)
#ifdef GENPC_HPR_PROF
__device__ unsigned long long g_hpr_prof[16];
#define HPR_PROF_DECL unsigned long long prof_c[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define HPR_PROF_ADD(k, v) do { prof_c[k] += (unsigned long long)(v); } while (0)      // per lane, in registers
#define HPR_PROF_FLUSH() do { for (int k_ = 0; k_ < 10; k_++) { unsigned long long v_ = prof_c[k_]; if (k_ >= 4 && k_ != 8) { for (int o_ = 32; o_ > 0; o_ >>= 1) { const unsigned long long x_ = __shfl_xor(v_, o_, 64); v_ = k_ == 5 ? (x_ > v_ ? x_ : v_) : v_ + x_; } } \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_hpr_prof[k_], v_); } } while (0)
#define HPR_PROF_CLOCK() __builtin_readcyclecounter()
#else
#define HPR_PROF_DECL do {} while (0)
#define HPR_PROF_ADD(k, v) do {} while (0)
#define HPR_PROF_FLUSH() do {} while (0)
#define HPR_PROF_CLOCK() 0ull
#endif
constexpr int kHprBatch = 64;      // tiles tested per round (one bit each)
constexpr int kHprRimStep = 8;     // after this many tiles ...
constexpr double kHprRimD2 = 1.0e6; // ... a polygon with a vertex farther than 1000 from the origin is a silhouette point's
constexpr int kHprRimTiles = 256;  // ... and is handed to the second pass if the cloud has at least this many tiles

// Parked points are stored in eight segments, one per XCD: the blocks of a view run on XCD view % 8 (hpr_block), one view
// after the other, so a segment holds its views' points grouped by view -- and the wave-per-point kernels send block b to
// segment b % 8 (the XCD it runs on), where consecutive waves then read ONE view's flipped points and tile records
// through that XCD's L2 (in one list in order of arrival every L2 saw eight to sixteen views at a time).
// base[x] = start of segment x (prefix sums of the views' listed points: upper bounds), status[32 + x] = its fill.
// The record lives in device memory, written by hpr_segs_kernel from the accept pass's per-view counts: the host never
// sees a count of this entry point (round 4 fetched six of them, a stream synchronisation each).
struct HprSegs {
    int base[9];
    int on;            // 0: one segment (views not a multiple of 8)
    int cap;           // slots the polygon store holds: a point whose slot lies beyond is listed (restarts from the box) instead of parked
};
__device__ __forceinline__ int hpr_seg_of(const HprSegs &sg, int view) { return sg.on ? (view & 7) : 0; }
// item b of a pass over `8 x longest segment` (or the one segment): its slot, -1 past the segment's end
__device__ __forceinline__ int hpr_seg_slot(const HprSegs &sg, const int *status, int b)
{
    const int x = sg.on ? (b & 7) : 0, k = sg.on ? (b >> 3) : b;
    const int slot = sg.base[x] + k;
    return k < status[32 + x] && slot < sg.cap ? slot : -1;
}
// items of such a pass
__device__ __forceinline__ int hpr_seg_items(const HprSegs &sg, const int *status)
{
    int longest = 0;
    for (int x = 0; x < 8; x++) longest = max(longest, status[32 + x]);
    return sg.on ? 8 * longest : longest;
}

// (also clears the work counters of the wave-per-point passes: kHprWorkWords ints, a 128-byte line per counter)
constexpr int kHprWorkLine = 32, kHprWorkWords = 16 * kHprWorkLine;
constexpr int kHprDecideChunk = 4;      // consecutive parked points a wave of hpr_decide_kernel takes per draw
__global__ __launch_bounds__(1024) void hpr_segs_kernel(int c, const int *__restrict__ hardcnt, HprSegs *__restrict__ out, int cap, int *__restrict__ work)
{
    __shared__ int s_per[8];
    if (threadIdx.x < kHprWorkWords) work[threadIdx.x] = 0;
    if (threadIdx.x < 8) s_per[threadIdx.x] = 0;
    __syncthreads();
    const int on = (c & 7) == 0 ? 1 : 0;
    int mine = 0;            // (thread t takes views t, t + 1024, ...: all of one segment)
    for (int v = threadIdx.x; v < c; v += 1024) mine += hardcnt[v];
    if (mine) atomicAdd(&s_per[on ? (threadIdx.x & 7) : 0], mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        int at = 0;            // (the sum is at most views x points < 2^31)
        for (int x = 0; x < 8; x++) { out->base[x] = at; at += s_per[x]; }
        out->base[8] = at;
        out->on = on;
        out->cap = cap;
    }
}

// status[1] != 0 (an internal error) reaches the caller without a host read of it: every count becomes -1
__global__ __launch_bounds__(256) void hpr_finish_kernel(int c, const int *__restrict__ status, int *__restrict__ cnt)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < c && status[1] != 0) cnt[v] = -1;
}

// status[0] = points handed to the wave-per-point pass, status[1] = error (2: a polygon outgrew kHprOverCap),
// status[2] = points left undecided by the split first kernel (continued by the wave-per-point pass)
__global__ __launch_bounds__(kHprThreads) void hpr_kernel(int n, const double *__restrict__ fl_all, const int *__restrict__ perm,
                                                         const HprTile *__restrict__ tiles_all, const int *__restrict__ hardlist,
                                                         const int *__restrict__ hardcnt, unsigned char *__restrict__ vis,
                                                         int *__restrict__ cnt, int *status, int *__restrict__ over_list, int no_cull, int max_clips,
                                                         int split, int4 *__restrict__ surv, double2 *__restrict__ surv_poly,
                                                         int *__restrict__ und, int straggle_from, int straggle_lanes, int nviews, int home_tiles, const HprSegs *__restrict__ segs_p, int chunk_w, int home_chunks)
{
    HPR_PROF_DECL;
    const HprSegs segs = *segs_p;
    __shared__ double2 s_poly[kHprMaxV * kHprThreads];
    __shared__ double4 s_stage[kHprThreads];      // 64 tile records while testing, then one tile's candidates
    __shared__ unsigned long long s_mask;
    static_assert(sizeof(HprTile) * kHprBatch <= sizeof(double4) * kHprThreads, "the tile records share the staging area");
    const int tid = threadIdx.x;
    int view, blk;
    hpr_block(ceil_div_dev(n, kHprThreads), nviews, view, blk);
    const double *fl = fl_all + (size_t)view * n * 3;
    // block g of a view owns the points of rank 128 g .. 128 g + 127 in the view's list of points the accept pass
    // left over; its tile order starts at the tile of its middle point (hpr_base_tile: the oracle's rule too)
    const int nhard = hardcnt[view];
    if (blk * kHprThreads >= nhard) return;
    const int *hl = hardlist + (size_t)view * n;
    const int rank = blk * kHprThreads + tid;
    const int pos = rank < nhard ? hl[rank] : -1;
    const int i = pos >= 0 ? perm[(size_t)view * n + pos] : -1;
    const int ntiles = ceil_div_dev(n, kHprThreads), own = hpr_base_tile(hl, nhard, rank);
    const HprTile *tiles = tiles_all + (size_t)view * ntiles;
    HprFrame f;
    bool active = false;
    int nv = 0;            // > 0 visible so far, 0 hidden, -1 handed to the second pass
    if (i >= 0) active = hpr_frame(fl + (size_t)pos * 3, f);
    double2 *poly = s_poly + tid;
    HprReach R = {};
    if (active) {
        nv = 4;
        poly[0 * kHprThreads] = make_double2(-kHprBox, -kHprBox);
        poly[1 * kHprThreads] = make_double2(kHprBox, -kHprBox);
        poly[2 * kHprThreads] = make_double2(kHprBox, kHprBox);
        poly[3 * kHprThreads] = make_double2(-kHprBox, kHprBox);
        R = hpr_reach(poly, kHprThreads, nv);
    }
    int nclips = 0;
    HprTile *s_rec = (HprTile *)s_stage;
    const int home = pos >= 0 ? pos / kHprThreads : -0x40000000;      // the tile this point lies in
    // Stage one tile of candidates and let the lanes that want it clip against it.  Pass 1 over the tile, the
    // same for every lane: which candidates are not provably out of reach (one bit each).  Pass 2: every lane
    // takes ITS marked candidates in order -- the lanes of a wave run their k-th marked candidate together, so
    // the wave pays for the longest list, not for the sum of all of them (clipping candidate by candidate as
    // they come costs 10x more: the lanes meet their cuts at different candidates).
    // Clip the polygon of this lane by the 128 candidates of one tile; `load(k)` returns candidate k of the tile
    // (rows past the end of the cloud: NaN).
    // Split form: a point this kernel does not finish is parked with its polygon as the exact path left it and the place
    // to resume from -- code = 256 x (home tiles done: 0 .. 3) + candidates of the next home tile already taken -- and a
    // wave of hpr_overflow_kernel continues from there (round 3 restarted such points from the box: 6.5 of 45 ms).
    auto park = [&](int code) {
        const int sx = hpr_seg_of(segs, view);
        const int slot = segs.base[sx] + atomicAdd(&status[32 + sx], 1);
        if (und) atomicAdd(&und[view], 1);
        if (slot >= segs.cap) {          // the polygon store is full: listed instead
            over_list[atomicAdd(&status[0], 1)] = view * n + rank;
            nv = -1;
            active = false;
            return;
        }
        atomicAdd(&status[2], 1);
        surv[slot] = make_int4(view * n + rank, nv, pos, code);
        double2 *sp = surv_poly + (size_t)slot * kHprMaxV;
        for (int k = 0; k < nv; k++) sp[k] = poly[k * kHprThreads];
        nv = -1;
        active = false;
    };
    auto clip_by_tile = [&](int tile, int rel, auto load) {
                // 32 candidates at a time: the marks are taken against the polygon as the previous 32 left it (while
                // the polygon is still the box every candidate is marked; after the nearest few it is tight and
                // almost none are)
                // (in the point's HOME tile the chunks start with its own: nearest neighbours first)
                const int rot = tile == home ? ((pos - home * kHprThreads) >> 5) : 0;
                for (int cc = 0; cc < (rel == 0 ? home_chunks : kHprThreads / 32) && active; cc++) {
                    const int c0 = ((cc + rot) & (kHprThreads / 32 - 1)) * 32;
                    for (int s0 = 0; s0 < 32 && active; s0 += chunk_w) {
                    // Marks: the candidates that CUT the polygon as it is now (a vertex strictly outside) -- the polygon in
                    // registers, every lane the same straight-line test.  The clip loop below runs as long as its slowest
                    // lane, at ~500 instructions a trip; with marks from the reach bounds (hpr_far: 17 of 32 candidates
                    // marked per lane, a few of them real cuts) a wave made 28 trips per 32 candidates with a third of its
                    // lanes busy (tools/hpr_phases.py).  A candidate that does not cut now cannot cut what is left later.
                    double2 pv[kHprMaxV];
#pragma unroll
                    for (int k = 0; k < kHprMaxV; k++) pv[k] = poly[(k < nv ? k : 0) * kHprThreads];
                    unsigned m = 0u;
#pragma unroll 1
                    for (int t = s0; t < s0 + chunk_w; t++) {
                        const double4 q = load(c0 + t);      // rows past the end are NaN
                        const double A = f.e1x * q.x + f.e1y * q.y + f.e1z * q.z;
                        const double B = f.e2x * q.x + f.e2y * q.y + f.e2z * q.z;
                        const double C = f.rho - (f.ux * q.x + f.uy * q.y + f.uz * q.z);
                        const bool self = q.x == f.px && q.y == f.py && q.z == f.pz;      // the point itself, or an exact duplicate
                        bool cut = false;
#pragma unroll
                        for (int k = 0; k < kHprMaxV; k++) cut |= pv[k].x * A + pv[k].y * B - C > 0.0;      // (NaN rows: false)
                        m |= (!self && cut) ? (1u << t) : 0u;
                    }
                    HPR_PROF_ADD(4, 1);                                   // marking rounds (summed over lanes)
                    HPR_PROF_ADD(7, __popc(m));                           // marked candidates (summed over lanes)
                    while (m && active) {
                        HPR_PROF_ADD(5, 1);                               // clip-loop trips (the wave's = its lanes' maximum)
                        HPR_PROF_ADD(6, 1);                               // ... summed over lanes
                        const int t = __ffs((int)m) - 1;
                        m &= m - 1;
                        const double4 q = load(c0 + t);      // (recomputed: the same values)
                        const double A = f.e1x * q.x + f.e1y * q.y + f.e1z * q.z;
                        const double B = f.e2x * q.x + f.e2y * q.y + f.e2z * q.z;
                        const double C = f.rho - (f.ux * q.x + f.uy * q.y + f.uz * q.z);
                        // outside vertices: how many, and where their (cyclic) run starts
                        // (eight vertices are fetched at a time: one LDS round trip instead of eight in a row;
                        // slots past nv are read -- they belong to this thread -- and ignored)
                        int out = 0, first = 0, runs = 0;
                        bool prev_out;
                        {
                            const double2 v = poly[(nv - 1) * kHprThreads];
                            prev_out = v.x * A + v.y * B - C > 0.0;
                        }
                        for (int k0 = 0; k0 < nv; k0 += 8) {
                            double2 v[8];
#pragma unroll
                            for (int j = 0; j < 8; j++) v[j] = poly[(k0 + j < kHprMaxV ? k0 + j : kHprMaxV - 1) * kHprThreads];
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const bool o = k0 + j < nv && v[j].x * A + v[j].y * B - C > 0.0;
                                out += o ? 1 : 0;
                                first = (o && !prev_out) ? k0 + j : first;
                                runs += (o && !prev_out) ? 1 : 0;
                                prev_out = k0 + j < nv ? o : prev_out;
                            }
                        }
                        if (!out) continue;
                        if (out == nv) { nv = 0; active = false; break; }
                        // (runs > 1: on a sliver, roundoff can put the outside vertices in two runs; the in-place clip
                        // assumes one -- the second pass clips like the oracle does, vertex by vertex)
                        if (nv - out + 2 > kHprMaxV || runs > 1) {
                            if (split && rel >= 0) { park(rel * 256 + cc * 32 + t); break; }      // (candidate t not taken yet)
                            over_list[atomicAdd(&status[0], 1)] = view * n + rank;
                            if (und) atomicAdd(&und[view], 1);
                            nv = -1;
                            active = false;
                            break;
                        }
                        const int mm = hpr_clip_inplace(poly, kHprThreads, nv, out, first, A, B, C);
                        if (mm < 3) { nv = 0; active = false; break; }      // no interior left: not strictly extreme
                        nv = mm;
                        // a polygon that has been cut this often is a sliver far from the origin that thousands
                        // of candidates shave a little more (seen: 1400 cuts): such a point holds its whole block
                        // up -- the second pass, a wave per point with the candidates tested in parallel, is the
                        // place for it
                        if (++nclips > max_clips) {
                            if (split && rel >= 0) { park(rel * 256 + cc * 32 + t + 1); break; }
                            over_list[atomicAdd(&status[0], 1)] = view * n + rank;
                            if (und) atomicAdd(&und[view], 1);
                            nv = -1;
                            active = false;
                            break;
                        }
                    }
                    }
                }
                };
    // Stage one tile of candidates in LDS and let the lanes that want it clip against it.
    auto take_tile = [&](int tile, bool wanted) {
        const int tile0 = tile * kHprThreads;
        const int tn = min(kHprThreads, n - tile0);
        {
            // rows past the end are NaN: they cut nothing
            double4 q = make_double4(__builtin_nan(""), 0.0, 0.0, 0.0);
            if (tid < tn) {
                const double *g = fl + (size_t)(tile0 + tid) * 3;
                q = make_double4(g[0], g[1], g[2], 0.0);
            }
            s_stage[tid] = q;
        }
        __syncthreads();
        if (active && wanted) clip_by_tile(tile, -1, [&](int k) { return s_stage[k]; });
        __syncthreads();
    };
    // Phase 1: every point's home tile and its two neighbours, whatever the group's starting tile is (when few
    // points are left over, the 128 of a block lie many tiles apart; a polygon that has not met its nearest
    // neighbours first stays large for a long time).
    // Each lane reads ITS tiles straight from memory (its neighbours in the wave read the same or the next tile:
    // the lines are shared in L1/L2): staging them through LDS would take the block's tiles one after the other
    // with a few lanes busy on each -- measured 1.0 of a block's 2.7 ms.
    [[maybe_unused]] const unsigned long long prof_t0 = HPR_PROF_CLOCK();
    for (int rel = 0; rel < home_tiles && active; rel++) {            // home, home + 1, home - 1: nearest first
        const int tile = home + (rel == 0 ? 0 : (rel == 1 ? 1 : -1));
        if (tile < 0 || tile >= ntiles) continue;
        const int tile0 = tile * kHprThreads;
        clip_by_tile(tile, rel, [&](int k) {
            const int j = tile0 + k;
            if (j >= n) return make_double4(__builtin_nan(""), 0.0, 0.0, 0.0);
            const double *g = fl + (size_t)j * 3;
            return make_double4(g[0], g[1], g[2], 0.0);
        });
    }
    // Verify phase.  The polygon now reflects the nearest neighbours; later candidates mostly shave its corners.
    // Take its centroid (a*, b*) -- an interior point -- and test that ONE normal n = u + a* e1 + b* e2 against
    // every other point: if a* A + b* B - C < 0 with room to spare for all of them, (a*, b*) is strictly feasible,
    // the final polygon is not empty and the point is visible -- one dot product per candidate instead of the
    // polygon machinery, and tiles skipped with the same cone bound (for a one-vertex polygon).  A lane that
    // meets a counter-example goes on to phase 2 with its polygon untouched (the decision there is the same
    // computation as without this phase).  This is the early accept of hpr_accept_kernel with a better normal.
    // (Interior point: the centroid; for a polygon that runs out to the box, a point near its bounded end.)
    [[maybe_unused]] const unsigned long long prof_t1 = HPR_PROF_CLOCK();
    HPR_PROF_ADD(0, prof_t1 - prof_t0);            // home tiles
    HPR_PROF_ADD(8, 1);                            // waves
    if (!(no_cull & 32) && !(split && !(no_cull & 512))) {
        bool trying = active && nv >= 3;
        double2 ctr = make_double2(0.0, 0.0);
        if (trying) {
            // the two vertices nearest the origin (v0, v1) and the centroid
            double2 v0 = make_double2(0.0, 0.0), v1 = v0;
            double r0 = __builtin_inf(), r1 = __builtin_inf();
            for (int k = 0; k < nv; k++) {
                const double2 v = poly[k * kHprThreads];
                const double r2 = v.x * v.x + v.y * v.y;
                ctr.x += v.x;
                ctr.y += v.y;
                if (r2 < r0) { r1 = r0; v1 = v0; r0 = r2; v0 = v; }
                else if (r2 < r1) { r1 = r2; v1 = v; }
            }
            ctr.x /= (double)nv;
            ctr.y /= (double)nv;
            if (!(ctr.x * ctr.x + ctr.y * ctr.y < 1.0e6)) {
                // The polygon runs out to the box (a point on the silhouette): its centroid is somewhere near
                // the box.  Any point of the segment from a vertex to the centroid is interior: step from the
                // vertex nearest the origin towards the centroid by the length of the polygon's near part.
                const double dx = ctr.x - v0.x, dy = ctr.y - v0.y;
                const double len = sqrt(dx * dx + dy * dy);
                double h = sqrt((v1.x - v0.x) * (v1.x - v0.x) + (v1.y - v0.y) * (v1.y - v0.y));
                h = h < 0.5 * len ? h : 0.5 * len;
                trying = len > 0.0 && h > 0.0;
                if (trying) {
                    ctr.x = v0.x + h * (dx / len);
                    ctr.y = v0.y + h * (dy / len);
                }
            }
        }
        const double nx = f.ux + ctr.x * f.e1x + ctr.y * f.e2x, ny = f.uy + ctr.x * f.e1y + ctr.y * f.e2y,
                     nz = f.uz + ctr.x * f.e1z + ctr.y * f.e2z;
        const double nn = 1.0 + ctr.x * ctr.x + ctr.y * ctr.y;
        const double thr = 1e-10 * f.rho * nn;
        // a one-vertex "polygon" in LDS slot kHprMaxV - 1?  No: hpr_tile_needed reads through a pointer -- give it
        // a private copy (stride 0 is fine: one element)
        const double cl = sqrt(nn) * (1.0 + 1e-15), cpsi1 = 1.0 / cl, spsi1 = sqrt(nn - 1.0) / cl * (1.0 + 1e-15);
        for (int step0 = 0; step0 < 2 * ntiles; step0 += kHprBatch) {
            if (__syncthreads_count(trying) == 0) break;
            if (tid < kHprBatch) {
                const int tile = hpr_tile_of(step0 + tid, own);
                if (tile >= 0 && tile < ntiles) s_rec[tid] = tiles[tile];
            }
            if (tid == 0) s_mask = 0ull;
            __syncthreads();
            unsigned long long mine = 0ull;
            if (trying) {
                for (int b = 0; b < kHprBatch; b++) {
                    const int tile = hpr_tile_of(step0 + b, own);
                    if (tile < 0 || tile >= ntiles) continue;
                    if ((no_cull & 1) || hpr_tile_needed(f, cpsi1, spsi1, &ctr, 0, 1, s_rec[b])) mine |= 1ull << b;
                }
            }
            {
                unsigned long long w = mine;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) w |= (unsigned long long)__shfl_xor((long long)w, o, kWave);
                if ((tid & (kWave - 1)) == 0 && w) atomicOr(&s_mask, w);
            }
            __syncthreads();
            unsigned long long todo = s_mask;
            while (todo) {
                const int b = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int tile0 = hpr_tile_of(step0 + b, own) * kHprThreads;
                const int tn = min(kHprThreads, n - tile0);
                {
                    double4 q = make_double4(__builtin_nan(""), 0.0, 0.0, 0.0);
                    if (tid < tn) {
                        const double *g = fl + (size_t)(tile0 + tid) * 3;
                        q = make_double4(g[0], g[1], g[2], 0.0);
                    }
                    s_stage[tid] = q;
                }
                __syncthreads();
                if (trying && ((mine >> b) & 1ull)) {
                    bool bad = false;
#pragma unroll 8
                    for (int t = 0; t < kHprThreads; t++) {
                        const double4 q = s_stage[t];          // NaN rows: the comparison is false
                        const double sv = (nx * q.x + ny * q.y + nz * q.z) - f.rho;
                        const bool self = q.x == f.px && q.y == f.py && q.z == f.pz;
                        bad |= !self && sv > -thr;
                    }
                    if (bad) trying = false;
                }
                __syncthreads();
            }
        }
        if (trying) {              // no counter-example anywhere: visible
            active = false;        // (nv >= 3 stays: reported as visible below)
        }
    }
    // Split form: every other tile (outward from the group's starting tile) is the wave-per-point pass's: only about
    // one listed point in eight is still undecided here, and a block that carried them on waited for its slowest
    // lane with the other seven eighths idle (1.5 of a block's 2.7 ms).  The undecided lanes park
    // their polygons in memory; hpr_overflow_kernel continues from them, one wave each.
    [[maybe_unused]] const unsigned long long prof_t2 = HPR_PROF_CLOCK();
    HPR_PROF_ADD(1, prof_t2 - prof_t1);            // verify
    if (split && active) {
        const unsigned long long bal = __ballot(true);
        const int lane = tid & (kWave - 1);
        int base = 0;
        if (lane == __ffsll((long long)bal) - 1) {
            const int sx = hpr_seg_of(segs, view);
            base = segs.base[sx] + atomicAdd(&status[32 + sx], __popcll(bal));
            atomicAdd(&status[2], min(__popcll(bal), max(segs.cap - base, 0)));
            if (und) atomicAdd(&und[view], __popcll(bal));      // (a wave's lanes share the view: one block = one view)
        }
        base = __shfl(base, __ffsll((long long)bal) - 1, kWave);
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (slot >= segs.cap) {          // the polygon store is full: listed instead
            over_list[atomicAdd(&status[0], 1)] = view * n + rank;
        } else {
            surv[slot] = make_int4(view * n + rank, nv, pos, home_chunks < kHprThreads / 32 ? home_chunks * 32 : home_tiles * 256);
            double2 *sp = surv_poly + (size_t)slot * kHprMaxV;
            for (int k = 0; k < nv; k++) sp[k] = poly[k * kHprThreads];
        }
        nv = -1;               // decided later
        active = false;
    }
    // split == 0 (large clouds: few views, thousands of tiles -- there the lanes of a block want the same tiles and
    // staging them once per block through LDS beats per-lane reads: 2 x 165546 points 40 ms against 49):
    // Phase 2: all other tiles, outward from the group's starting tile, in batches of 1, 1, 2, 4, ... 64 whose
    // records are tested first (the polygon is tight by now, most tiles are out of its reach)
    for (int step0 = 0, bsz = 1; step0 < 2 * ntiles; step0 += bsz, bsz = step0 < kHprBatch ? step0 : kHprBatch) {
        // A polygon that still runs out to the box after the eight nearest tiles belongs to a point on the
        // silhouette: it is cut by points all along the rim and holds its whole wave up for as many tiles as
        // the cloud has -- in a large cloud such points go to the second pass, which puts a wave on each
        // (measured: 2 x 165546 points 77 -> 51 ms; 64 x 10000 points 29 -> 32 ms, hence the size rule).
        if (step0 == kHprRimStep && ntiles >= kHprRimTiles && active && R.d2 > kHprRimD2 && !(no_cull & 8)) {
            over_list[atomicAdd(&status[0], 1)] = view * n + rank;
                            if (und) atomicAdd(&und[view], 1);
            nv = -1;
            active = false;
        }
        const int nactive = __syncthreads_count(active);
        if (nactive == 0) break;
        // After the first few batches of tiles the points a block still carries restart in the wave-per-point pass, one
        // wave each: a block that walks on for some of its 128 points holds two waves (and their LDS) for every tile any
        // of them needs -- block times at 2 x 165546: median 3.1 M ticks, slowest 23 M = the launch -- while the
        // wave-per-point pass tests 64 candidates against one polygon at a time and clips with all lanes.  Measured
        // (hand-off from batch 4, all remaining points): 2 x 165546 21.8 -> 14.8 ms (radius 100: 30.6 -> 16.2), 64 x 10000
        // 12.1 -> 7.3 (15.7 -> 7.3); from batch 16: 9.4, from 64: 12.6.  (The split form for many views of a small cloud
        // keeps its own second kernel: 1024 x 10000 65 / 39.6 / 43.9 ms against 74 / 36.9 / 42.7 this way.)
        if (step0 >= straggle_from && nactive <= straggle_lanes && !(no_cull & 8)) {
            if (active) {
                over_list[atomicAdd(&status[0], 1)] = view * n + rank;
                if (und) atomicAdd(&und[view], 1);
                nv = -1;
                active = false;
            }
            break;
        }
        // which of the next tiles can still cut somebody's polygon (tested against the polygon as it is now:
        // it only shrinks, so a tile found out of reach stays out of reach)
        if (tid < bsz) {
            const int tile = hpr_tile_of(step0 + tid, own);
            if (tile >= 0 && tile < ntiles) s_rec[tid] = tiles[tile];
        }
        if (tid == 0) s_mask = 0ull;
        __syncthreads();
        unsigned long long mine = 0ull;
        if (active) {
            R = hpr_reach(poly, kHprThreads, nv);
            const double l = sqrt(1.0 + R.d2) * (1.0 + 1e-15), cpsi = 1.0 / l, spsi = sqrt(R.d2) / l * (1.0 + 1e-15);
            for (int b = 0; b < bsz; b++) {
                const int tile = hpr_tile_of(step0 + b, own);
                if (tile < 0 || tile >= ntiles) continue;
                if (tile >= home - 1 && tile <= home + 1) continue;          // taken in phase 1
                if ((no_cull & 1) || hpr_tile_needed(f, cpsi, spsi, poly, kHprThreads, nv, s_rec[b])) mine |= 1ull << b;
            }
        }
        {
            unsigned long long w = mine;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) w |= (unsigned long long)__shfl_xor((long long)w, o, kWave);
            if ((tid & (kWave - 1)) == 0 && w) atomicOr(&s_mask, w);
        }
        __syncthreads();
        unsigned long long todo = s_mask;
        while (todo) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            take_tile(hpr_tile_of(step0 + b, own), ((mine >> b) & 1ull) != 0ull);
        }
    }
    HPR_PROF_ADD(2, HPR_PROF_CLOCK() - prof_t2);    // park / walk
    HPR_PROF_FLUSH();
    const bool seen = i >= 0 && nv > 0;
    if (i >= 0 && nv >= 0) vis[(size_t)view * n + i] = seen ? 1 : 0;
    const int c = __syncthreads_count(seen);
    if (tid == 0 && c) atomicAdd(&cnt[view], c);
}

// viewpoint_select only needs the view that sees the MOST points.  After hpr_kernel a view's count is a lower bound
// (accepted + verified + decided in phase 1) and `und` says how many of its points are still undecided: a view whose
// upper bound stays below the best lower bound cannot win (nor tie), and the second kernel and the wave-per-point
// pass skip its points.  alive[v] = 1: the view's final count is exact.
__global__ __launch_bounds__(1024) void hpr_prune_kernel(int c, const int *__restrict__ cnt, const int *__restrict__ und,
                                                        unsigned char *__restrict__ alive)
{
    __shared__ int s_max[16];
    int m = 0;
    for (int v = threadIdx.x; v < c; v += 1024) m = max(m, cnt[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, kWave));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_max[0];
    for (int w = 1; w < 16; w++) m = max(m, s_max[w]);
    for (int v = threadIdx.x; v < c; v += 1024) alive[v] = cnt[v] + und[v] >= m ? 1 : 0;
}


// second pass: one WAVE per listed point, polygon in LDS (two buffers of kHprOverCap vertices).  Lane b tests
// tile b of a batch of 64; in a tile that can reach the polygon the 64 lanes test 64 candidates at once, and
// the few that are not provably out of reach are taken one by one, in order, the vertex test spread over the
// lanes and the clip itself done by lane 0.  Same candidates, same order, same arithmetic as the first pass
// (and the oracle).
// hpr_clip and hpr_reach by all 64 lanes of a wave (the wave-per-point pass: its heaviest points clip 1000-2000 times
// polygons of a hundred vertices; with one lane clipping and every lane walking the whole polygon for the reach, a
// clip cost 36 k ticks and one point set the launch's 24 ms at 2 x 165546).  Every output vertex is computed by the
// expressions of hpr_clip from the same operands, the maxima of hpr_reach are exact: the polygons are the same bits.
__device__ __forceinline__ int hpr_clip_wave(const double2 *src, int nv, double A, double B, double C, double2 *dst, int lane)
{
    int base = 0;
    for (int k0 = 0; k0 < nv; k0 += kWave) {
        const int k = k0 + lane;
        bool in0 = false, cross = false;
        double2 cur = make_double2(0.0, 0.0), x = cur;
        if (k < nv) {
            const int k2 = k + 1 < nv ? k + 1 : 0;
            cur = src[k];
            const double2 nxt = src[k2];
            const double s0 = cur.x * A + cur.y * B - C;
            const double s1 = nxt.x * A + nxt.y * B - C;
            in0 = !(s0 > 0.0);
            cross = (s0 > 0.0) != (s1 > 0.0) && s0 != 0.0 && s1 != 0.0;
            if (cross) {
                const double t = s0 / (s0 - s1);
                x.x = cur.x + t * (nxt.x - cur.x);
                x.y = cur.y + t * (nxt.y - cur.y);
            }
        }
        const unsigned long long bi = __ballot(in0), bc = __ballot(cross);
        const unsigned long long lt = (1ull << lane) - 1ull;
        int pos = base + __popcll(bi & lt) + __popcll(bc & lt);
        if (in0) dst[pos++] = cur;
        if (cross) dst[pos] = x;
        base += __popcll(bi) + __popcll(bc);
    }
    return base;
}

__device__ __forceinline__ HprReach hpr_reach_wave(const double2 *p, int nv, int lane)
{
    if (nv <= 48) return hpr_reach(p, 1, nv);          // (every lane the same loop: broadcast reads)
    double m[9] = {0.0, -__builtin_inf(), -__builtin_inf(), -__builtin_inf(), -__builtin_inf(), -__builtin_inf(), -__builtin_inf(),
                   -__builtin_inf(), -__builtin_inf()};
    for (int k = lane; k < nv; k += kWave) {
        const double2 v = p[k];
        const double r2 = v.x * v.x + v.y * v.y, s = v.x + v.y, t = v.x - v.y;
        m[0] = r2 > m[0] ? r2 : m[0];
        m[1] = v.x > m[1] ? v.x : m[1];   m[2] = -v.x > m[2] ? -v.x : m[2];
        m[3] = v.y > m[3] ? v.y : m[3];   m[4] = -v.y > m[4] ? -v.y : m[4];
        m[5] = s > m[5] ? s : m[5];       m[8] = -s > m[8] ? -s : m[8];
        m[6] = t > m[6] ? t : m[6];       m[7] = -t > m[7] ? -t : m[7];
    }
#pragma unroll
    for (int q = 0; q < 9; q++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(m[q], o, kWave);
            m[q] = other > m[q] ? other : m[q];
        }
    }
    HprReach r;
    r.d2 = m[0]; r.xp = m[1]; r.xn = m[2]; r.yp = m[3]; r.yn = m[4]; r.pp = m[5]; r.pn = m[6]; r.np = m[7]; r.nn = m[8];
    return r;
}

// ---- decisions without the walk (wave-per-point pass) ----
// The points that reach the wave-per-point pass have met their nearest neighbours (home tiles); what is left of their
// polygons P is decided by very few of the remaining candidates, but the walk below finds those only by clipping with
// everything that still touches the polygon on the way (measured on 1024 x 10000: 50 - 90 clips for a point that ends
// hidden, every tile visited for one that ends visible: 19 k instructions per point).  So first, on a SCRATCH copy of P:
// take an interior point c, find the candidate whose constraint c violates MOST (one dot product per candidate, tiles
// culled for the single normal) -- none: c is strictly feasible with the verify margin, the point is visible -- else test
// whether that constraint (alone, or with one found earlier) excludes all of P, else clip the scratch polygon by it and
// repeat.  Both decisions are proofs about P, which is the exact path's own state, so they are what the walk would end
// with (below); anything undecided after kHprLpIters rounds takes the walk from P as before.  Measured on the CPU
// (tools/hpr_lp_study.py, scan / blob views): 99.7 % of the points decided, 1.2 - 1.4 rounds on average.
//   visible: the verify phase's argument (hpr_kernel), unchanged.
//   hidden: lam s1 + (1 - lam) s2 > m on every vertex of P for some lam in [0, 1] (s_i = a A_i + b B_i - C_i), hence on all
//     of P (affine).  Every vertex w the walk can still hold once it has taken both candidates satisfies s_i(w) <= eta, eta
//     = the walk's own rounding: a kept vertex has fl(s_i) <= 0, a crossing is computed on an edge inside P with a position
//     error of ~1e-16 of the edge's ends, later clips only take convex combinations of vertices that carry such errors
//     already; eta <= (12 + 3 G) u M after G clips, with M = the largest |a A_i| + |b B_i| + |C_i| over P's vertices (the
//     ends of those edges) and G <= n.  m = (1e-9 + 4e-15 n) M (at least ten times eta for any n, 10^4 times for the
//     tens of clips a walk really makes) therefore leaves the walk no vertex at all: it ends with the polygon empty,
//     whatever order it takes the candidates in and whichever of them it skips (it skips only candidates that cut nothing:
//     s_i <= 0 on all its vertices already).
constexpr int kHprLpIters = 4;
constexpr int kHprLpMaxV = 32;         // vertices of P (one per lane in the certificates); scratch polygons: 64

// an interior point of a convex polygon (every lane the same loop): the centroid; for a polygon that runs out to the
// box, a point near its bounded end (as the verify phase of hpr_kernel)
__device__ __forceinline__ bool hpr_interior(const double2 *src, int nv, double2 &c)
{
    double2 ctr = make_double2(0.0, 0.0), v0 = ctr, v1 = ctr;
    double r0 = __builtin_inf(), r1 = __builtin_inf();
    for (int k = 0; k < nv; k++) {
        const double2 v = src[k];
        const double r2 = v.x * v.x + v.y * v.y;
        ctr.x += v.x;
        ctr.y += v.y;
        if (r2 < r0) { r1 = r0; v1 = v0; r0 = r2; v0 = v; }
        else if (r2 < r1) { r1 = r2; v1 = v; }
    }
    ctr.x /= (double)nv;
    ctr.y /= (double)nv;
    if (!(ctr.x * ctr.x + ctr.y * ctr.y < 1.0e6)) {
        const double dx = ctr.x - v0.x, dy = ctr.y - v0.y;
        const double len = sqrt(dx * dx + dy * dy);
        double h = sqrt((v1.x - v0.x) * (v1.x - v0.x) + (v1.y - v0.y) * (v1.y - v0.y));
        h = h < 0.5 * len ? h : 0.5 * len;
        if (!(len > 0.0 && h > 0.0)) return false;
        ctr.x = v0.x + h * (dx / len);
        ctr.y = v0.y + h * (dy / len);
    }
    c = ctr;
    return ctr.x == ctr.x && ctr.y == ctr.y;
}

__device__ __forceinline__ double hpr_wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(v, o, kWave);
        v = other > v ? other : v;
    }
    return v;
}

// does the constraint (A, B, C) alone, or some mix of it with (A1, B1, C1), exclude every vertex of P by the margin?
// (lane k holds vertex k; nv <= kHprLpMaxV)
__device__ __forceinline__ bool hpr_excluded(const double2 *P, int nv, bool pair, double A1, double B1, double C1, double A, double B,
                                             double C, int lane, double margin)
{
    const bool mine = lane < nv;
    const double2 v = mine ? P[lane] : make_double2(0.0, 0.0);
    const double s2 = v.x * A + v.y * B - C, M2 = fabs(v.x * A) + fabs(v.y * B) + fabs(C);
    const double s1 = pair ? v.x * A1 + v.y * B1 - C1 : 0.0, M1 = pair ? fabs(v.x * A1) + fabs(v.y * B1) + fabs(C1) : 0.0;
    const double m = margin * hpr_wave_max(mine ? (M1 > M2 ? M1 : M2) : 0.0);
    if (!(m < __builtin_inf())) return false;
    if (__ballot(mine && !(s2 > m)) == 0ull) return true;            // the new constraint alone
    if (!pair) return false;
    // lam s1 + (1 - lam) s2 > m  <=>  s2 + lam (s1 - s2) > m: an interval of lam per vertex
    double lo = 0.0, hi = 1.0;
    if (mine) {
        const double d = s1 - s2;
        if (d > 0.0) lo = (m - s2) / d;
        else if (d < 0.0) hi = (m - s2) / d;
        else if (!(s2 > m)) lo = 2.0;
    }
    lo = hpr_wave_max(lo);
    hi = -hpr_wave_max(-hi);
    lo = lo > 0.0 ? lo : 0.0;
    hi = hi < 1.0 ? hi : 1.0;
    if (!(lo < hi)) return false;
    const double lam = 0.5 * (lo + hi);
    const double g = lam * s1 + (1.0 - lam) * s2;          // the certificate itself, evaluated as such
    return __ballot(mine && !(g > m)) == 0ull;
}

// One point's decision rounds (see above): P = its polygon as the exact path left it (nv <= kHprLpMaxV vertices), scratch =
// room for two polygons of 64 vertices.  -> 0 undecided, 1 visible, 2 hidden.
__device__ __forceinline__ int hpr_decide(const HprFrame &f, const double2 *P, int nv, double2 *scratch, const double *__restrict__ fl,
                                          int n, int ntiles, const HprTile *__restrict__ tiles, int home, int pos, int no_cull, int lane)
{
    // the hidden proof's margin, relative to M (above): the walk's rounding grows with the clips behind a vertex -- a vertex
    // made by the G-th clip sits on an edge whose ends carry (12 + 3 (G - 1)) u M already -- and G <= n
    const double margin = 1e-9 + 4e-15 * (double)n;
    const double2 *poly = P;
    int pn = nv, nprev = 0;
    double pA[kHprLpIters], pB[kHprLpIters], pC[kHprLpIters];
    for (int it = 0; it < kHprLpIters; it++) {
        double2 c;
        if (!hpr_interior(poly, pn, c)) break;
        const double nx = f.ux + c.x * f.e1x + c.y * f.e2x, ny = f.uy + c.x * f.e1y + c.y * f.e2y,
                     nz = f.uz + c.x * f.e1z + c.y * f.e2z;
        const double nn = 1.0 + c.x * c.x + c.y * c.y;
        const double thr = 1e-10 * f.rho * nn;
        const double cl = sqrt(nn) * (1.0 + 1e-15), cpsi1 = 1.0 / cl, spsi1 = sqrt(nn - 1.0) / cl * (1.0 + 1e-15);
        // the candidate c violates most (distance to its line); key < 0: none comes within the margin
        double key = -1.0, kA = 0.0, kB = 0.0, kC = 0.0;
        int kj = INT_MAX;
        // Tiles outward from the point's own: what hides a point is near it in direction, and the first candidate found
        // to exclude all of P ends the scan (two thirds of the parked points end hidden) -- so the own tile and its two
        // neighbours are taken before any tile record is looked at.  Two tiles at a time (twelve loads in flight: this
        // loop waits on memory).  A lane tests only its most violated candidate so far for exclusion, once per pair.
        auto scan_pair = [&](int ta, int tb) -> bool {
            constexpr int kPer = kHprThreads / kWave;
            double q[2 * kPer][3];
            int jj[2 * kPer];
#pragma unroll
            for (int e = 0; e < 2 * kPer; e++) {
                const int tile = e < kPer ? ta : tb;
                const int j = tile * kHprThreads + (e % kPer) * kWave + lane;
                jj[e] = (tile >= 0 && j < n) ? j : -1;
                const int jl = jj[e] >= 0 ? j : pos;           // (a harmless address: the point itself)
                q[e][0] = fl[(size_t)jl * 3 + 0];
                q[e][1] = fl[(size_t)jl * 3 + 1];
                q[e][2] = fl[(size_t)jl * 3 + 2];
            }
            bool fresh = false;
#pragma unroll
            for (int e = 0; e < 2 * kPer; e++) {
                const double qx = q[e][0], qy = q[e][1], qz = q[e][2];
                // (a sieve with the verify margin behind it: fused multiply-adds, and the point itself recognised by its
                // position -- what passes is valued below with the walk's own operations)
                const double sv = __builtin_fma(nz, qz, __builtin_fma(ny, qy, nx * qx)) - f.rho;
                if (jj[e] >= 0 && jj[e] != pos && sv > -thr) {           // (NaN rows: false)
                    const double A = f.e1x * qx + f.e1y * qy + f.e1z * qz;
                    const double B = f.e2x * qx + f.e2y * qy + f.e2z * qz;
                    const double l2 = A * A + B * B;
                    const double k2 = sv > 0.0 ? (l2 > 0.0 ? sv * sv / l2 : __builtin_inf()) : 0.0;
                    if (k2 > key || (k2 == key && jj[e] < kj)) {
                        key = k2; kj = jj[e]; kA = A; kB = B;
                        kC = f.rho - (f.ux * qx + f.uy * qy + f.uz * qz);
                        fresh = true;
                    }
                }
            }
            // does the lane's candidate alone exclude all of P (hpr_excluded's test, a lane per candidate)?
            bool excl = false;
            if (fresh) {
                double smin = __builtin_inf(), mmax = 0.0;
                for (int k = 0; k < nv; k++) {
                    const double2 v = P[k];
                    const double sk = v.x * kA + v.y * kB - kC, mk = fabs(v.x * kA) + fabs(v.y * kB) + fabs(kC);
                    smin = sk < smin ? sk : smin;
                    mmax = mk > mmax ? mk : mmax;
                }
                excl = mmax < __builtin_inf() && smin > margin * mmax;
            }
            return __ballot(excl) != 0ull;
        };
        for (int ph = 0, t0 = 0; t0 < 2 * ntiles;) {
            const int tl = hpr_tile_of(t0 + lane, home);
            bool need = tl >= 0 && tl < ntiles;
            if (ph == 0) need = need && lane < 3;          // the own tile and its neighbours: no questions asked
            else need = need && t0 + lane >= 3 && ((no_cull & 1) || hpr_tile_needed(f, cpsi1, spsi1, &c, 0, 1, tiles[tl]));
            unsigned long long todo = __ballot(need);
            while (todo) {
                const int b0 = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                int b1 = -1;
                if (todo) {
                    b1 = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                }
                if (scan_pair(hpr_tile_of(t0 + b0, home), b1 >= 0 ? hpr_tile_of(t0 + b1, home) : -1)) return 2;
            }
            if (ph == 0) ph = 1;
            else t0 += kWave;
        }
        double bkey = key;
        int bj = kj;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ok = __shfl_xor(bkey, o, kWave);
            const int oj = __shfl_xor(bj, o, kWave);
            if (ok > bkey || (ok == bkey && oj < bj)) { bkey = ok; bj = oj; }
        }
        if (bkey < 0.0) return 1;                 // c is strictly feasible: visible
        const int owner = __ffsll((long long)__ballot(kj == bj && key == bkey)) - 1;
        const double A = __shfl(kA, owner, kWave), B = __shfl(kB, owner, kWave), C = __shfl(kC, owner, kWave);
        bool hidden = hpr_excluded(P, nv, false, 0.0, 0.0, 0.0, A, B, C, lane, margin);
#pragma unroll
        for (int q = kHprLpIters - 2; q >= 0; q--)
            if (q < nprev && !hidden) hidden = hpr_excluded(P, nv, true, pA[q], pB[q], pC[q], A, B, C, lane, margin);
        if (hidden) return 2;
        if (it + 1 == kHprLpIters) break;
        double2 *dst = scratch + (it & 1) * 64;
        const int m = hpr_clip_wave(poly, pn, A, B, C, dst, lane);
        __syncthreads();
        if (m < 3 || m > 62) break;
        poly = dst;
        pn = m;
#pragma unroll
        for (int q = 0; q < kHprLpIters - 1; q++)
            if (q == nprev) { pA[q] = A; pB[q] = B; pC[q] = C; }
        nprev++;
    }
    return 0;
}

// The parked points' first try in a kernel of its own: the rounds above wait on memory (a wave lives ~20 us, most of it
// in a dozen dependent round trips), so what matters is how many waves a CU holds -- this kernel needs a third of the
// registers of hpr_overflow_kernel, whose walk and clip code the 98 % of the points decided here never run.  Undecided
// points (and the ones that still have home tiles to take after the try) are listed for hpr_overflow_kernel by slot.
__device__ __forceinline__ void hpr_decide_item(int slot, int n, const double *__restrict__ fl_all, const HprTile *__restrict__ tiles_all,
                                                unsigned char *__restrict__ vis, int *__restrict__ cnt, int *status,
                                                const int *__restrict__ perm, int no_cull, const unsigned char *__restrict__ alive,
                                                const int4 *__restrict__ surv, const double2 *__restrict__ surv_poly,
                                                int *__restrict__ slots, const HprSegs &segs, double2 *s_p, double2 *s_scratch)
{
    const int lane = threadIdx.x;
    const int4 rec = surv[slot];
    const int view = rec.x / n, nv = rec.y, pos = rec.z;
    if (alive && !alive[view]) return;        // (wave-uniform) a view that cannot be the best any more
    const int i = perm[(size_t)view * n + pos];
    const double *fl = fl_all + (size_t)view * n * 3;
    const int ntiles = ceil_div_dev(n, kHprThreads);
    HprFrame f;
    if (!hpr_frame(fl + (size_t)pos * 3, f)) return;
    if (lane < nv) s_p[lane] = surv_poly[(size_t)slot * kHprMaxV + lane];
    __syncthreads();
    int d = 0;
    if (nv >= 3 && !(no_cull & (32 | 128)))
        d = hpr_decide(f, s_p, nv, s_scratch, fl, n, ntiles, tiles_all + (size_t)view * ntiles, pos / kHprThreads, pos, no_cull, lane);
    if (lane == 0) {
        if (d) {
            vis[(size_t)view * n + i] = d == 1 ? 1 : 0;
            if (d == 1) atomicAdd(&cnt[view], 1);
        } else {
            slots[atomicAdd(&status[5], 1)] = slot;
        }
    }
}

// The launch does not know how many points were parked (the count never leaves the device): a fixed grid of one-wave
// blocks, a multiple of 8 of them, strides over the items -- block b stays with segment b % 8, the XCD it runs on.
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(3, 8))) void hpr_decide_kernel(int n, const double *__restrict__ fl_all, const HprTile *__restrict__ tiles_all,
                                                          unsigned char *__restrict__ vis, int *__restrict__ cnt, int *status,
                                                          const int *__restrict__ perm, int no_cull, const unsigned char *__restrict__ alive,
                                                          const int4 *__restrict__ surv, const double2 *__restrict__ surv_poly,
                                                          int *__restrict__ slots, const HprSegs *__restrict__ segs_p, int *work)
{
    __shared__ double2 s_p[kHprMaxV + 6];
    __shared__ double2 s_scratch[128];
    const HprSegs segs = *segs_p;
    // this block's segment, its first slot and its fill (wave-uniform: the loop stays scalar)
    const int x = segs.on ? (int)(blockIdx.x & 7) : 0, stride = segs.on ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int first = segs.base[x];
    const int fill = __builtin_amdgcn_readfirstlane(min(status[32 + x], segs.cap - first));
    // The first chunk of kHprDecideChunk consecutive points is the block's own; the following ones are drawn from the
    // segment's counter in order of arrival, so that the waves in flight work on a narrow window of the segment -- a few
    // views' flipped points and tile records in that XCD's L2, as when every item was a block of its own and the dispatcher
    // handed them out in order.  (A fixed stride lets the waves drift apart over hundreds of views: 9.2 ms against 6.7 at
    // 1024 x 10000; a draw per POINT from eight counters in one cache line serialised the launch: 18.8 ms -- the counters
    // have a 128-byte line each and a draw covers four points.)
    int k0 = (segs.on ? (int)(blockIdx.x >> 3) : (int)blockIdx.x) * kHprDecideChunk;
    while (k0 < fill) {
        int nxt = 0;
        if (threadIdx.x == 0) nxt = atomicAdd(&work[x * kHprWorkLine], 1);          // (issued now, read after the chunk)
        for (int k = k0; k < min(k0 + kHprDecideChunk, fill); k++) {
            hpr_decide_item(first + k, n, fl_all, tiles_all, vis, cnt, status, perm, no_cull, alive, surv, surv_poly, slots, segs, s_p, s_scratch);
            __syncthreads();
        }
        k0 = (stride + __builtin_amdgcn_readfirstlane(nxt)) * kHprDecideChunk;
    }
}

// CAP = vertices a polygon may reach: the pass runs with 128 first (4 KiB of LDS per wave: forty waves per CU instead
// of the five that two 16 KiB buffers allow) and hands the few polygons that outgrow that to a second launch with
// kHprOverCap (list2, counted in status[3]); results do not depend on the tier.
template <int CAP>
__device__ __forceinline__ void hpr_overflow_item(int item, int n, const double *__restrict__ fl_all,
                                                  const HprTile *__restrict__ tiles_all, unsigned char *__restrict__ vis,
                                                  int *__restrict__ cnt, int *status, const int *__restrict__ list,
                                                  const int *__restrict__ perm, const int *__restrict__ hardlist,
                                                  const int *__restrict__ hardcnt, int no_cull,
                                                  const unsigned char *__restrict__ alive, int *__restrict__ list2,
                                                  const int4 *__restrict__ surv, const double2 *__restrict__ surv_poly,
                                                  double2 *__restrict__ gbuf, int gcap, const int *__restrict__ slots, const HprSegs &segs,
                                                  double2 (*s_lds)[CAP > 0 ? CAP : 1])
{
    // CAP == 0: the polygon lives in global memory, gcap vertices per buffer (n + 8: a polygon has at most one edge per
    // other point and four of the box -- this tier cannot overflow; round 3 returned an error beyond 1024 vertices)
    double2 *s_buf[2];
    s_buf[0] = CAP > 0 ? s_lds[0] : gbuf + (size_t)blockIdx.x * 2 * gcap;          // (a pair of buffers per BLOCK)
    s_buf[1] = CAP > 0 ? s_lds[1] : gbuf + ((size_t)blockIdx.x * 2 + 1) * gcap;
    const int cap = CAP > 0 ? CAP : gcap;
    const int lane = threadIdx.x;
    // surv != null: the points the split first kernel left undecided continue HERE from their saved polygons (home tiles and
    // verification are behind them), one wave each
    const bool cont = surv != nullptr;
    // (a parked point's record carries its position: one dependent load less in front of the first useful one)
    // (slots: the parked points hpr_decide_kernel left undecided -- their first try is behind them)
    const int slot = !cont ? 0 : (slots ? slots[item] : hpr_seg_slot(segs, status, item));
    if (slot < 0) return;
    const int4 rec = cont ? surv[slot] : make_int4(list[item], 4, -1, 0);
    const int id = rec.x, view = id / n, rank = id - view * n;
    if (alive && !alive[view]) return;        // (wave-uniform) a view that cannot be the best any more
    const int *hl = hardlist + (size_t)view * n;
    const int pos = cont ? rec.z : hl[rank], i = perm[(size_t)view * n + pos];
    const double *fl = fl_all + (size_t)view * n * 3;
    const int ntiles = ceil_div_dev(n, kHprThreads);
    const HprTile *tiles = tiles_all + (size_t)view * ntiles;
    HprFrame f;
    if (!hpr_frame(fl + (size_t)pos * 3, f)) return;      // (cannot happen: the first pass listed it)
    int cur = 0, nv = 4;
    // home tiles the exact path has behind it (0 .. 3) and candidates of the next one already taken (hpr_kernel's park)
    const int rel0 = cont ? rec.w >> 8 : 0, skip0 = cont ? rec.w & 255 : 0;
    if (cont) {
        nv = rec.y;
        if (lane < nv) s_buf[0][lane] = surv_poly[(size_t)slot * kHprMaxV + lane];
    } else if (lane == 0) {
        s_buf[0][0] = make_double2(-kHprBox, -kHprBox);
        s_buf[0][1] = make_double2(kHprBox, -kHprBox);
        s_buf[0][2] = make_double2(kHprBox, kHprBox);
        s_buf[0][3] = make_double2(-kHprBox, kHprBox);
    }
    __syncthreads();
    HprReach R = hpr_reach(s_buf[0], 1, nv);
    bool failed = false;
    const int home = pos / kHprThreads;
    // one tile of candidates against the polygon (32 candidates at a time)
    auto take_tile = [&](int tile, int skip) {
        const int tile0 = tile * kHprThreads;
            // the first pass's order: 32-candidate chunks, in the home tile starting with the point's own
            // (any other tile is taken in ascending order: 64 candidates at a time there)
            const bool is_home = tile == home;
            const int rot = is_home ? ((pos - home * kHprThreads) >> 5) : 0;
            const int width = is_home ? 32 : kWave;
            for (int cc = 0; cc < kHprThreads / width && nv > 0; cc++) {
                if ((cc + 1) * width <= skip) continue;              // (taken before the point was parked)
                const int j = tile0 + (is_home ? ((cc + rot) & (kHprThreads / 32 - 1)) * 32 : cc * kWave) + lane;
                double A = 0.0, B = 0.0, C = 0.0;
                bool pass = false;
                if (lane < width && cc * width + lane >= skip && j < n) {
                    const double qx = fl[(size_t)j * 3 + 0], qy = fl[(size_t)j * 3 + 1], qz = fl[(size_t)j * 3 + 2];
                    A = f.e1x * qx + f.e1y * qy + f.e1z * qz;
                    B = f.e2x * qx + f.e2y * qy + f.e2z * qz;
                    C = f.rho - (f.ux * qx + f.uy * qy + f.uz * qz);
                    const bool self = qx == f.px && qy == f.py && qz == f.pz;
                    pass = !self && qx == qx && !hpr_far(R, A, B, C);
                }
                // every lane tests ITS candidate against the whole polygon (broadcast reads); only the ones
                // that cut anything are then taken one by one (a candidate that does not cut the polygon now
                // cannot cut what is left of it later) -- the tail of this kernel is a point with 20000
                // candidates in reach of its polygon of which 1400 cut: 29 ms when each was taken serially
                if (pass) {
                    const double2 *pv = s_buf[cur];
                    bool cuts = false;
                    for (int k = 0; k < nv; k++) cuts |= pv[k].x * A + pv[k].y * B - C > 0.0;
                    pass = cuts;
                }
                unsigned long long mask = __ballot(pass);
                bool clipped = false;
                while (mask && nv > 0) {
                    const int from = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const double Aj = __shfl(A, from, kWave), Bj = __shfl(B, from, kWave), Cj = __shfl(C, from, kWave);
                    const double2 *src = s_buf[cur];
                    int out = 0;
                    for (int k = lane; k < nv; k += kWave) out += (src[k].x * Aj + src[k].y * Bj - Cj > 0.0) ? 1 : 0;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) out += __shfl_xor(out, o, kWave);
                    if (!out) continue;
                    if (out == nv) { nv = 0; break; }
                    if (nv - out + 2 > cap) {
                        if (lane == 0) {
                            if (CAP > 0 && CAP < kHprOverCap) list2[atomicAdd(&status[3], 1)] = id;      // the large-polygon launch takes it
                            else if (CAP == kHprOverCap) list2[atomicAdd(&status[4], 1)] = id;          // the global-memory tier takes it
                            else atomicExch(&status[1], 2);                                            // (cannot happen: gcap = n + 8)
                        }
                        failed = true;
                        nv = 0;
                        break;
                    }
                    const int m = hpr_clip_wave(src, nv, Aj, Bj, Cj, s_buf[cur ^ 1], lane);
                    __syncthreads();
                    if (m < 3) { nv = 0; break; }
                    nv = m;
                    cur ^= 1;
                    clipped = true;
                }
                // (the reach bounds only feed conservative rejections, and a clip only shrinks the polygon: refreshed once
                // per 64 candidates instead of after every clip -- half the instructions of a clip; the slowest walker,
                // 1400 clips, is what the launch waits for)
                if (clipped && nv > 0) R = hpr_reach_wave(s_buf[cur], nv, lane);
            }
    };
    bool lp_ran = false;
    // -> 0 undecided, 1 visible, 2 hidden
    auto lp_try = [&]() -> int {
        if (!(CAP >= 128 && nv >= 3 && nv <= kHprLpMaxV && !(no_cull & (32 | 128)))) return 0;
        lp_ran = true;
        const int d = hpr_decide(f, s_buf[cur], nv, s_buf[cur ^ 1], fl, n, ntiles, tiles, home, pos, no_cull, lane);
        if (d) return d;
        __syncthreads();          // (the walk reuses the scratch buffer)
        return 0;
    };
    auto decided = [&](int d) {
        if (lane == 0) {
            vis[(size_t)view * n + i] = d == 1 ? 1 : 0;
            if (d == 1) atomicAdd(&cnt[view], 1);
        }
    };
    // the first pass's order: the home tile and its neighbours, then outward from the group's tile.  A parked point is
    // tried before its remaining home tiles are taken (most are decided from the polygon they bring), and again after.
    if (cont && rel0 < 3 && !slots) {
        const int d = lp_try();
        if ((no_cull & 256) && lane == 0) atomicAdd(&status[8 + d], 1);        // measurement: [8] undecided, [9] visible, [10] hidden
        if (d) { decided(d); return; }
    }
    for (int rel = rel0; rel < 3 && nv > 0; rel++) {
        const int tile = home + (rel == 0 ? 0 : (rel == 1 ? 1 : -1));
        if (tile >= 0 && tile < ntiles) take_tile(tile, rel == rel0 ? skip0 : 0);
    }
    if (!(slots && rel0 >= 3)) {
        const int d = lp_try();
        if ((no_cull & 256) && lane == 0) atomicAdd(&status[12 + (nv > 0 ? d : 3)], 1);      // second try: [12] undecided, [13], [14], [15] died in the home tiles
        if (d) { decided(d); return; }
    }
    if (nv >= 3 && !(no_cull & 32) && !cont && !lp_ran) {
        const double2 *src = s_buf[cur];
        double2 ctr = make_double2(0.0, 0.0), v0 = ctr, v1 = ctr;
        double r0 = __builtin_inf(), r1 = __builtin_inf();
        for (int k = 0; k < nv; k++) {
            const double2 v = src[k];
            const double r2 = v.x * v.x + v.y * v.y;
            ctr.x += v.x;
            ctr.y += v.y;
            if (r2 < r0) { r1 = r0; v1 = v0; r0 = r2; v0 = v; }
            else if (r2 < r1) { r1 = r2; v1 = v; }
        }
        ctr.x /= (double)nv;
        ctr.y /= (double)nv;
        const bool open = !(ctr.x * ctr.x + ctr.y * ctr.y < 1.0e6);
        const double dx = ctr.x - v0.x, dy = ctr.y - v0.y;
        const double len = sqrt(dx * dx + dy * dy);
        double h = sqrt((v1.x - v0.x) * (v1.x - v0.x) + (v1.y - v0.y) * (v1.y - v0.y));
        h = h < 0.5 * len ? h : 0.5 * len;
        for (int trial = 0; trial < (open ? 4 : 1); trial++) {
            double2 c = ctr;
            if (open) {
                if (!(len > 0.0 && h > 0.0)) break;
                c.x = v0.x + h * (dx / len);
                c.y = v0.y + h * (dy / len);
                h *= 0.125;
            }
            const double nx = f.ux + c.x * f.e1x + c.y * f.e2x, ny = f.uy + c.x * f.e1y + c.y * f.e2y,
                         nz = f.uz + c.x * f.e1z + c.y * f.e2z;
            const double thr = 1e-10 * f.rho * (1.0 + c.x * c.x + c.y * c.y);
            bool bad = false;
            for (int j0 = 0; j0 < n && !bad; j0 += kWave) {
                const int j = j0 + lane;
                bool b1 = false;
                if (j < n) {
                    const double qx = fl[(size_t)j * 3 + 0], qy = fl[(size_t)j * 3 + 1], qz = fl[(size_t)j * 3 + 2];
                    const double sv = (nx * qx + ny * qy + nz * qz) - f.rho;
                    const bool self = qx == f.px && qy == f.py && qz == f.pz;
                    b1 = !self && sv > -thr;
                }
                bad = __ballot(b1) != 0ull;
            }
            if (!bad) {
                if (lane == 0) {
                    vis[(size_t)view * n + i] = 1;
                    atomicAdd(&cnt[view], 1);
                }
                return;
            }
        }
    }
    const int own = hpr_base_tile(hl, hardcnt[view], rank);
    [[maybe_unused]] const unsigned long long walk_t0 = HPR_PROF_CLOCK();
    [[maybe_unused]] const int walk_nv0 = nv;
    [[maybe_unused]] const double walk_d2 = R.d2;
    for (int step0 = 0, bsz = 1; step0 < 2 * ntiles && nv > 0; step0 += bsz, bsz = step0 < kWave ? step0 : kWave) {
        bool need = false;
        if (lane < bsz) {
            const int tile = hpr_tile_of(step0 + lane, own);
            const double l = sqrt(1.0 + R.d2) * (1.0 + 1e-15), cpsi = 1.0 / l, spsi = sqrt(R.d2) / l * (1.0 + 1e-15);
            if (tile >= 0 && tile < ntiles && !(tile >= home - 1 && tile <= home + 1))
                need = (no_cull & 1) || hpr_tile_needed(f, cpsi, spsi, s_buf[cur], 1, nv, tiles[tile]);
        }
        unsigned long long todo = __ballot(need);
        while (todo && nv > 0) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            take_tile(hpr_tile_of(step0 + b, own), 0);
        }
    }
#ifdef GENPC_HPR_PROF
    if (lane == 0 && walk_nv0 > 0) {
        const unsigned long long dt = HPR_PROF_CLOCK() - walk_t0;
        atomicAdd(&g_hpr_prof[10], 1ull);                       // walkers
        atomicAdd(&g_hpr_prof[11], dt);                         // their ticks
        atomicMax(&g_hpr_prof[12], dt);                         // the slowest
        if (dt > 200000ull) atomicAdd(&g_hpr_prof[13], 1ull);   // walkers over 100 us
        if (walk_d2 > 1.0e6) { atomicAdd(&g_hpr_prof[14], 1ull); atomicAdd(&g_hpr_prof[15], dt); }      // open polygons: count, ticks
        if (dt > 200000ull && nv > 0) atomicAdd(&g_hpr_prof[9], 1ull);      // slow walkers that end visible
    }
#endif
    if (failed) return;
    if (lane == 0) {
        vis[(size_t)view * n + i] = nv > 0 ? 1 : 0;
        if (nv > 0) atomicAdd(&cnt[view], 1);
    }
}

// A fixed grid of one-wave blocks over a count that lives on the device (`count`: a word of status; null: the parked
// points by segment).  A wave's first item is its block index, the following ones it draws from `work` (zero at launch) --
// walkers take from 10 us to over a millisecond, so the items are handed out as waves come free; the draw is issued
// before the item is worked on and read after it.
template <int CAP>
__global__ __launch_bounds__(kWave) void hpr_overflow_kernel(int n, const double *__restrict__ fl_all,
                                                             const HprTile *__restrict__ tiles_all, unsigned char *__restrict__ vis,
                                                             int *__restrict__ cnt, int *status, const int *__restrict__ list,
                                                             const int *__restrict__ perm, const int *__restrict__ hardlist,
                                                             const int *__restrict__ hardcnt, int no_cull,
                                                             const unsigned char *__restrict__ alive, int *__restrict__ list2,
                                                             const int4 *__restrict__ surv, const double2 *__restrict__ surv_poly,
                                                             double2 *__restrict__ gbuf, int gcap, const int *__restrict__ slots,
                                                             const HprSegs *__restrict__ segs_p, const int *count, int *work)
{
    // (work: this pass's counter, a line of its own)
    __shared__ double2 s_lds[2][CAP > 0 ? CAP : 1];
    const HprSegs segs = *segs_p;
    const int items = __builtin_amdgcn_readfirstlane(count ? *count : hpr_seg_items(segs, status));
    int item = blockIdx.x;
    while (item < items) {
        int nxt = 0;
        if (threadIdx.x == 0) nxt = atomicAdd(work, 1);
        hpr_overflow_item<CAP>(item, n, fl_all, tiles_all, vis, cnt, status, list, perm, hardlist, hardcnt, no_cull, alive, list2, surv,
                               surv_poly, gbuf, gcap, slots, segs, s_lds);
        __syncthreads();
        item = (int)gridDim.x + __builtin_amdgcn_readfirstlane(nxt);          // (lane 0's draw)
    }
}

}  // namespace genpc

using namespace genpc;

static int hpr_run(int c, int n, const float *points, const double *eyes, double radius, unsigned char *visible, int *counts,
                   int *second_pass_points, void *stream_, int best_only, unsigned char *exact)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (c < 0 || n < 0 || !(radius > 0.0)) {
        set_error("genpc_hpr_visibility: bad size or radius");
        return -1;
    }
    if (second_pass_points) *second_pass_points = 0;
    if (c == 0) return 1;
    if (!counts || (n > 0 && (!points || !eyes || !visible))) {
        set_error("genpc_hpr_visibility: null pointer");
        return -1;
    }
    if ((long long)c * n > (long long)INT_MAX || c > 4096) {
        set_error("genpc_hpr_visibility: views x points too large (at most 4096 views per call)");
        return -1;
    }
    if (!check(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)c, stream), "hipMemsetAsync(hpr counts)")) return 0;
    if (n == 0) return 1;
    const size_t total = (size_t)c * n;
    size_t sort_bytes = 0;
    if (!check(rocprim::radix_sort_pairs(nullptr, sort_bytes, (const unsigned *)nullptr, (unsigned *)nullptr, (const int *)nullptr, (int *)nullptr,
                                         (size_t)c * n, 0u, 32u, stream),
               "hpr sort size"))
        return 0;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 256;
    const size_t o_fl = off; off += up(total * 3 * sizeof(double));
    const size_t o_list = off; off += up(total * sizeof(int));
    const size_t o_k0 = off; off += up(total * 4);
    const size_t o_k1 = off; off += up(total * 4);
    const size_t o_i0 = off; off += up(total * 4);
    const size_t o_i1 = off; off += up(total * 4);
    const size_t o_tmp = off; off += up(sort_bytes);
    const int ntiles = ceil_div(n, kHprThreads);
    const size_t o_tiles = off; off += up((size_t)c * ntiles * sizeof(HprTile));
    const size_t o_hard = off; off += up(total);
    const size_t o_hl = off; off += up(total * sizeof(int));
    const size_t o_hc = off; off += up((size_t)c * sizeof(int));
    const size_t o_und = off; off += up((size_t)c * sizeof(int));
    const size_t o_alive = off; off += up((size_t)c);
    const size_t o_dup = off; off += up((size_t)n);
    const size_t o_work = off; off += up((size_t)kHprWorkWords * sizeof(int));
    // Split form (the first kernel stops after the home tiles + verification and saves the undecided points' polygons; a
    // wave per point continues from them): for clouds of < 256 tiles.  Round 3, first half: only for >= 4 M (view, point)
    // pairs, with a dense one-thread-per-survivor second kernel (1024 x 10000: 128 -> 88 ms; 64 x 10000 14.3 -> 16.6).  With
    // the wave-per-point continuation: 1024 x 10000 64 / 39 / 44 -> 58.6 / 32 / 37.5 ms (blob / two scans), 64 x 10000
    // 7.3 -> 6.9.  A few views of a large cloud keep the one-kernel form (2 x 165546: 14.9 against 17.8 ms): there the lanes
    // of a block want the same tiles, and staging them once per block through LDS beats per-wave reads
    // (GENPC_HPR_SPLIT=0/1 overrides).
    static const int env_split = tune_env("GENPC_HPR_SPLIT", -1, "hidden-point removal: 1 two-kernel form (first kernel stops after the home tiles, a wave per point continues), 0 one kernel, -1 pick by size");
    const int split = env_split >= 0 ? env_split : (ntiles < kHprRimTiles && (long long)c * n >= 100000ll ? 1 : 0);
    const size_t o_surv = off; off += split ? up(total * sizeof(int4)) : 0;
    // (the survivors' polygons are sized after the accept pass has counted the listed points: a second workspace)
    char *ws = (char *)workspace(17, off, stream);
    if (!ws) return 0;
    int *status = (int *)ws;
    unsigned *bounds = (unsigned *)(ws + 64);
    double *fl = (double *)(ws + o_fl);
    int *list = (int *)(ws + o_list);
    unsigned *k0 = (unsigned *)(ws + o_k0), *k1 = (unsigned *)(ws + o_k1);
    int *i0 = (int *)(ws + o_i0), *i1 = (int *)(ws + o_i1);
    if (!check(hipMemsetAsync(status, 0, 256, stream), "hipMemsetAsync(hpr status)")) return 0;
    if (!check(hipMemsetAsync(bounds, 0xff, 12, stream), "hipMemsetAsync(hpr bounds)")) return 0;
    const int g256 = ceil_div(n, 256);
    // later copies of exact duplicates (the key / index buffers of the view sort are free until then)
    unsigned char *dup = (unsigned char *)(ws + o_dup);
    {
        int *ia = i0, *ib = i1;
        for (int axis = 2; axis >= 0; axis--) {
            hipLaunchKernelGGL(hpr_dupkey_kernel, dim3(g256), dim3(256), 0, stream, n, points, axis, axis == 2 ? (const int *)nullptr : (const int *)ia, k0, ia);
            size_t sb = sort_bytes;
            if (!check(rocprim::radix_sort_pairs(ws + o_tmp, sb, (const unsigned *)k0, k1, (const int *)ia, ib, (size_t)n, 0u, 32u, stream), "hpr duplicate sort"))
                return 0;
            std::swap(ia, ib);
        }
        hipLaunchKernelGGL(hpr_dupmark_kernel, dim3(g256), dim3(256), 0, stream, n, points, (const int *)ia, dup);
    }
    hipLaunchKernelGGL(hpr_bounds_kernel, dim3(g256 < 1024 ? g256 : 1024), dim3(256), 0, stream, n, points, bounds);
    hipLaunchKernelGGL(hpr_key_kernel, dim3(g256, c), dim3(256), 0, stream, n, points, (const unsigned *)bounds, eyes, k0, i0);
    int key_bits = 20;
    while ((1 << (key_bits - 20)) < c) key_bits++;
    if (!check(rocprim::radix_sort_pairs(ws + o_tmp, sort_bytes, (const unsigned *)k0, k1, (const int *)i0, i1, total, 0u, (unsigned)key_bits, stream),
               "hpr radix sort"))
        return 0;
    HprTile *tiles = (HprTile *)(ws + o_tiles);
    static const int no_cull = tune_env("GENPC_HPR_NOCULL", 0, "hidden-point removal: measurement mask (1 every tile, 8 no silhouette hand-off, 16 no early accept, 32 no verify, 64 no hand-off of much-cut polygons, 128 no decisions without the walk in the wave-per-point pass, 256 count those decisions on stderr, 512 two-kernel form WITH the first kernel's own verify phase)");      // measurement knob: 1 = every tile examined, 8 = no silhouette hand-off, 16 = no early accept, 32 = no verify phase, 64 = no hand-off of much-cut polygons (results unchanged)
    hipLaunchKernelGGL(hpr_flip_kernel, dim3(g256, c), dim3(256), 0, stream, n, points, (const int *)i1, eyes, radius, fl, (const unsigned char *)dup);
    hipLaunchKernelGGL(hpr_tile_kernel, dim3(ntiles, c), dim3(kHprThreads), 0, stream, n, (const double *)fl, tiles);
    static const int env_clips = tune_env("GENPC_HPR_MAXCLIPS", 0, "hidden-point removal: clips after which a polygon goes to the wave-per-point pass (0 = pick)");
    // (large clouds: 96 -- 2 x 165546 points 46 -> 40 ms; many views of a small cloud: the second pass fills up
    //  instead -- 1024 x 10000 points 154 ms with 256, 168 with 128, 180 with 96)
    const int max_clips = (no_cull & 64) ? 0x7fffffff : (env_clips > 0 ? env_clips : (ntiles >= kHprRimTiles ? 48 : 256));      // (large clouds: 96 -> 48 once the wave-per-point pass clipped by all lanes: 2 x 165546 24.9 -> 21.5 ms)
    unsigned char *hard = (unsigned char *)(ws + o_hard);
    int *hardlist = (int *)(ws + o_hl), *hardcnt = (int *)(ws + o_hc);
    hipLaunchKernelGGL(hpr_accept_kernel, dim3(ntiles * c), dim3(kHprThreads), 0, stream, n, (const double *)fl, (const int *)i1,
                       (const HprTile *)tiles, hard, visible, counts, (no_cull & 16) ? 1 : 0, c);
    hipLaunchKernelGGL(hpr_compact_kernel, dim3(c), dim3(1024), 0, stream, n, (const unsigned char *)hard, hardlist, hardcnt);
    // The undecided points' polygons: at most one per point the accept pass left over.  That count stays on the device
    // (round 4 fetched it to size the store, the first of six stream synchronisations of this entry point): the store is
    // sized for every (view, point) pair up to GENPC_HPR_PARK_MB, and a point whose slot would lie beyond it is listed
    // for the wave-per-point pass instead of parked (it restarts from the box there: same result, computed again).
    int4 *surv = (int4 *)(ws + o_surv);
    // best_only (viewpoint_select): views that cannot be the best any more are dropped after the first polygon kernel
    const bool prune = best_only && split;
    int *und = prune ? (int *)(ws + o_und) : nullptr;
    unsigned char *alive = prune ? (unsigned char *)(ws + o_alive) : nullptr;
    if (prune && !check(hipMemsetAsync(und, 0, sizeof(int) * (size_t)c, stream), "hipMemsetAsync(hpr und)")) return 0;
    HprSegs *segs = (HprSegs *)(ws + 192);          // (the 256-byte header: status words 0 .. 15 and 32 .. 44, the bounds in words 16 .. 21, this record in 48 .. 58)
    static_assert(sizeof(HprSegs) <= 64, "the segment record shares the header with the status words");
    static const int env_park = tune_env("GENPC_HPR_PARK_MB", 3072, "hidden-point removal, two-kernel form: MiB of polygon store (256 bytes per parked point; points beyond it restart in the wave-per-point pass)");
    const size_t poly_bytes = (size_t)kHprMaxV * sizeof(double2);
    const size_t park_cap = split ? std::min<size_t>(total, std::max<size_t>(1, ((size_t)(env_park > 0 ? env_park : 1) << 20) / poly_bytes)) : 0;
    double2 *surv_poly = nullptr;
    if (split) {
        surv_poly = (double2 *)workspace(20, park_cap * poly_bytes, stream);
        if (!surv_poly) return 0;
    }
    int *work = (int *)(ws + o_work);          // line x: the first try's counter of segment x; lines 8 .. 11: the other passes'
    hipLaunchKernelGGL(hpr_segs_kernel, dim3(1), dim3(1024), 0, stream, c, (const int *)hardcnt, segs, (int)park_cap, work);
    static const int env_sf = tune_env("GENPC_HPR_STRAGGLE_FROM", 4, "hidden-point removal: tile batches a block walks before the hand-over");
    static const int env_sl = tune_env("GENPC_HPR_STRAGGLE_LANES", kHprThreads, "hidden-point removal: hand a block's points over when at most this many lanes are still undecided");
    static const int env_st = tune_env("GENPC_HPR_STRAGGLE_TILES", 0, "hidden-point removal: clouds of at least this many tiles hand undecided points over early");
    const int straggle_from = ntiles >= env_st ? env_sf : 0x7fffffff, straggle_lanes = env_sl;
    // split form: home tiles (the point's own, the next, the previous) the first kernel takes before it tries to verify and
    // parks what is left; the wave-per-point pass decides most parked points from the polygon they bring and takes the
    // remaining home tiles only for the others
    static const int env_ht = tune_env("GENPC_HPR_HOME_TILES", 1, "hidden-point removal, two-kernel form: home tiles (1..3) the first kernel clips by before it parks a point");
    static const int env_cw = tune_env("GENPC_HPR_CHUNK", 32, "hidden-point removal, first kernel: candidates marked at a time against the polygon as the previous ones left it (8, 16 or 32)");
    const int chunk_w = env_cw == 8 || env_cw == 16 ? env_cw : 32;
    const int home_tiles = split ? (env_ht < 1 ? 1 : (env_ht > 3 ? 3 : env_ht)) : 3;
    static const int env_hc = tune_env("GENPC_HPR_HOME_CHUNKS", 2, "hidden-point removal, two-kernel form with one home tile: 32-candidate chunks of it (1..4) the first kernel clips by");
    const int home_chunks = split && home_tiles == 1 ? (env_hc < 1 ? 1 : (env_hc > 4 ? 4 : env_hc)) : 4;
    hipLaunchKernelGGL(hpr_kernel, dim3(ntiles * c), dim3(kHprThreads), 0, stream, n, (const double *)fl, (const int *)i1,
                       (const HprTile *)tiles, (const int *)hardlist, (const int *)hardcnt, visible, counts, status, list, no_cull,
                       max_clips, split, surv, surv_poly, und, straggle_from, straggle_lanes, c, home_tiles, (const HprSegs *)segs, chunk_w, home_chunks);
    if (!check(hipGetLastError(), "hpr launch")) return 0;
    if (prune) hipLaunchKernelGGL(hpr_prune_kernel, dim3(1), dim3(1024), 0, stream, c, (const int *)counts, (const int *)und, alive);
    // The wave-per-point passes.  How many points each of them finds is known on the device only: every pass is a fixed
    // grid of one-wave blocks -- as many as the chip holds of that kernel, or as there can be items -- that reads its count
    // from `status` and leaves at once when it is zero (an empty pass costs its launch, ~5 us).
    int *list2 = (int *)k0;          // (the sort's key buffer is free by now; at most views x points entries)
    int *list3 = (int *)k1;          // (likewise: the points whose polygons outgrow the 1024-vertex tier)
    const int cus = num_cus();
    auto grid_of = [&](int per_cu) { return (int)std::min<size_t>(total, (size_t)per_cu * cus); };
    static const int env_dw = tune_env("GENPC_HPR_DECIDE_WAVES", 64, "hidden-point removal: one-wave blocks per CU of the parked points' first try");
    static const int env_ww = tune_env("GENPC_HPR_WALK_WAVES", 40, "hidden-point removal: one-wave blocks per CU of the wave-per-point pass (128-vertex tier)");
    // status words: [0] listed, [2] parked, [3] polygons over 128 vertices, [4] over kHprOverCap, [5] parked points the first try
    // left undecided, [32 .. 39] the segments' fills (the passes' work counters: `work`, a cache line each)
    if (split) {
        static const int env_dk = tune_env("GENPC_HPR_DECIDE_KERNEL", 1, "hidden-point removal: 1 = the parked points' first try in a kernel of its own (more waves per CU), 0 = inside the wave-per-point kernel");
        const int *slots = nullptr;
        const int *count = nullptr;          // (null: the parked points by segment)
        if (env_dk && !(no_cull & (32 | 128))) {
            int *sl = i0;          // (the sort's index buffer is free by now; at most views x points entries)
            const int gd = (grid_of(env_dw > 0 ? env_dw : 64) + 7) & ~7;          // block b -> segment b % 8
            hipLaunchKernelGGL(hpr_decide_kernel, dim3(gd), dim3(kWave), 0, stream, n, (const double *)fl, (const HprTile *)tiles, visible,
                               counts, status, (const int *)i1, no_cull, (const unsigned char *)alive, (const int4 *)surv,
                               (const double2 *)surv_poly, sl, (const HprSegs *)segs, work);
            if (!check(hipGetLastError(), "hpr decide launch")) return 0;
            slots = sl;
            count = status + 5;
        }
        hipLaunchKernelGGL(hpr_overflow_kernel<128>, dim3(grid_of(env_ww > 0 ? env_ww : 40)), dim3(kWave), 0, stream, n, (const double *)fl, (const HprTile *)tiles,
                           visible, counts, status, (const int *)nullptr, (const int *)i1, (const int *)hardlist, (const int *)hardcnt,
                           no_cull, (const unsigned char *)alive, list2, (const int4 *)surv, (const double2 *)surv_poly, (double2 *)nullptr, 0, slots,
                           (const HprSegs *)segs, count, work + 8 * kHprWorkLine);
        if (!check(hipGetLastError(), "hpr continuation launch")) return 0;
    }
    {
        static const bool one_tier = tune_env("GENPC_HPR_ONE_TIER", 0, "hidden-point removal: 1 = every listed point straight to the 1024-vertex tier") != 0;
        if (!one_tier) {
            hipLaunchKernelGGL(hpr_overflow_kernel<128>, dim3(grid_of(env_ww > 0 ? env_ww : 40)), dim3(kWave), 0, stream, n, (const double *)fl, (const HprTile *)tiles,
                               visible, counts, status, (const int *)list, (const int *)i1, (const int *)hardlist, (const int *)hardcnt,
                               no_cull, (const unsigned char *)alive, list2, (const int4 *)nullptr, (const double2 *)nullptr, (double2 *)nullptr, 0,
                               (const int *)nullptr, (const HprSegs *)segs, (const int *)(status + 0), work + 9 * kHprWorkLine);
            if (!check(hipGetLastError(), "hpr second pass launch")) return 0;
        } else {
            hipLaunchKernelGGL(hpr_overflow_kernel<kHprOverCap>, dim3(grid_of(5)), dim3(kWave), 0, stream, n, (const double *)fl,
                               (const HprTile *)tiles, visible, counts, status, (const int *)list, (const int *)i1, (const int *)hardlist,
                               (const int *)hardcnt, no_cull, (const unsigned char *)alive, list3, (const int4 *)nullptr,
                               (const double2 *)nullptr, (double2 *)nullptr, 0, (const int *)nullptr, (const HprSegs *)segs,
                               (const int *)(status + 0), work + 9 * kHprWorkLine);
        }
        // polygons over 128 vertices (from either launch above)
        hipLaunchKernelGGL(hpr_overflow_kernel<kHprOverCap>, dim3(grid_of(5)), dim3(kWave), 0, stream, n, (const double *)fl,
                           (const HprTile *)tiles, visible, counts, status, (const int *)list2, (const int *)i1, (const int *)hardlist,
                           (const int *)hardcnt, no_cull, (const unsigned char *)alive, list3, (const int4 *)nullptr,
                           (const double2 *)nullptr, (double2 *)nullptr, 0, (const int *)nullptr, (const HprSegs *)segs,
                           (const int *)(status + 3), work + 10 * kHprWorkLine);
        if (!check(hipGetLastError(), "hpr large-polygon launch")) return 0;
        // polygons over 1024 vertices (exactly co-spherical input, lattices seen from their centre): a third tier with the
        // polygon in global memory, n + 8 vertices per buffer -- it cannot overflow -- a pair of buffers per wave of the grid
        {
            static const int env_gm = tune_env("GENPC_HPR_GLOBAL_MB", 256, "hidden-point removal: MiB of global-memory polygon buffers of the third tier (two buffers of n + 8 vertices per wave)");
            const int gcap = n + 8;
            const size_t per = (size_t)2 * gcap * sizeof(double2);
            const int g3 = (int)std::min<size_t>(std::min<size_t>(total, 1024), std::max<size_t>(1, ((size_t)(env_gm > 0 ? env_gm : 1) << 20) / per));
            double2 *gbuf = (double2 *)workspace(30, (size_t)g3 * per, stream);
            if (!gbuf) return 0;
            hipLaunchKernelGGL(hpr_overflow_kernel<0>, dim3(g3), dim3(kWave), 0, stream, n, (const double *)fl, (const HprTile *)tiles,
                               visible, counts, status, (const int *)list3, (const int *)i1, (const int *)hardlist,
                               (const int *)hardcnt, no_cull, (const unsigned char *)alive, (int *)nullptr, (const int4 *)nullptr,
                               (const double2 *)nullptr, gbuf, gcap, (const int *)nullptr, (const HprSegs *)segs, (const int *)(status + 4), work + 11 * kHprWorkLine);
            if (!check(hipGetLastError(), "hpr global-polygon launch")) return 0;
        }
        hipLaunchKernelGGL(hpr_finish_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, stream, c, (const int *)status, counts);
        // the counters are fetched (the entry's only stream synchronisation) when somebody asks for them
        const bool tiers = tune_env("GENPC_HPR_TIERS", 0, "hidden-point removal: 1 = report on stderr how many points took the wave-per-point tiers") != 0;
        if (second_pass_points || tiers || (no_cull & 256)) {
            int st[48] = {0};
            if (!check(hipMemcpyAsync(st, status, sizeof st, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(hpr status)")) return 0;
            if (!check(hipStreamSynchronize(stream), "hipStreamSynchronize(hpr)")) return 0;
            if (second_pass_points) *second_pass_points = st[0] + st[2];      // every point a wave took over (parked or listed)
            if (st[1]) {
                set_error("genpc_hpr_visibility: internal error (a normal-cone polygon outgrew its buffer)");
                return 0;
            }
            if (tiers)
                fprintf(stderr, "hpr: %d points parked, %d listed for the wave-per-point pass, %d polygons over 128 vertices, %d over %d (global-memory tier)\n",
                        st[2], st[0], st[3], st[4], kHprOverCap);
            if (no_cull & 256)
                fprintf(stderr, "hpr: parked points, first try: %d undecided, %d visible, %d hidden; after the home tiles: %d undecided, %d visible, %d hidden, %d died in the tiles\n",
                        st[8], st[9], st[10], st[12], st[13], st[14], st[15]);
        }
    }
    if (exact) {
        if (prune) {
            if (!check(hipMemcpyAsync(exact, alive, (size_t)c, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(hpr exact)")) return 0;
        } else if (!check(hipMemsetAsync(exact, 1, (size_t)c, stream), "hipMemsetAsync(hpr exact)")) return 0;
    }
    return 1;
}

#ifdef GENPC_HPR_PROF
extern "C" __attribute__((visibility("default"))) int genpc_hpr_prof_read(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(genpc::g_hpr_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 0;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(genpc::g_hpr_prof), z, sizeof z) != hipSuccess) return 0;
    }
    return 1;
}
#endif

GENPC_API int genpc_hpr_visibility(int c, int n, const float *points, const double *eyes, double radius, unsigned char *visible,
                                   int *counts, int *second_pass_points, void *stream_)
{
    return hpr_run(c, n, points, eyes, radius, visible, counts, second_pass_points, stream_, 0, nullptr);
}

GENPC_API int genpc_hpr_best_view_counts(int c, int n, const float *points, const double *eyes, double radius,
                                         unsigned char *visible, int *counts, unsigned char *exact, int *second_pass_points,
                                         void *stream_)
{
    return hpr_run(c, n, points, eyes, radius, visible, counts, second_pass_points, stream_, 1, exact);
}
