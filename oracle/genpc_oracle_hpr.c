/*
 * genpc_oracle_hpr.c -- CPU restatement of the EXACT hidden-point-removal operator in the form the
 * gfx950 library computes it (genpc_amd/csrc/hpr.hip).  TEST INFRASTRUCTURE ONLY, same rules as
 * genpc_oracle.c.
 *
 * The reference calls open3d's PointCloud.hidden_point_removal(camera, radius) (DepthPrompting.py:273-290):
 * Katz' operator -- spherical flipping p' = v + 2 (radius - |v|) v / |v|, v = p - camera, then the convex
 * hull of the flipped points and the origin (qhull); visible = hull vertices.  oracle/hpr.py restates that
 * with qhull itself (scipy).  This file restates the same SET without a hull:
 *
 *   p'_i is a vertex of conv({p'_j} U {0})  <=>  some plane through p'_i has every other p'_j and the origin
 *   strictly on one side  <=>  there is a normal n with n.p'_i > 0 and n.p'_j < n.p'_i for all j.
 *   n.p'_i > 0 lets n be scaled to n = u_i + a e1 + b e2 with u_i = p'_i/|p'_i| and (e1, e2) an orthonormal
 *   basis of the plane normal to u_i; then n.p'_i = |p'_i| and every other point is ONE LINEAR constraint on
 *   (a, b):   a (e1.p'_j) + b (e2.p'_j) <= |p'_i| - u_i.p'_j .
 *   The feasible (a, b) form a convex polygon (the cross-section of p'_i's normal cone; for a large radius
 *   it is the power cell of point i among the directions of the cloud, weighted by depth): the point is
 *   visible iff the polygon is not empty after clipping by every other point.
 *
 * The polygon starts as the square |a|, |b| <= HPR_BOX (normals tilted from u_i by more than atan(HPR_BOX) =
 * 89.994 degrees are not considered: the one deviation from the hull definition, besides roundoff -- qhull
 * merges facets within its own tolerance).  Exact duplicates: only the copy with the lowest index takes part (it can be
 * visible and it clips); the other copies are hidden and clip nothing -- qhull reports exactly one copy of a
 * coincident group as a hull vertex too (which one is its own business), so the COUNTS agree with open3d's.
 * The points are ordered by the 2-D Morton code of their DIRECTION from the eye (20-bit keys, ties by index:
 * points in the same direction -- a surface and what it hides -- are neighbours) and cut into tiles of HPR_TILE; points whose own direction already separates them are accepted at once (see below); the
 * others are taken, in Morton order, in groups of HPR_TILE, and a point takes the candidates tile by tile: its
 * home tile and the two next to it, then outward from its group's starting tile (s, s+1, s-1, s+2, ...) -- the
 * order the GPU streams them in.  The order has no influence beyond the last bits of the polygon vertices.
 * Everything is double, operation order spelled out, -ffp-contract=off.
 * Pinned against oracle/hpr.py (qhull) in tests/test_oracle_numpy.py.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

#define HPR_BOX 1.0e4
#define HPR_MAXV 4096
#define HPR_TILE 128

typedef struct { uint32_t key; int idx; } mkey_t;

static int cmp_mkey(const void *a, const void *b)
{
    const mkey_t *x = (const mkey_t *)a, *y = (const mkey_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

static uint32_t spread10(uint32_t v)          /* 10 bits -> every second bit */
{
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

/* perm[pos] = index of the pos-th point in the view's order: 2-D Morton code of the point's DIRECTION from the
 * eye, in a frame whose axis points at the centre of the cloud's bounding box (hpr.hip: hpr_bounds_kernel,
 * hpr_view_frame, hpr_key_kernel, stable sort).  The order only decides which candidates a point meets first. */
static void view_order(int n, const float *pts, const double *eye, int *perm)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            const float v = pts[3 * i + k];
            if (!(fabsf(v) < INFINITY)) continue;
            if (v < lo[k]) lo[k] = v;
            if (v > hi[k]) hi[k] = v;
        }
    const double lx = (double)lo[0], ly = (double)lo[1], lz = (double)lo[2];
    const double hx = (double)hi[0], hy = (double)hi[1], hz = (double)hi[2];
    double wx = (lx + hx) * 0.5 - eye[0], wy = (ly + hy) * 0.5 - eye[1], wz = (lz + hz) * 0.5 - eye[2];
    const double dist = sqrt(wx * wx + wy * wy + wz * wz);
    const double hd = 0.5 * sqrt((hx - lx) * (hx - lx) + (hy - ly) * (hy - ly) + (hz - lz) * (hz - lz));
    if (dist > 0.0 && dist < INFINITY) { wx /= dist; wy /= dist; wz /= dist; }
    else { wx = 0.0; wy = 0.0; wz = 1.0; }
    const double ax = fabs(wx), ay = fabs(wy), az = fabs(wz);
    double x, y, z;
    if (ax <= ay && ax <= az) { x = 0.0; y = wz; z = -wy; }
    else if (ay <= az)        { x = -wz; y = 0.0; z = wx; }
    else                      { x = wy; y = -wx; z = 0.0; }
    const double l = sqrt(x * x + y * y + z * z);
    const double e1x = x / l, e1y = y / l, e1z = z / l;
    const double e2x = wy * e1z - wz * e1y, e2y = wz * e1x - wx * e1z, e2z = wx * e1y - wy * e1x;
    double s = 1.0;
    if (dist > hd) {
        s = 1.05 * hd / dist;
        s = s < 1.0 ? s : 1.0;
    }
    s = s > 0.0 ? s : 1.0;
    mkey_t *keys = (mkey_t *)malloc(sizeof(mkey_t) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        const double vx = (double)pts[3 * i + 0] - eye[0];
        const double vy = (double)pts[3 * i + 1] - eye[1];
        const double vz = (double)pts[3 * i + 2] - eye[2];
        const double r = sqrt(vx * vx + vy * vy + vz * vz);
        uint32_t key = 0xfffffu;
        if (r > 0.0 && r < INFINITY) {
            const double dx = vx / r, dy = vy / r, dz = vz / r;
            const double px = dx * e1x + dy * e1y + dz * e1z;
            const double py = dx * e2x + dy * e2y + dz * e2z;
            double qx = floor((px / s + 1.0) * 512.0), qy = floor((py / s + 1.0) * 512.0);
            qx = qx < 0.0 ? 0.0 : (qx > 1023.0 ? 1023.0 : qx);
            qy = qy < 0.0 ? 0.0 : (qy > 1023.0 ? 1023.0 : qy);
            key = spread10((uint32_t)qx) | (spread10((uint32_t)qy) << 1);
        }
        keys[i].key = key;
        keys[i].idx = i;
    }
    qsort(keys, (size_t)n, sizeof(mkey_t), cmp_mkey);
    for (int i = 0; i < n; i++) perm[i] = keys[i].idx;
    free(keys);
}

/* dup[i] = 1 when a point with the same coordinates (numerically: -0 == +0) and a lower index exists */
static int dup_cmp(const void *a, const void *b, void *ctx)
{
    const float *dup_pts = (const float *)ctx;
    const int i = *(const int *)a, j = *(const int *)b;
    for (int k = 0; k < 3; k++) {
        const float x = dup_pts[3 * (size_t)i + k], y = dup_pts[3 * (size_t)j + k];
        if (x < y) return -1;
        if (x > y) return 1;
        if (x != y) return (x != x) - (y != y) ? ((x != x) ? 1 : -1) : 0;      /* NaNs last, equal among themselves */
    }
    return (i > j) - (i < j);
}
static void mark_duplicates(int n, const float *pts, uint8_t *dup)
{
    int *idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) { idx[i] = i; dup[i] = 0; }
    qsort_r(idx, (size_t)n, sizeof(int), dup_cmp, (void *)pts);
    for (int p = 1; p < n; p++) {
        const float *a = pts + 3 * (size_t)idx[p], *b = pts + 3 * (size_t)idx[p - 1];
        if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) dup[idx[p]] = 1;
    }
    free(idx);
}

/* p' of every point for one camera, in the order of perm (open3d: |v| = 0 -> 1e-4) */
static void flip_points(int n, const float *pts_in, const int *perm, const double *eye, double radius, double *fl,
                        const uint8_t *dup)
{
    for (int i = 0; i < n; i++) {
        if (dup[perm[i]]) { fl[3 * i + 0] = fl[3 * i + 1] = fl[3 * i + 2] = NAN; continue; }      /* hidden, clips nothing */
        const float *pts = pts_in + 3 * (size_t)perm[i] - 3 * (size_t)i;      /* so that pts[3 i + k] is point perm[i] */
        const double vx = (double)pts[3 * i + 0] - eye[0];
        const double vy = (double)pts[3 * i + 1] - eye[1];
        const double vz = (double)pts[3 * i + 2] - eye[2];
        double r = sqrt(vx * vx + vy * vy + vz * vz);
        if (r == 0.0) r = 0.0001;
        const double k = 2.0 * (radius - r) / r;
        fl[3 * i + 0] = vx + k * vx;
        fl[3 * i + 1] = vy + k * vy;
        fl[3 * i + 2] = vz + k * vz;
    }
}

/* clip the polygon (a[], b[], *nv vertices, counter-clockwise or clockwise, convex) by A a + B b <= C */
static void clip(double *a, double *b, int *nv, double A, double B, double C, double *ta, double *tb)
{
    const int n = *nv;
    int m = 0;
    for (int k = 0; k < n; k++) {
        const int k2 = k + 1 < n ? k + 1 : 0;
        const double s0 = a[k] * A + b[k] * B - C;
        const double s1 = a[k2] * A + b[k2] * B - C;
        if (!(s0 > 0.0)) { ta[m] = a[k]; tb[m] = b[k]; m++; }
        if ((s0 > 0.0) != (s1 > 0.0) && s0 != 0.0 && s1 != 0.0) {
            const double t = s0 / (s0 - s1);
            ta[m] = a[k] + t * (a[k2] - a[k]);
            tb[m] = b[k] + t * (b[k2] - b[k]);
            m++;
        }
    }
    memcpy(a, ta, sizeof(double) * (size_t)m);
    memcpy(b, tb, sizeof(double) * (size_t)m);
    *nv = m;
}

/* vis[i] = 1 if point i is visible from `eye`; returns the number visible, -1 on bad input, -2 if a
 * polygon outgrew HPR_MAXV vertices.  max_vertices (optional): the largest polygon met on the way. */
ORACLE_API int oracle_hpr_visibility(int n, const float *pts, const double *eye, double radius, uint8_t *vis,
                                     int *max_vertices)
{
    if (n < 0 || !pts || !eye || !vis || !(radius > 0.0)) return -1;
    double *fl = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    int *perm = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (!fl || !perm) { free(fl); free(perm); return -1; }
    view_order(n, pts, eye, perm);
    uint8_t *dup = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    if (!dup) { free(fl); free(perm); return -1; }
    mark_duplicates(n, pts, dup);
    flip_points(n, pts, perm, eye, radius, fl, dup);
    free(dup);
    /* Early accept (hpr.hip: hpr_accept_kernel): the point's own direction u is already a separating normal
     * when u.p'_j < |p'_i| for every other point (margin 1e-8 |p'_i|): the origin of the (a, b) plane is
     * strictly feasible, the point is visible and no polygon is built. */
    uint8_t *hard = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int *hl = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (!hard || !hl) { free(fl); free(perm); free(hard); free(hl); return -1; }
    int total = 0, bad = 0, mv = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
    for (int i = 0; i < n; i++) {
        const double px = fl[3 * i], py = fl[3 * i + 1], pz = fl[3 * i + 2];
        const double rho = sqrt(px * px + py * py + pz * pz);
        vis[perm[i]] = 0;
        if (!(rho > 0.0) || !(rho < INFINITY)) continue;        /* coincides with the origin / not finite */
        const double ux = px / rho, uy = py / rho, uz = pz / rho;
        const double thr = 1e-8 * rho;
        int ok = 1;
        for (int j = 0; j < n && ok; j++) {
            const double qx = fl[3 * j], qy = fl[3 * j + 1], qz = fl[3 * j + 2];
            if (qx == px && qy == py && qz == pz) continue;
            const double C = rho - (ux * qx + uy * qy + uz * qz);
            if (C < thr) ok = 0;
        }
        if (ok) { vis[perm[i]] = 1; total++; }
        else hard[i] = 1;
    }
    /* the rest, in Morton order, in groups of HPR_TILE that share a starting tile (that of the group's middle
     * point): the blocks of hpr_kernel */
    int nhard = 0;
    for (int i = 0; i < n; i++)
        if (hard[i]) hl[nhard++] = i;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : total) reduction(| : bad) reduction(max : mv)
    for (int r = 0; r < nhard; r++) {
        const int i = hl[r];
        double pa[HPR_MAXV + 2], pb[HPR_MAXV + 2], ta[HPR_MAXV + 2], tb[HPR_MAXV + 2];
        const double px = fl[3 * i], py = fl[3 * i + 1], pz = fl[3 * i + 2];
        const double rho = sqrt(px * px + py * py + pz * pz);
        const double ux = px / rho, uy = py / rho, uz = pz / rho;
        /* e1 = normalise(u x axis of u's smallest component), e2 = u x e1 */
        const double ax = fabs(ux), ay = fabs(uy), az = fabs(uz);
        double e1x, e1y, e1z;
        if (ax <= ay && ax <= az) { e1x = 0.0; e1y = uz; e1z = -uy; }
        else if (ay <= az)        { e1x = -uz; e1y = 0.0; e1z = ux; }
        else                      { e1x = uy; e1y = -ux; e1z = 0.0; }
        const double l = sqrt(e1x * e1x + e1y * e1y + e1z * e1z);
        e1x /= l; e1y /= l; e1z /= l;
        const double e2x = uy * e1z - uz * e1y, e2y = uz * e1x - ux * e1z, e2z = ux * e1y - uy * e1x;
        int nv = 4;
        pa[0] = -HPR_BOX; pb[0] = -HPR_BOX;
        pa[1] = HPR_BOX;  pb[1] = -HPR_BOX;
        pa[2] = HPR_BOX;  pb[2] = HPR_BOX;
        pa[3] = -HPR_BOX; pb[3] = HPR_BOX;
        int mid = (r / HPR_TILE) * HPR_TILE + HPR_TILE / 2;
        if (mid > nhard - 1) mid = nhard - 1;
        const int ntiles = (n + HPR_TILE - 1) / HPR_TILE, own = hl[mid] / HPR_TILE, home = i / HPR_TILE;
        /* candidate order: the point's home tile, the next, the previous, then every other tile
         * outward from the group's starting tile */
        for (int seq = 0; seq < 3 + 2 * ntiles && nv > 0; seq++) {
            int tile;
            if (seq < 3) {
                tile = home + (seq == 0 ? 0 : (seq == 1 ? 1 : -1));
            } else {
                const int step = seq - 3;
                tile = (step & 1) ? own + (step + 1) / 2 : own - step / 2;
                if (tile >= home - 1 && tile <= home + 1) continue;
            }
            if (tile < 0 || tile >= ntiles) continue;
            /* 32-candidate chunks; in the home tile they start with the point's own chunk (nearest first) */
            const int rot = tile == home ? ((i - home * HPR_TILE) >> 5) : 0;
            for (int jj = 0; jj < HPR_TILE && nv > 0; jj++) {
                const int j = tile * HPR_TILE + (((jj >> 5) + rot) & (HPR_TILE / 32 - 1)) * 32 + (jj & 31);
                if (j >= n) continue;
                const double qx = fl[3 * j], qy = fl[3 * j + 1], qz = fl[3 * j + 2];
                if (qx == px && qy == py && qz == pz) continue;        /* the point itself, or an exact duplicate */
                const double A = e1x * qx + e1y * qy + e1z * qz;
                const double B = e2x * qx + e2y * qy + e2z * qz;
                const double C = rho - (ux * qx + uy * qy + uz * qz);
                int any = 0;
                for (int k = 0; k < nv; k++) any |= (pa[k] * A + pb[k] * B - C > 0.0);
                if (!any) continue;
                if (nv + 2 > HPR_MAXV) { bad = 1; nv = 0; break; }
                clip(pa, pb, &nv, A, B, C, ta, tb);
                if (nv < 3) nv = 0;            /* no interior left: not strictly extreme */
                if (nv > mv) mv = nv;
            }
        }
        if (nv > 0) { vis[perm[i]] = 1; total++; }
    }
    free(hard);
    free(hl);
    free(fl);
    free(perm);
    if (max_vertices) *max_vertices = mv;
    return bad ? -2 : total;
}
