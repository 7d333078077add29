// icp.hip -- batched point-to-point ICP and the scale-search scoring of reg_xyz.py
//  for gfx950.
//
// The reference runs 11 + 1000 open3d ICP solves on the CPU, one candidate scale at
// a time, each followed by two chamfer_3DDist calls on ~1-5 k-point clouds with
// numpy <-> torch <-> GPU round trips (reg_xyz.py:60-96,146-173) -- launch- and
// transfer-bound.  Here K candidates that share the source and target clouds and
// differ only in their initial 4x4 transform are solved together: per ICP pass
//   icp_transform_kernel   p' = T_k p for all K candidates (T in double, p' fp32)
//   nn_forward_kernel      one batched NN launch, B = K (chamfer.hip)
//   icp_accum_kernel       Kabsch sums over the correspondences with d2 <= r^2
//                          (fp64 wave/block reduction, 17 atomics per block)
//   icp_update_kernel      one thread per candidate: fitness / rmse, open3d's
//                          convergence test, Horn's quaternion solve (4x4 Jacobi),
//                          T_k <- update T_k
// with no host synchronisation; converged candidates idle.  The 1000-candidate
// anisotropic scale search scores every candidate with ONE batched NN launch
// (B = 1000): its score (reg_xyz.py:77-83) depends on the scaled source only, not
// on the ICP result, so only the winner needs an ICP solve.
// Numerics follow oracle/genpc_oracle_geom.c (oracle_icp): same fp32 NN, sums in
// double; sums are reduced in a different order (1e-12 relative).
#include "nn.h"
#include "../../include/genpc_hip.h"

#include <math.h>

extern "C" int genpc_nm_distance(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                                 int *result_i, void *stream);
extern "C" int genpc_chamfer_forward(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                                     int *idx1, float *dist2, int *idx2, void *stream);

namespace genpc {

constexpr int kIBlock = 256;

struct IcpState {
    double prev_fitness, prev_rmse;
    int done, iters;
};

// p' = T p: the product in double, rounded to fp32 (oracle: icp_evaluate)
__device__ __forceinline__ void icp_map_point(const double *m, const float *__restrict__ p, float &ox, float &oy, float &oz)
{
    const double x = p[0], y = p[1], z = p[2];
    ox = (float)(m[0] * x + m[1] * y + m[2] * z + m[3]);
    oy = (float)(m[4] * x + m[5] * y + m[6] * z + m[7]);
    oz = (float)(m[8] * x + m[9] * y + m[10] * z + m[11]);
}

__device__ __forceinline__ void icp_map_point(const double *m, float sx, float sy, float sz, float &ox, float &oy, float &oz)
{
    const float p[3] = {sx, sy, sz};
    icp_map_point(m, p, ox, oy, oz);
}

__global__ __launch_bounds__(kIBlock) void icp_transform_kernel(int ns, const float *__restrict__ source,
                                                                const double *__restrict__ T,
                                                                float *__restrict__ pts)
{
    const int c = blockIdx.y;
    const double *M = T + (size_t)c * 16;
    double m[12];
#pragma unroll
    for (int k = 0; k < 12; k++) m[k] = M[k];
    for (int j = blockIdx.x * kIBlock + threadIdx.x; j < ns; j += gridDim.x * kIBlock) {
        float *o = pts + ((size_t)c * ns + j) * 3;
        icp_map_point(m, source + (size_t)j * 3, o[0], o[1], o[2]);
    }
}

// out[c, j, :] = in[j, :] * scale[c, :]   (double product rounded to fp32); scale == nullptr: plain replicate
__global__ __launch_bounds__(kIBlock) void replicate_scale_kernel(int n, const float *__restrict__ in,
                                                                  const float *__restrict__ scale,
                                                                  float *__restrict__ out)
{
    const int c = blockIdx.y;
    double s[3] = {1.0, 1.0, 1.0};
    if (scale) {
        s[0] = scale[c * 3 + 0]; s[1] = scale[c * 3 + 1]; s[2] = scale[c * 3 + 2];
    }
    for (int j = blockIdx.x * kIBlock + threadIdx.x; j < n; j += gridDim.x * kIBlock) {
        float *o = out + ((size_t)c * n + j) * 3;
        if (scale) {
            o[0] = (float)((double)in[(size_t)j * 3 + 0] * s[0]);
            o[1] = (float)((double)in[(size_t)j * 3 + 1] * s[1]);
            o[2] = (float)((double)in[(size_t)j * 3 + 2] * s[2]);
        } else {
            o[0] = in[(size_t)j * 3 + 0]; o[1] = in[(size_t)j * 3 + 1]; o[2] = in[(size_t)j * 3 + 2];
        }
    }
}

// accum[c, 17]: n, sum p[3], sum q[3], sum p q^T [9], sum d2
__global__ __launch_bounds__(kIBlock) void icp_accum_kernel(int ns, const float *__restrict__ pts,
                                                            const float *__restrict__ target,
                                                            const float *__restrict__ d, const int *__restrict__ idx,
                                                            float md2, double *__restrict__ accum)
{
    __shared__ double red[17][kIBlock / kWave];
    const int c = blockIdx.y;
    double a[17];
#pragma unroll
    for (int k = 0; k < 17; k++) a[k] = 0.0;
    for (int j = blockIdx.x * kIBlock + threadIdx.x; j < ns; j += gridDim.x * kIBlock) {
        const float dj = d[(size_t)c * ns + j];
        if (!(dj <= md2)) continue;
        const float *p = pts + ((size_t)c * ns + j) * 3;
        const float *q = target + (size_t)idx[(size_t)c * ns + j] * 3;
        const double pp[3] = {p[0], p[1], p[2]}, qq[3] = {q[0], q[1], q[2]};
        a[0] += 1.0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            a[1 + r] += pp[r];
            a[4 + r] += qq[r];
#pragma unroll
            for (int s = 0; s < 3; s++) a[7 + r * 3 + s] += pp[r] * qq[s];
        }
        a[16] += (double)dj;
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        double x = a[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
        if (lane == 0) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 17) {
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < kIBlock / kWave; w++) x += red[threadIdx.x][w];
        atomicAdd(&accum[(size_t)c * 17 + threadIdx.x], x);
    }
}

// Cyclic Jacobi on the symmetric 4x4.  Converged when the off-diagonal mass is rounding noise RELATIVE to the matrix
// (an absolute threshold of 1e-300 is never met: a rotated-away element comes back as noise of size 1e-16 |A|, and all 30
// sweeps ran -- 35 us per ICP pass in one lane, fp64 divisions and square roots; now 5-7 sweeps).
// 1 / sqrt(x), x normal and positive: the hardware estimate and two Newton steps (relative error ~1e-16)
__device__ __forceinline__ double rsqrt_f64(double x)
{
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double e = __builtin_fma(-x * y, y, 1.0);        // 1 - x y^2
        y = __builtin_fma(y * 0.5, e, y);
    }
    return y;
}

// (Every index below is a compile-time constant once the loops are unrolled: the matrices live in registers -- as arrays indexed by
// loop variables they sat in scratch memory, and the update was 12 us of one lane per pass.)
__device__ __forceinline__ void jacobi4(double (&A)[4][4], double (&V)[4][4])
{
    double norm2 = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            V[i][j] = (i == j) ? 1.0 : 0.0;
            norm2 += A[i][j] * A[i][j];
        }
    const double tiny2 = 1e-32 * norm2;          // an element below 1e-16 |A|: rounding noise of the rotations themselves
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
        if (!(off > tiny2)) break;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                if (!(A[p][q] * A[p][q] > tiny2)) continue;
                // the rotation that zeroes A[p][q]: with d = A[q][q] - A[p][p], b = 2 A[p][q], r = |(d, b)|:  cos 2phi = |d| / r,
                // sin 2phi = sgn(d) b / r, c = sqrt((1 + cos 2phi) / 2), s = sin 2phi / (2 c) -- the rotation of the textbook form
                // (theta = d / b, t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), c = 1 / sqrt(t^2 + 1), s = t c) with two reciprocal
                // square roots instead of three divisions and two square roots (each a ~40-instruction dependent chain in fp64:
                // the update was 12 us of one lane per pass)
                const double d = A[q][q] - A[p][p], b = 2.0 * A[p][q];
                const double inv_r = rsqrt_f64(d * d + b * b);
                const double c2 = 0.5 + 0.5 * fabs(d) * inv_r;
                const double inv_c = rsqrt_f64(c2);
                const double c = c2 * inv_c, s = (d >= 0 ? 0.5 : -0.5) * b * inv_r * inv_c;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                A[p][q] = A[q][p] = 0.0;          // (what the rotation was chosen for)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// Horn's closed-form absolute orientation from the Kabsch sums
__device__ __forceinline__ void kabsch_from_sums(const double *sums, double *U)
{
    const double n = sums[0], inv_n = 1.0 / n;
    const double mp[3] = {sums[1] * inv_n, sums[2] * inv_n, sums[3] * inv_n};
    const double mq[3] = {sums[4] * inv_n, sums[5] * inv_n, sums[6] * inv_n};
    double S[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) S[a][b] = sums[7 + a * 3 + b] - n * mp[a] * mq[b];
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double V[4][4];
    jacobi4(N, V);
    // the eigenvector of the largest eigenvalue (the first of equals)
    double lam = N[0][0], w = V[0][0], x = V[1][0], y = V[2][0], z = V[3][0];
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (N[i][i] > lam) {
            lam = N[i][i];
            w = V[0][i]; x = V[1][i]; y = V[2][i]; z = V[3][i];
        }
    const double inn = rsqrt_f64(w * w + x * x + y * y + z * z);
    w *= inn; x *= inn; y *= inn; z *= inn;
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                            {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                            {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) U[a * 4 + b] = R[a][b];
        U[a * 4 + 3] = mq[a] - (R[a][0] * mp[0] + R[a][1] * mp[1] + R[a][2] * mp[2]);
    }
    U[12] = U[13] = U[14] = 0.0;
    U[15] = 1.0;
}

// One candidate, after the pass's sums: fitness / rmse, open3d's convergence test, the Kabsch step.  (Does not clear the sums.)
__device__ void icp_update_one(const double *A, double *Tc, IcpState &S, double *stats, int ns, int pass, int max_iter,
                               double rel_fitness, double rel_rmse)
{
    if (pass == 0) {
        S.done = 0;
        S.iters = 0;
    }
    if (S.done) return;
    const double n = A[0];
    const double fitness = n / ns, rmse = n > 0 ? sqrt(A[16] / n) : 0.0;
    if (pass > 0) {
        S.iters += 1;
        if (fabs(S.prev_fitness - fitness) < rel_fitness && fabs(S.prev_rmse - rmse) < rel_rmse) S.done = 1;
    }
    S.prev_fitness = fitness;
    S.prev_rmse = rmse;
    stats[0] = fitness;
    stats[1] = rmse;
    stats[2] = (double)S.iters;
    if (S.done) return;
    if (pass >= max_iter || n < 1.0) {
        S.done = 1;
        return;
    }
    double U[16], Tn[16];
    kabsch_from_sums(A, U);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) acc += U[a * 4 + q] * Tc[q * 4 + b];
            Tn[a * 4 + b] = acc;
        }
#pragma unroll
    for (int q = 0; q < 16; q++) Tc[q] = Tn[q];
}

__global__ void icp_update_kernel(int k, int ns, double *__restrict__ accum, double *__restrict__ T,
                                  IcpState *__restrict__ state, double *__restrict__ stats, int pass, int max_iter,
                                  double rel_fitness, double rel_rmse)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    double *A = accum + (size_t)c * 17;
    icp_update_one(A, T + (size_t)c * 16, state[c], stats + (size_t)c * 3, ns, pass, max_iter, rel_fitness, rel_rmse);
    for (int q = 0; q < 17; q++) A[q] = 0.0;
}

// scores[c] = mean sqrt(d1[c,:]) + w * mean sqrt(d2[c,:])   (reg_xyz.py:81-83,167-169)
__global__ __launch_bounds__(kIBlock) void cd_score_kernel(int n1, const float *__restrict__ d1, int n2,
                                                           const float *__restrict__ d2, float w,
                                                           float *__restrict__ scores)
{
    __shared__ double red[2][kIBlock / kWave];
    const int c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int j = threadIdx.x; j < n1; j += kIBlock) a += (double)sqrtf(d1[(size_t)c * n1 + j]);
    for (int j = threadIdx.x; j < n2; j += kIBlock) b += (double)sqrtf(d2[(size_t)c * n2 + j]);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, kWave);
        b += __shfl_xor(b, off, kWave);
    }
    if (lane == 0) {
        red[0][wave] = a;
        red[1][wave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double x = 0.0, y = 0.0;
        for (int q = 0; q < kIBlock / kWave; q++) {
            x += red[0][q];
            y += red[1][q];
        }
        const float m1 = (float)(x / n1), m2 = (float)(y / n2);
        scores[c] = m1 + m2 * w;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The whole solve of one candidate in ONE workgroup (round 6).  The five launches per pass above cost ~45 us per pass at the
// pipeline's sizes (3-5 k points a side: every kernel is a few microseconds of work behind a launch and a device-wide kernel
// boundary), 31 passes a solve, six solves a completed scan: ~8 ms of a 38 ms scan and 930 of its 2100 launches (with six scans
// in flight the device runs ~90 k kernels per second whatever the number of queues: the launches are the limit there).
//
// ICP uses a correspondence only when its squared distance is <= max_dist^2, so the nearest-neighbour search needs no more than
// the targets within max_dist of the query: the target cloud goes ONCE into a uniform grid in LDS (cell >= 1.001 max_dist: a
// target within max_dist lies in the 27 cells around the query's), and a pass is, per wave and 64 queries at a time,
//   p' = T p (registers) -> the home cell's points -> three rounds (one plane of 9 neighbour cells each): the (query, cell) pairs
//   whose box can still hold a point nearer than the query's best so far (or than max_dist) go to a wave-private list in LDS and are
//   worked off one pair per lane (ds_min_u64 on distance bits << 32 | index: the smallest distance, the lowest index among equals,
//   which is the order of the exhaustive search) -> Kabsch sums of the inliers in double registers
// -- no workgroup barrier inside --, then a fixed-order reduction (lane tree, waves in order: the sums do not depend on timing,
// unlike the atomics above) and the update by one thread.  Distances: sqdist<FMA>, the arithmetic of the exhaustive search.
// Boxes are widened by 4e-7 of the largest coordinate and compared with 1e-5 relative slack: a cell is skipped only when no
// point of it can tie or beat the bound in fp32.
constexpr int kFT = 1024;
constexpr int kFWaves = kFT / kWave;
constexpr int kFItems = 512;                 // (query, neighbour cell) pairs of one wave per round of the list
constexpr int kFAxis = 64;                   // cells per axis at most
constexpr size_t kFFixed = (size_t)kFT * 16 + (size_t)kFT * 8 + (size_t)kFWaves * kFItems * 4 + (size_t)17 * kFWaves * 8 + 16 * 8 + 17 * 8;

struct IcpGrid {
    float lo[3], hi[3];
    float h, inv_h, eps;
    int c[3];
};

__device__ __forceinline__ int icp_cell_of(const IcpGrid &G, float x, float y, float z)
{
    const int ix = (int)fminf(fmaxf((x - G.lo[0]) * G.inv_h, 0.0f), (float)(G.c[0] - 1));
    const int iy = (int)fminf(fmaxf((y - G.lo[1]) * G.inv_h, 0.0f), (float)(G.c[1] - 1));
    const int iz = (int)fminf(fmaxf((z - G.lo[2]) * G.inv_h, 0.0f), (float)(G.c[2] - 1));
    return (iz * G.c[1] + iy) * G.c[0] + ix;
}
// squared distance from p to the (widened) box of cell i along one axis
__device__ __forceinline__ float icp_axis_gap(const IcpGrid &G, int a, int i, float p)
{
    const float lo = G.lo[a] + (float)i * G.h - G.eps;
    const float hi = (i == G.c[a] - 1 ? G.hi[a] : G.lo[a] + (float)(i + 1) * G.h) + G.eps;
    const float g = fmaxf(fmaxf(lo - p, p - hi), 0.0f);
    return g * g;
}

// key of a candidate: distance bits << 32 | index << 16 | position in P -- the smallest distance, the lowest index among equals
template <int FMA>
__device__ __forceinline__ unsigned long long icp_key(const float4 v, unsigned pos, float px, float py, float pz)
{
    const float d = sqdist<FMA>(v.x - px, v.y - py, v.z - pz);
    return ((unsigned long long)__float_as_uint(d) << 32) | ((unsigned)__float_as_int(v.w) << 16) | pos;
}
// the points P[b, e) against one query, four at a time (four LDS reads in flight; the last group repeats the last point)
template <int FMA>
__device__ __forceinline__ unsigned long long icp_scan_cell(const float4 *__restrict__ P, unsigned b, unsigned e, float px, float py, float pz,
                                                            unsigned long long key)
{
    for (unsigned t = b; t < e; t += 4) {
        const unsigned l = e - 1;
        const unsigned t1 = t + 1 < l ? t + 1 : l, t2 = t + 2 < l ? t + 2 : l, t3 = t + 3 < l ? t + 3 : l;
        const float4 v0 = P[t], v1 = P[t1], v2 = P[t2], v3 = P[t3];
        unsigned long long k0 = icp_key<FMA>(v0, t, px, py, pz), k1 = icp_key<FMA>(v1, t1, px, py, pz);
        const unsigned long long k2 = icp_key<FMA>(v2, t2, px, py, pz), k3 = icp_key<FMA>(v3, t3, px, py, pz);
        k0 = k0 < k2 ? k0 : k2;
        k1 = k1 < k3 ? k1 : k3;
        k0 = k0 < k1 ? k0 : k1;
        key = k0 < key ? k0 : key;
    }
    return key;
}

#define ICP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// tools/icp_timeline.py builds a private copy with -DGENPC_ICP_TIMELINE: thread 0 of candidate 0 stamps the wall clock (100 MHz)
// at the start, after the grid, and per pass after the search, the reduction and the update.
#ifdef GENPC_ICP_TIMELINE
__device__ unsigned long long g_icp_tl[2 + 3 * 64];
#define ICP_STAMP(i) do { if (tid == 0 && cand == 0 && (i) < 2 + 3 * 64) g_icp_tl[i] = wall_clock64(); } while (0)
#else
#define ICP_STAMP(i) do { } while (0)
#endif

template <int FMA>
__global__ __launch_bounds__(kFT) void icp_fused_kernel(int ns, const float *__restrict__ source, int nt, const float *__restrict__ target,
                                                        float md2, double max_dist, const double *__restrict__ init, int max_iter,
                                                        double rel_fitness, double rel_rmse, int cells_cap, double *__restrict__ out_T,
                                                        double *__restrict__ stats)
{
    extern __shared__ __align__(16) unsigned char icp_lds[];
    __shared__ IcpGrid G;
    __shared__ IcpState s_state;
    __shared__ float s_bb[6][kFWaves];
    __shared__ unsigned s_scan[kFWaves];
    const int cand = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int ntp = (nt + 3) & ~3;
    float4 *P = (float4 *)icp_lds;                                   // [ntp] targets in cell order: x, y, z, index
    unsigned *cend = (unsigned *)(P + ntp);                          // [cells_cap] end of each cell's run in P
    float4 *Qw = (float4 *)(cend + cells_cap) + wave * kWave;        // per wave: the 64 queries of the batch
    unsigned long long *keys = (unsigned long long *)((float4 *)(cend + cells_cap) + kFT) + wave * kWave;      // per wave: their best so far
    unsigned *items = (unsigned *)((unsigned long long *)((float4 *)(cend + cells_cap) + kFT) + kFT) + wave * kFItems;     // per wave: query | cell << 6
    double *red = (double *)((unsigned *)((unsigned long long *)((float4 *)(cend + cells_cap) + kFT) + kFT) + kFWaves * kFItems);   // [17][waves]
    double *s_T = red + 17 * kFWaves;                                // [16]
    double *s_sums = s_T + 16;                                       // [17]

    ICP_STAMP(0);
    // ---- the grid of the target cloud
    {
        float b[6] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        for (int t = tid; t < nt; t += kFT)
            for (int a = 0; a < 3; a++) {
                const float v = target[(size_t)t * 3 + a];
                b[a] = fminf(b[a], v);
                b[3 + a] = fmaxf(b[3 + a], v);
            }
#pragma unroll
        for (int a = 0; a < 6; a++) {
            float v = b[a];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float o = __shfl_xor(v, off, kWave);
                v = a < 3 ? fminf(v, o) : fmaxf(v, o);
            }
            if (lane == 0) s_bb[a][wave] = v;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float mag = 0.0f, ext[3];
        for (int a = 0; a < 3; a++) {
            float lo = __builtin_inff(), hi = -__builtin_inff();
            for (int w = 0; w < kFWaves; w++) {
                lo = fminf(lo, s_bb[a][w]);
                hi = fmaxf(hi, s_bb[3 + a][w]);
            }
            G.lo[a] = lo;
            G.hi[a] = hi;
            ext[a] = hi - lo;
            mag = fmaxf(mag, fmaxf(fabsf(lo), fabsf(hi)));
        }
        const bool finite = mag < 1e30f;
        G.eps = finite ? 4e-7f * mag + 1e-30f : 0.0f;
        float h = (float)(max_dist * 1.001) + 4.0f * G.eps;
        if (!(h > 1e-30f) || !(h < 1e30f) || !finite) h = __builtin_inff();
        for (;;) {
            long long cells = 1;
            for (int a = 0; a < 3; a++) {
                const float q = h < __builtin_inff() ? ext[a] / h : 0.0f;
                G.c[a] = q >= (float)(kFAxis - 1) ? kFAxis : (int)q + 1;
                cells *= G.c[a];
            }
            if (cells <= cells_cap) break;
            h *= 1.25f;
        }
        G.h = h < __builtin_inff() ? h : 0.0f;             // (one cell: its box is the cloud's, h plays no part)
        G.inv_h = h < __builtin_inff() ? 1.0f / h : 0.0f;
        for (int q = 0; q < 16; q++) s_T[q] = init[(size_t)cand * 16 + q];
    }
    __syncthreads();
    const int cells = G.c[0] * G.c[1] * G.c[2];
    for (int q = tid; q < cells; q += kFT) cend[q] = 0;
    __syncthreads();
    for (int t = tid; t < nt; t += kFT)
        atomicAdd(&cend[icp_cell_of(G, target[(size_t)t * 3 + 0], target[(size_t)t * 3 + 1], target[(size_t)t * 3 + 2])], 1u);
    __syncthreads();
    {
        // exclusive scan of the cell counts: a run of `per` cells per thread
        const int per = (cells + kFT - 1) / kFT;
        const int c0 = tid * per, c1 = c0 + per < cells ? c0 + per : cells;
        unsigned sum = 0;
        for (int q = c0; q < c1; q++) sum += cend[q];
        unsigned incl = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned o = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) s_scan[wave] = incl;
        __syncthreads();
        unsigned base = incl - sum;
        for (int w = 0; w < wave; w++) base += s_scan[w];
        for (int q = c0; q < c1; q++) {
            const unsigned n = cend[q];
            cend[q] = base;
            base += n;
        }
    }
    __syncthreads();
    for (int t = tid; t < nt; t += kFT) {
        const float x = target[(size_t)t * 3 + 0], y = target[(size_t)t * 3 + 1], z = target[(size_t)t * 3 + 2];
        const unsigned pos = atomicAdd(&cend[icp_cell_of(G, x, y, z)], 1u);        // afterwards cend[c] is the END of cell c's run
        P[pos] = make_float4(x, y, z, __int_as_float(t));
    }
    __syncthreads();

    ICP_STAMP(1);
    const unsigned long long kNone = ((unsigned long long)0x7f800000u << 32) | 0xffffffffu;      // (+inf, nothing)
    const int c0 = G.c[0], c1 = G.c[1], c2 = G.c[2];
    for (int pass = 0; pass <= max_iter; pass++) {
        double m[12];
#pragma unroll
        for (int q = 0; q < 12; q++) m[q] = s_T[q];
        double acc[17];
#pragma unroll
        for (int q = 0; q < 17; q++) acc[q] = 0.0;
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;          // the batch's source point (loaded one batch ahead)
        if (wave * kWave + lane < ns) {
            const float *sp = source + (size_t)(wave * kWave + lane) * 3;
            sx = sp[0]; sy = sp[1]; sz = sp[2];
        }
        for (int b0 = wave * kWave; b0 < ns; b0 += kFT) {
            const int j = b0 + lane;
            const bool valid = j < ns;
            float px = 0.0f, py = 0.0f, pz = 0.0f;
            int cx = -1, cy = -1, cz = -1;
            unsigned long long key = kNone;
            if (valid) icp_map_point(m, sx, sy, sz, px, py, pz);
            if (j + kFT < ns) {
                const float *sp = source + (size_t)(j + kFT) * 3;
                sx = sp[0]; sy = sp[1]; sz = sp[2];
            }
            if (valid) {
                // cell coordinates, -1 and c for "before" / "beyond" the grid (NaN: -1)
                cx = (int)fminf(fmaxf(floorf((px - G.lo[0]) * G.inv_h), -1.0f), (float)c0);
                cy = (int)fminf(fmaxf(floorf((py - G.lo[1]) * G.inv_h), -1.0f), (float)c1);
                cz = (int)fminf(fmaxf(floorf((pz - G.lo[2]) * G.inv_h), -1.0f), (float)c2);
                if (cx >= 0 && cx < c0 && cy >= 0 && cy < c1 && cz >= 0 && cz < c2) {
                    const int cell = (cz * c1 + cy) * c0 + cx;
                    key = icp_scan_cell<FMA>(P, cell ? cend[cell - 1] : 0u, cend[cell], px, py, pz, key);
                }
            }
            Qw[lane] = make_float4(px, py, pz, 0.0f);
            keys[lane] = key;
            // the neighbour cells that can still hold a point that ties or beats the best so far (or max_dist)
            unsigned mask = 0;
            if (valid) {
                const float lim = fminf(__uint_as_float((unsigned)(key >> 32)), md2);
                float gx[3], gy[3], gz[3];
#pragma unroll
                for (int o = 0; o < 3; o++) {
                    gx[o] = cx + o - 1 >= 0 && cx + o - 1 < c0 ? icp_axis_gap(G, 0, cx + o - 1, px) : __builtin_inff();
                    gy[o] = cy + o - 1 >= 0 && cy + o - 1 < c1 ? icp_axis_gap(G, 1, cy + o - 1, py) : __builtin_inff();
                    gz[o] = cz + o - 1 >= 0 && cz + o - 1 < c2 ? icp_axis_gap(G, 2, cz + o - 1, pz) : __builtin_inff();
                }
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    // the plane's nine runs first (18 LDS reads in flight), then the tests
                    unsigned rb[9], re[9];
                    const int nz = min(max(cz + pl - 1, 0), c2 - 1);
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        const int nx = min(max(cx + k % 3 - 1, 0), c0 - 1), ny = min(max(cy + k / 3 - 1, 0), c1 - 1);
                        const int cell = (nz * c1 + ny) * c0 + nx;
                        re[k] = cend[cell];
                        rb[k] = cend[cell > 0 ? cell - 1 : 0];
                        if (cell == 0) rb[k] = 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        if (pl == 1 && k == 4) continue;
                        const float g = gz[pl] + gy[k / 3] + gx[k % 3];            // (+inf: outside the grid)
                        if (g * 0.99999f <= lim && rb[k] != re[k]) mask |= 1u << (pl * 9 + k);
                    }
                }
            }
            int cnt = __popc(mask);
            ICP_WAVE_SYNC();
            for (;;) {
                const int incl = wave_scan_incl(cnt);
                const int total = __builtin_amdgcn_readlane(incl, kWave - 1);
                if (total == 0) break;
                int w = incl - cnt;
                while (mask && w < kFItems) {
                    const int sl = __ffs(mask) - 1;
                    mask &= mask - 1;
                    const int cell = ((cz + sl / 9 - 1) * c1 + cy + (sl / 3) % 3 - 1) * c0 + cx + sl % 3 - 1;
                    items[w++] = (unsigned)lane | ((unsigned)cell << 6);
                }
                cnt = __popc(mask);
                ICP_WAVE_SYNC();
                const int todo = total < kFItems ? total : kFItems;
                for (int i = lane; i < todo; i += kWave) {
                    const unsigned it = items[i];
                    const int q = it & 63, cell = it >> 6;
                    const float4 Q = Qw[q];
                    const unsigned long long k = icp_scan_cell<FMA>(P, cell ? cend[cell - 1] : 0u, cend[cell], Q.x, Q.y, Q.z, kNone);
                    atomicMin(&keys[q], k);
                }
                ICP_WAVE_SYNC();
            }
            key = keys[lane];
            const float dj = __uint_as_float((unsigned)(key >> 32));
            if (valid && dj <= md2) {
                const float4 qv = P[(unsigned)key & 0xffffu];
                const double pp[3] = {px, py, pz}, qq[3] = {qv.x, qv.y, qv.z};
                acc[0] += 1.0;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    acc[1 + a] += pp[a];
                    acc[4 + a] += qq[a];
#pragma unroll
                    for (int b = 0; b < 3; b++) acc[7 + a * 3 + b] += pp[a] * qq[b];
                }
                acc[16] += (double)dj;
            }
            __builtin_amdgcn_wave_barrier();          // (the next batch rewrites Qw / keys)
        }
        ICP_STAMP(2 + pass * 3);
        // fixed-order reduction: the lanes on the DPP network, then the waves in order
#pragma unroll
        for (int q = 0; q < 17; q++) {
            const double x = wave_sum63(acc[q]);
            if (lane == kWave - 1) red[q * kFWaves + wave] = x;
        }
        __syncthreads();
        if (wave == 0) {
            if (lane < 17) {
                double x = 0.0;
#pragma unroll
                for (int w = 0; w < kFWaves; w++) x += red[lane * kFWaves + w];
                s_sums[lane] = x;
            }
            ICP_WAVE_SYNC();
            ICP_STAMP(3 + pass * 3);
            if (lane == 0) icp_update_one(s_sums, s_T, s_state, stats + (size_t)cand * 3, ns, pass, max_iter, rel_fitness, rel_rmse);
        }
        __syncthreads();
        ICP_STAMP(4 + pass * 3);
        if (s_state.done) break;
    }
    if (tid < 16) out_T[(size_t)cand * 16 + tid] = s_T[tid];
}

// LDS of the one-workgroup solve for nt targets with `cells` grid cells
static size_t icp_fused_lds(int nt, int cells) { return (size_t)((nt + 3) & ~3) * 16 + (size_t)cells * 4 + kFFixed; }

static int gx(int n)
{
    int g = ceil_div(n, kIBlock);
    return g > 64 ? 64 : (g < 1 ? 1 : g);
}

}  // namespace genpc

GENPC_API int genpc_icp_batch(int k, int ns, const float *source, int nt, const float *target, double max_dist,
                              const double *init, int max_iter, double rel_fitness, double rel_rmse, double *out_T,
                              double *stats, void *stream)
{
    using namespace genpc;
    if (k <= 0 || ns <= 0 || nt <= 0 || max_iter < 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    // the one-workgroup solve (one launch) when the target cloud and its grid fit a compute unit's LDS
    static const int env_fused = tune_env("GENPC_ICP_FUSED", 1, "ICP: 1 = the whole solve of a candidate in one workgroup (target grid in LDS; clouds up to ~7000 target points), 0 = five launches per pass");
    if (env_fused) {
        int cells = 8192;
        while (cells >= 1024 && icp_fused_lds(nt, cells) + 1024 > (size_t)160 * 1024) cells >>= 1;
        if (cells >= 1024 && nt < 65536) {
            const size_t lds = icp_fused_lds(nt, cells);
            const int fma = arith_mode() != 0 ? 1 : 0;
            static size_t set_bytes[2] = {0, 0};
            if (lds > set_bytes[fma]) {
                const hipError_t e = fma ? hipFuncSetAttribute((const void *)icp_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                         : hipFuncSetAttribute((const void *)icp_fused_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (!check(e, "hipFuncSetAttribute(icp_fused_kernel)")) return 0;
                set_bytes[fma] = lds;
            }
            const float md2f = (float)(max_dist * max_dist);
            if (fma)
                hipLaunchKernelGGL((icp_fused_kernel<1>), dim3(k), dim3(kFT), lds, st, ns, source, nt, target, md2f, max_dist, init, max_iter,
                                   rel_fitness, rel_rmse, cells, out_T, stats);
            else
                hipLaunchKernelGGL((icp_fused_kernel<0>), dim3(k), dim3(kFT), lds, st, ns, source, nt, target, md2f, max_dist, init, max_iter,
                                   rel_fitness, rel_rmse, cells, out_T, stats);
            return check(hipGetLastError(), "icp (one workgroup) launch") ? 1 : 0;
        }
    }
    // scratch: accum[k,17] | state[k] | pts[k,ns,3] | target_rep[k,nt,3] | d[k,ns] | idx[k,ns]
    size_t off = 0;
    const size_t o_acc = off; off += ((size_t)k * 17 * 8 + 255) / 256 * 256;
    const size_t o_state = off; off += ((size_t)k * sizeof(IcpState) + 255) / 256 * 256;
    const size_t o_pts = off; off += ((size_t)k * ns * 12 + 255) / 256 * 256;
    const size_t o_tgt = off; off += ((size_t)k * nt * 12 + 255) / 256 * 256;
    const size_t o_d = off; off += ((size_t)k * ns * 4 + 255) / 256 * 256;
    const size_t o_i = off; off += ((size_t)k * ns * 4 + 255) / 256 * 256;
    char *ws = (char *)workspace(5, off, st);
    if (!ws) return 0;
    double *accum = (double *)(ws + o_acc);
    IcpState *state = (IcpState *)(ws + o_state);
    float *pts = (float *)(ws + o_pts), *tgt = (float *)(ws + o_tgt), *d = (float *)(ws + o_d);
    int *idx = (int *)(ws + o_i);
    if (!check(hipMemsetAsync(accum, 0, (size_t)k * 17 * 8, st), "hipMemsetAsync(icp accum)")) return 0;
    if (!check(hipMemcpyAsync(out_T, init, (size_t)k * 16 * 8, hipMemcpyDeviceToDevice, st), "copy init")) return 0;
    const float *tq = target;
    if (k > 1) {
        hipLaunchKernelGGL(replicate_scale_kernel, dim3(gx(nt), k), dim3(kIBlock), 0, st, nt, target,
                           (const float *)nullptr, tgt);
        tq = tgt;
    }
    // the target does not move: its duplicate mask (nn_dedupe.hip) is made once for all passes
    // (one row: the k candidates share the target)
    unsigned *dup_t = (unsigned *)workspace(27, nn_dedupe_mask_words(1, nt) * sizeof(unsigned), st);
    if (!dup_t) return 0;
    {
        const float *dp[2] = {target, nullptr};
        const int dn[2] = {nt, 0};
        unsigned *dm[2] = {dup_t, nullptr};
        if (!launch_nn_dedupe(1, 1, dp, dn, dm, nullptr, st)) return 0;
    }
    const float md2 = (float)(max_dist * max_dist);
    for (int pass = 0; pass <= max_iter; pass++) {
        hipLaunchKernelGGL(icp_transform_kernel, dim3(gx(ns), k), dim3(kIBlock), 0, st, ns, source,
                           (const double *)out_T, pts);
        if (nn_forward(k, 1, pts, ns, tq, nt, d, idx, nullptr, 0, nullptr, 0, nullptr, nullptr, st, __builtin_inff(), dup_t, nullptr, 1) != 1)
            return 0;
        hipLaunchKernelGGL(icp_accum_kernel, dim3(gx(ns), k), dim3(kIBlock), 0, st, ns, (const float *)pts, target,
                           (const float *)d, (const int *)idx, md2, accum);
        hipLaunchKernelGGL(icp_update_kernel, dim3(ceil_div(k, 64)), dim3(64), 0, st, k, ns, accum, out_T, state, stats,
                           pass, max_iter, rel_fitness, rel_rmse);
    }
    return check(hipGetLastError(), "icp launch") ? 1 : 0;
}

#ifdef GENPC_ICP_TIMELINE
extern "C" __attribute__((visibility("default"))) int genpc_icp_timeline_read(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(genpc::g_icp_tl), sizeof(unsigned long long) * (2 + 3 * 64)) == hipSuccess ? 1 : 0;
}
#endif

GENPC_API int genpc_scale_search_scores(int k, int ns, const float *source, int nt, const float *target,
                                        const float *scales, float cd_inv_weight, float *scores, void *stream)
{
    using namespace genpc;
    if (k <= 0 || ns <= 0 || nt <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    size_t off = 0;
    const size_t o_src = off; off += ((size_t)k * ns * 12 + 255) / 256 * 256;
    const size_t o_tgt = off; off += ((size_t)k * nt * 12 + 255) / 256 * 256;
    const size_t o_d1 = off; off += ((size_t)k * ns * 4 + 255) / 256 * 256;
    const size_t o_i1 = off; off += ((size_t)k * ns * 4 + 255) / 256 * 256;
    const size_t o_d2 = off; off += ((size_t)k * nt * 4 + 255) / 256 * 256;
    const size_t o_i2 = off; off += ((size_t)k * nt * 4 + 255) / 256 * 256;
    char *ws = (char *)workspace(6, off, st);
    if (!ws) return 0;
    float *src = (float *)(ws + o_src), *tgt = (float *)(ws + o_tgt);
    float *d1 = (float *)(ws + o_d1), *d2 = (float *)(ws + o_d2);
    int *i1 = (int *)(ws + o_i1), *i2 = (int *)(ws + o_i2);
    hipLaunchKernelGGL(replicate_scale_kernel, dim3(gx(ns), k), dim3(kIBlock), 0, st, ns, source, scales, src);
    hipLaunchKernelGGL(replicate_scale_kernel, dim3(gx(nt), k), dim3(kIBlock), 0, st, nt, target,
                       (const float *)nullptr, tgt);
    // every candidate is a scaled copy of the one source against a copy of the one target: equal points stay equal under
    // any scale, so ONE row of duplicate marks per cloud (nn_dedupe.hip) serves all k candidates
    const size_t w_s = (nn_dedupe_mask_words(1, ns) + 63) & ~(size_t)63;
    unsigned *dup_s = (unsigned *)workspace(27, (w_s + nn_dedupe_mask_words(1, nt)) * sizeof(unsigned), st);
    if (!dup_s) return 0;
    unsigned *dup_t = dup_s + w_s;
    {
        const float *dp[2] = {target, source};
        const int dn[2] = {nt, ns};
        unsigned *dm[2] = {dup_t, dup_s};
        if (!launch_nn_dedupe(1, 2, dp, dn, dm, nullptr, st)) return 0;
    }
    if (nn_forward(k, 2, src, ns, tgt, nt, d1, i1, tgt, nt, src, ns, d2, i2, st, __builtin_inff(), dup_t, dup_s, 1) != 1) return 0;
    hipLaunchKernelGGL(cd_score_kernel, dim3(k), dim3(kIBlock), 0, st, ns, (const float *)d1, nt, (const float *)d2,
                       cd_inv_weight, scores);
    return check(hipGetLastError(), "scale_search_scores launch") ? 1 : 0;
}
