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
        const double x = source[(size_t)j * 3 + 0], y = source[(size_t)j * 3 + 1], z = source[(size_t)j * 3 + 2];
        float *o = pts + ((size_t)c * ns + j) * 3;
        o[0] = (float)(m[0] * x + m[1] * y + m[2] * z + m[3]);
        o[1] = (float)(m[4] * x + m[5] * y + m[6] * z + m[7]);
        o[2] = (float)(m[8] * x + m[9] * y + m[10] * z + m[11]);
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
__device__ void jacobi4(double A[4][4], double V[4][4])
{
    double norm2 = 0.0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            V[i][j] = (i == j) ? 1.0 : 0.0;
            norm2 += A[i][j] * A[i][j];
        }
    const double tiny2 = 1e-36 * norm2;          // an element below 1e-18 |A|: no bit of any eigenvector depends on it
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0;
        for (int i = 0; i < 4; i++)
            for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
        if (!(off > tiny2)) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                if (!(A[p][q] * A[p][q] > tiny2)) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                A[p][q] = A[q][p] = 0.0;          // (what the rotation was chosen for)
                for (int k = 0; k < 4; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// Horn's closed-form absolute orientation from the Kabsch sums
__device__ void kabsch_from_sums(const double *sums, double *U)
{
    const double n = sums[0];
    const double mp[3] = {sums[1] / n, sums[2] / n, sums[3] / n};
    const double mq[3] = {sums[4] / n, sums[5] / n, sums[6] / n};
    double S[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) S[a][b] = sums[7 + a * 3 + b] - n * mp[a] * mq[b];
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double V[4][4];
    jacobi4(N, V);
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (N[i][i] > N[best][best]) best = i;
    double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
    const double nn = sqrt(w * w + x * x + y * y + z * z);
    w /= nn; x /= nn; y /= nn; z /= nn;
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                            {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                            {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) U[a * 4 + b] = R[a][b];
        U[a * 4 + 3] = mq[a] - (R[a][0] * mp[0] + R[a][1] * mp[1] + R[a][2] * mp[2]);
    }
    U[12] = U[13] = U[14] = 0.0;
    U[15] = 1.0;
}

__global__ void icp_update_kernel(int k, int ns, double *__restrict__ accum, double *__restrict__ T,
                                  IcpState *__restrict__ state, double *__restrict__ stats, int pass, int max_iter,
                                  double rel_fitness, double rel_rmse)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    double *A = accum + (size_t)c * 17;
    IcpState &S = state[c];
    if (pass == 0) {
        S.done = 0;
        S.iters = 0;
    }
    if (!S.done) {
        const double n = A[0];
        const double fitness = n / ns, rmse = n > 0 ? sqrt(A[16] / n) : 0.0;
        if (pass > 0) {
            S.iters += 1;
            if (fabs(S.prev_fitness - fitness) < rel_fitness && fabs(S.prev_rmse - rmse) < rel_rmse) S.done = 1;
        }
        S.prev_fitness = fitness;
        S.prev_rmse = rmse;
        stats[c * 3 + 0] = fitness;
        stats[c * 3 + 1] = rmse;
        stats[c * 3 + 2] = (double)S.iters;
        if (!S.done) {
            if (pass >= max_iter || n < 1.0) {
                S.done = 1;
            } else {
                double U[16], Tn[16];
                double *Tc = T + (size_t)c * 16;
                kabsch_from_sums(A, U);
                for (int a = 0; a < 4; a++)
                    for (int b = 0; b < 4; b++) {
                        double acc = 0.0;
                        for (int q = 0; q < 4; q++) acc += U[a * 4 + q] * Tc[q * 4 + b];
                        Tn[a * 4 + b] = acc;
                    }
                for (int q = 0; q < 16; q++) Tc[q] = Tn[q];
            }
        }
    }
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
