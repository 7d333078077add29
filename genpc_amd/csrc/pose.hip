// pose.hip -- the SE(3)+scale alignment loop of optim_registration/diff_obj_pose.py
//  for gfx950: 7-DoF pose model (6D rotation, translation,
// log-scale), Chamfer half of compute_loss_function, analytic backward, Adam.
//
// The reference runs this through torch autograd: per iteration ~40 tiny kernels,
// two chamfer_3DDist calls (each computing BOTH directions and dropping one,
// utils/loss_util.py:35-38), two backward launches with 6 atomics per point, and a
// host round trip for the tqdm postfix.  Here an iteration is four launches with no
// host synchronisation:
//   pose_transform_kernel   pts = (R ((v - c) s)^T)^T + c + t
//   nn_forward_kernel       ONE bidirectional NN (chamfer.hip): d1/i1 and d2/i2 are
//                           exactly the two partial-matching terms
//   pose_grad_kernel        d loss / d (R, s, t) reduced straight from (d, idx): the
//                           per-point gradient is never materialised, no atomics on
//                           point buffers; fp64 block reduction, one fp64 atomic per
//                           block and quantity
//   pose_update_kernel      (one wave) orthogonality term, 6D Gram-Schmidt backward,
//                           Adam for the three parameter groups, loss history,
//                           best-of-starts bookkeeping
// HBM traffic per iteration is O(N): ~24 B/point for the transform, ~28 B/point for
// the gradient pass; the NN launch dominates (VALU-bound, see chamfer.hip).
#include "common.h"
#include "../../include/genpc_hip.h"

#include <math.h>

// chamfer.hip
extern "C" int genpc_chamfer_forward(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                                     int *idx1, float *dist2, int *idx2, void *stream);

namespace genpc {

int genpc_mean3(int b, int n, const float *v, float *out, double *accum, hipStream_t st);

constexpr int kQBlock = 256;

// pytorch3d.transforms.rotation_6d_to_matrix (rows b1, b2, b1 x b2); F.normalize eps 1e-12
__device__ __forceinline__ void rot6d_to_matrix(const float *d6, float *R)
{
    const float a1x = d6[0], a1y = d6[1], a1z = d6[2], a2x = d6[3], a2y = d6[4], a2z = d6[5];
    float n1 = sqrtf(a1x * a1x + a1y * a1y + a1z * a1z);
    n1 = n1 > 1e-12f ? n1 : 1e-12f;
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
    float b2x = a2x - dt * b1x, b2y = a2y - dt * b1y, b2z = a2z - dt * b1z;
    float n2 = sqrtf(b2x * b2x + b2y * b2y + b2z * b2z);
    n2 = n2 > 1e-12f ? n2 : 1e-12f;
    b2x /= n2; b2y /= n2; b2z /= n2;
    R[0] = b1x; R[1] = b1y; R[2] = b1z;
    R[3] = b2x; R[4] = b2y; R[5] = b2z;
    R[6] = b1y * b2z - b1z * b2y;
    R[7] = b1z * b2x - b1x * b2z;
    R[8] = b1x * b2y - b1y * b2x;
}

__device__ __forceinline__ void pose_point(const float *R, float s, const float *c, const float *t, float vx, float vy,
                                           float vz, float *o)
{
    const float lx = __fmul_rn(vx - c[0], s), ly = __fmul_rn(vy - c[1], s), lz = __fmul_rn(vz - c[2], s);
    o[0] = __fadd_rn(__fadd_rn(__fmaf_rn(R[2], lz, __fmaf_rn(R[1], ly, __fmul_rn(R[0], lx))), c[0]), t[0]);
    o[1] = __fadd_rn(__fadd_rn(__fmaf_rn(R[5], lz, __fmaf_rn(R[4], ly, __fmul_rn(R[3], lx))), c[1]), t[1]);
    o[2] = __fadd_rn(__fadd_rn(__fmaf_rn(R[8], lz, __fmaf_rn(R[7], ly, __fmul_rn(R[6], lx))), c[2]), t[2]);
}

// diff_obj_pose.py:419-423
// blockIdx.y = batch element; element e reads v + e*n*3, center + e*cstride,
// params + e*pstride (strides in floats) and writes pts + e*n*3.
__global__ __launch_bounds__(kQBlock) void pose_transform_kernel(int n, const float *__restrict__ v,
                                                                 const float *__restrict__ center, int cstride,
                                                                 const float *__restrict__ params, int pstride,
                                                                 float *__restrict__ pts)
{
    const int e = blockIdx.y;
    v += (size_t)e * n * 3;
    pts += (size_t)e * n * 3;
    center += (size_t)e * cstride;
    params += (size_t)e * pstride;
    float R[9];
    rot6d_to_matrix(params, R);
    const float s = expf(params[9]);
    const float c[3] = {center[0], center[1], center[2]};
    const float t[3] = {params[6], params[7], params[8]};
    for (int j = blockIdx.x * kQBlock + threadIdx.x; j < n; j += gridDim.x * kQBlock) {
        float o[3];
        pose_point(R, s, c, t, v[(size_t)j * 3 + 0], v[(size_t)j * 3 + 1], v[(size_t)j * 3 + 2], o);
        pts[(size_t)j * 3 + 0] = o[0];
        pts[(size_t)j * 3 + 1] = o[1];
        pts[(size_t)j * 3 + 2] = o[2];
    }
}

// accum[0..8] = dL/dR (row-major), [9] = dL/ds, [10..12] = dL/dt, [13] = sum sqrt(d1),
// [14] = sum sqrt(d2).  Thread t < nc: term of complete point t (pts -> partial);
// nc <= t < nc+np: term of partial point t-nc (partial -> pts), attributed to the
// complete point it matched.
__global__ __launch_bounds__(kQBlock) void pose_grad_kernel(int nc, const float *__restrict__ v,
                                                            const float *__restrict__ center, int cstride,
                                                            const float *__restrict__ params, int pstride, int np,
                                                            const float *__restrict__ partial,
                                                            const float *__restrict__ d1, const int *__restrict__ i1,
                                                            const float *__restrict__ d2, const int *__restrict__ i2,
                                                            float cd_weight, double *__restrict__ accum)
{
    __shared__ double red[15][kQBlock / kWave];
    const int e = blockIdx.y;
    v += (size_t)e * nc * 3;
    partial += (size_t)e * np * 3;
    d1 += (size_t)e * nc; i1 += (size_t)e * nc;
    d2 += (size_t)e * np; i2 += (size_t)e * np;
    center += (size_t)e * cstride;
    params += (size_t)e * pstride;
    accum += (size_t)e * 16;
    float R[9];
    rot6d_to_matrix(params, R);
    const float s = expf(params[9]);
    const float c[3] = {center[0], center[1], center[2]};
    const float t[3] = {params[6], params[7], params[8]};
    double a[15];
#pragma unroll
    for (int k = 0; k < 15; k++) a[k] = 0.0;
    for (int e = blockIdx.x * kQBlock + threadIdx.x; e < nc + np; e += gridDim.x * kQBlock) {
        int j, k;
        float d;
        double w;
        if (e < nc) {
            j = e; k = i1[e]; d = d1[e];
            a[13] += (double)sqrtf(d);
            w = (double)cd_weight / nc;
        } else {
            k = e - nc; j = i2[k]; d = d2[k];
            a[14] += (double)sqrtf(d);
            w = (double)cd_weight * 0.5 / np;
        }
        if (d == 0.0f) continue;     // torch: 0.5/sqrt(0) * 0 = NaN; no gradient here
        w *= 1.0 / sqrt((double)d);  // d sqrt(d)/dd * 2 (from d |p-q|^2 / dp)
        const float vx = v[(size_t)j * 3 + 0], vy = v[(size_t)j * 3 + 1], vz = v[(size_t)j * 3 + 2];
        float p[3];
        pose_point(R, s, c, t, vx, vy, vz, p);
        const double g[3] = {w * (double)(p[0] - partial[(size_t)k * 3 + 0]),
                             w * (double)(p[1] - partial[(size_t)k * 3 + 1]),
                             w * (double)(p[2] - partial[(size_t)k * 3 + 2])};
        const double l[3] = {(double)(vx - c[0]), (double)(vy - c[1]), (double)(vz - c[2])};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            a[10 + r] += g[r];
#pragma unroll
            for (int q = 0; q < 3; q++) a[r * 3 + q] += g[r] * (double)s * l[q];
            a[9] += g[r] * ((double)R[r * 3 + 0] * l[0] + (double)R[r * 3 + 1] * l[1] + (double)R[r * 3 + 2] * l[2]);
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 15; k++) {
        double x = a[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
        if (lane == 0) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 15) {
        double x = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kQBlock / kWave; w2++) x += red[threadIdx.x][w2];
        atomicAdd(&accum[threadIdx.x], x);
    }
}

struct PoseState {       // device-resident
    float params[10];
    float m[10];
    float v[10];
    float grad[10];
    float loss[3];       // total, cd, ortho_err
    float local_best;
    float best_loss;
    float best_params[10];
    int step;            // Adam step of the current start (1-based after the first update)
};

// One thread: finish the gradient (orthogonality term + 6D backward), optionally
// take the Adam step, record the loss, clear the accumulators.
__global__ void pose_update_kernel(int b, PoseState *__restrict__ S, double *__restrict__ accum, int nc, int np,
                                   float cd_weight, float reg_weight, float lr, int do_step,
                                   float *__restrict__ history_slot, int history_stride)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b) return;
    S += e;
    accum += (size_t)e * 16;
    if (history_slot) history_slot += (size_t)e * history_stride;
    float Rf[9];
    rot6d_to_matrix(S->params, Rf);
    const double s = (double)expf(S->params[9]);
    double gR[9];
    for (int k = 0; k < 9; k++) gR[k] = accum[k];
    const double cd = accum[13] / nc + 0.5 * accum[14] / np;
    double E[9], err2 = 0.0;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double e = 0.0;
            for (int k = 0; k < 3; k++) e += (double)Rf[a * 3 + k] * (double)Rf[b * 3 + k];
            e -= (a == b) ? 1.0 : 0.0;
            E[a * 3 + b] = e;
            err2 += e * e;
        }
    const double err = sqrt(err2);
    if (err > 0.0)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double acc = 0.0;
                for (int k = 0; k < 3; k++) acc += E[a * 3 + k] * (double)Rf[k * 3 + b];
                gR[a * 3 + b] += (double)reg_weight * 2.0 * acc / err;
            }
    // Gram-Schmidt backward (double)
    const float *d6 = S->params;
    const double a1[3] = {d6[0], d6[1], d6[2]}, a2[3] = {d6[3], d6[4], d6[5]};
    const double n1 = sqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    const double b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const double dt = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const double u[3] = {a2[0] - dt * b1[0], a2[1] - dt * b1[1], a2[2] - dt * b1[2]};
    const double n2 = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const double b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const double *g1 = gR, *g2 = gR + 3, *g3 = gR + 6;
    double gb1[3], gb2[3];
    gb1[0] = g1[0] + (b2[1] * g3[2] - b2[2] * g3[1]);
    gb1[1] = g1[1] + (b2[2] * g3[0] - b2[0] * g3[2]);
    gb1[2] = g1[2] + (b2[0] * g3[1] - b2[1] * g3[0]);
    gb2[0] = g2[0] + (g3[1] * b1[2] - g3[2] * b1[1]);
    gb2[1] = g2[1] + (g3[2] * b1[0] - g3[0] * b1[2]);
    gb2[2] = g2[2] + (g3[0] * b1[1] - g3[1] * b1[0]);
    const double dot2 = gb2[0] * b2[0] + gb2[1] * b2[1] + gb2[2] * b2[2];
    const double gu[3] = {(gb2[0] - dot2 * b2[0]) / n2, (gb2[1] - dot2 * b2[1]) / n2, (gb2[2] - dot2 * b2[2]) / n2};
    const double gub1 = gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2];
    const double ga2[3] = {gu[0] - gub1 * b1[0], gu[1] - gub1 * b1[1], gu[2] - gub1 * b1[2]};
    for (int k = 0; k < 3; k++) gb1[k] += -dt * gu[k] - gub1 * a2[k];
    const double dot1 = gb1[0] * b1[0] + gb1[1] * b1[1] + gb1[2] * b1[2];
    float grad[10];
    for (int k = 0; k < 3; k++) grad[k] = (float)((gb1[k] - dot1 * b1[k]) / n1);
    for (int k = 0; k < 3; k++) grad[3 + k] = (float)ga2[k];
    for (int k = 0; k < 3; k++) grad[6 + k] = (float)accum[10 + k];
    grad[9] = (float)(accum[9] * s);
    const float loss = (float)((double)cd_weight * cd + (double)reg_weight * err);
    for (int k = 0; k < 10; k++) S->grad[k] = grad[k];
    S->loss[0] = loss;
    S->loss[1] = (float)cd;
    S->loss[2] = (float)err;
    for (int k = 0; k < 15; k++) accum[k] = 0.0;
    if (history_slot) *history_slot = loss;
    if (!do_step) return;
    if (loss < S->local_best) S->local_best = loss;       // diff_obj_pose.py:549-551
    // torch.optim.Adam, three groups: lr, 0.2 lr, 0.1 lr (diff_obj_pose.py:524-528)
    const int step = ++S->step;
    const double be1 = 0.9, be2 = 0.999, eps = 1e-8;
    const double bc1 = 1.0 - pow(be1, (double)step), bc2 = 1.0 - pow(be2, (double)step);
    for (int k = 0; k < 10; k++) {
        const double l = k < 6 ? (double)lr : (k < 9 ? (double)lr * 0.2 : (double)lr * 0.1);
        S->m[k] = (float)(be1 * S->m[k] + (1.0 - be1) * grad[k]);
        S->v[k] = (float)(be2 * S->v[k] + (1.0 - be2) * (double)grad[k] * grad[k]);
        const double denom = sqrt((double)S->v[k]) / sqrt(bc2) + eps;
        S->params[k] = (float)(S->params[k] - (l / bc1) * (S->m[k] / denom));
    }
}

// start < 0: global init.  Otherwise begin start `start` (get_init_rot('y', 90*start),
// trans 0, log_scale log(0.75): diff_obj_pose.py:367,519).
__global__ void pose_begin_kernel(int b, PoseState *__restrict__ S, double *__restrict__ accum, int start)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b) return;
    S += e;
    accum += (size_t)e * 16;
    if (start < 0) {
        S->best_loss = __builtin_inff();
        for (int k = 0; k < 10; k++) S->best_params[k] = 0.0f;
        for (int k = 0; k < 15; k++) accum[k] = 0.0;
        return;
    }
    const double th = start * 90.0 * M_PI / 180.0;
    const float init[10] = {(float)cos(th), 0.0f, (float)sin(th), 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, logf(0.75f)};
    for (int k = 0; k < 10; k++) {
        S->params[k] = init[k];
        S->m[k] = 0.0f;
        S->v[k] = 0.0f;
    }
    S->local_best = __builtin_inff();
    S->step = 0;
}

// end of a start: keep the FINAL parameters of the start with the lowest loss seen
// (diff_obj_pose.py:570-576).  final != 0: also emit T = [[sR, t],[0,1]] (:464-468).
__global__ void pose_end_kernel(int b, PoseState *__restrict__ S, int final, float *__restrict__ transform,
                                float *__restrict__ best_params)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b) return;
    S += e;
    if (transform) transform += (size_t)e * 16;
    if (best_params) best_params += (size_t)e * 10;
    if (!final) {
        if (S->local_best < S->best_loss) {
            S->best_loss = S->local_best;
            for (int k = 0; k < 10; k++) S->best_params[k] = S->params[k];
        }
        return;
    }
    float R[9];
    rot6d_to_matrix(S->best_params, R);
    const float s = expf(S->best_params[9]);
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) transform[a * 4 + b] = R[a * 3 + b] * s;
        transform[a * 4 + 3] = S->best_params[6 + a];
    }
    transform[12] = transform[13] = transform[14] = 0.0f;
    transform[15] = 1.0f;
    if (best_params)
        for (int k = 0; k < 10; k++) best_params[k] = S->best_params[k];
}

static int lin_grid(long long n)
{
    long long g = ceil_div64(n, kQBlock);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace genpc

GENPC_API int genpc_pose_transform(int n, const float *v, const float *center, const float *params, float *pts,
                                   void *stream)
{
    using namespace genpc;
    if (n <= 0) return 1;
    hipLaunchKernelGGL(pose_transform_kernel, dim3(lin_grid(n), 1), dim3(kQBlock), 0, (hipStream_t)stream, n, v, center,
                       0, params, 0, pts);
    return check(hipGetLastError(), "pose_transform launch") ? 1 : 0;
}

GENPC_API int genpc_pose_cd_grad(int nc, const float *v, const float *center, const float *params, int np,
                                 const float *partial, const float *d1, const int *i1, const float *d2, const int *i2,
                                 float cd_weight, float reg_weight, float *loss_out, float *grad, void *stream)
{
    using namespace genpc;
    if (nc <= 0 || np <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace(3, 256 + sizeof(PoseState), st);
    if (!ws) return 0;
    double *accum = (double *)ws;
    PoseState *S = (PoseState *)(ws + 256);
    if (!check(hipMemsetAsync(accum, 0, 16 * sizeof(double), st), "hipMemsetAsync(accum)")) return 0;
    if (!check(hipMemcpyAsync(S->params, params, 10 * sizeof(float), hipMemcpyDeviceToDevice, st), "copy params"))
        return 0;
    hipLaunchKernelGGL(pose_grad_kernel, dim3(lin_grid((long long)nc + np), 1), dim3(kQBlock), 0, st, nc, v, center, 0,
                       params, 0, np, partial, d1, i1, d2, i2, cd_weight, accum);
    hipLaunchKernelGGL(pose_update_kernel, dim3(1), dim3(64), 0, st, 1, S, accum, nc, np, cd_weight, reg_weight, 0.0f, 0,
                       (float *)nullptr, 0);
    if (!check(hipMemcpyAsync(grad, S->grad, 10 * sizeof(float), hipMemcpyDeviceToDevice, st), "copy grad")) return 0;
    if (!check(hipMemcpyAsync(loss_out, S->loss, 3 * sizeof(float), hipMemcpyDeviceToDevice, st), "copy loss")) return 0;
    return check(hipGetLastError(), "pose_cd_grad launch") ? 1 : 0;
}

GENPC_API int genpc_pose_optimize_cd_batch(int b, int nc, const float *complete, int np, const float *partial,
                                           float lr, int iters, int starts, float *transform, float *history,
                                           float *best_params, void *stream)
{
    using namespace genpc;
    if (b <= 0 || nc <= 0 || np <= 0 || iters < 0 || starts < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    // scratch: accum[b,16] | state[b] | center[b,4] | pts[b,nc,3] | d1 | d2 | i1 | i2
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = 0;
    const size_t o_acc = off; off += up((size_t)b * 16 * sizeof(double));
    const size_t o_state = off; off += up((size_t)b * sizeof(PoseState));
    const size_t o_center = off; off += up((size_t)b * 4 * sizeof(float));
    const size_t o_pts = off; off += up((size_t)b * nc * 12);
    const size_t o_d1 = off; off += up((size_t)b * nc * 4);
    const size_t o_d2 = off; off += up((size_t)b * np * 4);
    const size_t o_i1 = off; off += up((size_t)b * nc * 4);
    const size_t o_i2 = off; off += up((size_t)b * np * 4);
    char *ws = (char *)workspace(4, off, st);
    if (!ws) return 0;
    double *accum = (double *)(ws + o_acc);
    PoseState *S = (PoseState *)(ws + o_state);
    float *center = (float *)(ws + o_center);
    float *pts = (float *)(ws + o_pts);
    float *d1 = (float *)(ws + o_d1), *d2 = (float *)(ws + o_d2);
    int *i1 = (int *)(ws + o_i1), *i2 = (int *)(ws + o_i2);
    constexpr int kStateFloats = (int)(sizeof(PoseState) / sizeof(float));
    static_assert(sizeof(PoseState) % sizeof(float) == 0, "PoseState must be float-addressable");

    // center = mean(vert_pos) per scan (diff_obj_pose.py:362)
    if (!genpc_mean3(b, nc, complete, center, accum, st)) return 0;

    const int gb = ceil_div(b, 64);
    hipLaunchKernelGGL(pose_begin_kernel, dim3(gb), dim3(64), 0, st, b, S, accum, -1);
    const int g_t = lin_grid(nc), g_g = lin_grid((long long)nc + np);
    const int hstride = starts * (iters + 1);
    for (int s = 0; s < starts; s++) {
        hipLaunchKernelGGL(pose_begin_kernel, dim3(gb), dim3(64), 0, st, b, S, accum, s);
        for (int it = 0; it <= iters; it++) {
            hipLaunchKernelGGL(pose_transform_kernel, dim3(g_t, b), dim3(kQBlock), 0, st, nc, complete,
                               (const float *)center, 4, (const float *)S->params, kStateFloats, pts);
            if (genpc_chamfer_forward(b, nc, pts, np, partial, d1, i1, d2, i2, stream) != 1) return 0;
            hipLaunchKernelGGL(pose_grad_kernel, dim3(g_g, b), dim3(kQBlock), 0, st, nc, complete, (const float *)center,
                               4, (const float *)S->params, kStateFloats, np, partial, (const float *)d1,
                               (const int *)i1, (const float *)d2, (const int *)i2, 3.0f, accum);
            hipLaunchKernelGGL(pose_update_kernel, dim3(gb), dim3(64), 0, st, b, S, accum, nc, np, 3.0f, 0.001f, lr, 1,
                               history ? history + (size_t)s * (iters + 1) + it : (float *)nullptr, hstride);
        }
        hipLaunchKernelGGL(pose_end_kernel, dim3(gb), dim3(64), 0, st, b, S, 0, (float *)nullptr, (float *)nullptr);
    }
    hipLaunchKernelGGL(pose_end_kernel, dim3(gb), dim3(64), 0, st, b, S, 1, transform, best_params);
    return check(hipGetLastError(), "pose_optimize_cd launch") ? 1 : 0;
}

GENPC_API int genpc_pose_optimize_cd(int nc, const float *complete, int np, const float *partial, float lr,
                                     int iters, int starts, float *transform, float *history, float *best_params,
                                     void *stream)
{
    return genpc_pose_optimize_cd_batch(1, nc, complete, np, partial, lr, iters, starts, transform, history,
                                        best_params, stream);
}

namespace genpc {

// accum[e*16 + 0..2] += sum of v[e, :, 0..2]
__global__ __launch_bounds__(kQBlock) void mean3_accum_kernel(int n, const float *__restrict__ v,
                                                              double *__restrict__ accum)
{
    __shared__ double red[3][kQBlock / kWave];
    const int e = blockIdx.y;
    v += (size_t)e * n * 3;
    accum += (size_t)e * 16;
    double a[3] = {0.0, 0.0, 0.0};
    for (int j = blockIdx.x * kQBlock + threadIdx.x; j < n; j += gridDim.x * kQBlock) {
        a[0] += (double)v[(size_t)j * 3 + 0];
        a[1] += (double)v[(size_t)j * 3 + 1];
        a[2] += (double)v[(size_t)j * 3 + 2];
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double x = a[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
        if (lane == 0) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double x = 0.0;
        for (int w2 = 0; w2 < kQBlock / kWave; w2++) x += red[threadIdx.x][w2];
        atomicAdd(&accum[threadIdx.x], x);
    }
}

__global__ void mean3_finish_kernel(int b, int n, double *__restrict__ accum, float *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * 3) return;
    const int e = t / 3, k = t % 3;
    out[e * 4 + k] = (float)(accum[(size_t)e * 16 + k] / n);
    accum[(size_t)e * 16 + k] = 0.0;
}

int genpc_mean3(int b, int n, const float *v, float *out, double *accum, hipStream_t st)
{
    if (!check(hipMemsetAsync(accum, 0, (size_t)b * 16 * sizeof(double), st), "hipMemsetAsync(mean)")) return 0;
    hipLaunchKernelGGL(mean3_accum_kernel, dim3(lin_grid(n), b), dim3(kQBlock), 0, st, n, v, accum);
    hipLaunchKernelGGL(mean3_finish_kernel, dim3(ceil_div(b * 3, 64)), dim3(64), 0, st, b, n, accum, out);
    return check(hipGetLastError(), "mean3 launch") ? 1 : 0;
}

}  // namespace genpc
