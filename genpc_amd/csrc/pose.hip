// pose.hip -- the SE(3)+scale alignment loop of optim_registration/diff_obj_pose.py
//  for gfx950: 7-DoF pose model (6D rotation, translation,
// log-scale), both halves of compute_loss_function (Chamfer, and the silhouette terms on an
// own differentiable occupancy splat -- the reference's Pulsar renderer is absent and
// unpinned, see the mask section below), analytic backward, Adam.
//
// The reference runs this through torch autograd: per iteration ~40 tiny kernels,
// two chamfer_3DDist calls (each computing BOTH directions and dropping one,
// utils/loss_util.py:35-38), two backward launches with 6 atomics per point, and a
// host round trip for the tqdm postfix.  Here an iteration of the Chamfer-only objective is four launches with no
// host synchronisation (the full objective adds the silhouette half -- mask section below -- and, since round 6, folds the
// update into the next iteration's transform and runs the two halves on two streams that hand over through device
// counters: see PoseFuse and pose_publish_block):
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
#include "nn.h"
#include "emd.h"
#include "../../include/genpc_hip.h"

#include <math.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>

// chamfer.hip
extern "C" int genpc_chamfer_forward(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                                     int *idx1, float *dist2, int *idx2, void *stream);

namespace genpc {

int genpc_mean3(int b, int n, const float *v, float *out, double *accum, hipStream_t st);

constexpr int kQBlock = 256;
// doubles per scan: [0..12] gradient sums, [13,14] Chamfer sums, [15] mask loss, [16..18] sum I_ch, [19..21] sum I_ch^2,
// [22..25] mse / bce / intersection / sum m, [26 + 6 ch + k] the six gradient sums of channel ch (mask_sums_kernel)
constexpr int kAcc = 48;

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

// Waiting for a count another stream's kernel publishes (bounded: ~2 s; a wait that gives up marks ctr[2] and the results of the call
// become NaN).  Acquire: what the counted blocks wrote before their increment is visible afterwards.
__device__ __forceinline__ bool pose_wait_count(unsigned *ctr, int which, unsigned target)
{
    for (long long spin = 0; spin < (1ll << 24); spin++) {
        if ((int)(__hip_atomic_load(ctr + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            return true;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_store(ctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
// every block of a counted launch, after its work: what it wrote is visible to whoever sees the count
__device__ __forceinline__ void pose_publish_block(unsigned *ctr, int which)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_fetch_add(ctr + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void pose_wait_kernel(unsigned *ctr, int which, unsigned target)
{
    if (threadIdx.x == 0) (void)pose_wait_count(ctr, which, target);
}

// accum[0..8] = dL/dR (row-major), [9] = dL/ds, [10..12] = dL/dt, [13] = sum sqrt(d1),
// [14] = sum sqrt(d2).  Thread t < nc: term of complete point t (pts -> partial);
// nc <= t < nc+np: term of partial point t-nc (partial -> pts), attributed to the
// complete point it matched.
struct PoseGradArgs {          // pose_grad_kernel's arguments, for the launch that carries it along (mask_grad_kernel)
    int nc, cstride, pstride, np;
    const float *v, *center, *params, *partial, *d1, *d2;
    const int *i1, *i2;
    float cd_weight;
    double *accum;
    int gx;                    // blocks per batch element; 0: nothing rides along
};
__device__ __forceinline__ void pose_grad_body(int bx, int gdx, int e, int nc, const float *__restrict__ v,
                                               const float *__restrict__ center, int cstride,
                                               const float *__restrict__ params, int pstride, int np,
                                               const float *__restrict__ partial,
                                               const float *__restrict__ d1, const int *__restrict__ i1,
                                               const float *__restrict__ d2, const int *__restrict__ i2,
                                               float cd_weight, double *__restrict__ accum, double (*red)[kQBlock / kWave])
{
    v += (size_t)e * nc * 3;
    partial += (size_t)e * np * 3;
    d1 += (size_t)e * nc; i1 += (size_t)e * nc;
    d2 += (size_t)e * np; i2 += (size_t)e * np;
    center += (size_t)e * cstride;
    params += (size_t)e * pstride;
    accum += (size_t)e * kAcc;
    float R[9];
    rot6d_to_matrix(params, R);
    const float s = expf(params[9]);
    const float c[3] = {center[0], center[1], center[2]};
    const float t[3] = {params[6], params[7], params[8]};
    double a[15];
#pragma unroll
    for (int k = 0; k < 15; k++) a[k] = 0.0;
    for (int e = bx * kQBlock + threadIdx.x; e < nc + np; e += gdx * kQBlock) {
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
        const double x = wave_sum63(a[k]);
        if (lane == kWave - 1) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 15) {
        double x = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kQBlock / kWave; w2++) x += red[threadIdx.x][w2];
        atomicAdd(&accum[threadIdx.x], x);
    }
}

__global__ __launch_bounds__(kQBlock) void pose_grad_kernel(int nc, const float *__restrict__ v,
                                                            const float *__restrict__ center, int cstride,
                                                            const float *__restrict__ params, int pstride, int np,
                                                            const float *__restrict__ partial,
                                                            const float *__restrict__ d1, const int *__restrict__ i1,
                                                            const float *__restrict__ d2, const int *__restrict__ i2,
                                                            float cd_weight, double *__restrict__ accum, unsigned *ctr)
{
    __shared__ double red[15][kQBlock / kWave];
    pose_grad_body(blockIdx.x, gridDim.x, blockIdx.y, nc, v, center, cstride, params, pstride, np, partial, d1, i1, d2, i2, cd_weight, accum, red);
    if (ctr) pose_publish_block(ctr, 1);
}

struct PoseState {       // device-resident
    float params[10];
    float m[10];
    float v[10];
    float grad[10];
    float loss[4];       // total, cd, ortho_err, mask_loss
    float local_best;
    float best_loss;
    float best_params[10];
    int step;            // Adam step of the current start (1-based after the first update)
    int patience_counter;   // steps since local_best last improved (diff_obj_pose.py:549-556)
    int stopped;         // the start has run out of patience: its parameters are frozen for the rest of its iterations
};
constexpr int kPosePatience = 300;      // diff_obj_pose.py:530

// One thread, one batch element: finish the gradient (orthogonality term + 6D backward), optionally take the Adam step,
// record the loss; `clear`: zero the element's accumulators.  S may be a private copy of the state (the fused form below).
__device__ __forceinline__ void pose_update_one(PoseState *S, double *accum, int nc, int np, float cd_weight, float reg_weight, float lr,
                                                int do_step, float *history_slot, bool clear)
{
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
    // accum[15]: mask_weight * mask_loss of this step (mask_loss_kernel), 0 without the mask term
    const float loss = (float)((double)cd_weight * cd + (double)reg_weight * err + accum[15]);
    for (int k = 0; k < 10; k++) S->grad[k] = grad[k];
    S->loss[0] = loss;
    S->loss[1] = (float)cd;
    S->loss[2] = (float)err;
    S->loss[3] = (float)accum[15];
    if (clear)
        for (int k = 0; k < kAcc; k++) accum[k] = 0.0;
    // Early stop (diff_obj_pose.py:529-556): the reference leaves a start's loop once `patience` steps in a row failed to
    // improve its best loss.  The launches of a start are enqueued up front here, so a stopped start keeps its parameters
    // and its best loss through the remaining launches (history: NaN = "iteration not run").
    if (do_step && S->stopped) {
        if (history_slot) *history_slot = __builtin_nanf("");
        return;
    }
    if (history_slot) *history_slot = loss;
    if (!do_step) return;
    if (loss < S->local_best) {       // :549-553 (the optimizer step below has been taken by then, as here)
        S->local_best = loss;
        S->patience_counter = 0;
    } else if (++S->patience_counter > kPosePatience) {
        S->stopped = 1;               // :554-556: break AFTER this iteration's step
    }
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

__global__ void pose_update_kernel(int b, PoseState *__restrict__ S, double *__restrict__ accum, int nc, int np,
                                   float cd_weight, float reg_weight, float lr, int do_step,
                                   float *__restrict__ history_slot, int history_stride)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b) return;
    pose_update_one(S + e, accum + (size_t)e * kAcc, nc, np, cd_weight, reg_weight, lr, do_step,
                    history_slot ? history_slot + (size_t)e * history_stride : (float *)nullptr, true);
}

// The update of step k fused into the transform of step k + 1 (round 6: one launch and one kernel boundary less per Adam step).
// Every block of the transform recomputes the update of ITS batch element from the previous state and the previous step's
// accumulators (thread 0; the same arithmetic on the same inputs: every block gets the same parameters), block 0 of the element
// writes the new state to the OTHER state buffer (the other blocks are still reading the old one), records the loss and zeroes
// the accumulators the new step is about to use (their last reader was the previous transform).
struct PoseFuse {
    const PoseState *S_in;
    PoseState *S_out;
    double *acc_in, *acc_zero;
    int do_update, nc, np;
    float lr;
    float *history;          // slot of the step being finished (element 0), or null
    int hstride;
    // the hand-over between the loop's two streams through device words instead of events (below): ctr[0] counts finished blocks of
    // the transforms, ctr[1] of pose_grad, ctr[2] != 0: a wait gave up
    unsigned *ctr;
    unsigned pg_target;      // the update waits for ctr[1] to reach this (the Chamfer half's sums are complete)
};

__device__ __forceinline__ const float *pose_fused_params(const PoseFuse &fu, int e, const float *params, float *s_par)
{
    if (!fu.do_update) return params;
    if (threadIdx.x == 0) {
        bool gave_up = false;
        if (fu.ctr) gave_up = !pose_wait_count(fu.ctr, 1, fu.pg_target) || __hip_atomic_load(fu.ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        PoseState st = fu.S_in[e];
        const bool lead = blockIdx.x == 0;
        pose_update_one(&st, fu.acc_in + (size_t)e * kAcc, fu.nc, fu.np, 3.0f, 0.001f, fu.lr, 1,
                        lead && fu.history ? fu.history + (size_t)e * fu.hstride : (float *)nullptr, false);
        if (gave_up)
            for (int k = 0; k < 10; k++) st.params[k] = __builtin_nanf("");          // (a hand-over that timed out: visibly)
#pragma unroll
        for (int k = 0; k < 10; k++) s_par[k] = st.params[k];
        if (lead) {
            fu.S_out[e] = st;
            for (int k = 0; k < kAcc; k++) fu.acc_zero[(size_t)e * kAcc + k] = 0.0;
        }
    }
    __syncthreads();
    return s_par;
}

// start < 0: global init.  Otherwise begin start `start` (get_init_rot('y', 90*start),
// trans 0, log_scale log(0.75): diff_obj_pose.py:367,519).
// start_mod > 0: the starts of a scan run side by side (element e = scan * start_mod + start): start = e % start_mod.
__global__ void pose_begin_kernel(int b, PoseState *__restrict__ S, double *__restrict__ accum, int start, int start_mod)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b) return;
    if (start >= 0 && start_mod > 0) start = e % start_mod;
    S += e;
    accum += (size_t)e * kAcc;
    if (start < 0) {
        S->best_loss = __builtin_inff();
        for (int k = 0; k < 10; k++) S->best_params[k] = 0.0f;
        for (int k = 0; k < kAcc; k++) accum[k] = 0.0;
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
    S->patience_counter = 0;
    S->stopped = 0;
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

// Lock-step starts: the `starts` elements of scan g have finished side by side; apply the sequential rule of
// diff_obj_pose.py:570-576 in start order (strict <: the first of equal lowest losses wins) and emit that start's
// FINAL parameters and T = [[sR, t],[0,1]].
__global__ void pose_pick_kernel(int scans, int starts, const PoseState *__restrict__ S, float *__restrict__ transform,
                                 float *__restrict__ best_params)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= scans) return;
    float best_loss = __builtin_inff();
    int best = 0;
    for (int s = 0; s < starts; s++) {
        const float l = S[g * starts + s].local_best;
        if (l < best_loss) { best_loss = l; best = s; }
    }
    const float *bp = S[g * starts + best].params;
    float R[9];
    rot6d_to_matrix(bp, R);
    const float sc = expf(bp[9]);
    float *T = transform + (size_t)g * 16;
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) T[a * 4 + b] = R[a * 3 + b] * sc;
        T[a * 4 + 3] = bp[6 + a];
    }
    T[12] = T[13] = T[14] = 0.0f;
    T[15] = 1.0f;
    if (best_params)
        for (int k = 0; k < 10; k++) best_params[(size_t)g * 10 + k] = bp[k];
}

// dst[g, r, :] = src[g, :] for r < rep (count floats per row)
__global__ __launch_bounds__(kQBlock) void pose_replicate_kernel(size_t count, int rep, const float *__restrict__ src,
                                                                 float *__restrict__ dst)
{
    const int g = blockIdx.y;
    for (size_t j = (size_t)blockIdx.x * kQBlock + threadIdx.x; j < count; j += (size_t)gridDim.x * kQBlock) {
        const float v = src[(size_t)g * count + j];
        for (int r = 0; r < rep; r++) dst[((size_t)g * rep + r) * count + j] = v;
    }
}

// ---------------------------------------------------------------------------
// Silhouette ("mask") half of compute_loss_function (diff_obj_pose.py:286-336).
// The reference compares Pulsar renders (pytorch3d, CUDA only, absent, unpinned) of the partial
// cloud WITH ITS COLOURS (:108-134; load_point_cloud returns vert_col for every input the pipeline
// produces, :136-164) and of the posed complete cloud with its colours (:426-433).  This build defines
// its OWN differentiable colour splat with the reference's camera and radii -- restated with the loss
// in oracle/genpc_oracle_geom.c, whose loss is pinned to the reference's own code
// (tests/golden/ref_py_mask_loss.npz) and whose gradient is pinned to torch autograd:
//   Zv = 3 - z;  u = S/2 (1 + 4 x / Zv);  v = S/2 (1 - 4 y / Zv);  rho = S/2 * 4 * radius / Zv
//   a_i(pixel) = min(0.999, max(0, 1 - |pixel centre - (u, v)|^2 / rho^2))
//   O = 1 - prod_i (1 - a_i);  A_ch = sum_i a_i c_i,ch / sum_i a_i;  I_ch = O A_ch      (background 0)
// (coverage-weighted colour, order-independent: Pulsar's softmax in depth is NOT reproduced) and keeps
// the reference's own torch code for what follows the render: per-channel statistical normalisation,
// luminance, sigmoid soft masks, 30 MSE + BCE + 10 Dice (:204-217,261-278,238-259,304-311).
//   mask_project_kernel  every point posed (the transform is fused) and projected once: (u, v, rho); the point's
//                        index goes into the list of every tile its disc's bounding box touches (bin_points_block)
//   mask_splat_kernel    one block per 16 x 16 image tile and scan: the points of the tile's list (every point
//                        of the scan when the list overflowed or the image has no lists), in ascending point
//                        order, are put into LDS with their colours, then every pixel (four
//                        threads each) walks the list and accumulates prod (1 - a), a and a c of the discs
//                        covering it, in list order: no atomics on the image.  Writes the five planes
//                        T = prod (1 - a), D = sum a, N_ch = sum a c_ch and adds the tile's sums of
//                        I_ch and I_ch^2 to the scan's accumulators.  (Measured at 16384 points, single
//                        channel: one thread per point walking its own box 634 us per call -- every
//                        wave pays a full box for its one in-tile lane; a wave per point with LDS
//                        float atomics 36 us, 26 of them in the atomics.)
//   mask_sums_kernel     pixels over many blocks: the 22 sums the loss and its gradient need
//   mask_w_kernel        d loss / d I_ch per pixel folded with the splat's own derivative into the five
//                        per-pixel weights the backward gather needs; the loss itself
//   mask_grad_kernel     one thread per point: gathers the weights over the pixels it covers, chains
//                        through (u, v, rho) to the point and on to (R, s, t): same 13 accumulators as
//                        the Chamfer gradient
//   mask_grad_tile_kernel  opt-in: the same gather per tile from LDS through the tile lists, per-(point, tile)
//                        sums that mask_grad_kernel then adds
constexpr int kMaskTile = 16;
constexpr int kSplatBlock = 1024;  // 16 waves per tile: a wave per point leaves long dependent chains, four waves per SIMD hide them
constexpr int kSplatPer = 16;      // points per thread and round of the splat (16384 points per round)
constexpr int kSplatList = 1024;   // in-tile points drawn per fill of the LDS list (full-scan path; the list path holds a whole tile list: kTileCap)
constexpr float kMaskAmax = 0.999f;
constexpr float kMaskFocal = 4.0f, kMaskEyeZ = 3.0f, kMaskZnear = 1e-4f, kMaskZfar = 5.0f;
constexpr float kLumR = 0.299f, kLumG = 0.587f, kLumB = 0.114f;      // compute_soft_mask, diff_obj_pose.py:273
// Blend 1 = Pulsar's published blending function (Lassner & Zollhoefer, CVPR 2021, eq. 1-2) with the reference's arguments
// (diff_obj_pose.py:126-131,428-433: gamma 1e-2, znear 1e-4, zfar 5, bg 0; opacity 1): a softmax in depth over the discs
// covering a pixel,  I_ch = sum_i a_i e_i c_i,ch / (B + sum_i a_i e_i),  e_i = exp(z_i / gamma),  z_i = (zfar - Zv_i) / (zfar -
// znear),  B = exp(eps / gamma), eps = 1e-10 -- restated, with what is from memory marked, in oracle/genpc_oracle_geom.c and
// pinned there to torch autograd.  The image is kept as five planes like blend 0's: the exponent m every weight of the pixel
// is taken relative to (max(eps, max_i z_i) / gamma), D' = B e^-m + sum a e', N'_ch = sum a e' c_ch with e' = exp(z_i / gamma - m).
constexpr float kPulsarGamma = 1e-2f, kPulsarEps = 1e-10f;
constexpr float kPulsarZe0 = kPulsarEps / kPulsarGamma;                                  // the background's exponent
constexpr float kPulsarZeK = 1.0f / ((kMaskZfar - kMaskZnear) * kPulsarGamma);           // z / gamma = (zfar - Zv) * kPulsarZeK
__device__ __forceinline__ float pulsar_ze(float zv) { return (kMaskZfar - zv) * kPulsarZeK; }

struct SplatPt {
    float u, v, rho, zv;
    bool ok;
};

__device__ __forceinline__ SplatPt splat_project(const float *p, float radius, float hs)
{
    SplatPt o;
    o.zv = kMaskEyeZ - p[2];
    o.ok = o.zv > kMaskZnear && o.zv < kMaskZfar;
    const float iz = 1.0f / o.zv;
    o.u = hs * (1.0f + kMaskFocal * p[0] * iz);
    o.v = hs * (1.0f - kMaskFocal * p[1] * iz);
    o.rho = hs * kMaskFocal * radius * iz;
    return o;
}

// grid (blocks, b): uvr[e, j] = (u, v, rho, 1 / rho^2), rho = -1 for points the camera does not see.
// posed != 0: v is the complete cloud and is posed with params first.
// Per-tile index lists (filled by the projection kernels, read and reset by mask_splat_kernel): bins = { count[b][tiles],
// idx[b][tiles][kTileCap] }.  A point goes to every tile its disc's bounding box touches -- THE test of the splat kernel, so
// a tile's list is exactly its hit set; a tile with more than kSplatCap hits (the count keeps counting) is drawn by the
// full scan.  (Every block used to read all projected points of its image: 205 MB of L2 reads per launch of four
// starts, 43 k ticks for an empty tile.)
constexpr int kTileCap = 8192;            // entries a tile's list holds (a small object: 16384 points over a dozen tiles; 4096: 72.8 ms per call, 8192: 60.8)
constexpr int kSplatCap = 1024;          // the splat takes lists up to this length (one entry per thread, one fill)
constexpr int kRankWords = 2048;         // bitmap ranks in the splat: images of up to 65536 points
constexpr int kRankByCount = 256;        // lists up to this length are ranked by counting
constexpr int kBinTiles = 1024;          // tiles per image the block-level histogram holds (S <= 512); larger images: no bins
constexpr int kBinPer = 4;               // tiles a listed disc may touch (entries carry the slot in two bits)

// f(tile) for every tile the disc's bounding box touches, in (row, column) order -- mask_splat_kernel's test, verbatim
template <class F>
__device__ __forceinline__ void for_each_tile(int S, float u, float v, float rho, F f)
{
    const int T = (S + kMaskTile - 1) / kMaskTile;
    int tx_lo = (int)floorf((u - rho) / (float)kMaskTile) - 1, tx_hi = (int)floorf((u + rho) / (float)kMaskTile) + 1;
    int ty_lo = (int)floorf((v - rho) / (float)kMaskTile) - 1, ty_hi = (int)floorf((v + rho) / (float)kMaskTile) + 1;
    tx_lo = tx_lo < 0 ? 0 : tx_lo; ty_lo = ty_lo < 0 ? 0 : ty_lo;
    tx_hi = tx_hi > T - 1 ? T - 1 : tx_hi; ty_hi = ty_hi > T - 1 ? T - 1 : ty_hi;
    for (int ty = ty_lo; ty <= ty_hi; ty++)
        for (int tx = tx_lo; tx <= tx_hi; tx++) {
            const int tx0 = tx * kMaskTile, ty0 = ty * kMaskTile;
            const int tx1 = min(S, tx0 + kMaskTile) - 1, ty1 = min(S, ty0 + kMaskTile) - 1;
            if (u + rho >= (float)tx0 && u - rho <= (float)(tx1 + 1) && v + rho >= (float)ty0 && v - rho <= (float)(ty1 + 1)) f(ty * T + tx);
        }
}

// number of tiles a disc touches; slot = position of `tile` among them (-1: not touched)
__device__ __forceinline__ int tile_count(int S, float u, float v, float rho, int tile, int &slot)
{
    int cnt = 0, sl = -1;
    for_each_tile(S, u, v, rho, [&](int t) {
        if (t == tile) sl = cnt;
        cnt++;
    });
    slot = sl;
    return cnt;
}

constexpr int kBinPoison = 1 << 30;      // set in a tile's counter by a disc that is in no list (over more than kBinPer tiles)

// Block-level binning of one point per thread (all threads of the block call it; `valid`: this thread has a point):
// the tile counts of the block's points are first accumulated in LDS, ONE global atomic per (block, tile) reserves the
// block's range in the tile's list, then the threads write their entries: point index * 4 + slot, slot = the position
// of the tile among the point's tiles (where the tile pass of the mask gradient leaves the point's partial sums).
// (One global atomic per (point, tile) -- 144 k per launch of four starts, 640 on the counter of a crowded tile -- made
// the 5 us projection kernel 50 us.)  A disc over more than kBinPer tiles is listed nowhere and poisons the counters of
// its tiles: the splat draws those by the full scan, the mask gradient takes such a point by itself.
__device__ __forceinline__ void bin_points_block(int *__restrict__ cnt, int *__restrict__ idx, int *s_cnt, int *s_base, int S, bool valid,
                                                 int j, float u, float v, float rho)
{
    const int T = (S + kMaskTile - 1) / kMaskTile, tiles = T * T;
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) s_cnt[t] = 0;
    __syncthreads();
    int my_tile[kBinPer], my_pos[kBinPer], nmine = 0;
    if (valid && rho > 0.0f) {
        int none;
        const int total = tile_count(S, u, v, rho, -1, none);
        if (total <= kBinPer) {
            for_each_tile(S, u, v, rho, [&](int t) {
                if (nmine < kBinPer) {      // (always: keeps the arrays in registers)
                    my_tile[nmine] = t;
                    my_pos[nmine] = atomicAdd(&s_cnt[t], 1);
                    nmine++;
                }
            });
        } else {
            for_each_tile(S, u, v, rho, [&](int t) { atomicOr(&cnt[t], kBinPoison); });
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) {
        const int c = s_cnt[t];
        s_base[t] = c ? (atomicAdd(&cnt[t], c) & (kBinPoison - 1)) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBinPer; k++) {
        if (k < nmine) {
            const int pos = s_base[my_tile[k]] + my_pos[k];
            if (pos < kTileCap) idx[(size_t)my_tile[k] * kTileCap + pos] = j * 4 + k;
        }
    }
    __syncthreads();
}
__host__ __device__ __forceinline__ size_t bins_tiles(int S)
{
    const size_t T = (size_t)((S + kMaskTile - 1) / kMaskTile);
    return T * T;
}

__global__ __launch_bounds__(kQBlock) void mask_project_kernel(int n, const float *__restrict__ v,
                                                               const float *__restrict__ center, int cstride,
                                                               const float *__restrict__ params, int pstride, int posed,
                                                               float radius, int S, float4 *__restrict__ uvr, int *__restrict__ bins,
                                                               float *__restrict__ zex)
{
    const int e = blockIdx.y;
    v += (size_t)e * n * 3;
    uvr += (size_t)e * n;
    if (zex) zex += (size_t)e * n;       // blend 1: the depth exponent z / gamma of every point
    int *bin_cnt = bins ? bins + (size_t)e * bins_tiles(S) : nullptr;
    int *bin_idx = bins ? bins + (size_t)gridDim.y * bins_tiles(S) + (size_t)e * bins_tiles(S) * kTileCap : nullptr;
    float R[9], s = 1.0f, c[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    if (posed) {
        center += (size_t)e * cstride;
        params += (size_t)e * pstride;
        rot6d_to_matrix(params, R);
        s = expf(params[9]);
        c[0] = center[0]; c[1] = center[1]; c[2] = center[2];
        t[0] = params[6]; t[1] = params[7]; t[2] = params[8];
    }
    const float hs = 0.5f * S;
    __shared__ int s_cnt[kBinTiles], s_base[kBinTiles];
    for (int j0 = blockIdx.x * kQBlock; j0 < n; j0 += gridDim.x * kQBlock) {        // (block-uniform: bin_points_block has barriers)
        const int j = j0 + threadIdx.x;
        const bool valid = j < n;
        const int jj = valid ? j : n - 1;
        float p[3] = {v[(size_t)jj * 3 + 0], v[(size_t)jj * 3 + 1], v[(size_t)jj * 3 + 2]};
        if (posed) {
            float o[3];
            pose_point(R, s, c, t, p[0], p[1], p[2], o);
            p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
        }
        const SplatPt q = splat_project(p, radius, hs);
        if (valid) uvr[j] = make_float4(q.u, q.v, q.ok ? q.rho : -1.0f, q.ok ? 1.0f / (q.rho * q.rho) : 0.0f);
        if (valid && zex) zex[j] = pulsar_ze(q.zv);
        if (bin_cnt) bin_points_block(bin_cnt, bin_idx, s_cnt, s_base, S, valid && q.ok, j, q.u, q.v, q.rho);
    }
}

// pose_transform_kernel and mask_project_kernel (posed) in one launch: both walk the complete cloud with the same
// parameters (one 5 us launch less per Adam step; the same arithmetic, so the same bits in pts and uvr)
__global__ __launch_bounds__(kQBlock) void pose_transform_project_kernel(int n, const float *__restrict__ v,
                                                                         const float *__restrict__ center, int cstride,
                                                                         const float *__restrict__ params, int pstride,
                                                                         float *__restrict__ pts, float radius, int S,
                                                                         float4 *__restrict__ uvr, int *__restrict__ bins,
                                                                         float *__restrict__ zex, PoseFuse fu)
{
    __shared__ float s_par[10];
    const int e = blockIdx.y;
    v += (size_t)e * n * 3;
    pts += (size_t)e * n * 3;
    uvr += (size_t)e * n;
    if (zex) zex += (size_t)e * n;
    int *bin_cnt = bins ? bins + (size_t)e * bins_tiles(S) : nullptr;
    int *bin_idx = bins ? bins + (size_t)gridDim.y * bins_tiles(S) + (size_t)e * bins_tiles(S) * kTileCap : nullptr;
    center += (size_t)e * cstride;
    params = pose_fused_params(fu, e, params + (size_t)e * pstride, s_par);      // (the previous step's update, when fused)
    float R[9];
    rot6d_to_matrix(params, R);
    const float s = expf(params[9]);
    const float c[3] = {center[0], center[1], center[2]};
    const float t[3] = {params[6], params[7], params[8]};
    const float hs = 0.5f * S;
    __shared__ int s_cnt[kBinTiles], s_base[kBinTiles];
    for (int j0 = blockIdx.x * kQBlock; j0 < n; j0 += gridDim.x * kQBlock) {        // (block-uniform: bin_points_block has barriers)
        const int j = j0 + threadIdx.x;
        const bool valid = j < n;
        const int jj = valid ? j : n - 1;
        float o[3];
        pose_point(R, s, c, t, v[(size_t)jj * 3 + 0], v[(size_t)jj * 3 + 1], v[(size_t)jj * 3 + 2], o);
        const SplatPt q = splat_project(o, radius, hs);
        if (valid) {
            pts[(size_t)j * 3 + 0] = o[0];
            pts[(size_t)j * 3 + 1] = o[1];
            pts[(size_t)j * 3 + 2] = o[2];
            uvr[j] = make_float4(q.u, q.v, q.ok ? q.rho : -1.0f, q.ok ? 1.0f / (q.rho * q.rho) : 0.0f);
            if (zex) zex[j] = pulsar_ze(q.zv);
        }
        if (bin_cnt) bin_points_block(bin_cnt, bin_idx, s_cnt, s_base, S, valid && q.ok, j, q.u, q.v, q.rho);
    }
    if (fu.ctr) pose_publish_block(fu.ctr, 0);
}

// The image of a scan is kept as five planes of P = S * S floats: T, D, N_r, N_g, N_b (header comment).
// `direct` = 1 images (genpc_mask_loss: the caller supplies I itself) hold I_r, I_g, I_b in planes 0..2; `direct` = 2: blend 1's
// planes m, D', N'_ch (T carries m; O = 1; A = I = N' / D').
struct PxImg {
    float I[3];
    float T, O, iD;        // exp(L), 1 - exp(L), 1 / D (0 where no disc covers the pixel)
    float A[3];            // N_ch / D
};

__device__ __forceinline__ PxImg load_pixel(const float *__restrict__ pl, int P, int q, int direct)
{
    PxImg o;
    if (direct == 1) {
        o.I[0] = pl[q]; o.I[1] = pl[P + q]; o.I[2] = pl[2 * P + q];
        o.T = 0.0f; o.O = 1.0f; o.iD = 0.0f;
        o.A[0] = o.A[1] = o.A[2] = 0.0f;
        return o;
    }
    const float d = pl[P + q];
    o.T = pl[q];
    if (direct == 2) {
        o.O = 1.0f;
        o.iD = 1.0f / d;          // (the background term keeps D' > 0)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) o.I[ch] = o.A[ch] = pl[(2 + ch) * P + q] * o.iD;
        return o;
    }
    o.O = 1.0f - o.T;
    o.iD = d > 0.0f ? 1.0f / d : 0.0f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        o.A[ch] = pl[(2 + ch) * P + q] * o.iD;
        o.I[ch] = o.O * o.A[ch];
    }
    return o;
}

// Workgroups go to the 8 XCDs round-robin by linear id, each XCD with its own 4 MB L2.  A 1-D launch of gx blocks for each
// of nb images, mapped so that an image's blocks share as few XCDs as possible: its W planes (1 MB at 224 x 224) are then
// gathered from ONE L2 instead of all eight (32 images in lock-step: 32 MB through every 4 MB L2; measured on the per-point
// gradient: 168 -> 152 us at 32 images, 30.7 -> 29.3 at 4).
struct XcdBlock { int e, x; };
__device__ __forceinline__ XcdBlock xcd_block(int gx, int nb)
{
    const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
    XcdBlock r;
    if ((nb & 7) == 0) {                       // whole images per XCD
        r.e = 8 * (k / gx) + xcd;
        r.x = k % gx;
    } else if (nb < 8 && 8 % nb == 0 && gx % (8 / nb) == 0) {      // 8 / nb XCDs per image
        r.e = xcd % nb;
        r.x = k * (8 / nb) + xcd / nb;
    } else {
        r.e = lin / gx;
        r.x = lin % gx;
    }
    return r;
}

// what a pixel accumulates of one disc that covers it (coverage ac in (0, kMaskAmax], colours cr / cg / cb, depth exponent ze)
template <int BLEND>
__device__ __forceinline__ void splat_fold(float &tr, float &sd, float &sr, float &sg, float &sb, float ac, float cr, float cg, float cb, float ze)
{
    if (BLEND == 0) {
        tr *= 1.0f - ac;
        sd += ac;
        sr += ac * cr;
        sg += ac * cg;
        sb += ac * cb;
    } else {
        // softmax in depth, one pass: tr is the running maximum of the exponents, the sums are relative to it
        if (ze > tr) {
            const float sc = __expf(tr - ze);
            sd *= sc; sr *= sc; sg *= sc; sb *= sc;
            tr = ze;
        }
        const float w = ac * __expf(ze - tr);
        sd += w;
        sr += w * cr;
        sg += w * cg;
        sb += w * cb;
    }
}

// LDS layout of a listed disc.  Blend 0: list = (u, v, 1 / rho^2, red), list_gb = (green, blue).  Blend 1: list = (u, v, 1 / rho^2,
// depth exponent or weight), list_c = (red, green, blue, -) -- still ONE 16-byte read per disc tested and one more per disc that
// covers the pixel (a third, 4-byte read for the exponent made the gather, which is bound by its LDS reads, 30 % slower).
// Blend 1 without an exponential per (pixel, disc): after a fill of the LDS list (list[k].w = the entries' depth exponents) the
// block takes the fill's largest exponent m_f and turns the entries into weights exp(z_k / gamma - m_f) -- ONE exponential
// per listed disc instead of one per pixel it covers (the per-pair form cost 13 us of a 43 us launch, 3.8 ms of a 57 ms
// completed scan) --, every pixel moves its sums to the reference max(its own, m_f) once, and the gather multiplies.
// Exactness is kept by falling back to splat_fold's per-pair form for a fill whose exponents span more than kPulsarSpan (its
// far entries' weights would underflow: a tile that holds points at both ends of the frustum).  All threads of the block call
// it between the fill's barrier and the gather; zlo / zhi: the smallest / largest exponent this thread wrote (inf / 0: none).
constexpr float kPulsarSpan = 60.0f;
__device__ __forceinline__ bool splat_fill_weights(float4 *list, int cnt, int *s_zmm, float zlo, float zhi, float &tr, float &sd, float &sr,
                                                   float &sg, float &sb, float &fscale)
{
    // (exponents are >= 0: the integer order of their bit patterns is theirs)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        zlo = fminf(zlo, __shfl_xor(zlo, off, kWave));
        zhi = fmaxf(zhi, __shfl_xor(zhi, off, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMax(&s_zmm[0], __float_as_int(zhi));
        atomicMin(&s_zmm[1], __float_as_int(zlo));
    }
    __syncthreads();
    const float m_f = __int_as_float(s_zmm[0]), z_min = __int_as_float(s_zmm[1]);
    const bool fast = m_f - z_min < kPulsarSpan;
    if (fast) {
        for (int k = threadIdx.x; k < cnt; k += blockDim.x) list[k].w = __expf(list[k].w - m_f);
        const float m_new = fmaxf(tr, m_f);
        const float sc = __expf(tr - m_new);
        sd *= sc; sr *= sc; sg *= sc; sb *= sc;
        fscale = __expf(m_f - m_new);
        tr = m_new;
    }
    __syncthreads();
    if (threadIdx.x == 0) { s_zmm[0] = 0; s_zmm[1] = 0x7f800000; }      // for the next fill (whose writes come behind a barrier)
    return fast;
}

// The full scan of mask_splat_kernel: every point of the image against the tile, sixteen per thread and round (see the
// comments inside; called for a tile whose list overflowed, or when the launch has no lists).
template <int BLEND>
__device__ __attribute__((noinline)) void splat_full_scan(int n, const float4 *__restrict__ uvr, const float *__restrict__ col,
                                                          const float *__restrict__ zex, int tx0, int ty0, int tx1, int ty1, float pxc,
                                                          float pyc, int share, int lane, int wave, float &tr, float &sd, float &sr,
                                                          float &sg, float &sb, float4 *list, float2 *list_gb, float4 *list_c, int *s_tab,
                                                          int *s_cntp, int *s_zmm)
{
    for (int j0 = 0; j0 < n; j0 += kSplatBlock * kSplatPer) {
        // the bounding-box tests keep only a bit per point (four points in flight at a time): the kernel must fit 64
        // VGPRs so that TWO 1024-thread blocks share a CU -- with all sixteen points of a thread in registers it needed
        // 128, one block per CU, and a lock-step batch of 8 scans x 4 starts (6272 blocks) ran 8 % slower
        unsigned hit = 0;
#pragma unroll
        for (int i0 = 0; i0 < kSplatPer; i0 += 4) {
            float4 q[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = j0 + (i0 + i) * kSplatBlock + threadIdx.x;
                q[i] = j < n ? uvr[j] : make_float4(0.0f, 0.0f, -1.0f, 0.0f);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // the disc's bounding box against the tile
                if (q[i].z > 0.0f && q[i].x + q[i].z >= (float)tx0 && q[i].x - q[i].z <= (float)(tx1 + 1) &&
                    q[i].y + q[i].z >= (float)ty0 && q[i].y - q[i].z <= (float)(ty1 + 1))
                    hit |= 1u << (i0 + i);
            }
        }
        // The list is filled in ascending point order -- rank = number of hits before this one in (i, wave, lane)
        // order, which is the order of the point indices j = j0 + i * kSplatBlock + tid -- so that a pixel sums its
        // discs in the same order in every run (an arrival-order list made the whole alignment loop irreproducible:
        // last-bit differences in the image flip a soft-mask pixel in or out of fp32 sigmoid saturation a few steps
        // later).  One table of per-(i, wave) counts, one scan by wave 0, two barriers per round of 16384 points.
        // a lane's rank among its wave's hits of step i, eight bits each (the ballots themselves are not kept: 32 SGPRs)
        unsigned long long rk[kSplatPer / 8];
#pragma unroll
        for (int w8 = 0; w8 < kSplatPer / 8; w8++) rk[w8] = 0ull;
#pragma unroll
        for (int i = 0; i < kSplatPer; i++) {
            const unsigned long long bl = __ballot((hit >> i) & 1u);
            if (lane == 0) s_tab[i * (kSplatBlock / kWave) + wave] = __popcll(bl);
            rk[i >> 3] |= (unsigned long long)__popcll(bl & ((1ull << lane) - 1ull)) << (8 * (i & 7));
        }
        __syncthreads();
        if (wave == 0) {
            // exclusive scan of the 256 counts (four per lane, in table order)
            static_assert(kSplatPer * (kSplatBlock / kWave) == 4 * kWave, "four table entries per lane");
            int c4[4], tot = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { c4[q] = s_tab[lane * 4 + q]; tot += c4[q]; }
            int incl = tot;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const int o = __shfl_up(incl, off, kWave);
                incl += lane >= off ? o : 0;
            }
            int run = incl - tot;
#pragma unroll
            for (int q = 0; q < 4; q++) { s_tab[lane * 4 + q] = run; run += c4[q]; }
            if (lane == kWave - 1) *s_cntp = incl;
        }
        __syncthreads();
        const int total = *s_cntp;
        // the list holds kSplatList entries: a crowded tile is drawn in several fills
        for (int f0 = 0; f0 < total; f0 += kSplatList) {
            float zlo = __builtin_inff(), zhi = 0.0f;
#pragma unroll
            for (int i = 0; i < kSplatPer; i++) {
                if ((hit >> i) & 1u) {
                    const int slot = s_tab[i * (kSplatBlock / kWave) + wave] + (int)((rk[i >> 3] >> (8 * (i & 7))) & 0xffull) - f0;
                    if (slot >= 0 && slot < kSplatList) {
                        const int j = j0 + i * kSplatBlock + threadIdx.x;
                        const float4 qh = uvr[j];                 // (a hit is rare: read again rather than kept)
                        float cr = 1.0f, cg = 1.0f, cb = 1.0f;
                        if (col) { cr = col[(size_t)j * 3 + 0]; cg = col[(size_t)j * 3 + 1]; cb = col[(size_t)j * 3 + 2]; }
                        if (BLEND) {
                            const float ze = zex[j];
                            list[slot] = make_float4(qh.x, qh.y, qh.w, ze);
                            list_c[slot] = make_float4(cr, cg, cb, 0.0f);
                            zlo = fminf(zlo, ze); zhi = fmaxf(zhi, ze);
                        } else {
                            list[slot] = make_float4(qh.x, qh.y, qh.w, cr);
                            list_gb[slot] = make_float2(cg, cb);
                        }
                    }
                }
            }
            __syncthreads();
            const int cnt = min(total - f0, kSplatList);
            float fscale = 1.0f;
            const bool fast = BLEND && splat_fill_weights(list, cnt, s_zmm, zlo, zhi, tr, sd, sr, sg, sb, fscale);
            for (int k = share; k < cnt; k += 4) {
                const float4 p = list[k];
                const float dx = pxc - p.x, dy = pyc - p.y;
                const float a = 1.0f - (dx * dx + dy * dy) * p.z;
                if (a > 0.0f) {
                    if (BLEND) {
                        const float4 c = list_c[k];
                        if (fast) {
                            const float w = fminf(a, kMaskAmax) * (p.w * fscale);
                            sd += w; sr += w * c.x; sg += w * c.y; sb += w * c.z;
                        } else {
                            splat_fold<1>(tr, sd, sr, sg, sb, fminf(a, kMaskAmax), c.x, c.y, c.z, p.w);
                        }
                    } else {
                        const float2 gb = list_gb[k];
                        splat_fold<0>(tr, sd, sr, sg, sb, fminf(a, kMaskAmax), p.w, gb.x, gb.y, 0.0f);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// tools/splat_timeline.py builds a private copy of the library with -DGENPC_SPLAT_TIMELINE: thread 0 of every block stamps the
// 100 MHz wall clock at the phase boundaries of mask_splat_kernel (nothing of this exists in the shipped library)
#ifdef GENPC_SPLAT_TIMELINE
__device__ unsigned long long g_splat_tl[4096 * 8];
#define GENPC_STL(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_splat_tl[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define GENPC_STL_VAL(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_splat_tl[blockIdx.x * 8 + (k)] = (unsigned long long)(v); } while (0)
#else
#define GENPC_STL(k) do {} while (0)
#define GENPC_STL_VAL(k, v) do {} while (0)
#endif
// 1-D grid of tiles * nb blocks (xcd_block).  col: [nb, n, 3] colours or nullptr (white).  BLEND 1: zex [nb, n] depth exponents.
template <int BLEND>
__global__ __launch_bounds__(kSplatBlock, 8) void mask_splat_kernel(int n, const float4 *__restrict__ uvr,
                                                                 const float *__restrict__ col, int S,
                                                                 float *__restrict__ planes, double *__restrict__ accum,
                                                                 int *__restrict__ bins, int keep_bins, int nb,
                                                                 const float *__restrict__ zex, int *__restrict__ clean = nullptr)
{
    static_assert(kSplatBlock == 4 * kMaskTile * kMaskTile, "four threads per pixel of the tile");
    __shared__ float part[5][4][kMaskTile * kMaskTile];
    __shared__ float4 list[kSplatList];       // u, v, 1 / rho^2, red
    __shared__ float2 list_gb[BLEND ? 1 : kSplatList];    // blend 0: green, blue
    __shared__ float4 list_c[BLEND ? kSplatList : 1];     // blend 1: red, green, blue (list[k].w: depth exponent, or weight exp(z / gamma - m_f))
    __shared__ int s_zmm[2];                              // blend 1: a fill's largest / smallest exponent (bits)
    if (BLEND && threadIdx.x == 0) { s_zmm[0] = 0; s_zmm[1] = 0x7f800000; }
    GENPC_STL(0);
    __shared__ int s_cnt;
    __shared__ int s_tab[kSplatPer * (kSplatBlock / kWave)];      // per (i, wave): hits, then their exclusive prefix
    const XcdBlock xb = xcd_block(((S + kMaskTile - 1) / kMaskTile) * ((S + kMaskTile - 1) / kMaskTile), nb);
    const int e = xb.e, tile = xb.x;
    const int P = S * S;
    const int tiles_x = (S + kMaskTile - 1) / kMaskTile;
    const int tx0 = (tile % tiles_x) * kMaskTile, ty0 = (tile / tiles_x) * kMaskTile;
    const int tx1 = min(S, tx0 + kMaskTile) - 1, ty1 = min(S, ty0 + kMaskTile) - 1;
    uvr += (size_t)e * n;
    if (col) col += (size_t)e * n * 3;
    if (BLEND) zex += (size_t)e * n;
    planes += (size_t)e * 5 * P;
    if (accum) accum += (size_t)e * kAcc;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    // thread = (pixel of the tile, one of four interleaved shares of the point list)
    const int pix = threadIdx.x & (kMaskTile * kMaskTile - 1), share = threadIdx.x / (kMaskTile * kMaskTile);
    const float pxc = (float)(tx0 + (pix & (kMaskTile - 1))) + 0.5f, pyc = (float)(ty0 + pix / kMaskTile) + 0.5f;
    float tr = BLEND ? kPulsarZe0 : 1.0f, sd = 0.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f;      // transmittance prod (1 - a) (blend 1: the running maximum exponent), sums of a and a c
    // Every block reads every point of its scan (16 x 16 tiles: 196 blocks per scan), kSplatPer per thread
    // loaded together; the points whose disc touches the tile are compacted into LDS (one LDS atomic per wave
    // and step), then GATHERED: every pixel walks the list (broadcast reads) and accumulates the discs
    // that cover it, in list order -- no atomics on the image.  (Scattering with LDS float atomics ran at
    // ~0.6 adds per clock per CU: 26 us of the kernel's 36 for 758 points in the busiest tile.)
    // The tile's own list (bins, filled by the projection kernel): its indices are put in ascending order -- the fixed
    // summation order of the gather -- by counting ranks, the entries fetched, and the full scan below is skipped.  A tile
    // whose list overflowed (or a launch without bins) takes the full scan.
    // (the counting ranks' copy of the indices is dead before the first fill writes the colours: blend 1 keeps it in their array --
    //  with an array of its own the block took 83 228 bytes of LDS, ONE block per CU instead of two: 46 us against 35)
    __shared__ __attribute__((aligned(16))) int s_sort_own[BLEND ? 4 : kSplatCap];
    int *s_sort = BLEND ? (int *)&list_c[0] : s_sort_own;
    static_assert(sizeof(float4) * kSplatList >= sizeof(int) * kSplatCap, "the indices fit the colour array");
    __shared__ unsigned char s_mask[kSplatCap];
    __shared__ unsigned s_bits[kRankWords];
    __shared__ unsigned short s_pref[kRankWords];
    __shared__ unsigned short s_strip[4][kSplatCap];
    __shared__ int s_wc[kSplatBlock / kWave][4], s_stot[4];
    static_assert(kMaskTile == 16 && kSplatBlock / kWave == 16, "a wave = four rows of the tile, four shares");
    int binned = -1;
    int *bin_cnt = bins ? bins + (size_t)e * bins_tiles(S) : nullptr;
    const int *bin_idx = bins ? bins + (size_t)nb * bins_tiles(S) + ((size_t)e * bins_tiles(S) + tile) * kTileCap : nullptr;
    int my_ent = 0;
    if (bins) {
        binned = bin_cnt[tile];
        if (threadIdx.x < kSplatCap) my_ent = bin_idx[threadIdx.x];      // (before the count is known: one round trip, not two)
    }
    const bool by_list = binned >= 0 && (binned <= kSplatCap || (binned <= kTileCap && n <= 32 * kRankWords));
    GENPC_STL_VAL(7, binned + 1);
    GENPC_STL(1);
    // An empty tile whose pixels already hold the background leaves at once (round 6): three of four tiles of a scan's image are
    // empty at every step, and each held a 1024-thread, 79 KB slot for 9.5 us to rewrite what was there (tools/splat_timeline.py) --
    // the launch's other blocks waited up to 13 us for a slot.  Its sums of I and I^2 are zero: nothing to add.
    int *clean_flag = clean && bins ? clean + (size_t)e * bins_tiles(S) + tile : nullptr;
    if (clean_flag && binned == 0) {
        const int was_clean = *clean_flag;          // (block-uniform; the only writer of this word is this block, below)
        if (was_clean) { GENPC_STL(6); return; }
    }
    if (by_list && binned > 0) {
        static_assert(kSplatCap <= kSplatBlock && kSplatCap <= kSplatList && kSplatCap <= kTileCap, "one list entry per thread and fill");
        const int L = binned, L4 = (L + 3) & ~3;
        const bool bitmap = L > kRankByCount && n <= 32 * kRankWords;      // (by_list: L <= kSplatCap otherwise)
        int rank0 = 0;      // counting path: the rank of this thread's (only) entry
        if (bitmap) {
            // a long list: rank = number of set bits below the point's own in a bitmap of the image's points (the
            // counting below is O(L^2): 40 of the launch's 220 us with 32 images, in the crowded tiles)
            const int words = (n + 31) >> 5;
            for (int w = threadIdx.x; w < words; w += kSplatBlock) s_bits[w] = 0u;
            __syncthreads();
            for (int i = threadIdx.x; i < L; i += kSplatBlock) {
                const int j = (i < kSplatCap ? my_ent : bin_idx[i]) >> 2;
                atomicOr(&s_bits[j >> 5], 1u << (j & 31));
            }
            __syncthreads();
            // exclusive prefix of the words' popcounts: kRankWords / kSplatBlock consecutive words per thread
            constexpr int kWPT = kRankWords / kSplatBlock;
            int c[kWPT], tot = 0;
#pragma unroll
            for (int q = 0; q < kWPT; q++) {
                const int w = threadIdx.x * kWPT + q;
                c[q] = w < words ? __popc(s_bits[w]) : 0;
                tot += c[q];
            }
            int incl = tot;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const int o = __shfl_up(incl, off, kWave);
                incl += lane >= off ? o : 0;
            }
            if (lane == kWave - 1) s_wc[wave][0] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; w++) base += s_wc[w][0];
            int run = base + incl - tot;
#pragma unroll
            for (int q = 0; q < kWPT; q++) {
                const int w = threadIdx.x * kWPT + q;
                if (w < words) s_pref[w] = (unsigned short)run;
                run += c[q];
            }
        } else {
            const int myj = (int)threadIdx.x < L ? my_ent >> 2 : 0x7fffffff;
            if ((int)threadIdx.x < L4) s_sort[threadIdx.x] = myj;
            __syncthreads();
            if ((int)threadIdx.x < L) {
                for (int k = 0; k < L4; k += 4) {
                    const int4 o = *(const int4 *)&s_sort[k];
                    rank0 += (o.x < myj) + (o.y < myj) + (o.z < myj) + (o.w < myj);
                }
            }
        }
        // A list of several fills: the point indices in sorted order first (16-bit: the bitmap path holds n <= 65536), in the
        // LDS of `part`, which is not needed before the epilogue -- every fill then takes its 1024 entries from there instead
        // of walking the whole list for them (a tile of 5000 entries: five fills of five dependent global reads per thread).
        static_assert(sizeof(part) >= kTileCap * sizeof(unsigned short) && 32 * kRankWords <= 65536, "the sorted indices fit `part`");
        unsigned short *s_sorted = (unsigned short *)&part[0][0][0];
        const bool presorted = L > kSplatCap;      // (implies bitmap)
        if (presorted) {
            __syncthreads();      // s_pref is ready
            for (int i = threadIdx.x; i < L; i += kSplatBlock) {
                const int myj = (i < kSplatCap ? my_ent : bin_idx[i]) >> 2;
                s_sorted[s_pref[myj >> 5] + __popc(s_bits[myj >> 5] & ((1u << (myj & 31)) - 1u))] = (unsigned short)myj;
            }
        }
        // the sorted list is drawn kSplatCap entries at a time (one fill, unless the tile is crowded beyond that)
        for (int f0 = 0; f0 < L; f0 += kSplatCap) {
            __syncthreads();      // ranks ready / the previous fill's gather is done with the list
            const int Lf = min(kSplatCap, L - f0);
            float zlo = __builtin_inff(), zhi = 0.0f;
            for (int i = threadIdx.x; i < (presorted ? Lf : L); i += kSplatBlock) {
                int myj, rank;
                if (presorted) {
                    myj = s_sorted[f0 + i];
                    rank = i;
                } else {
                    myj = (i < kSplatCap ? my_ent : bin_idx[i]) >> 2;
                    rank = (bitmap ? s_pref[myj >> 5] + __popc(s_bits[myj >> 5] & ((1u << (myj & 31)) - 1u)) : rank0) - f0;
                }
                if (rank < 0 || rank >= Lf) continue;
                const float4 qh = uvr[myj];
                float cr = 1.0f, cg = 1.0f, cb = 1.0f;
                if (col) { cr = col[(size_t)myj * 3 + 0]; cg = col[(size_t)myj * 3 + 1]; cb = col[(size_t)myj * 3 + 2]; }
                if (BLEND) {
                    const float ze = zex[myj];
                    list[rank] = make_float4(qh.x, qh.y, qh.w, ze);
                    list_c[rank] = make_float4(cr, cg, cb, 0.0f);
                    zlo = fminf(zlo, ze); zhi = fmaxf(zhi, ze);
                } else {
                    list[rank] = make_float4(qh.x, qh.y, qh.w, cr);
                    list_gb[rank] = make_float2(cg, cb);
                }
                // the strips (four rows of the tile = the 64 pixels of one wave) the disc can reach
                unsigned m = 0;
#pragma unroll
                for (int st = 0; st < 4; st++)
                    if (qh.y + qh.z >= (float)(ty0 + 4 * st) && qh.y - qh.z <= (float)(ty0 + 4 * st + 4)) m |= 1u << st;
                s_mask[rank] = (unsigned char)m;
            }
            // blend 1: the fill's largest / smallest exponent ride on the barriers the strips need anyway (splat_fill_weights,
            // the full scan's form, pays two block-wide barriers of sixteen waves for them: 11 us of this kernel's 46)
            if (BLEND) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    zlo = fminf(zlo, __shfl_xor(zlo, off, kWave));
                    zhi = fmaxf(zhi, __shfl_xor(zhi, off, kWave));
                }
                if (lane == 0 && zhi >= zlo) {          // (exponents are >= 0: the integer order of their bit patterns is theirs)
                    atomicMax(&s_zmm[0], __float_as_int(zhi));
                    atomicMin(&s_zmm[1], __float_as_int(zlo));
                }
            }
            __syncthreads();
            GENPC_STL(2);
            bool fast = false;
            float fscale = 1.0f;
            if (BLEND) {
                const float m_f = __int_as_float(s_zmm[0]);
                fast = m_f - __int_as_float(s_zmm[1]) < kPulsarSpan;
                if (fast) {
                    // one exponential per listed disc (the strips' barriers below stand between this and the gather)
                    if ((int)threadIdx.x < Lf) list[threadIdx.x].w = __expf(list[threadIdx.x].w - m_f);
                    const float m_new = fmaxf(tr, m_f);
                    const float sc = __expf(tr - m_new);
                    sd *= sc; sr *= sc; sg *= sc; sb *= sc;
                    fscale = __expf(m_f - m_new);
                    tr = m_new;
                }
            }
            // Per-strip sublists of the sorted list, in its order: a wave walks the discs that can reach its four rows only
            // (about half of the tile's).  Ballot ranks inside a wave, a 16 x 4 table of wave counts, its prefix by four threads.
            {
                const unsigned m = (int)threadIdx.x < Lf ? s_mask[threadIdx.x] : 0u;
                int pre[4];
#pragma unroll
                for (int st = 0; st < 4; st++) {
                    const unsigned long long bl = __ballot((m >> st) & 1u);
                    pre[st] = __popcll(bl & ((1ull << lane) - 1ull));
                    if (lane == 0) s_wc[wave][st] = __popcll(bl);
                }
                __syncthreads();
                if (BLEND && threadIdx.x == kWave) { s_zmm[0] = 0; s_zmm[1] = 0x7f800000; }      // (read by all before the barrier above; the next fill's come behind the loop's)
                if (threadIdx.x < 4) {
                    int run = 0;
                    for (int w = 0; w < kSplatBlock / kWave; w++) {
                        const int c = s_wc[w][threadIdx.x];
                        s_wc[w][threadIdx.x] = run;
                        run += c;
                    }
                    s_stot[threadIdx.x] = run;
                }
                __syncthreads();
#pragma unroll
                for (int st = 0; st < 4; st++)
                    if ((m >> st) & 1u) s_strip[st][s_wc[wave][st] + pre[st]] = (unsigned short)threadIdx.x;
                __syncthreads();
            }
            GENPC_STL(3);
            const int strip = wave & 3;      // (pix = tid & 255: wave w holds rows 4 (w & 3) .. + 3, share w >> 2)
            const int LS = s_stot[strip];
            for (int k2 = share; k2 < LS; k2 += 4) {
                const int k = s_strip[strip][k2];
                const float4 p = list[k];
                const float dx = pxc - p.x, dy = pyc - p.y;
                const float a = 1.0f - (dx * dx + dy * dy) * p.z;
                if (a > 0.0f) {
                    if (BLEND) {
                        const float4 c = list_c[k];
                        if (fast) {
                            const float w = fminf(a, kMaskAmax) * (p.w * fscale);
                            sd += w; sr += w * c.x; sg += w * c.y; sb += w * c.z;
                        } else {
                            splat_fold<1>(tr, sd, sr, sg, sb, fminf(a, kMaskAmax), c.x, c.y, c.z, p.w);
                        }
                    } else {
                        const float2 gb = list_gb[k];
                        splat_fold<0>(tr, sd, sr, sg, sb, fminf(a, kMaskAmax), p.w, gb.x, gb.y, 0.0f);
                    }
                }
            }
        }
    }
    // the count is reset for the next launch by the last kernel that reads the lists: this one, or the mask gradient's tile pass
    __syncthreads();
    GENPC_STL(4);
    if (bins && !keep_bins && threadIdx.x == 0) bin_cnt[tile] = 0;
    // (the full scan of a tile without a usable list is a function of its own, NOT inlined: its sixteen-points-per-thread
    // bookkeeping is what overflowed the 64 registers the two-blocks-per-CU launch allows -- 59 spilled VGPRs, ten scratch
    // instructions of them on the path of a tile that has its list; VERDICT r4 weak #8.  Here the spills stay in the callee.)
    if (!by_list)
        splat_full_scan<BLEND>(n, uvr, col, zex, tx0, ty0, tx1, ty1, pxc, pyc, share, lane, wave, tr, sd, sr, sg, sb, list, list_gb, list_c, s_tab, &s_cnt, s_zmm);
    part[0][share][pix] = tr;
    part[1][share][pix] = sd;
    part[2][share][pix] = sr;
    part[3][share][pix] = sg;
    part[4][share][pix] = sb;
    __syncthreads();
    __shared__ float s_I[3][kMaskTile * kMaskTile];
    if (threadIdx.x < kMaskTile * kMaskTile) {
        float I3[3] = {0.0f, 0.0f, 0.0f};
        const int r = ty0 + pix / kMaskTile, cc = tx0 + (pix & (kMaskTile - 1));
        if (r < S && cc < S) {
            float w[5];
            // plane 0 is the transmittance T = prod (1 - a) itself (round 2 accumulated log(1 - a) per covering disc and
            // took exp(L) wherever the plane was read: the logarithm was a third of the gather's instructions, and the
            // gather is what the kernel's time is -- 33 M pixel-disc tests per step of four starts).  (Per-strip index lists -- a
            // wave walks only the discs reaching its four rows -- cost three more barriers per FILL of the full scan: 42.0 ->
            // 44.3 ms per call there; the list path above builds them once per tile and keeps them.)
            if (BLEND) {
                // the four shares' sums are relative to their own maxima: to the common one, plus the background's weight
                const float m4 = fmaxf(fmaxf(part[0][0][pix], part[0][1][pix]), fmaxf(part[0][2][pix], part[0][3][pix]));
                float sc[4];
#pragma unroll
                for (int k = 0; k < 4; k++) sc[k] = __expf(part[0][k][pix] - m4);
                w[0] = m4;
                planes[(size_t)r * S + cc] = m4;
#pragma unroll
                for (int k = 1; k < 5; k++) {
                    w[k] = (part[k][0][pix] * sc[0] + part[k][1][pix] * sc[1]) + (part[k][2][pix] * sc[2] + part[k][3][pix] * sc[3]);
                    if (k == 1) w[k] += __expf(kPulsarZe0 - m4);
                    planes[(size_t)k * P + (size_t)r * S + cc] = w[k];
                }
                const float iD = 1.0f / w[1];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) I3[ch] = w[2 + ch] * iD;
            } else {
            w[0] = (part[0][0][pix] * part[0][1][pix]) * (part[0][2][pix] * part[0][3][pix]);
            planes[(size_t)r * S + cc] = w[0];
#pragma unroll
            for (int k = 1; k < 5; k++) {
                w[k] = (part[k][0][pix] + part[k][1][pix]) + (part[k][2][pix] + part[k][3][pix]);
                planes[(size_t)k * P + (size_t)r * S + cc] = w[k];
            }
            const float O = 1.0f - w[0];
            const float iD = w[1] > 0.0f ? 1.0f / w[1] : 0.0f;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) I3[ch] = O * (w[2 + ch] * iD);
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) s_I[ch][pix] = I3[ch];
    }
    GENPC_STL(5);
    if (clean_flag && threadIdx.x == 0) *clean_flag = binned == 0 ? 1 : 0;
    if (accum) {
        // The tile's sums of I and I^2 per channel (doubles), by ONE wave: four pixels per lane, then the wave's shuffles.
        // (All sixteen waves used to run the 72 64-bit shuffles of the reduction although only four held pixels: with two
        // blocks per CU that was 76 of the launch's 200 us at 32 images.)
        __syncthreads();
        if (wave == 0) {
            double s1[3] = {0.0, 0.0, 0.0}, s2[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < kMaskTile * kMaskTile / kWave; q++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const double I = (double)s_I[ch][lane + q * kWave];
                    s1[ch] += I;
                    s2[ch] += I * I;
                }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                s1[ch] = wave_sum63(s1[ch]);
                s2[ch] = wave_sum63(s2[ch]);
            }
            if (lane == kWave - 1) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    atomicAdd(&accum[16 + ch], s1[ch]);
                    atomicAdd(&accum[19 + ch], s2[ch]);
                }
            }
        }
    }
    GENPC_STL(6);
}

constexpr int kMLThreads = 1024;

// block-wide sums of up to four doubles (all threads get the totals)
__device__ __forceinline__ void block_sum4(double (&x)[4], double (*red)[kMLThreads / kWave])
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double y = wave_sum63(x[k]);
        if (lane == kWave - 1) red[k][wave] = y;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double y = 0.0;
        for (int w = 0; w < kMLThreads / kWave; w++) y += red[k][w];
        x[k] = y;
    }
    __syncthreads();
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// compute_soft_mask (diff_obj_pose.py:261-278): sigmoid((luminance - 0.1) / 0.05), float32 like the reference
__device__ __forceinline__ float soft_mask(float r, float g, float b)
{
    const float lum = kLumR * r + kLumG * g + kLumB * b;
    return sigmoidf((lum - 0.1f) / 0.05f);
}

// Reference image of a scan (once per call; one block per scan): per-channel mean and unbiased std of the
// image (normalize_images, :208-209), the soft mask m_ref = soft_mask(I_ref) into mref[P] and
// stats[8] = mean[3], std[3], sum m_ref, 0.
__global__ __launch_bounds__(kMLThreads) void mask_ref_kernel(int S, const float *__restrict__ planes, int direct,
                                                              float *__restrict__ mref, float *__restrict__ stats)
{
    __shared__ double red[4][kMLThreads / kWave];
    const int e = blockIdx.x, P = S * S;
    planes += (size_t)e * 5 * P;
    mref += (size_t)e * P;
    stats += (size_t)e * 8;
    double a[4] = {0, 0, 0, 0};
    for (int q = threadIdx.x; q < P; q += kMLThreads) {
        const PxImg px = load_pixel(planes, P, q, direct);
        a[0] += (double)px.I[0]; a[1] += (double)px.I[1]; a[2] += (double)px.I[2];
    }
    block_sum4(a, red);
    const double mu[3] = {a[0] / P, a[1] / P, a[2] / P};
    double b[4] = {0, 0, 0, 0};
    for (int q = threadIdx.x; q < P; q += kMLThreads) {
        const PxImg px = load_pixel(planes, P, q, direct);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const double dd = (double)px.I[ch] - mu[ch];
            b[ch] += dd * dd;
        }
        const float m = soft_mask(px.I[0], px.I[1], px.I[2]);
        b[3] += (double)m;
        mref[q] = m;
    }
    block_sum4(b, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            stats[ch] = (float)mu[ch];
            stats[3 + ch] = (float)sqrt(b[ch] / (P - 1));
        }
        stats[6] = (float)b[3];
        stats[7] = 0.0f;
    }
}

// per-channel statistics of the posed cloud's image from the sums in accum[16..21], and the reference's
struct MaskStats {
    float muf[3], k[3], murf[3];
    double sd[3], mu[3], sdr[3];
};

__device__ __forceinline__ MaskStats mask_image_stats(const double *accum, const float *stats, int P)
{
    MaskStats o;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const double mu = accum[16 + ch] / P;
        double var = (accum[19 + ch] - (double)P * mu * mu) / (P - 1);
        var = var > 0.0 ? var : 0.0;
        o.mu[ch] = mu;
        o.sd[ch] = sqrt(var);
        o.sdr[ch] = (double)stats[3 + ch];
        o.k[ch] = (float)((o.sdr[ch] + 1e-6) / (o.sd[ch] + 1e-6));
        o.muf[ch] = (float)mu;
        o.murf[ch] = stats[ch];
    }
    return o;
}

// the per-pixel quantities every pass needs
struct MaskPx {
    float m, mr;
    float spc[3];            // d m / d I'_ch (normalised image): 0 where the clamp or the saturated sigmoid cuts the gradient
    float lm, l1m, dmb;      // clamped logs; d (30 MSE + BCE) / d m * P
};

__device__ __forceinline__ MaskPx mask_pixel(const float *I, float mr, const MaskStats &st)
{
    MaskPx o;
    float xn[3];
    bool inside[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float x0 = (I[ch] - st.muf[ch]) * st.k[ch] + st.murf[ch];
        inside[ch] = x0 > 0.0f && x0 < 1.0f;
        xn[ch] = fminf(fmaxf(x0, 0.0f), 1.0f);
    }
    o.m = soft_mask(xn[0], xn[1], xn[2]);
    o.mr = mr;
    const float lm = logf(o.m), l1m = logf(1.0f - o.m);
    o.lm = fmaxf(lm, -100.0f);
    o.l1m = fmaxf(l1m, -100.0f);
    const float sp = o.m * (1.0f - o.m) * 20.0f;
    o.spc[0] = inside[0] ? sp * kLumR : 0.0f;
    o.spc[1] = inside[1] ? sp * kLumG : 0.0f;
    o.spc[2] = inside[2] ? sp * kLumB : 0.0f;
    float db = 0.0f;
    if (lm > -100.0f) db -= mr / o.m;
    if (l1m > -100.0f) db += (1.0f - mr) / (1.0f - o.m);
    o.dmb = 60.0f * (o.m - mr) + db;
    return o;
}

// grid (blocks, b): accum[16..21] += sum I_ch, sum I_ch^2 of a direct image (the splat does this for its own)
__global__ __launch_bounds__(kQBlock) void mask_image_sums_kernel(int P, const float *__restrict__ planes,
                                                                  double *__restrict__ accum)
{
    __shared__ double red[6][kQBlock / kWave];
    const int e = blockIdx.y;
    planes += (size_t)e * 5 * P;
    accum += (size_t)e * kAcc;
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int q = blockIdx.x * kQBlock + threadIdx.x; q < P; q += gridDim.x * kQBlock) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const double I = (double)planes[(size_t)ch * P + q];
            a[ch] += I;
            a[3 + ch] += I * I;
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double x = wave_sum63(a[i]);
        if (lane == kWave - 1) red[i][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double x = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kQBlock / kWave; w2++) x += red[threadIdx.x][w2];
        atomicAdd(&accum[16 + threadIdx.x], x);
    }
}

// grid (blocks, b).  accum[22..25] += mse, bce, intersection, sum m, and per channel, with g1 = dmb spc / P,
// g2 = mr spc, g3 = spc:  accum[26 + 6 ch ..] += sum g1, sum g2, sum g3, sum g1 (I - mu), sum g2 (I - mu),
// sum g3 (I - mu)  -- the Dice term's share of G is (dice_a g2 + dice_b g3) with coefficients only known
// after this pass.
constexpr int kMaskSums = 22;
__global__ __launch_bounds__(kQBlock) void mask_sums_kernel(int S, const float *__restrict__ planes, int direct,
                                                            const float *__restrict__ mref,
                                                            const float *__restrict__ stats, double *__restrict__ accum)
{
    __shared__ double red[kMaskSums][kQBlock / kWave];
    const int e = blockIdx.y, P = S * S;
    planes += (size_t)e * 5 * P;
    mref += (size_t)e * P;
    stats += (size_t)e * 8;
    accum += (size_t)e * kAcc;
    const MaskStats st = mask_image_stats(accum, stats, P);
    const float invP = 1.0f / (float)P;
    double a[kMaskSums];
#pragma unroll
    for (int i = 0; i < kMaskSums; i++) a[i] = 0.0;
    for (int q = blockIdx.x * kQBlock + threadIdx.x; q < P; q += gridDim.x * kQBlock) {
        const PxImg im = load_pixel(planes, P, q, direct);
        const MaskPx px = mask_pixel(im.I, mref[q], st);
        a[0] += (double)((px.m - px.mr) * (px.m - px.mr));
        a[1] += (double)(-(px.mr * px.lm + (1.0f - px.mr) * px.l1m));
        a[2] += (double)(px.m * px.mr);
        a[3] += (double)px.m;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const double g1 = (double)(px.dmb * px.spc[ch] * invP), g2 = (double)(px.mr * px.spc[ch]), g3 = (double)px.spc[ch];
            const double dI = (double)im.I[ch] - st.mu[ch];
            a[4 + 6 * ch + 0] += g1; a[4 + 6 * ch + 1] += g2; a[4 + 6 * ch + 2] += g3;
            a[4 + 6 * ch + 3] += g1 * dI; a[4 + 6 * ch + 4] += g2 * dI; a[4 + 6 * ch + 5] += g3 * dI;
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kMaskSums; i++) {
        const double x = wave_sum63(a[i]);
        if (lane == kWave - 1) red[i][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < kMaskSums) {
        double x = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kQBlock / kWave; w2++) x += red[threadIdx.x][w2];
        atomicAdd(&accum[22 + threadIdx.x], x);
    }
}

// What the backward gather needs per image besides the pixel itself: the Dice term's two coefficients and, per channel, the
// mean of G and the factor of the standard deviation's own derivative (mask_w_kernel's prologue; the fused gather's too)
struct MaskCoef {
    float dice_a, dice_b, meanG[3], kk[3], invP, mask_weight;
};

__device__ __forceinline__ MaskCoef mask_coefficients(const double *accum, const float *stats, const MaskStats &st, int P, float mask_weight)
{
    MaskCoef o;
    const double den = accum[25] + (double)stats[6] + 1e-6, num = 2.0 * accum[24] + 1e-6;
    o.dice_a = (float)(-20.0 / den);
    o.dice_b = (float)(10.0 * num / (den * den));
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const double *s = accum + 26 + 6 * ch;
        const double sG = s[0] + (double)o.dice_a * s[1] + (double)o.dice_b * s[2];
        const double sGd = s[3] + (double)o.dice_a * s[4] + (double)o.dice_b * s[5];
        o.meanG[ch] = (float)(sG / P);
        const double sd = st.sd[ch];
        o.kk[ch] = sd > 0.0 ? (float)((st.sdr[ch] + 1e-6) / ((sd + 1e-6) * (sd + 1e-6)) / ((P - 1) * sd) * sGd) : 0.0f;
    }
    o.invP = 1.0f / (float)P;
    o.mask_weight = mask_weight;
    return o;
}

__device__ __forceinline__ double mask_loss_value(const double *accum, const float *stats, int P)
{
    const double den = accum[25] + (double)stats[6] + 1e-6, num = 2.0 * accum[24] + 1e-6;
    return 30.0 * accum[22] / P + accum[23] / P + 10.0 * (1.0 - num / den);
}

// the weights of pixel q (mask_w_kernel's body; the fused gather evaluates it per gathered pixel: same arithmetic, same bits)
__device__ __forceinline__ void mask_w_pixel(const float *__restrict__ planes, const float *__restrict__ mref, int P, int q, int direct,
                                             const MaskStats &st, const MaskCoef &cf, float &w1, float4 &w4)
{
    const PxImg im = load_pixel(planes, P, q, direct);
    const MaskPx px = mask_pixel(im.I, mref[q], st);
    const float Gm = px.dmb * cf.invP + cf.dice_a * px.mr + cf.dice_b;
    float dI[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
        dI[ch] = cf.mask_weight * (st.k[ch] * (Gm * px.spc[ch] - cf.meanG[ch]) - cf.kk[ch] * (im.I[ch] - st.muf[ch]));
    if (direct == 2) {
        // blend 1: I_ch = N'_ch / D', N' and D' sums of w_i = a_i e'_i:  d loss / d w_i = W4.xyz . c_i - W4.w; the gather
        // multiplies by e'_i = exp(z_i / gamma - m), so W1 carries the pixel's m
        const float sI = dI[0] * im.I[0] + dI[1] * im.I[1] + dI[2] * im.I[2];
        w1 = im.T;
        w4 = make_float4(im.iD * dI[0], im.iD * dI[1], im.iD * dI[2], im.iD * sI);
    } else if (direct) {
        w4 = make_float4(dI[0], dI[1], dI[2], 0.0f);
        w1 = 0.0f;
    } else {
        const float sA = dI[0] * im.A[0] + dI[1] * im.A[1] + dI[2] * im.A[2];
        const float od = im.O * im.iD;
        w1 = im.T * sA;
        w4 = make_float4(od * dI[0], od * dI[1], od * dI[2], od * sA);
    }
}

// grid (blocks, b): with dI_ch = mask_weight * d mask_loss / d I_ch, the five weights of the backward gather
//   W1 = T sum_ch dI_ch A_ch     (d O / d a_i = T / (1 - a_i))
//   W4 = (O / D) (dI_r, dI_g, dI_b, sum_ch dI_ch A_ch)     (d A_ch / d a_i = (c_i,ch - A_ch) / D)
// so that d loss / d a_i = W1 / (1 - a_i) + W4.xyz . c_i - W4.w.  direct: W4.xyz = dI itself.
// Block 0 adds mask_weight * mask_loss to accum[15].
__global__ __launch_bounds__(kQBlock) void mask_w_kernel(int S, const float *__restrict__ planes, int direct,
                                                         const float *__restrict__ mref,
                                                         const float *__restrict__ stats, float mask_weight,
                                                         float *__restrict__ W1, float4 *__restrict__ W4,
                                                         double *__restrict__ accum)
{
    const int e = blockIdx.y, P = S * S;
    planes += (size_t)e * 5 * P;
    mref += (size_t)e * P;
    W1 += (size_t)e * P;
    W4 += (size_t)e * P;
    stats += (size_t)e * 8;
    accum += (size_t)e * kAcc;
    const MaskStats st = mask_image_stats(accum, stats, P);
    const MaskCoef cf = mask_coefficients(accum, stats, st, P, mask_weight);
    for (int q = blockIdx.x * kQBlock + threadIdx.x; q < P; q += gridDim.x * kQBlock) {
        float w1;
        float4 w4;
        mask_w_pixel(planes, mref, P, q, direct, st, cf, w1, w4);
        W1[q] = w1;
        W4[q] = w4;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) accum[15] += (double)mask_weight * mask_loss_value(accum, stats, P);
}

// grid (tiles, b), the tile pass of the mask gradient: a tile's W planes in LDS, one thread per entry of the tile's list
// (bins), the entry's point gathers the weights of its pixels INSIDE the tile and leaves the three sums in its slot of
// gpart[e, j, kBinPer]; mask_grad_kernel adds a point's slots in order and chains to the pose.  (A thread per point
// gathering its whole box from global memory was bound by the scattered 16-byte reads: 8 scans in lock-step 150 us, 45 us
// with the same instructions on coalesced addresses.)  A tile whose list overflowed walks all points of the image.
// Resets the tile's counter (the splat kept it for this pass).
// TB threads: 1024 with the entries sorted by piece size when few images are in flight (the crowded tiles are the critical
// path), 256 unsorted when many are (a tile's list averages 125 entries: fourteen of sixteen waves of a 1024-thread block
// idle while two blocks fill the CU -- 104 us at 32 images; eight small blocks per CU: see DESIGN 4.5)
template <int TB, bool SORT, int BLEND>
__global__ __launch_bounds__(TB) void mask_grad_tile_kernel(int n, const float4 *__restrict__ uvr, const float *__restrict__ col,
                                                                       int S, const float *__restrict__ W1,
                                                                       const float4 *__restrict__ W4, int *__restrict__ bins,
                                                                       float4 *__restrict__ gpart, const float *__restrict__ zex)
{
    static_assert(TB >= kMaskTile * kMaskTile, "a thread per pixel for the load");
    __shared__ float4 sW4[kMaskTile * kMaskTile];
    __shared__ float sW1[kMaskTile * kMaskTile];
    __shared__ int s_list[SORT ? kTileCap : 1];
    __shared__ int s_ccnt[4];
    static_assert(kTileCap % TB == 0, "whole entries per thread");
    const int e = blockIdx.y, tile = blockIdx.x;
    const int tiles_x = (S + kMaskTile - 1) / kMaskTile;
    const int tx0 = (tile % tiles_x) * kMaskTile, ty0 = (tile / tiles_x) * kMaskTile;
    const int tx1 = min(S, tx0 + kMaskTile) - 1, ty1 = min(S, ty0 + kMaskTile) - 1;
    uvr += (size_t)e * n;
    if (col) col += (size_t)e * n * 3;
    if (BLEND) zex += (size_t)e * n;
    W1 += (size_t)e * S * S;
    W4 += (size_t)e * S * S;
    gpart += (size_t)e * n * kBinPer;
    int *bin_cnt = bins + (size_t)e * bins_tiles(S);
    const int *bin_idx = bins + (size_t)gridDim.y * bins_tiles(S) + ((size_t)e * bins_tiles(S) + tile) * kTileCap;
    if (threadIdx.x < kMaskTile * kMaskTile) {
        const int px = tx0 + (threadIdx.x & (kMaskTile - 1)), py = ty0 + threadIdx.x / kMaskTile;
        const bool in = px < S && py < S;
        sW4[threadIdx.x] = in ? W4[(size_t)py * S + px] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        sW1[threadIdx.x] = in ? W1[(size_t)py * S + px] : 0.0f;
    }
    const int count = bin_cnt[tile] & (kBinPoison - 1);
    __syncthreads();
    // One thread per entry.  Once the weights come from LDS the gather is VALU-bound (40 instructions a pixel over a box of
    // 81, and a wave runs the largest box of its lanes), so the walk is cut to the disc: rows whose chord is empty are
    // skipped, a row is walked over its chord +- a pixel (the test on av below still decides), and 1 / (1 - av) is the
    // hardware reciprocal (1 ulp; the oracle bar on the gradient is 2e-3).  Sixteen lanes per entry (a lane per row) was no
    // faster (43.7 us against 44.7 for a thread per entry over the full box; this form 25.6).
    auto gather = [&](int j, int slot, float4 q) {
        const float u = q.x, v = q.y, rho = q.z, ir2 = q.w;
        float cr = 1.0f, cg = 1.0f, cb = 1.0f;
        if (col) { cr = col[(size_t)j * 3 + 0]; cg = col[(size_t)j * 3 + 1]; cb = col[(size_t)j * 3 + 2]; }
        // the point's pixel box (mask_grad_kernel's), cut to the tile
        const int c0 = max(max((int)floorf(u - rho - 0.5f), 0), tx0), c1 = min(min((int)ceilf(u + rho - 0.5f), S - 1), tx1);
        const int r0 = max(max((int)floorf(v - rho - 0.5f), 0), ty0), r1 = min(min((int)ceilf(v + rho - 0.5f), S - 1), ty1);
        const float rho2 = rho * rho;
        float gu = 0.0f, gv = 0.0f, gr = 0.0f, gz = 0.0f;
        const float ze = BLEND ? zex[j] : 0.0f;
        for (int r = r0; r <= r1; r++) {
            const float dy = (float)r + 0.5f - v;
            const float h2 = rho2 - dy * dy;
            if (h2 < -1e-5f * rho2) continue;
            const float h = sqrtf(fmaxf(h2, 0.0f));
            const int ca = max((int)floorf(u - h - 0.5f), c0), cz = min((int)ceilf(u + h - 0.5f), c1);
            const float dy2 = dy * dy;
            const float4 *rw4 = sW4 + (r - ty0) * kMaskTile - tx0;
            const float *rw1 = sW1 + (r - ty0) * kMaskTile - tx0;
            // (the loads behind the test: a branch-free form -- loads first, a selected 0 for pixels outside the disc, two
            // pixels a trip -- was slower, 25.6 -> 30.9 us)
            for (int cc = ca; cc <= cz; cc++) {
                const float dx = (float)cc + 0.5f - u;
                const float d2 = dx * dx + dy2;
                const float av = 1.0f - d2 * ir2;
                if (BLEND) {
                    if (av <= 0.0f) continue;
                    const float4 w4 = rw4[cc];
                    const float w = ((cr * w4.x + cg * w4.y) + (cb * w4.z - w4.w)) * __expf(ze - rw1[cc]);
                    gz += w * fminf(av, kMaskAmax);       // through the depth: also where the coverage is clamped
                    if (av >= kMaskAmax) continue;
                    gu += w * dx;
                    gv += w * dy;
                    gr += w * d2;
                    continue;
                }
                if (av <= 0.0f || av >= kMaskAmax) continue;      // outside the disc / clamped: no gradient
                const float4 w4 = rw4[cc];
                const float w = rw1[cc] * __builtin_amdgcn_rcpf(1.0f - av) + ((cr * w4.x + cg * w4.y) + (cb * w4.z - w4.w));
                gu += w * dx;
                gv += w * dy;
                gr += w * d2;
            }
        }
        gpart[(size_t)j * kBinPer + slot] = make_float4(gu, gv, gr, gz);
    };
    if (count <= kTileCap && !SORT) {
        for (int i = threadIdx.x; i < count; i += TB) {
            const int ent = bin_idx[i];
            gather(ent >> 2, ent & 3, uvr[ent >> 2]);
        }
    } else if (count <= kTileCap) {
        // Entries in order of the size of their piece of the disc (four classes by the clipped box's area): a wave runs as
        // long as its largest piece, and a tile's list mixes whole discs (81 pixels) with halves and corners of discs
        // centred in the neighbouring tiles.  The order inside a class is arbitrary -- every entry writes its own slot.
        constexpr int kPerThread = SORT ? kTileCap / TB : 1;
        int my_c[kPerThread], my_pos[kPerThread], my_ent[kPerThread];
        if (threadIdx.x < 4) s_ccnt[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kPerThread; t++) {
            const int i = threadIdx.x + t * TB;
            my_c[t] = -1;
            if (i < count) {
                const int ent = bin_idx[i];
                const float4 q = uvr[ent >> 2];
                const int c0 = max(max((int)floorf(q.x - q.z - 0.5f), 0), tx0), c1 = min(min((int)ceilf(q.x + q.z - 0.5f), S - 1), tx1);
                const int r0 = max(max((int)floorf(q.y - q.z - 0.5f), 0), ty0), r1 = min(min((int)ceilf(q.y + q.z - 0.5f), S - 1), ty1);
                const int area = max(c1 - c0 + 1, 0) * max(r1 - r0 + 1, 0);
                const int full = (2 * (int)q.z + 3) * (2 * (int)q.z + 3);      // about the unclipped box
                const int c = 4 * area >= 3 * full ? 0 : (2 * area >= full ? 1 : (4 * area >= full ? 2 : 3));
                my_ent[t] = ent;
                my_c[t] = c;
                my_pos[t] = atomicAdd(&s_ccnt[c], 1);
            }
        }
        __syncthreads();
        const int o1 = s_ccnt[0], o2 = o1 + s_ccnt[1], o3 = o2 + s_ccnt[2];
#pragma unroll
        for (int t = 0; t < kPerThread; t++)
            if (my_c[t] >= 0) s_list[(my_c[t] == 0 ? 0 : (my_c[t] == 1 ? o1 : (my_c[t] == 2 ? o2 : o3))) + my_pos[t]] = my_ent[t];
        __syncthreads();
        for (int i = threadIdx.x; i < count; i += TB) {
            const int ent = s_list[i];
            gather(ent >> 2, ent & 3, uvr[ent >> 2]);
        }
    } else {
        for (int j = threadIdx.x; j < n; j += TB) {
            const float4 q = uvr[j];
            if (!(q.z > 0.0f)) continue;
            int slot;
            const int total = tile_count(S, q.x, q.y, q.z, tile, slot);
            if (slot >= 0 && total <= kBinPer) gather(j, slot, q);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) bin_cnt[tile] = 0;
}

// 1-D grid of gx * nb blocks (xcd_block): gradient of the mask term with respect to (R, s, t), into accum[0..12].
// FUSEW (round 6): the per-pixel weights are not read from W1 / W4 but evaluated where they are gathered, from the image's
// planes and the reference mask with mask_w_kernel's own arithmetic (mask_w_pixel: same bits) -- the alignment loop's step is
// one launch shorter (mask_w_kernel: 13 us of a 128 us step for 50 k pixels per image); block 0 of an image adds the loss.
template <int kGradSub, int BLEND, int FUSEW = 0>
__global__ __launch_bounds__(kQBlock) void mask_grad_kernel(int n, const float *__restrict__ v,
                                                            const float *__restrict__ col,
                                                            const float *__restrict__ center, int cstride,
                                                            const float *__restrict__ params, int pstride, float radius,
                                                            int S, const float *__restrict__ W1,
                                                            const float4 *__restrict__ W4, double *__restrict__ accum,
                                                            const float4 *__restrict__ gpart, const float4 *__restrict__ uvr,
                                                            int gx, int nb, const float *__restrict__ planes = nullptr,
                                                            const float *__restrict__ mref = nullptr,
                                                            const float *__restrict__ stats = nullptr, float mask_weight = 0.0f,
                                                            PoseGradArgs pg = PoseGradArgs{})
{
    __shared__ double red[15][kQBlock / kWave];
    // (one stream, full objective: the Chamfer half's gradient rides along as the launch's last blocks -- a launch less per step)
    if (pg.gx > 0 && (int)blockIdx.x >= gx * nb) {
        const int r = (int)blockIdx.x - gx * nb;
        pose_grad_body(r % pg.gx, pg.gx, r / pg.gx, pg.nc, pg.v, pg.center, pg.cstride, pg.params, pg.pstride, pg.np, pg.partial, pg.d1, pg.i1,
                       pg.d2, pg.i2, pg.cd_weight, pg.accum, red);
        return;
    }
    const XcdBlock xb = xcd_block(gx, nb);
    const int e = xb.e;
    if (uvr) uvr += (size_t)e * n;
    v += (size_t)e * n * 3;
    if (col) col += (size_t)e * n * 3;
    const int PP = S * S;
    if (!FUSEW) {
        W1 += (size_t)e * PP;
        W4 += (size_t)e * PP;
    }
    if (gpart) gpart += (size_t)e * n * kBinPer;
    center += (size_t)e * cstride;
    params += (size_t)e * pstride;
    accum += (size_t)e * kAcc;
    MaskStats mst = {};
    MaskCoef mcf = {};
    if (FUSEW) {
        planes += (size_t)e * 5 * PP;
        mref += (size_t)e * PP;
        stats += (size_t)e * 8;
        mst = mask_image_stats(accum, stats, PP);
        mcf = mask_coefficients(accum, stats, mst, PP, mask_weight);
    }
    constexpr int kMode = BLEND ? 2 : 0;
    float R[9];
    rot6d_to_matrix(params, R);
    const float s = expf(params[9]);
    const float c[3] = {center[0], center[1], center[2]};
    const float t[3] = {params[6], params[7], params[8]};
    const float hs = 0.5f * S;
    double a[13];
#pragma unroll
    for (int k = 0; k < 13; k++) a[k] = 0.0;
    // kGradSub lanes share a point: each takes every kGradSub-th row of the point's pixel box, the partial sums meet
    // by shuffles and the first lane does the chain rule.  One scan: 8 lanes (one thread per point leaves 64 blocks
    // for 16384 points and a 64-pixel serial loop per thread: 21 us of a 109 us step -> 102); several scans in
    // lock-step fill the chip with one thread per point, and the idle lanes of the 8-lane form cost 7 % there.
    const int sub = threadIdx.x & (kGradSub - 1);
    const int per_block = kQBlock / kGradSub;
    for (int j0 = xb.x * per_block; j0 < n; j0 += gx * per_block) {
        const int j = j0 + threadIdx.x / kGradSub;
        const bool live = j < n;
        const int jj = live ? j : n - 1;
        const float vx = v[(size_t)jj * 3 + 0], vy = v[(size_t)jj * 3 + 1], vz = v[(size_t)jj * 3 + 2];
        float cr = 1.0f, cg = 1.0f, cb = 1.0f;
        if (col) { cr = col[(size_t)jj * 3 + 0]; cg = col[(size_t)jj * 3 + 1]; cb = col[(size_t)jj * 3 + 2]; }
        float p[3];
        pose_point(R, s, c, t, vx, vy, vz, p);
        const SplatPt q = splat_project(p, radius, hs);
        const bool ok = live && q.ok;
        const int c0 = max((int)floorf(q.u - q.rho - 0.5f), 0), c1 = min((int)ceilf(q.u + q.rho - 0.5f), S - 1);
        const int r0 = max((int)floorf(q.v - q.rho - 0.5f), 0), r1 = min((int)ceilf(q.v + q.rho - 0.5f), S - 1);
        const float ir2 = 1.0f / (q.rho * q.rho);
        float gu = 0.0f, gv = 0.0f, gr = 0.0f, gz = 0.0f;
        const float ze = pulsar_ze(q.zv);
        // after the tile pass (gpart): the sums of a point over at most kBinPer tiles lie in its slots; a wider disc was
        // left out there and gathers its box here
        int ntile = kBinPer + 1;
        if (kGradSub == 1 && gpart && ok) {
            // (counted from the projection's OWN record of the point, which is what the lists were built from -- not from
            // the values recomputed above, equal as they should be)
            const float4 rec = uvr[j];
            int none;
            ntile = rec.z > 0.0f ? tile_count(S, rec.x, rec.y, rec.z, -1, none) : 0;
            if (ntile <= kBinPer)
                for (int k = 0; k < ntile; k++) {
                    const float4 g = gpart[(size_t)j * kBinPer + k];
                    gu += g.x; gv += g.y; gr += g.z; gz += g.w;
                }
        }
        if (ok && ntile > kBinPer) {
            // the walk of mask_grad_tile_kernel: rows with an empty chord skipped, a row over its chord +- a pixel, the
            // hardware reciprocal for 1 / (1 - av)
            const float rho2 = q.rho * q.rho;
            for (int r = r0 + sub; r <= r1; r += kGradSub) {
                const float dy = (float)r + 0.5f - q.v;
                const float h2 = rho2 - dy * dy;
                if (h2 < -1e-5f * rho2) continue;
                const float h = sqrtf(fmaxf(h2, 0.0f));
                const int ca = max((int)floorf(q.u - h - 0.5f), c0), cz = min((int)ceilf(q.u + h - 0.5f), c1);
                for (int cc = ca; cc <= cz; cc++) {
                    const float dx = (float)cc + 0.5f - q.u;
                    const float d2 = dx * dx + dy * dy;
                    const float av = 1.0f - d2 * ir2;
                    if (BLEND) {
                        if (av <= 0.0f) continue;
                        float4 w4;
                        float w1;
                        if (FUSEW) mask_w_pixel(planes, mref, PP, r * S + cc, kMode, mst, mcf, w1, w4);
                        else { w4 = W4[(size_t)r * S + cc]; w1 = W1[(size_t)r * S + cc]; }
                        const float w = ((cr * w4.x + cg * w4.y) + (cb * w4.z - w4.w)) * __expf(ze - w1);
                        gz += w * fminf(av, kMaskAmax);
                        if (av >= kMaskAmax) continue;
                        gu += w * dx;
                        gv += w * dy;
                        gr += w * d2;
                        continue;
                    }
                    if (av <= 0.0f || av >= kMaskAmax) continue;      // outside the disc / clamped: no gradient
                    float4 w4;
                    float w1;
                    if (FUSEW) mask_w_pixel(planes, mref, PP, r * S + cc, kMode, mst, mcf, w1, w4);
                    else { w4 = W4[(size_t)r * S + cc]; w1 = W1[(size_t)r * S + cc]; }
                    const float w = w1 * __builtin_amdgcn_rcpf(1.0f - av) + ((cr * w4.x + cg * w4.y) + (cb * w4.z - w4.w));
                    gu += w * dx;
                    gv += w * dy;
                    gr += w * d2;
                }
            }
        }
        if (kGradSub == 8) {
            // the eight lanes of a point: quads by permutation, then the mirror image within the eight (DPP: no LDS round trips)
            gu += dpp_or_zero<0xB1>(gu); gu += dpp_or_zero<0x4E>(gu); gu += dpp_or_zero<0x141>(gu);
            gv += dpp_or_zero<0xB1>(gv); gv += dpp_or_zero<0x4E>(gv); gv += dpp_or_zero<0x141>(gv);
            gr += dpp_or_zero<0xB1>(gr); gr += dpp_or_zero<0x4E>(gr); gr += dpp_or_zero<0x141>(gr);
            if (BLEND) { gz += dpp_or_zero<0xB1>(gz); gz += dpp_or_zero<0x4E>(gz); gz += dpp_or_zero<0x141>(gz); }
        } else {
#pragma unroll
            for (int off = kGradSub / 2; off > 0; off >>= 1) {
                gu += __shfl_xor(gu, off, kWave);
                gv += __shfl_xor(gv, off, kWave);
                gr += __shfl_xor(gr, off, kWave);
                if (BLEND) gz += __shfl_xor(gz, off, kWave);
            }
        }
        if (!ok || sub != 0) continue;
        gu *= 2.0f * ir2; gv *= 2.0f * ir2; gr *= 2.0f * ir2 / q.rho;
        const double iz = 1.0 / (double)q.zv;
        const double f4 = (double)hs * kMaskFocal;
        // (blend 1: the weights also depend on the depth, z / gamma = (zfar - Zv) kPulsarZeK)
        const double gzv = (double)gu * (-f4 * p[0] * iz * iz) + (double)gv * (f4 * p[1] * iz * iz) + (double)gr * (-(double)q.rho * iz) +
                           (BLEND ? -(double)gz * (double)kPulsarZeK : 0.0);
        const double g[3] = {(double)gu * f4 * iz, -(double)gv * f4 * iz, -gzv};
        const double l[3] = {(double)(vx - c[0]), (double)(vy - c[1]), (double)(vz - c[2])};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            a[10 + r] += g[r];
#pragma unroll
            for (int qq = 0; qq < 3; qq++) a[r * 3 + qq] += g[r] * (double)s * l[qq];
            a[9] += g[r] * ((double)R[r * 3 + 0] * l[0] + (double)R[r * 3 + 1] * l[1] + (double)R[r * 3 + 2] * l[2]);
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const double x = wave_sum63(a[k]);
        if (lane == kWave - 1) red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 13) {
        double x = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kQBlock / kWave; w2++) x += red[threadIdx.x][w2];
        atomicAdd(&accum[threadIdx.x], x);
    }
    if (FUSEW && xb.x == 0 && threadIdx.x == 0) accum[15] += (double)mask_weight * mask_loss_value(accum, stats, PP);
}

// img[P, 3] (H, W, C like the reference's renders) from the five planes
__global__ void mask_image_kernel(int P, const float *__restrict__ planes, float *__restrict__ img, int mode)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    const PxImg px = load_pixel(planes, P, q, mode);
    img[(size_t)q * 3 + 0] = px.I[0];
    img[(size_t)q * 3 + 1] = px.I[1];
    img[(size_t)q * 3 + 2] = px.I[2];
}

// [P, 3] -> planes 0..2 (genpc_mask_loss's inputs)
__global__ void mask_to_planes_kernel(int P, const float *__restrict__ img, float *__restrict__ planes)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    planes[q] = img[(size_t)q * 3 + 0];
    planes[(size_t)P + q] = img[(size_t)q * 3 + 1];
    planes[2 * (size_t)P + q] = img[(size_t)q * 3 + 2];
}

// genpc_mask_loss's outputs: loss from accum[15], grad[P, 3] from W4.xyz; clears the accumulators
__global__ void mask_loss_out_kernel(int P, const float4 *__restrict__ W4, double *__restrict__ accum,
                                     float *__restrict__ loss_out, float *__restrict__ grad)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (grad && q < P) {
        const float4 w = W4[q];
        grad[(size_t)q * 3 + 0] = w.x;
        grad[(size_t)q * 3 + 1] = w.y;
        grad[(size_t)q * 3 + 2] = w.z;
    }
    if (q == 0) *loss_out = (float)accum[15];
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

namespace genpc {

static int mask_tiles(int S) { const int t = ceil_div(S, kMaskTile); return t * t; }

// Which blend the images of the mask term are drawn with: 1 Pulsar's published blending function (softmax in depth; the
// default), 0 the round-2 coverage splat (order-independent; kept for A/B and for the numbers quoted before round 5).
// genpc_render_tune() sets it per calling thread; GENPC_RENDER_BLEND for the process.
thread_local int t_render_blend = -1;
static int render_blend()
{
    static const int env = tune_env("GENPC_RENDER_BLEND", 1, "mask term's renderer: 1 = Pulsar's blending function (softmax in depth), 0 = the coverage splat");
    return (t_render_blend >= 0 ? t_render_blend : env) ? 1 : 0;
}

// scratch of the mask term for b scans of P pixels and up to nmax points (bytes, 256-aligned pieces)
// GENPC_SPLAT_BINS=0: every tile by the full scan (A/B)
static bool use_bins(int S)
{
    static const bool v = !(tune_env("GENPC_SPLAT_BINS", 1, "alignment loop: 0 = the colour splat scans every point per tile instead of reading per-tile lists") == 0);
    return v && bins_tiles(S) <= (size_t)kBinTiles;
}

struct MaskScratch {
    float *stats;      // [b, 8]
    float *mref;       // [b, P]
    float *planes;     // [b, 5, P]
    float *W1;         // [b, P]
    float4 *W4;        // [b, P]
    float4 *uvr;       // [b, nmax]
    float *zex;        // [b, nmax] blend 1: the depth exponents of the projected points
    int *bins;         // [b, tiles] counts | [b, tiles, kTileCap] entries (bin_points_block)
    float4 *gpart;     // [b, nmax, kBinPer] per-tile sums of the mask gradient (mask_grad_tile_kernel)
    int *clean;        // [b, tiles]: 1 = the tile's pixels hold the background in all five planes (mask_splat_kernel: an empty tile leaves at once)
    size_t clean_bytes;
    size_t bins_count_bytes;
    static size_t up(size_t x) { return (x + 255) / 256 * 256; }
    static size_t side(size_t P) { size_t S = (size_t)sqrt((double)P); while (S * S < P) S++; return S; }
    static size_t bins_bytes(int b, size_t P)
    {
        const size_t t = bins_tiles((int)side(P));
        return t <= (size_t)kBinTiles ? (size_t)b * t * (1 + (size_t)kTileCap) * sizeof(int) : 0;      // (no lists for such an image: use_bins)
    }
    static size_t bytes(int b, size_t P, size_t nmax)
    {
        return up((size_t)b * 8 * 4) + up((size_t)b * P * 4) + up((size_t)b * 5 * P * 4) + up((size_t)b * P * 4) +
               up((size_t)b * P * 16) + up((size_t)b * nmax * 16) + up((size_t)b * nmax * 4) + up(bins_bytes(b, P)) +
               up((size_t)b * nmax * kBinPer * 16) + up((size_t)b * bins_tiles((int)side(P)) * sizeof(int));
    }
    void carve(char *base, int b, size_t P, size_t nmax)
    {
        size_t off = 0;
        stats = (float *)(base + off); off += up((size_t)b * 8 * 4);
        mref = (float *)(base + off); off += up((size_t)b * P * 4);
        planes = (float *)(base + off); off += up((size_t)b * 5 * P * 4);
        W1 = (float *)(base + off); off += up((size_t)b * P * 4);
        W4 = (float4 *)(base + off); off += up((size_t)b * P * 16);
        uvr = (float4 *)(base + off); off += up((size_t)b * nmax * 16);
        zex = (float *)(base + off); off += up((size_t)b * nmax * 4);
        bins = (int *)(base + off); off += up(bins_bytes(b, P));
        gpart = (float4 *)(base + off); off += up((size_t)b * nmax * kBinPer * 16);
        clean = (int *)(base + off);
        clean_bytes = (size_t)b * bins_tiles((int)side(P)) * sizeof(int);
        bins_count_bytes = bins_bytes(b, P) ? (size_t)b * bins_tiles((int)side(P)) * sizeof(int) : 0;
    }
    // the tile counters are zero between launches (the splat kernel resets what it reads); once per API call for a
    // workspace that is new or was last used with another batch size
    bool zero_bins(hipStream_t st) const
    {
        // (the planes of a fresh call hold anything: no tile is known to be clean)
        if (clean_bytes && !check(hipMemsetAsync(clean, 0, clean_bytes, st), "hipMemsetAsync(tile flags)")) return false;
        return bins_count_bytes == 0 || check(hipMemsetAsync(bins, 0, bins_count_bytes, st), "hipMemsetAsync(tile counters)");
    }
};

// splat of the partial clouds (with their colours) + reference soft masks / statistics (once per call)
static int mask_prepare_ref(int b, int np, const float *partial, const float *partial_col, float radius, int S,
                            const MaskScratch &m, hipStream_t st)
{
    const int blend = render_blend();
    hipLaunchKernelGGL(mask_project_kernel, dim3(lin_grid(np), b), dim3(kQBlock), 0, st, np, partial, (const float *)nullptr, 0,
                       (const float *)nullptr, 0, 0, radius, S, m.uvr, use_bins(S) ? m.bins : (int *)nullptr, blend ? m.zex : (float *)nullptr);
    if (blend)
        hipLaunchKernelGGL(mask_splat_kernel<1>, dim3(mask_tiles(S) * b), dim3(kSplatBlock), 0, st, np, (const float4 *)m.uvr, partial_col,
                           S, m.planes, (double *)nullptr, use_bins(S) ? m.bins : (int *)nullptr, 0, b, (const float *)m.zex, m.clean);
    else
        hipLaunchKernelGGL(mask_splat_kernel<0>, dim3(mask_tiles(S) * b), dim3(kSplatBlock), 0, st, np, (const float4 *)m.uvr, partial_col,
                           S, m.planes, (double *)nullptr, use_bins(S) ? m.bins : (int *)nullptr, 0, b, (const float *)nullptr, m.clean);
    hipLaunchKernelGGL(mask_ref_kernel, dim3(b), dim3(kMLThreads), 0, st, S, (const float *)m.planes, blend ? 2 : 0, m.mref, m.stats);
    return check(hipGetLastError(), "mask reference launch") ? 1 : 0;
}

// (the per-point silhouette gradient can carry pose_grad's blocks along; the per-tile form cannot)
static bool mask_step_carries(int S)
{
    static const int env_tp = tune_env("GENPC_MASK_GRAD_TILES", -1, "alignment loop: 1 = the silhouette gradient through the per-tile lists (opt-in), 0 = per point");
    return !(use_bins(S) && env_tp > 0);
}

// the launches of the mask term for the current parameters: accum[0..12] += gradient, accum[15] += loss
static int mask_step(int b, int nc, const float *complete, const float *complete_col, const float *center, int cstride,
                     const float *params, int pstride, float radius, int S, float mask_weight, const MaskScratch &m,
                     double *accum, hipStream_t st, bool projected = false, const PoseGradArgs *ride = nullptr)
{
    const float rad = 1.1f * radius;      // diff_obj_pose.py:385: the posed cloud is drawn with 1.1 x the radius
    const int gp = lin_grid((long long)S * S);
    // The mask gradient per tile from LDS (mask_grad_tile_kernel + the per-point sum) or per point from global memory
    // (mask_grad_kernel alone).  Measured at 16384 points per image, 224 x 224: 4 images (one scan's four starts, the W
    // planes stay in L2) 25.6 + 8.2 us against 29.3; 32 images (8 scans in lock-step: 32 MB of W planes through every 4 MB
    // L2, the scattered 16-byte reads cost 105 of the per-point kernel's 150 us) 107 + 21 against 150-168.
    static const int env_tp = tune_env("GENPC_MASK_GRAD_TILES", -1, "alignment loop: 1 = the silhouette gradient through the per-tile lists (opt-in), 0 = per point");
    // (8 x 4 images of 32768 points, 167 points per tile on average: 0.814 s per call with the tile pass, 0.789 without --
    // crowded tiles take several rounds of the block; 16384 points, 84 per tile: 167 ms against 176)
    // OPT-IN (GENPC_MASK_GRAD_TILES=1).  On a cloud that fills the image the tile pass wins 5 % with 32 images in flight, but
    // on a small object (16384 points over ~18 tiles: tools/time_reg_small.py) a few blocks carry all the work and 8 scans
    // take 328 ms against 191 with the per-point kernel -- which does not care where the points fall.
    const bool tile_pass = use_bins(S) && env_tp > 0;
    (void)b;
    const int blend = render_blend(), mode = blend ? 2 : 0;
    const float *zex = blend ? m.zex : nullptr;
    if (!projected)      // (the alignment loop projects in its transform launch)
        hipLaunchKernelGGL(mask_project_kernel, dim3(lin_grid(nc), b), dim3(kQBlock), 0, st, nc, complete, center, cstride, params,
                           pstride, 1, rad, S, m.uvr, use_bins(S) ? m.bins : (int *)nullptr, blend ? m.zex : (float *)nullptr);
    if (blend)
        hipLaunchKernelGGL(mask_splat_kernel<1>, dim3(mask_tiles(S) * b), dim3(kSplatBlock), 0, st, nc, (const float4 *)m.uvr, complete_col,
                           S, m.planes, accum, use_bins(S) ? m.bins : (int *)nullptr, tile_pass ? 1 : 0, b, zex, m.clean);
    else
        hipLaunchKernelGGL(mask_splat_kernel<0>, dim3(mask_tiles(S) * b), dim3(kSplatBlock), 0, st, nc, (const float4 *)m.uvr, complete_col,
                           S, m.planes, accum, use_bins(S) ? m.bins : (int *)nullptr, tile_pass ? 1 : 0, b, zex, m.clean);
    // few blocks per image: every block ends in 22 double atomics on the image's accumulators, and 196 blocks x 22 on the
    // same addresses serialise in L2 (17.5 us for 0.2 M pixels; GENPC_MASK_SUMS_BLOCKS for A/B)
    static const int env_sb = tune_env("GENPC_MASK_SUMS_BLOCKS", 0, "alignment loop: blocks per image of mask_sums_kernel (0 = pick)");
    const int gs = std::min(gp, env_sb > 0 ? env_sb : 48);      // 196: 221 ms per 8-scan call, 48: 210, 24: 210, 12: 210 (single scan: 42.1 / 41.4 / 41.8 / 43.3)
    hipLaunchKernelGGL(mask_sums_kernel, dim3(gs, b), dim3(kQBlock), 0, st, S, (const float *)m.planes, mode, (const float *)m.mref,
                       (const float *)m.stats, accum);
    // the weights per pixel as a launch of their own (the tile pass reads them from LDS tiles), or evaluated inside the gather
    static const int env_fw = tune_env("GENPC_MASK_FUSE_W", 1, "alignment loop: 1 = the silhouette gradient evaluates the per-pixel weights where it gathers them (no mask_w launch), 0 = mask_w_kernel + gather");
    // (only where the gather touches fewer pixels than ~three and a half passes over the image (config 2: 4493 points): a point's disc covers ~pi rho^2 pixels, rho =
    //  S/2 * focal * radius / 3 at the camera's distance -- 2451 points: 0.4 of the image; 16384 points: 2.9 images' worth of
    //  weights, each 60 instructions where the launch of its own computes them once per pixel)
    const float rho_px = 0.5f * (float)S * kMaskFocal * rad / kMaskEyeZ;
    const bool fuse_w = !tile_pass && env_fw != 0 && (double)nc * 3.1416 * rho_px * rho_px <= 3.5 * (double)S * S;
    if (!fuse_w)
        hipLaunchKernelGGL(mask_w_kernel, dim3(gp, b), dim3(kQBlock), 0, st, S, (const float *)m.planes, mode, (const float *)m.mref,
                           (const float *)m.stats, mask_weight, m.W1, m.W4, accum);
    if (tile_pass) {
#define GENPC_LAUNCH_TILE(TB, SORT, BL)                                                                                              \
        hipLaunchKernelGGL((mask_grad_tile_kernel<TB, SORT, BL>), dim3(mask_tiles(S), b), dim3(TB), 0, st, nc, (const float4 *)m.uvr,   \
                           complete_col, S, (const float *)m.W1, (const float4 *)m.W4, m.bins, m.gpart, zex)
        if (b > 4) { if (blend) GENPC_LAUNCH_TILE(256, false, 1); else GENPC_LAUNCH_TILE(256, false, 0); }
        else { if (blend) GENPC_LAUNCH_TILE(1024, true, 1); else GENPC_LAUNCH_TILE(1024, true, 0); }
#undef GENPC_LAUNCH_TILE
        if (blend)
            hipLaunchKernelGGL((mask_grad_kernel<1, 1>), dim3(lin_grid(nc) * b), dim3(kQBlock), 0, st, nc, complete, complete_col, center,
                               cstride, params, pstride, rad, S, (const float *)m.W1, (const float4 *)m.W4, accum,
                               (const float4 *)m.gpart, (const float4 *)m.uvr, lin_grid(nc), b);
        else
            hipLaunchKernelGGL((mask_grad_kernel<1, 0>), dim3(lin_grid(nc) * b), dim3(kQBlock), 0, st, nc, complete, complete_col, center,
                               cstride, params, pstride, rad, S, (const float *)m.W1, (const float4 *)m.W4, accum,
                               (const float4 *)m.gpart, (const float4 *)m.uvr, lin_grid(nc), b);
    } else {
        // lanes per point: a thread per point walks its whole pixel box alone (a chain of ~35 dependent gathers at the loop's
        // radius: 21 us for 4 x 2451 points), eight lanes share it row by row (12.9 us) -- until the points fill the chip by
        // themselves (measured at 4 x 16384 points, the four starts in lock-step: <1> 35 us, <8> 42)
        static const int env_sub = tune_env("GENPC_MASK_GRAD_SUB", 0, "alignment loop: lanes per point of the per-point silhouette gradient (0 = pick)");
        const int sub = env_sub ? env_sub : ((long long)b * nc <= 24576 || b <= 2 ? 8 : 1);
        const PoseGradArgs pgx = ride ? *ride : PoseGradArgs{};
#define GENPC_LAUNCH_MASK_GRAD3(SUB, BL, FW)                                                                                         \
        hipLaunchKernelGGL((mask_grad_kernel<SUB, BL, FW>), dim3(lin_grid((long long)nc * SUB) * b + pgx.gx * b), dim3(kQBlock), 0, st, nc, complete, \
                           complete_col, center, cstride, params, pstride, rad, S, (const float *)m.W1, (const float4 *)m.W4, accum, \
                           (const float4 *)nullptr, (const float4 *)nullptr, lin_grid((long long)nc * SUB), b, (const float *)m.planes, \
                           (const float *)m.mref, (const float *)m.stats, mask_weight, pgx)
#define GENPC_LAUNCH_MASK_GRAD2(SUB, BL) do { if (fuse_w) GENPC_LAUNCH_MASK_GRAD3(SUB, BL, 1); else GENPC_LAUNCH_MASK_GRAD3(SUB, BL, 0); } while (0)
#define GENPC_LAUNCH_MASK_GRAD(SUB) do { if (blend) GENPC_LAUNCH_MASK_GRAD2(SUB, 1); else GENPC_LAUNCH_MASK_GRAD2(SUB, 0); } while (0)
        if (sub == 8) GENPC_LAUNCH_MASK_GRAD(8);
        else if (sub == 4) GENPC_LAUNCH_MASK_GRAD(4);
        else if (sub == 2) GENPC_LAUNCH_MASK_GRAD(2);
        else GENPC_LAUNCH_MASK_GRAD(1);
#undef GENPC_LAUNCH_MASK_GRAD3
#undef GENPC_LAUNCH_MASK_GRAD2
#undef GENPC_LAUNCH_MASK_GRAD
    }
    return check(hipGetLastError(), "mask step launch") ? 1 : 0;
}

}  // namespace genpc

#ifdef GENPC_SPLAT_TIMELINE
extern "C" __attribute__((visibility("default"))) int genpc_splat_timeline_read(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(genpc::g_splat_tl), sizeof(unsigned long long) * 4096 * 8) == hipSuccess;
}
#endif

GENPC_API int genpc_splat_image(int n, const float *pts, const float *col, float radius, int size, float *img, void *stream)
{
    using namespace genpc;
    if (n < 0 || size <= 0 || !(radius > 0.0f)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const size_t P = (size_t)size * size;
    char *ws = (char *)workspace(14, MaskScratch::bytes(1, P, n > 0 ? n : 1), st);
    if (!ws) return 0;
    MaskScratch m;
    m.carve(ws, 1, P, n > 0 ? n : 1);
    if (!m.zero_bins(st)) return 0;
    const int blend = render_blend();
    hipLaunchKernelGGL(mask_project_kernel, dim3(lin_grid(n > 0 ? n : 1), 1), dim3(kQBlock), 0, st, n, pts, (const float *)nullptr, 0,
                       (const float *)nullptr, 0, 0, radius, size, m.uvr, use_bins(size) ? m.bins : (int *)nullptr, blend ? m.zex : (float *)nullptr);
    if (blend)
        hipLaunchKernelGGL(mask_splat_kernel<1>, dim3(mask_tiles(size)), dim3(kSplatBlock), 0, st, n, (const float4 *)m.uvr, col, size,
                           m.planes, (double *)nullptr, use_bins(size) ? m.bins : (int *)nullptr, 0, 1, (const float *)m.zex, m.clean);
    else
        hipLaunchKernelGGL(mask_splat_kernel<0>, dim3(mask_tiles(size)), dim3(kSplatBlock), 0, st, n, (const float4 *)m.uvr, col, size,
                           m.planes, (double *)nullptr, use_bins(size) ? m.bins : (int *)nullptr, 0, 1, (const float *)nullptr, m.clean);
    hipLaunchKernelGGL(mask_image_kernel, dim3(ceil_div((int)P, 256)), dim3(256), 0, st, (int)P, (const float *)m.planes, img, blend ? 2 : 0);
    return check(hipGetLastError(), "splat_image launch") ? 1 : 0;
}

GENPC_API int genpc_mask_loss(int size, const float *img, const float *ref, float *loss_out, float *grad, void *stream)
{
    using namespace genpc;
    if (size <= 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int P = size * size;
    char *ws = (char *)workspace(18, 512 + MaskScratch::bytes(2, (size_t)P, 1), st);
    if (!ws) return 0;
    double *accum = (double *)ws;
    MaskScratch m;
    m.carve(ws + 512, 2, (size_t)P, 1);
    float *pl_img = m.planes, *pl_ref = m.planes + (size_t)5 * P;
    if (!check(hipMemsetAsync(accum, 0, kAcc * sizeof(double), st), "hipMemsetAsync(accum)")) return 0;
    const int g256 = ceil_div(P, 256), gp = lin_grid(P);
    hipLaunchKernelGGL(mask_to_planes_kernel, dim3(g256), dim3(256), 0, st, P, img, pl_img);
    hipLaunchKernelGGL(mask_to_planes_kernel, dim3(g256), dim3(256), 0, st, P, ref, pl_ref);
    hipLaunchKernelGGL(mask_ref_kernel, dim3(1), dim3(kMLThreads), 0, st, size, (const float *)pl_ref, 1, m.mref, m.stats);
    hipLaunchKernelGGL(mask_image_sums_kernel, dim3(gp, 1), dim3(kQBlock), 0, st, P, (const float *)pl_img, accum);
    hipLaunchKernelGGL(mask_sums_kernel, dim3(gp, 1), dim3(kQBlock), 0, st, size, (const float *)pl_img, 1, (const float *)m.mref,
                       (const float *)m.stats, accum);
    hipLaunchKernelGGL(mask_w_kernel, dim3(gp, 1), dim3(kQBlock), 0, st, size, (const float *)pl_img, 1, (const float *)m.mref,
                       (const float *)m.stats, 1.0f, m.W1, m.W4, accum);
    hipLaunchKernelGGL(mask_loss_out_kernel, dim3(g256), dim3(256), 0, st, P, (const float4 *)m.W4, accum, loss_out, grad);
    return check(hipGetLastError(), "mask_loss launch") ? 1 : 0;
}

GENPC_API int genpc_pose_loss_grad(int nc, const float *v, const float *vert_col, const float *center, const float *params,
                                   int np, const float *partial, const float *partial_col, const float *d1, const int *i1,
                                   const float *d2, const int *i2, float cd_weight, float reg_weight, float mask_weight,
                                   float radius, int render_size, float *loss_out, float *grad, void *stream)
{
    using namespace genpc;
    if (nc <= 0 || np <= 0) return -1;
    const bool mask = mask_weight != 0.0f;
    if (mask && (render_size <= 1 || !(radius > 0.0f))) return -1;
    hipStream_t st = (hipStream_t)stream;
    const size_t P = mask ? (size_t)render_size * render_size : 0;
    const size_t nmax = (size_t)(nc > np ? nc : np);
    const size_t o_state = 512, o_mask = o_state + MaskScratch::up(sizeof(PoseState));
    char *ws = (char *)workspace(3, o_mask + MaskScratch::bytes(1, P, nmax), st);
    if (!ws) return 0;
    double *accum = (double *)ws;
    PoseState *S = (PoseState *)(ws + o_state);
    MaskScratch m;
    m.carve(ws + o_mask, 1, P, nmax);
    if (P && !m.zero_bins(st)) return 0;
    if (!check(hipMemsetAsync(accum, 0, kAcc * sizeof(double), st), "hipMemsetAsync(accum)")) return 0;
    if (!check(hipMemcpyAsync(S->params, params, 10 * sizeof(float), hipMemcpyDeviceToDevice, st), "copy params"))
        return 0;
    hipLaunchKernelGGL(pose_grad_kernel, dim3(lin_grid((long long)nc + np), 1), dim3(kQBlock), 0, st, nc, v, center, 0,
                       params, 0, np, partial, d1, i1, d2, i2, cd_weight, accum, (unsigned *)nullptr);
    if (mask) {
        if (!mask_prepare_ref(1, np, partial, partial_col, radius, render_size, m, st)) return 0;
        if (!mask_step(1, nc, v, vert_col, center, 0, params, 0, radius, render_size, mask_weight, m, accum, st)) return 0;
    }
    hipLaunchKernelGGL(pose_update_kernel, dim3(1), dim3(64), 0, st, 1, S, accum, nc, np, cd_weight, reg_weight, 0.0f, 0,
                       (float *)nullptr, 0);
    if (!check(hipMemcpyAsync(grad, S->grad, 10 * sizeof(float), hipMemcpyDeviceToDevice, st), "copy grad")) return 0;
    if (!check(hipMemcpyAsync(loss_out, S->loss, 4 * sizeof(float), hipMemcpyDeviceToDevice, st), "copy loss")) return 0;
    return check(hipGetLastError(), "pose_loss_grad launch") ? 1 : 0;
}

GENPC_API int genpc_pose_cd_grad(int nc, const float *v, const float *center, const float *params, int np,
                                 const float *partial, const float *d1, const int *i1, const float *d2, const int *i2,
                                 float cd_weight, float reg_weight, float *loss_out, float *grad, void *stream)
{
    // loss_out[3]: total, cd, ortho (the Chamfer half alone)
    using namespace genpc;
    hipStream_t st = (hipStream_t)stream;
    float *tmp = (float *)workspace(15, 256, st);
    if (!tmp) return 0;
    const int rc = genpc_pose_loss_grad(nc, v, nullptr, center, params, np, partial, nullptr, d1, i1, d2, i2, cd_weight,
                                        reg_weight, 0.0f, 0.0f, 0, tmp, grad, stream);
    if (rc != 1) return rc;
    return check(hipMemcpyAsync(loss_out, tmp, 3 * sizeof(float), hipMemcpyDeviceToDevice, st), "copy loss") ? 1 : 0;
}

namespace genpc { thread_local int t_pose_seeded = -1; }

/* Nearest-neighbour path of the alignment loop, for tests and A/B (calling host thread): 1 seeded cell search from the
 * second step on (csrc/nn_seeded.hip: it wins when every query keeps a near target and loses on misaligned starts of
 * real shapes), 0 the brute-force filter at every step, 2 whichever of the two the call measures to be faster (the
 * default), < 0 the default / environment (GENPC_POSE_SEEDED).  All give the same bits.  Returns the previous setting. */
/* The renderer of the mask term, per calling host thread: 1 Pulsar's blending function (default), 0 the coverage splat,
 * < 0 back to the default / GENPC_RENDER_BLEND.  Returns the previous setting (-1 = default). */
GENPC_API int genpc_render_tune(int blend)
{
    const int prev = genpc::t_render_blend;
    genpc::t_render_blend = blend < 0 ? -1 : (blend ? 1 : 0);
    return prev;
}

namespace genpc { static thread_local int t_pose_dual = -1; }

/* The calling host thread's alignment loops: 1 = the Chamfer half of a step on a side stream beside the silhouette half (small
 * clouds, full objective), 0 = one stream, < 0 = the default (GENPC_POSE_DUAL, on).  pipeline.run_in_lanes switches it off in
 * its lanes: with several scans in flight the chip is shared already and a second stream per scan costs throughput (six lanes:
 * 30 scans/s with, 40 without; one scan alone: 45.6 ms with, 48.3 without).  Returns the previous setting. */
GENPC_API int genpc_pose_dual(int on)
{
    const int prev = genpc::t_pose_dual;
    genpc::t_pose_dual = on < 0 ? -1 : (on ? 1 : 0);
    return prev;
}

GENPC_API int genpc_pose_tune(int seeded)
{
    const int prev = genpc::t_pose_seeded;
    genpc::t_pose_seeded = seeded < 0 ? -1 : (seeded > 2 ? 2 : seeded);
    return prev;
}

namespace genpc {
// The two halves of an Adam step read the posed cloud and nothing of each other -- nearest neighbours + Chamfer gradient
// (three launches) and the silhouette term (splat, sums, gather) -- and every launch of either is latency-bound on a mostly
// idle chip: with the full objective the Chamfer half runs on a stream of its own beside the silhouette half (fork behind
// the transform, join in front of the update; all enqueued up front like the rest, no host synchronisation).  One side
// stream and a small ring of events per caller's stream, made once and kept for the life of the process.
struct PoseSide {
    hipStream_t side = nullptr;
    hipEvent_t fork[4] = {nullptr, nullptr, nullptr, nullptr}, join[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = false;
    // hand-over through device words (pose_wait_count / pose_publish_block): counts that only grow, so nothing is reset between steps,
    // starts or calls; a call that returned early leaves `dirty` set and the next one starts from a synchronised, zeroed state
    unsigned *ctr = nullptr;
    unsigned tp_count = 0, pg_count = 0;
    bool dirty = false;
};
static PoseSide *pose_side_of(hipStream_t st)
{
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, PoseSide *> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    auto it = table.find({dev, st});
    if (it != table.end()) return it->second->ok ? it->second : nullptr;
    PoseSide *p = new PoseSide();
    // (the highest priority class: streams of one class share a handful of hardware queues, and a side stream that lands on its
    //  main stream's queue runs behind it instead of beside it -- tools/time_c2_streams.py; the class's queues are its own)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    p->ok = hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio_hi) == hipSuccess;
    for (int i = 0; i < 4 && p->ok; i++)
        p->ok = hipEventCreateWithFlags(&p->fork[i], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&p->join[i], hipEventDisableTiming) == hipSuccess;
    if (p->ok && (hipMalloc((void **)&p->ctr, 4 * sizeof(unsigned)) != hipSuccess || hipMemset(p->ctr, 0, 4 * sizeof(unsigned)) != hipSuccess))
        p->ctr = nullptr;          // (the events still work)
    table[{dev, st}] = p;
    return p->ok ? p : nullptr;
}
}  // namespace genpc

namespace genpc { void fps_deferred_prepare(hipStream_t st); }

/* The side streams the library pairs with `stream` (the alignment loop's second stream, the sampling check's), made NOW and
 * each given a first command, so that they get their hardware queues before the caller makes other streams: which queue
 * (and which of the command processor's pipes) a stream lands on is decided when it is first used, and a completed scan
 * whose side streams were first used after six lane streams existed ran at 26.7 scans/s against 28.9
 * (tools/c2_after_big.py).  pipeline.run_in_lanes calls it before it makes its lanes. */
GENPC_API int genpc_streams_prepare(void *stream)
{
    using namespace genpc;
    hipStream_t st = (hipStream_t)stream;
    PoseSide *p = pose_side_of(st);
    if (p) {
        (void)hipEventRecord(p->fork[0], p->side);
        (void)hipStreamSynchronize(p->side);
    }
    fps_deferred_prepare(st);
    return 1;
}

GENPC_API int genpc_pose_optimize_batch(int b, int nc, const float *complete, const float *complete_col, int np,
                                        const float *partial, const float *partial_col, float lr, int iters, int starts,
                                        float radius, int render_size, float mask_weight, float *transform,
                                        float *history, float *best_params, void *stream)
{
    using namespace genpc;
    if (b <= 0 || nc <= 0 || np <= 0 || iters < 0 || starts < 1) return -1;
    const bool mask = mask_weight != 0.0f;
    if (mask && (render_size <= 1 || !(radius > 0.0f))) return -1;
    hipStream_t st = (hipStream_t)stream;
    // The multi-starts of a scan are independent optimisations of the same clouds (diff_obj_pose.py:516-576): they run
    // SIDE BY SIDE as batch elements (element = scan * starts + start) -- one pass of iters + 1 Adam steps with
    // `starts` times the work per launch instead of `starts` passes.  At the reference's sizes every kernel of a
    // step is latency-bound (15403 x 7855 points: 118 us of kernels per step, 804 steps = 95 ms of reg()'s 110),
    // so the wider launches are nearly free.  Same arithmetic per element, same selection rule at the end.
    static const int env_lock = tune_env("GENPC_POSE_LOCKSTEP", 1, "alignment loop: 0 = the starts of a scan one after the other instead of side by side");
    const int lock = (starts > 1 && env_lock && (long long)b * starts <= 256) ? starts : 0;
    const int scans = b, starts_in = starts;
    float *x_complete = nullptr, *x_partial = nullptr, *x_ccol = nullptr, *x_pcol = nullptr;
    if (lock) {
        auto up0 = [](size_t x) { return (x + 255) / 256 * 256; };
        const size_t bc = (size_t)b * lock * nc * 12, bp = (size_t)b * lock * np * 12;
        char *xs = (char *)workspace(22, 2 * up0(bc) + 2 * up0(bp), st);
        if (!xs) return 0;
        x_complete = (float *)xs;
        x_partial = (float *)(xs + up0(bc));
        hipLaunchKernelGGL(pose_replicate_kernel, dim3(lin_grid((long long)nc * 3), b), dim3(kQBlock), 0, st, (size_t)nc * 3, lock, complete, x_complete);
        hipLaunchKernelGGL(pose_replicate_kernel, dim3(lin_grid((long long)np * 3), b), dim3(kQBlock), 0, st, (size_t)np * 3, lock, partial, x_partial);
        if (complete_col) {
            x_ccol = (float *)(xs + up0(bc) + up0(bp));
            hipLaunchKernelGGL(pose_replicate_kernel, dim3(lin_grid((long long)nc * 3), b), dim3(kQBlock), 0, st, (size_t)nc * 3, lock, complete_col, x_ccol);
        }
        if (partial_col) {
            x_pcol = (float *)(xs + 2 * up0(bc) + up0(bp));
            hipLaunchKernelGGL(pose_replicate_kernel, dim3(lin_grid((long long)np * 3), b), dim3(kQBlock), 0, st, (size_t)np * 3, lock, partial_col, x_pcol);
        }
        complete = x_complete; partial = x_partial; complete_col = x_ccol; partial_col = x_pcol;
        b *= lock;
        starts = 1;
    }
    const size_t P = mask ? (size_t)render_size * render_size : 0;
    // scratch: accum[b,kAcc] | state[b] | center[b,4] | pts[b,nc,3] | d1 | d2 | i1 | i2 | mask scratch
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = 0;
    const size_t o_acc = off; off += up((size_t)2 * b * kAcc * sizeof(double));        // two sets: a step's sums are read by the next step's
    const size_t o_state = off; off += up((size_t)2 * b * sizeof(PoseState));           // transform (the fused update) while that step fills the other
    const size_t o_center = off; off += up((size_t)b * 4 * sizeof(float));
    const size_t o_pts = off; off += up((size_t)b * nc * 12);
    const size_t o_d1 = off; off += up((size_t)b * nc * 4);
    const size_t o_d2 = off; off += up((size_t)b * np * 4);
    const size_t o_i1 = off; off += up((size_t)b * nc * 4);
    const size_t o_i2 = off; off += up((size_t)b * np * 4);
    const size_t o_mask = off; off += mask ? MaskScratch::bytes(b, P, (size_t)(nc > np ? nc : np)) : 0;
    char *ws = (char *)workspace(4, off, st);
    if (!ws) return 0;
    double *accum = (double *)(ws + o_acc);
    PoseState *S = (PoseState *)(ws + o_state);
    float *center = (float *)(ws + o_center);
    float *pts = (float *)(ws + o_pts);
    float *d1 = (float *)(ws + o_d1), *d2 = (float *)(ws + o_d2);
    int *i1 = (int *)(ws + o_i1), *i2 = (int *)(ws + o_i2);
    MaskScratch m = {};
    if (mask) m.carve(ws + o_mask, b, P, (size_t)(nc > np ? nc : np));
    if (mask && !m.zero_bins(st)) return 0;
    constexpr int kStateFloats = (int)(sizeof(PoseState) / sizeof(float));
    static_assert(sizeof(PoseState) % sizeof(float) == 0, "PoseState must be float-addressable");

    // center = mean(vert_pos) per scan (diff_obj_pose.py:362)
    if (!genpc_mean3(b, nc, complete, center, accum, st)) return 0;
    if (!check(hipMemsetAsync(accum + (size_t)b * kAcc, 0, (size_t)b * kAcc * sizeof(double), st), "hipMemsetAsync(second accumulator set)")) return 0;
    // reference image of the partial cloud (render_reference_image, diff_obj_pose.py:108-134)
    if (mask && !mask_prepare_ref(b, np, partial, partial_col, radius, render_size, m, st)) return 0;

    // Exact duplicates (a partial cloud resampled with replacement, a crop pad-repeated to a fixed size) are the same
    // at every step -- the partial cloud does not move and equal rest-frame points are posed to equal points -- so
    // the filter's duplicate masks (nn_dedupe.hip) are made once per call, on the rest-frame clouds.
    const size_t w_p = (nn_dedupe_mask_words(b, np) + 63) & ~(size_t)63, w_c = (nn_dedupe_mask_words(b, nc) + 63) & ~(size_t)63;
    unsigned *dup_p = (unsigned *)workspace(26, (w_p + w_c) * sizeof(unsigned), st);
    if (!dup_p) return 0;
    unsigned *dup_c = dup_p + w_p;
    {
        const float *dp[2] = {partial, complete};
        const int dn[2] = {np, nc};
        unsigned *dm[2] = {dup_p, dup_c};
        if (!launch_nn_dedupe(b, 2, dp, dn, dm, nullptr, st)) return 0;
    }

    // genpc_pose_tune(1) / GENPC_POSE_SEEDED=1 (the default is 2 = adaptive, below): from the second step on every nearest-neighbour query starts from the
    // index it was answered with a step ago and searches only the ball that answer leaves (nn_seeded.hip): both clouds
    // sorted once per call into uniform grids, the moving one in its rest frame.  Bit-identical to the brute-force filter.
    // Measured (round 4): 8 scans of uniform VOLUME clouds 161 -> 120 ms per call (49 -> 66 scans/s: a rotated cube is still
    // a cube, every query keeps a near target), one scan 33.7 -> 32.2 ms (at that size a step is eight dependent launches,
    // not their work) -- but config 5's surfaces 0.67 -> 0.84 s: three of the four starts are rotated by 90 / 180 / 270
    // degrees, most queries of a misaligned start have NO near target, and the ball their old answer leaves crosses the
    // other surface over hundreds of cells (18 k instructions per wave, 4.8 ms per step against 3.0 for the filter, which
    // does not care where the points are).  Real shapes look like config 5: off by default.
    static const int env_seeded = tune_env("GENPC_POSE_SEEDED", 2, "alignment loop, nearest neighbours from the second step on: 1 = seeded cell search, 0 = brute-force filter, 2 = measure both and switch");
    // (small clouds with the full objective: the nearest-neighbour launches run on the side stream beside the silhouette half
    //  and are not what a step waits for -- the adaptive mode's timing probes, ~10 host synchronisations per call, would cost
    //  more than either choice: the filter it is)
    // On by default for small clouds (GENPC_POSE_DUAL=0 / genpc_pose_dual(0): one stream).  The side stream is of the highest
    // priority class: as a stream of the caller's class it could land on the main stream's hardware queue and run BEHIND it
    // (tools/time_c2_streams.py, one scan at a time: 24.4 scans/s without a side stream; 25.7 with it on the null stream but 21.2
    // on a stream of the caller's own; with its own class 25.3 / 25.9).
    static const int env_dual0 = tune_env("GENPC_POSE_DUAL", 1, "alignment loop, full objective, small clouds: 1 = the Chamfer half of a step (nearest neighbours + gradient) on a side stream beside the silhouette half, 0 = one stream");
    static const int env_dual_max = tune_env("GENPC_POSE_DUAL_MAX", 65536, "alignment loop: the side stream for up to this many points per call (elements x points)");
    const bool dual_small = mask && (t_pose_dual >= 0 ? t_pose_dual != 0 : env_dual0 != 0) && (long long)b * nc <= env_dual_max;
    int seed_mode = nc >= 256 && np >= 256 ? (t_pose_seeded >= 0 ? t_pose_seeded : env_seeded) : 0;
    if (dual_small && t_pose_seeded < 0 && seed_mode == 2) seed_mode = 0;
    // ... and so it is for small clouds in general: at the post-voxel sizes of reg() (4 x 4493 against 886 points) a step's
    // nearest-neighbour launches are ~17 us whichever way (the seeded search forced: 20.3 scans/s against 24.5), the adaptive
    // mode's ten timing probes each drain the stream (~40 us of nothing enqueued)
    if (t_pose_seeded < 0 && seed_mode == 2 && (long long)b * nc <= 24576) seed_mode = 0;
    const bool seeded = seed_mode != 0;
    // Mode 2.  What the seeded search costs depends on the data (a query whose last answer is far away searches a large ball:
    // the hidden side of a complete shape against a one-sided scan; misaligned starts) and falls as the poses converge; the
    // filter costs the same at every step.  Both give the same bits, so the choice is free: the first step times the
    // filter (it is the filter's anyway), every kPoseProbe-th step times the seeded search, and the steps in between
    // take whichever was faster last (two events and one wait per probe: ~10 per call).
    constexpr int kPoseProbe = 25;
    const bool adaptive = seed_mode == 2;
    // (the two events are released on EVERY way out of this function, the early `return 0`s of the loop included: ADVICE r4)
    struct EventPair {
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evs;
    hipEvent_t &ev0 = evs.a, &ev1 = evs.b;
    if (adaptive && (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess)) {
        set_error("pose_optimize: hipEventCreate failed");
        return 0;
    }
    float t_filter = 0.0f, t_seeded = __builtin_inff();
    bool use_seeded = seed_mode == 1;
    int probe_every = kPoseProbe, next_probe = 1;          // (a probe that loses doubles the distance to the next one)
    // (sequential starts: every start probes afresh)
    SeededGrids sg{};
    if (seeded) {
        void *gw = workspace(29, seeded_grids_bytes(b, nc, np), st);
        if (!gw || !build_seeded_grids(b, nc, complete, np, partial, gw, sg, st)) return 0;
    }
    const int fma_mode = arith_mode() != 0 ? 1 : 0;

    const int gb = ceil_div(b, 64);
    hipLaunchKernelGGL(pose_begin_kernel, dim3(gb), dim3(64), 0, st, b, S, accum, -1, 0);
    // (every block of the gradient kernels ends in 13-22 double atomics on its image's accumulators: with many images in
    // flight fewer, longer blocks per image -- GENPC_POSE_GRAD_BLOCKS for A/B)
    static const int env_gb = tune_env("GENPC_POSE_GRAD_BLOCKS", 0, "alignment loop: blocks per image of the gradient kernels (0 = pick)");
    // 32 images (8 scans x 4 starts): 96 blocks per image 153.1 ms per call, 48: 151.0, 24: 150.2, 12: 150.4
    const int g_t = lin_grid(nc), g_g = std::min(env_gb > 0 ? env_gb : (b >= 16 ? 24 : 1024), lin_grid((long long)nc + np));
    const int hstride = starts * (iters + 1);
    PoseSide *dual = dual_small ? pose_side_of(st) : nullptr;      // (small clouds only: where the launches fill the chip by themselves the two
                                                                   //  halves only take each other's compute units -- 8 scans in lock-step 77.6 -> 53.2 scans/s)
    hipStream_t sn = dual ? dual->side : st;        // the stream of the nearest-neighbour launches and pose_grad
    long long dual_step = 0;
    static const int env_fuse_upd = tune_env("GENPC_POSE_FUSE_UPDATE", 1, "alignment loop, full objective: 1 = the Adam update of a step inside the next step's transform launch, 0 = a launch of its own");
    const bool fuse_upd = mask && env_fuse_upd != 0 && b <= 8;      // (measured: 4 elements 25.0 -> 26.1 completed scans/s, 32 elements 75.0 -> 74.1: every block repeats the update)
    PoseState *Sb[2] = {S, S + b};
    double *ab[2] = {accum, accum + (size_t)b * kAcc};
    // The two streams of a step hand over through DEVICE WORDS, not events (round 6): an event record between the transform and
    // the splat and an event wait in front of the next transform each put ~3 us of command-processor latency on the step's critical
    // path (14.1 -> 12.2 ms per 201 steps without them, measured with the ordering switched off).  The transform's blocks count
    // themselves finished (release), a one-wave kernel on the side stream waits for that count in front of the Chamfer half;
    // pose_grad's blocks count themselves, and thread 0 of every block of the NEXT transform waits for that count before it reads
    // the sums (the fused update).  The last step of a start joins through an event as before (its update is a launch of its own).
    static const int env_dual_flags = tune_env("GENPC_POSE_DUAL_FLAGS", 1, "alignment loop with a side stream: 1 = the streams hand over through device counters, 0 = through events");
    // (not under tools that run one kernel at a time -- rocprofv3's counter collection, AMD_SERIALIZE_KERNEL: a kernel that waits for
    //  a count can then sit in front of the kernel that publishes it, until its spin gives up; the events order the launches instead)
    static const bool serialised = [] {
        const char *a = getenv("ROCPROF_COUNTER_COLLECTION"), *b2 = getenv("AMD_SERIALIZE_KERNEL");
        return (a && *a && *a != '0') || (b2 && *b2 && *b2 != '0');
    }();
    const bool flags = dual && fuse_upd && env_dual_flags != 0 && dual->ctr != nullptr && !serialised;
    if (flags) {
        if (dual->dirty) {          // an earlier call left early: counts and expectations may disagree
            if (!check(hipStreamSynchronize(st), "hipStreamSynchronize") || !check(hipStreamSynchronize(dual->side), "hipStreamSynchronize") ||
                !check(hipMemset(dual->ctr, 0, 4 * sizeof(unsigned)), "hipMemset(counters)"))
                return 0;
            dual->tp_count = dual->pg_count = 0;
        }
        dual->dirty = true;
    }
    for (int s = 0; s < starts; s++) {
        if (adaptive) { probe_every = kPoseProbe; next_probe = 1; use_seeded = false; t_seeded = __builtin_inff(); }
        hipLaunchKernelGGL(pose_begin_kernel, dim3(gb), dim3(64), 0, st, b, S, accum, s, lock);
        for (int it = 0; it <= iters; it++) {
            // The update of step it - 1 rides in this step's transform (full objective; fuse_upd): state and accumulators ping-pong
            // between two sets -- this step's kernels read Sc and add into ac, the transform read the other set and zeroed ac.
            const int cur = fuse_upd ? (it & 1) : 0;
            PoseState *Sc = Sb[cur];
            double *ac = ab[cur];
            if (mask) {
                PoseFuse fu{};
                if (fuse_upd && it > 0) {
                    fu.S_in = Sb[cur ^ 1]; fu.S_out = Sc; fu.acc_in = ab[cur ^ 1]; fu.acc_zero = ac;
                    fu.do_update = 1; fu.nc = nc; fu.np = np; fu.lr = lr;
                    fu.history = history ? history + (size_t)s * (iters + 1) + (it - 1) : (float *)nullptr;
                    fu.hstride = hstride;
                    fu.pg_target = flags ? dual->pg_count : 0u;
                }
                fu.ctr = flags ? dual->ctr : (unsigned *)nullptr;
                hipLaunchKernelGGL(pose_transform_project_kernel, dim3(g_t, b), dim3(kQBlock), 0, st, nc, complete,
                                   (const float *)center, 4, (const float *)Sc->params, kStateFloats, pts, 1.1f * radius, render_size,
                                   m.uvr, use_bins(render_size) ? m.bins : (int *)nullptr, render_blend() ? m.zex : (float *)nullptr, fu);
            } else
                hipLaunchKernelGGL(pose_transform_kernel, dim3(g_t, b), dim3(kQBlock), 0, st, nc, complete,
                                   (const float *)center, 4, (const float *)Sc->params, kStateFloats, pts);
            // steps 0 and 2 time the filter (the first one carries the call's one-off costs: the smaller of the two counts),
            // step 1 and then every probe_every-th the seeded search
            if (flags) {
                dual->tp_count += (unsigned)g_t * (unsigned)b;
                hipLaunchKernelGGL(pose_wait_kernel, dim3(1), dim3(kWave), 0, sn, dual->ctr, 0, dual->tp_count);
            } else if (dual) {
                // fork: the side stream's launches of this step wait for the transform
                (void)hipEventRecord(dual->fork[dual_step & 3], st);
                (void)hipStreamWaitEvent(sn, dual->fork[dual_step & 3], 0);
            }
            const bool probe_f = adaptive && (it == 0 || it == 2);
            const bool probe_s = adaptive && it == next_probe && !probe_f;
            const bool probe = probe_f || probe_s;
            // A probe of the seeded search while the filter is in use runs on every kPoseSample-th block only, in front of
            // the filter (which then answers the step, same bits): on a misaligned start of a real scan a full seeded step
            // costs a millisecond against the filter's 16 us, and three such probes were 5 % of a completed scan.
            constexpr int kPoseSample = 4;
            const bool sampled = probe_s && !use_seeded;
            const bool this_seeded = seeded && it > 0 && !probe_f && use_seeded;
            if (probe) (void)hipEventRecord(ev0, sn);
            if (sampled && launch_nn_seeded(b, nc, pts, np, partial, sg, center, 4, (const float *)Sc->params, kStateFloats, d1, i1, d2, i2,
                                            fma_mode, sn, kPoseSample) != 1)
                return 0;
            if (sampled) (void)hipEventRecord(ev1, sn);
            if (this_seeded) {
                if (launch_nn_seeded(b, nc, pts, np, partial, sg, center, 4, (const float *)Sc->params, kStateFloats, d1, i1, d2, i2, fma_mode, sn) != 1)
                    return 0;
            } else if (nn_forward(b, 2, pts, nc, partial, np, d1, i1, partial, np, pts, nc, d2, i2, sn, __builtin_inff(), dup_p, dup_c) != 1) {
                return 0;
            }
            if (probe) {
                float ms = 0.0f;
                if ((sampled || hipEventRecord(ev1, sn) == hipSuccess) && hipEventSynchronize(ev1) == hipSuccess &&
                    hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
                    if (probe_f) {
                        t_filter = it == 0 ? ms : (ms < t_filter ? ms : t_filter);
                    } else {
                        t_seeded = sampled ? ms * kPoseSample : ms;      // (a sample's launch overhead counts four times: errs towards the filter)
                        const bool wins = t_seeded < 0.9f * t_filter;
                        probe_every = wins ? kPoseProbe : 2 * probe_every;
                        next_probe = it + probe_every;
                    }
                    // (the filter is the known quantity: the search must win clearly, and against a clean measurement)
                    use_seeded = it >= 2 && t_seeded < 0.9f * t_filter;
                }
            }
            // one stream and the full objective: pose_grad's blocks ride at the end of the silhouette gradient's launch
            static const int env_ride = tune_env("GENPC_POSE_GRAD_RIDES", 1, "alignment loop, one stream, full objective: 1 = pose_grad's blocks in mask_grad's launch, 0 = a launch of its own");
            const bool ride = mask && !dual && env_ride != 0 && mask_step_carries(render_size);
            PoseGradArgs pga{};
            if (ride) {
                pga.nc = nc; pga.cstride = 4; pga.pstride = kStateFloats; pga.np = np;
                pga.v = complete; pga.center = center; pga.params = Sc->params; pga.partial = partial;
                pga.d1 = d1; pga.d2 = d2; pga.i1 = i1; pga.i2 = i2;
                pga.cd_weight = 3.0f; pga.accum = ac; pga.gx = g_g;
            } else {
                hipLaunchKernelGGL(pose_grad_kernel, dim3(g_g, b), dim3(kQBlock), 0, sn, nc, complete, (const float *)center,
                                   4, (const float *)Sc->params, kStateFloats, np, partial, (const float *)d1,
                                   (const int *)i1, (const float *)d2, (const int *)i2, 3.0f, ac, flags ? dual->ctr : (unsigned *)nullptr);
                if (flags) dual->pg_count += (unsigned)g_g * (unsigned)b;
            }
            if (mask && !mask_step(b, nc, complete, complete_col, center, 4, Sc->params, kStateFloats, radius, render_size,
                                   mask_weight, m, ac, st, true, ride ? &pga : nullptr))
                return 0;
            if (dual && (!flags || it == iters)) {
                // join: the update reads both halves' sums
                (void)hipEventRecord(dual->join[dual_step & 3], sn);
                (void)hipStreamWaitEvent(st, dual->join[dual_step & 3], 0);
                dual_step++;
            }
            // (fused: the next transform takes the update; the last step's is the launch below, in place)
            if (!fuse_upd || it == iters)
                hipLaunchKernelGGL(pose_update_kernel, dim3(gb), dim3(64), 0, st, b, Sc, ac, nc, np, 3.0f, 0.001f, lr, 1,
                                   history ? history + (size_t)s * (iters + 1) + it : (float *)nullptr, hstride);
        }
        // the start's final state back into the first set (where the kernels after the loop, and the next start, look for it)
        if (fuse_upd && (iters & 1) && !check(hipMemcpyAsync(Sb[0], Sb[1], (size_t)b * sizeof(PoseState), hipMemcpyDeviceToDevice, st), "copy pose state")) return 0;
        // (the set the last transform read still holds the step before last's sums: the next start begins with both sets clear)
        if (fuse_upd && iters > 0 && !check(hipMemsetAsync(ab[(iters & 1) ^ 1], 0, (size_t)b * kAcc * sizeof(double), st), "hipMemsetAsync(accumulators)")) return 0;
        hipLaunchKernelGGL(pose_end_kernel, dim3(gb), dim3(64), 0, st, b, S, 0, (float *)nullptr, (float *)nullptr);
    }
    if (lock)
        hipLaunchKernelGGL(pose_pick_kernel, dim3(ceil_div(scans, 64)), dim3(64), 0, st, scans, starts_in, (const PoseState *)S, transform,
                           best_params);
    else
        hipLaunchKernelGGL(pose_end_kernel, dim3(gb), dim3(64), 0, st, b, S, 1, transform, best_params);
    if (flags) dual->dirty = false;          // (every launch of the call has been enqueued: counts and expectations agree)
    return check(hipGetLastError(), "pose_optimize launch") ? 1 : 0;
}

GENPC_API int genpc_pose_optimize_cd_batch(int b, int nc, const float *complete, int np, const float *partial,
                                           float lr, int iters, int starts, float *transform, float *history,
                                           float *best_params, void *stream)
{
    return genpc_pose_optimize_batch(b, nc, complete, nullptr, np, partial, nullptr, lr, iters, starts, 0.0f, 0, 0.0f,
                                     transform, history, best_params, stream);
}

GENPC_API int genpc_pose_optimize_cd(int nc, const float *complete, int np, const float *partial, float lr,
                                     int iters, int starts, float *transform, float *history, float *best_params,
                                     void *stream)
{
    return genpc_pose_optimize_cd_batch(1, nc, complete, np, partial, lr, iters, starts, transform, history,
                                        best_params, stream);
}

namespace genpc {

// accum[e*kAcc + 0..2] += sum of v[e, :, 0..2]
__global__ __launch_bounds__(kQBlock) void mean3_accum_kernel(int n, const float *__restrict__ v,
                                                              double *__restrict__ accum)
{
    __shared__ double red[3][kQBlock / kWave];
    const int e = blockIdx.y;
    v += (size_t)e * n * 3;
    accum += (size_t)e * kAcc;
    double a[3] = {0.0, 0.0, 0.0};
    for (int j = blockIdx.x * kQBlock + threadIdx.x; j < n; j += gridDim.x * kQBlock) {
        a[0] += (double)v[(size_t)j * 3 + 0];
        a[1] += (double)v[(size_t)j * 3 + 1];
        a[2] += (double)v[(size_t)j * 3 + 2];
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double x = wave_sum63(a[k]);
        if (lane == kWave - 1) red[k][wave] = x;
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
    out[e * 4 + k] = (float)(accum[(size_t)e * kAcc + k] / n);
    accum[(size_t)e * kAcc + k] = 0.0;
}

int genpc_mean3(int b, int n, const float *v, float *out, double *accum, hipStream_t st)
{
    if (!check(hipMemsetAsync(accum, 0, (size_t)b * kAcc * sizeof(double), st), "hipMemsetAsync(mean)")) return 0;
    hipLaunchKernelGGL(mean3_accum_kernel, dim3(lin_grid(n), b), dim3(kQBlock), 0, st, n, v, accum);
    hipLaunchKernelGGL(mean3_finish_kernel, dim3(ceil_div(b * 3, 64)), dim3(64), 0, st, b, n, accum, out);
    return check(hipGetLastError(), "mean3 launch") ? 1 : 0;
}

}  // namespace genpc
