// chamfer.hip -- brute-force nearest neighbour (Chamfer3D forward/backward) for
// gfx950.  Replaces loss_functions/Chamfer3D/chamfer3D.cu of the reference.
//
// Forward design (VALU-bound: ~6.5 fp32 issue slots per (query,target) pair,
// 12 B of target per 64*R pairs -- HBM is idle):
//   * a wave owns 64*R queries held in VGPRs (R per lane) and walks a slice of
//     the targets; target coordinates are wave-uniform, so they are fetched
//     with SCALAR loads (s_load_dwordxN through the scalar cache) and consumed
//     as SGPR operands -- no LDS, no barriers, no VGPRs spent on targets;
//   * both directions (A->B, B->A), all batch elements, all query chunks and
//     all target slices are ONE launch; the grid is sized from the problem
//     (the reference's fixed 32x16 grid leaves 94 % of an MI355X idle at B=1);
//   * the running minimum stays in registers for the whole slice (the
//     reference read-modify-writes result[] in global memory every 512 targets);
//   * the inner loop tracks only the minimum VALUE per chunk of kChunk targets
//     (v_min3_f32: half a slot per pair); the chunk that produced a new
//     minimum is remembered and re-scanned once at the end to recover the
//     FIRST index attaining it, which is the reference's strict-'<' tie-break
//     (chamfer3D.cu:36,46,56,66,119,126);
//   * when targets are split into S>1 slices for occupancy, per-slice
//     (min,argmin) go to scratch and a merge kernel folds them in slice order
//     with strict '<' (earlier slice == lower index wins), so results do not
//     depend on S.
// Arithmetic is written with explicit __fmaf_rn/__fmul_rn/__fadd_rn so the
// compiler cannot re-associate or contract differently from oracle/genpc_oracle.c.
#include "common.h"
#include "../../include/genpc_hip.h"

namespace genpc {

constexpr int kChunk = 32;      // targets per min-only chunk (re-scan granularity)
constexpr int kBlock = 256;     // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;

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
    float *out_d;      // [B, nq]            (S == 1) or scratch [S, B*nq]
    int *out_i;
    int nq, nt;
    int qchunks;       // ceil(nq / (64*R))
    int wave_begin;    // first global wave id of this direction
    int nwaves;        // slices * b * qchunks
};

struct NNArgs {
    NNDir dir[2];
    int ndir;
    int b;
    int slices;        // S
    int slice_len;     // targets per slice, multiple of kChunk
    int total_waves;
};

// One wave = one (direction, batch, query chunk, target slice) unit.
// Waves of a block take consecutive query chunks of the SAME slice so that
// they stream the same target bytes through the scalar cache together.
template <int R, int FMA>
__global__ __launch_bounds__(kBlock) void nn_forward_kernel(NNArgs a)
{
    const int lane = threadIdx.x & (kWave - 1);
    int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
    if (wid >= a.total_waves) return;
    const int d = (a.ndir > 1 && wid >= a.dir[1].wave_begin) ? 1 : 0;
    const NNDir &D = a.dir[d];
    wid -= D.wave_begin;
    if (wid >= D.nwaves) return;
    // wid = (slice * b + batch) * qchunks + qchunk
    const int qc = wid % D.qchunks;
    const int rest = wid / D.qchunks;
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
    const int q0 = qc * (kWave * R) + lane;
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

    int k = k_begin;
    // full chunks
    for (; k + kChunk <= k_end; k += kChunk) {
        const float *__restrict__ tp = T + (size_t)k * 3;
        float m[R];
#pragma unroll
        for (int r = 0; r < R; r++) m[r] = __builtin_inff();
#pragma unroll
        for (int c = 0; c < kChunk; c += 2) {
            const float ax = tp[c * 3 + 0], ay = tp[c * 3 + 1], az = tp[c * 3 + 2];
            const float bx = tp[c * 3 + 3], by = tp[c * 3 + 4], bz = tp[c * 3 + 5];
#pragma unroll
            for (int r = 0; r < R; r++) {
                float d0 = sqdist<FMA>(ax - qx[r], ay - qy[r], az - qz[r]);
                float d1 = sqdist<FMA>(bx - qx[r], by - qy[r], bz - qz[r]);
                m[r] = fminf(fminf(m[r], d0), d1);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            bool lt = m[r] < best[r];
            best[r] = lt ? m[r] : best[r];
            bchunk[r] = lt ? k : bchunk[r];
        }
    }
    // ragged tail (< kChunk targets)
    if (k < k_end) {
        float m[R];
#pragma unroll
        for (int r = 0; r < R; r++) m[r] = __builtin_inff();
        for (int kk = k; kk < k_end; kk++) {
            const float ax = T[(size_t)kk * 3 + 0], ay = T[(size_t)kk * 3 + 1], az = T[(size_t)kk * 3 + 2];
#pragma unroll
            for (int r = 0; r < R; r++) {
                float d0 = sqdist<FMA>(ax - qx[r], ay - qy[r], az - qz[r]);
                m[r] = fminf(m[r], d0);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            bool lt = m[r] < best[r];
            best[r] = lt ? m[r] : best[r];
            bchunk[r] = lt ? k : bchunk[r];
        }
    }

    // Recover the first index attaining best[r] inside the remembered chunk
    // (per-lane gather; kChunk pairs per query, once per slice).
    float *__restrict__ od = D.out_d + ((size_t)slice * a.b + batch) * nq;
    int *__restrict__ oi = D.out_i + ((size_t)slice * a.b + batch) * nq;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int base = bchunk[r];
        int idx = base;
        for (int c = kChunk - 1; c >= 0; c--) {
            int kk = base + c;
            if (kk < k_end) {
                const float tx = T[(size_t)kk * 3 + 0], ty = T[(size_t)kk * 3 + 1], tz = T[(size_t)kk * 3 + 2];
                float dd = sqdist<FMA>(tx - qx[r], ty - qy[r], tz - qz[r]);
                idx = (dd == best[r]) ? kk : idx;
            }
        }
        const int j = q0 + r * kWave;
        if (j < nq) {
            od[j] = best[r];
            oi[j] = idx;
        }
    }
}

// Folds S per-slice partials in slice order; strict '<' keeps the lower index.
__global__ __launch_bounds__(kBlock) void nn_merge_kernel(const float *__restrict__ pd, const int *__restrict__ pi,
                                                          float *__restrict__ out_d, int *__restrict__ out_i,
                                                          int total, int slices)
{
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= total) return;
    float best = pd[j];
    int bi = pi[j];
    for (int s = 1; s < slices; s++) {
        float v = pd[(size_t)s * total + j];
        int vi = pi[(size_t)s * total + j];
        bool lt = v < best;
        best = lt ? v : best;
        bi = lt ? vi : bi;
    }
    out_d[j] = best;
    out_i[j] = bi;
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

constexpr int kR = 4;   // queries per lane

template <int FMA>
static bool launch_forward(const NNArgs &a, hipStream_t st)
{
    int blocks = ceil_div(a.total_waves, kWavesPerBlock);
    hipLaunchKernelGGL((nn_forward_kernel<kR, FMA>), dim3(blocks), dim3(kBlock), 0, st, a);
    return check(hipGetLastError(), "nn_forward_kernel launch");
}

// Slices: enough waves to give every SIMD a few, never slices shorter than 8 chunks.
static int pick_slices(long long waves_unsplit, int nt_max)
{
    const long long want = (long long)kNumSIMD * 4;
    if (waves_unsplit >= want) return 1;
    long long s = ceil_div64(want, waves_unsplit > 0 ? waves_unsplit : 1);
    long long max_s = nt_max / (kChunk * 8);
    if (max_s < 1) max_s = 1;
    if (s > max_s) s = max_s;
    return (int)s;
}

static int nn_forward(int b, int ndir, const float *q0, int n0, const float *t0, int m0, float *d0, int *i0,
                      const float *q1, int n1, const float *t1, int m1, float *d1, int *i1, hipStream_t st)
{
    // A direction with no queries or no targets does nothing (the reference's
    // loops do not execute, outputs keep the caller's zeros).
    NNArgs a{};
    a.b = b;
    const float *qs[2] = {q0, q1};
    const float *ts[2] = {t0, t1};
    float *ds[2] = {d0, d1};
    int *is[2] = {i0, i1};
    int nqs[2] = {n0, n1}, nts[2] = {m0, m1};
    int nd = 0;
    long long waves = 0;
    int nt_max = 0;
    for (int d = 0; d < ndir; d++) {
        if (b <= 0 || nqs[d] <= 0 || nts[d] <= 0) continue;
        NNDir &D = a.dir[nd++];
        D.q = qs[d]; D.t = ts[d]; D.out_d = ds[d]; D.out_i = is[d];
        D.nq = nqs[d]; D.nt = nts[d];
        D.qchunks = ceil_div(D.nq, kWave * kR);
        waves += (long long)b * D.qchunks;
        if (D.nt > nt_max) nt_max = D.nt;
    }
    a.ndir = nd;
    if (nd == 0) return 1;
    const int S = pick_slices(waves, nt_max);
    a.slices = S;
    a.slice_len = ceil_div(ceil_div(nt_max, S), kChunk) * kChunk;
    // recompute S so that no slice is empty for the LONGEST target set; shorter
    // target sets may have empty trailing slices (they write +inf, merged away).
    a.slices = ceil_div(nt_max, a.slice_len);
    float *final_d[2] = {nullptr, nullptr};
    int *final_i[2] = {nullptr, nullptr};
    if (a.slices > 1) {
        size_t tot = 0;
        for (int d = 0; d < nd; d++) tot += (size_t)a.slices * b * a.dir[d].nq;
        char *ws = (char *)workspace(0, tot * 8, st);
        if (!ws) return 0;
        float *wd = (float *)ws;
        int *wi = (int *)(ws + tot * 4);
        size_t off = 0;
        for (int d = 0; d < nd; d++) {
            final_d[d] = a.dir[d].out_d;
            final_i[d] = a.dir[d].out_i;
            a.dir[d].out_d = wd + off;
            a.dir[d].out_i = wi + off;
            off += (size_t)a.slices * b * a.dir[d].nq;
        }
    }
    long long tw = 0;
    for (int d = 0; d < nd; d++) {
        // round each direction up to whole blocks so that the waves of a block
        // share a slice and a direction
        a.dir[d].wave_begin = (int)tw;
        long long w = (long long)a.slices * b * a.dir[d].qchunks;
        a.dir[d].nwaves = (int)w;
        tw += ceil_div64(w, kWavesPerBlock) * kWavesPerBlock;
    }
    if (tw > 0x7fffffffLL / 2) {
        set_error("chamfer: problem too large for one launch");
        return 0;
    }
    a.total_waves = (int)tw;
    bool ok = arith_mode() ? launch_forward<1>(a, st) : launch_forward<0>(a, st);
    if (!ok) return 0;
    if (a.slices > 1) {
        for (int d = 0; d < nd; d++) {
            int total = b * a.dir[d].nq;
            hipLaunchKernelGGL(nn_merge_kernel, dim3(ceil_div(total, kBlock)), dim3(kBlock), 0, st,
                               (const float *)a.dir[d].out_d, (const int *)a.dir[d].out_i, final_d[d], final_i[d],
                               total, a.slices);
        }
        if (!check(hipGetLastError(), "nn_merge_kernel launch")) return 0;
    }
    return 1;
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
