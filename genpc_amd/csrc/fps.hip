// fps.hip -- farthest point sampling for gfx950 (the metric of
// main.py:21-24 subsamples both clouds to 16384 points before CD/EMD, reg_xyz.py:215
// samples 20000 fused points).  The reference calls fpsample.fps_sampling, a
// third-party CPU extension with a RANDOM start index (not reproducible, unpinned);
// this is the deterministic counterpart used by the fixtures: start index 0, fp32
// squared distances, first arg-max.  Bit-exact with oracle_fps.
//
// FPS is k strictly sequential steps of (update N running minima, arg-max): what a step
// costs is the LATENCY of its dependency chain, not bandwidth (a step touches no memory but
// the hand-off).  Round 3 layout, built around that chain (round 2: 2.8 us per step):
//   * a cloud is owned by W <= 64 workgroups of 256 threads; every thread keeps R <= 16 points and their
//     running minima in REGISTERS for the whole run -- about 2-4 thousand points per workgroup, one wave per
//     SIMD, so the update of a step is R x 14 instructions (round 2: 16 points on each of 16 waves of one CU,
//     1.2 us of VALU issue per step);
//   * a thread carries the coordinates of its best point along, the wave's best is found with DPP row
//     rotations + v_readlane (no LDS round trips), the workgroup's best with one LDS exchange;
//   * the hand-off carries the winner's COORDINATES: {dist, gen | idx, x, y} as one 16-byte write-through
//     store and {z, gen} as one 8-byte one, each self-tagged with the step's generation (round 2 published
//     the index alone and every workgroup then fetched the pivot from memory: one more dependent miss);
//   * one wave per workgroup polls the cloud's W slots (a lane per slot), picks the global winner
//     and hands the pivot to its workgroup through LDS.  Slots are double-buffered by step parity: no
//     reset, no fences, no grid barrier.
// Several clouds -- of DIFFERENT sizes and sample counts -- run side by side in one launch (grid.y):
// the two subsamplings of the metric and the fused cloud's can share one pass (genpc_fps_multi).
// All workgroups of a cloud must be co-resident; launches are sized from the device's CU count at
// four 256-thread workgroups per CU at most (the hardware admits eight).  The polls are bounded all the same:
// a hand-off that times out aborts the cloud's run (every workgroup leaves its step loop), raises the error
// word and writes -1 to out[0]; genpc_amd/fps.py raises on it.
#include "common.h"
#include "../../include/genpc_hip.h"

namespace genpc {

constexpr int kFThreads = 256;
constexpr int kFWaves = kFThreads / kWave;
constexpr int kFMaxR = 16;
constexpr int kFMaxW = 64;
constexpr int kFMaxJobs = 8;
constexpr int kFPointsPerWg = 2048;      // target; the cap of 64 workgroups raises it for clouds beyond 131072 points

struct FpsJobs {
    const float *xyz[kFMaxJobs];
    int *out[kFMaxJobs];
    int n[kFMaxJobs], k[kFMaxJobs], W[kFMaxJobs];
    int slot0[kFMaxJobs];        // first slot of the job in the slot array (slots are per (job, parity, workgroup))
};

template <int FMA>
__device__ __forceinline__ float sqdist_f(float dx, float dy, float dz)
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

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// maximum over the wave, uniform result: four row rotations leave every row's maximum in all of its
// lanes, four v_readlane fold the rows (values are never NaN here)
__device__ __forceinline__ float wave_max_f32(float v)
{
    v = fmaxf(v, dpp_f32<0x128>(v));      // row_ror:8
    v = fmaxf(v, dpp_f32<0x124>(v));      // row_ror:4
    v = fmaxf(v, dpp_f32<0x122>(v));      // row_ror:2
    v = fmaxf(v, dpp_f32<0x121>(v));      // row_ror:1
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, dpp_i32<0x128>(v));
    v = min(v, dpp_i32<0x124>(v));
    v = min(v, dpp_i32<0x122>(v));
    v = min(v, dpp_i32<0x121>(v));
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

// the lane holding the wave's best (largest v, then lowest index; indices are distinct): uniform lane number.
// Branch-free: ties are the rule, not the exception, once most points are selected (distance 0 everywhere).
__device__ __forceinline__ int wave_best_lane(float v, int idx, float m)
{
    const int cand = v == m ? idx : 0x7fffffff;
    const int best = wave_min_i32(cand);
    return __ffsll((long long)__ballot(cand == best)) - 1;
}

__device__ __forceinline__ float lane_f32(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// 16-byte / 8-byte write-through stores and L1-bypassing loads (sc0 sc1: what agent-scope relaxed atomics
// lower to for 8 bytes; one instruction per granule, observed untorn on gfx950)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_sc_b128(void *p, uint4 v)
{
    const u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void store_sc_b64(void *p, uint2 v)
{
    const u32x2 w = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void load_slot(const void *pa, const void *pb, uint4 &a, uint2 &b)
{
    u32x4 wa;
    u32x2 wb;
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\t"
                 "global_load_dwordx2 %1, %3, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(wa), "=&v"(wb)
                 : "v"(pa), "v"(pb)
                 : "memory");
    a = make_uint4(wa.x, wa.y, wa.z, wa.w);
    b = make_uint2(wb.x, wb.y);
}

struct FpsSlot {          // 32 bytes per (job, parity, workgroup)
    uint4 a;              // dist bits, gen << 20 | idx, x bits, y bits
    uint2 b;              // z bits, gen
    uint2 pad;
};

template <int FMA, int R>
__global__ __launch_bounds__(kFThreads) void fps_kernel(FpsJobs jobs, FpsSlot *slots, int *__restrict__ err)
{
    __shared__ float s_w[kFWaves][5];      // per wave: dist, idx (bits), x, y, z of its best
    __shared__ float s_piv[4];             // pivot x, y, z and index (bits; -1: abort)
    const int job = blockIdx.y, wg = blockIdx.x, t = threadIdx.x;
    const int W = jobs.W[job];
    if (wg >= W) return;
    const int n = jobs.n[job], k = jobs.k[job];
    const int lane = t & (kWave - 1), wave = t >> 6;
    const float *__restrict__ X = jobs.xyz[job];
    int *__restrict__ out = jobs.out[job];
    FpsSlot *S = slots + jobs.slot0[job];
    const int per = (n + W - 1) / W;
    const int lo = wg * per;
    const int hi = min(n, lo + per);
    float px[R], py[R], pz[R], d[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = lo + t + r * kFThreads;
        const bool ok = i < hi;
        const int ii = ok ? i : (n - 1);
        px[r] = X[(size_t)ii * 3 + 0];
        py[r] = X[(size_t)ii * 3 + 1];
        pz[r] = X[(size_t)ii * 3 + 2];
        d[r] = ok ? __builtin_inff() : -1.0f;          // -1: never selected, never updated upward
    }
    float cx = X[0], cy = X[1], cz = X[2];
    int cur = 0;
    for (int s = 0; s < k; s++) {
        if (wg == 0 && t == 0) out[s] = cur;
        if (s == k - 1) break;                          // the last sample needs no successor
        float bv = -1.0f, bx = 0.0f, by = 0.0f, bz = 0.0f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float dd = sqdist_f<FMA>(px[r] - cx, py[r] - cy, pz[r] - cz);
            const float v = d[r] < dd ? d[r] : dd;      // padding slots stay at -1
            d[r] = v;
            const bool gt = v > bv;                     // ascending index within a thread: first max
            bv = gt ? v : bv;
            bi = gt ? lo + t + r * kFThreads : bi;
            bx = gt ? px[r] : bx;
            by = gt ? py[r] : by;
            bz = gt ? pz[r] : bz;
        }
        // wave's best -> LDS
        {
            const float m = wave_max_f32(bv);
            const int src = wave_best_lane(bv, bi, m);
            if (lane == 0) {
                s_w[wave][0] = m;
                s_w[wave][1] = __int_as_float(__builtin_amdgcn_readlane(bi, src));
                s_w[wave][2] = lane_f32(bx, src);
                s_w[wave][3] = lane_f32(by, src);
                s_w[wave][4] = lane_f32(bz, src);
            }
        }
        __syncthreads();
        if (wave == 0) {
            // workgroup's best (every lane computes it: four LDS broadcast reads)
            float m = s_w[0][0];
            int mi = __float_as_int(s_w[0][1]), mw = 0;
#pragma unroll
            for (int w = 1; w < kFWaves; w++) {
                const float v = s_w[w][0];
                const int i = __float_as_int(s_w[w][1]);
                const bool better = v > m || (v == m && i < mi);
                m = better ? v : m;
                mi = better ? i : mi;
                mw = better ? w : mw;
            }
            float wx = s_w[mw][2], wy = s_w[mw][3], wz = s_w[mw][4];
            bool timed_out = false;
            if (W > 1) {
                const unsigned gen = (unsigned)(s % 4095) + 1u;
                FpsSlot *slot = S + (size_t)(s & 1) * W;
                if (lane == 0) {
                    // a workgroup with no live point (m = -1) publishes index 0xFFFFF: it never wins
                    const unsigned pidx = m >= 0.0f ? (unsigned)mi : 0xFFFFFu;
                    store_sc_b128(&slot[wg].a, make_uint4(__float_as_uint(m), (gen << 20) | pidx, __float_as_uint(wx), __float_as_uint(wy)));
                    store_sc_b64(&slot[wg].b, make_uint2(__float_as_uint(wz), gen));
                }
                // poll the W slots of this step (a lane per slot), bounded
                uint4 a = make_uint4(0, 0, 0, 0);
                uint2 b = make_uint2(0, 0);
                int spins = 0;
                for (;;) {
                    bool ready = true;
                    if (lane < W) {
                        load_slot(&slot[lane].a, &slot[lane].b, a, b);
                        ready = (a.y >> 20) == gen && b.y == gen;
                    }
                    if (__all(ready)) break;
                    if (++spins > (1 << 21)) {
                        if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        timed_out = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const float v = lane < W ? __uint_as_float(a.x) : -2.0f;
                const int vi = lane < W ? (int)(a.y & 0xFFFFFu) : 0x7fffffff;
                m = wave_max_f32(v);
                const int src = wave_best_lane(v, vi, m);
                mi = __builtin_amdgcn_readlane(vi, src);
                wx = lane_f32(__uint_as_float(a.z), src);
                wy = lane_f32(__uint_as_float(a.w), src);
                wz = lane_f32(__uint_as_float(b.x), src);
            }
            if (lane == 0) {
                s_piv[0] = wx; s_piv[1] = wy; s_piv[2] = wz;
                s_piv[3] = __int_as_float(timed_out ? -1 : mi);
            }
        }
        __syncthreads();
        cx = s_piv[0]; cy = s_piv[1]; cz = s_piv[2];
        cur = __float_as_int(s_piv[3]);
        // aborted: this workgroup stops publishing, so its peers time out once and leave as well
        // (one bounded spin per workgroup, not one per remaining step)
        if (cur < 0) break;
    }
    // a hand-off that timed out (workgroups of a cloud not co-resident) poisons the
    // result visibly: index 0 is always 0 in a good run
    if (wg == 0 && t == 0 && (cur < 0 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) out[0] = -1;
}

template <int FMA>
static void launch_fps(int R, dim3 grid, hipStream_t st, const FpsJobs &jobs, FpsSlot *slots, int *err)
{
    switch (R) {
    case 1: case 2: hipLaunchKernelGGL((fps_kernel<FMA, 2>), grid, dim3(kFThreads), 0, st, jobs, slots, err); break;
    case 3: case 4: hipLaunchKernelGGL((fps_kernel<FMA, 4>), grid, dim3(kFThreads), 0, st, jobs, slots, err); break;
    case 5: case 6: case 7: case 8: hipLaunchKernelGGL((fps_kernel<FMA, 8>), grid, dim3(kFThreads), 0, st, jobs, slots, err); break;
    case 9: case 10: case 11: case 12: hipLaunchKernelGGL((fps_kernel<FMA, 12>), grid, dim3(kFThreads), 0, st, jobs, slots, err); break;
    default: hipLaunchKernelGGL((fps_kernel<FMA, 16>), grid, dim3(kFThreads), 0, st, jobs, slots, err); break;
    }
}

static int fps_workgroups(int n)
{
    int W = ceil_div(n, kFPointsPerWg);
    if (W > kFMaxW) W = kFMaxW;
    return W < 1 ? 1 : W;
}

}  // namespace genpc

GENPC_API int genpc_fps_multi(int c, const int *n, const int *k, const float *const *xyz, int *const *out_idx, void *stream)
{
    using namespace genpc;
    if (c <= 0) return 1;
    for (int j = 0; j < c; j++) {
        if (n[j] <= 0 || k[j] <= 0 || k[j] > n[j] || n[j] > (1 << 20) - 1 || n[j] > kFMaxW * kFMaxR * kFThreads) {
            fprintf(stderr, "genpc_fps: need 0 < k <= n <= %d\n", kFMaxW * kFMaxR * kFThreads);
            return -1;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    // co-residency budget: four 256-thread workgroups per CU, counted on THIS device (a partitioned or smaller
    // part has fewer than the 256 CUs of a full MI355X)
    int dev = 0, cus = 0;
    if (!check(hipGetDevice(&dev), "hipGetDevice")) return 0;
    if (!check(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return 0;
    if (cus < 1) cus = 1;
    const int budget = cus * 4;
    size_t total_slots = 0;
    for (int j = 0; j < c; j++) total_slots += 2 * (size_t)fps_workgroups(n[j]);
    char *ws = (char *)workspace(7, 256 + total_slots * sizeof(FpsSlot), st);
    if (!ws) return 0;
    int *err = (int *)ws;
    FpsSlot *slots = (FpsSlot *)(ws + 256);
    if (!check(hipMemsetAsync(ws, 0, 256 + total_slots * sizeof(FpsSlot), st), "hipMemsetAsync(fps)")) return 0;
    const bool fma = arith_mode() != 0;
    int slot0 = 0;
    for (int j0 = 0; j0 < c;) {
        // a launch takes up to kFMaxJobs clouds whose workgroups fit the budget together
        FpsJobs jobs = {};
        int nj = 0, wsum = 0, wmax = 0, rmax = 1;
        while (j0 + nj < c && nj < kFMaxJobs) {
            const int j = j0 + nj, W = fps_workgroups(n[j]);
            if (W > budget) {
                set_error("genpc_fps: the cloud needs more co-resident workgroups than the device admits");
                return 0;
            }
            if (nj > 0 && wsum + W > budget) break;
            jobs.xyz[nj] = xyz[j];
            jobs.out[nj] = out_idx[j];
            jobs.n[nj] = n[j];
            jobs.k[nj] = k[j];
            jobs.W[nj] = W;
            jobs.slot0[nj] = slot0;
            slot0 += 2 * W;
            wsum += W;
            wmax = W > wmax ? W : wmax;
            const int R = ceil_div(ceil_div(n[j], W), kFThreads);
            rmax = R > rmax ? R : rmax;
            nj++;
        }
        if (fma) launch_fps<1>(rmax, dim3(wmax, nj), st, jobs, slots, err);
        else launch_fps<0>(rmax, dim3(wmax, nj), st, jobs, slots, err);
        j0 += nj;
    }
    if (!check(hipGetLastError(), "fps launch")) return 0;
    return 1;
}

GENPC_API int genpc_fps(int c, int n, const float *xyz, int k, int *out_idx, void *stream)
{
    if (c <= 0 || k <= 0) return 1;
    // C clouds of one size: chunks of kFMaxJobs through the ragged entry
    for (int c0 = 0; c0 < c; c0 += genpc::kFMaxJobs) {
        const int cc = c - c0 < genpc::kFMaxJobs ? c - c0 : genpc::kFMaxJobs;
        int ns[genpc::kFMaxJobs], ks[genpc::kFMaxJobs];
        const float *xs[genpc::kFMaxJobs];
        int *os[genpc::kFMaxJobs];
        for (int j = 0; j < cc; j++) {
            ns[j] = n; ks[j] = k;
            xs[j] = xyz + (size_t)(c0 + j) * n * 3;
            os[j] = out_idx + (size_t)(c0 + j) * k;
        }
        const int rc = genpc_fps_multi(cc, ns, ks, xs, os, stream);
        if (rc != 1) return rc;
    }
    return 1;
}
