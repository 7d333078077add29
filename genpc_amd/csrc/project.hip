// project.hip -- point -> image projection, splat and colour gather for gfx950
// (SURVEY.md 8a rows a13-a15): DepthPrompting.getUvs / paintPixels and
// ScaleAdapter.colorPoint of the reference.  All three are O(N) streaming / scatter
// kernels bound by HBM (a13: 12 B read + 12..24 B written per point per camera).
//
// getUvs: the reference materialises cam.transform(points) for all 1024 cameras
// ([1024,N,3] fp32, 0.88 GB at N = 71k) and then consumes two rows
// (DepthPrompting.py:154-165).  Here any subset of cameras is projected in two
// passes over the points: pass 1 only reduces the per-camera bounding box of the
// NDC xy (lane = camera, points broadcast from LDS, no stores), pass 2 projects again
// (lane = point, cameras broadcast from LDS), rescales and stores uv / depth -- 12 B per
// (camera, point) through HBM instead of 28 with an in-place second pass; the cloud itself
// is read from HBM once.  `transformed` is optional.
// Arithmetic (fma order, IEEE division) matches oracle/genpc_oracle_geom.c bit for
// bit; min/max are exact, so uv is bit-exact too.
#include "common.h"
#include "fastdiv.h"
#include "../../include/genpc_hip.h"

#include <stdlib.h>
#include <algorithm>
#include <mutex>

namespace genpc {

constexpr int kPBlock = 256;

// float -> unsigned key with the same ordering
__device__ __forceinline__ unsigned f2key(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    // branch-free: the select form crashes hipcc 7.2's instruction selection inside project_write_kernel
    return __uint_as_float(k ^ ((unsigned)((int)~k >> 31) | 0x80000000u));
}

__device__ __forceinline__ void project_point(const float *v, float focal, float A, float B, float px, float py,
                                              float pz, float &ox, float &oy, float &oz)
{
    const float xc = __fadd_rn(__fmaf_rn(v[2], pz, __fmaf_rn(v[1], py, __fmul_rn(v[0], px))), v[3]);
    const float yc = __fadd_rn(__fmaf_rn(v[6], pz, __fmaf_rn(v[5], py, __fmul_rn(v[4], px))), v[7]);
    const float zc = __fadd_rn(__fmaf_rn(v[10], pz, __fmaf_rn(v[9], py, __fmul_rn(v[8], px))), v[11]);
    const float w = -zc;
    ox = __fmul_rn(focal, xc) / w;
    oy = __fmul_rn(focal, yc) / w;
    oz = __fmaf_rn(A, zc, B) / w;
}

// Pass 1: per-camera bounding box of the NDC xy, nothing stored per point.  Two kernels, exact result:
//
//  (a) project_bbox_approx_kernel -- every (camera, point) once, CHEAPLY: the camera-space coordinates with the
//      oracle's own operations (packed fp32: two points per instruction), the two quotients as a * v_rcp(w)
//      (relative error < 2^-22 against the correctly rounded quotient the oracle computes), running float
//      min/max per (camera, slice of the cloud).  Lane = camera (view matrix in registers), wave = slice (256 of
//      them), its points broadcast from LDS (SoA so that a 16-byte read delivers four x's as two register pairs).
//  (b) project_box_exact_kernel, one block per camera -- a slice can hold the exact extreme only if its
//      approximate extreme is within 2^-19 (relative to the largest |quotient|) of the best one: typically four
//      or five of the 256.  A camera whose approximate pass met a non-finite quotient takes all its slices.
//  (c) same kernel -- the candidate slices again with IEEE divisions (lane = point), exact min/max through
//      order-preserving keys: the camera's final box.
//
// Rounding is monotone, so the maximum of the rounded quotients is attained at the point of the largest true
// quotient, which (b) can not drop: |approx - exact| <= 2^-22 |exact| for every point, the window is 8 x wider
// and carries an absolute 2^-120 for flushed denormals.  (A brute-force exact pass, round 2, spent 116 us on two
// IEEE divisions per pair; a single pass that divides exactly only inside a running window divides every time
// because a wave divides whenever one lane asks.)
constexpr int kBoxBlocks = 64;                       // blocks per camera group: 16 groups x 64 = 1024 blocks at C = 1024
constexpr int kBoxSlices = kBoxBlocks * (kPBlock / kWave);      // 256 slices of the cloud, one per wave
constexpr int kBoxChunk = 512;                       // points a wave stages per LDS round (6 KiB, wave-private)

__device__ __forceinline__ int slice_len(int n) { return ((n + kBoxSlices - 1) / kBoxSlices + 3) & ~3; }   // multiple of 4

__global__ __launch_bounds__(kPBlock) void project_bbox_approx_kernel(int c, int n, const float *__restrict__ view, float focal,
                                                                      const float *__restrict__ xyz, float *__restrict__ part)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float s_pts[kPBlock / kWave][3][kBoxChunk];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int cam = blockIdx.y * kWave + lane;
    const int slice = blockIdx.x * (kPBlock / kWave) + wave;
    float *sx = s_pts[wave][0], *sy = s_pts[wave][1], *sz = s_pts[wave][2];
    const float *V = view + (size_t)(cam < c ? cam : c - 1) * 12;
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = V[k];
    const v2f v0 = {v[0], v[0]}, v1 = {v[1], v[1]}, v2 = {v[2], v[2]}, v3 = {v[3], v[3]};
    const v2f v4 = {v[4], v[4]}, v5 = {v[5], v[5]}, v6 = {v[6], v[6]}, v7 = {v[7], v[7]};
    const v2f v8 = {v[8], v[8]}, v9 = {v[9], v[9]}, v10 = {v[10], v[10]}, v11 = {v[11], v[11]};
    const v2f foc = {focal, focal}, zero = {0.0f, 0.0f};
    const int per = slice_len(n);
    const int p_begin = min(n, slice * per), p_end = min(n, p_begin + per);
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    v2f bad = {0.0f, 0.0f};                 // becomes NaN when a quotient is not finite
    // the staging area is the wave's own: LDS operations of one wave complete in order, no block barrier
    for (int p0 = p_begin; p0 < p_end; p0 += kBoxChunk) {
        const int cnt = min(kBoxChunk, p_end - p0);
        for (int i = lane; i < cnt * 3; i += kWave) {
            const float val = xyz[(size_t)p0 * 3 + i];
            const int pt = i / 3, comp = i - pt * 3;
            (comp == 0 ? sx : (comp == 1 ? sy : sz))[pt] = val;
        }
        // groups of four are read whole: pad the last one with copies of the chunk's first point
        const int groups = (cnt + 3) >> 2;
        if (lane < groups * 4 - cnt) {
            const float x0 = xyz[(size_t)p0 * 3 + 0], y0 = xyz[(size_t)p0 * 3 + 1], z0 = xyz[(size_t)p0 * 3 + 2];
            sx[cnt + lane] = x0; sy[cnt + lane] = y0; sz[cnt + lane] = z0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
        for (int g = 0; g < groups; g++) {
            const float4 X = *(const float4 *)&sx[g * 4], Y = *(const float4 *)&sy[g * 4], Z = *(const float4 *)&sz[g * 4];
            const v2f px[2] = {{X.x, X.y}, {X.z, X.w}}, py[2] = {{Y.x, Y.y}, {Y.z, Y.w}}, pz[2] = {{Z.x, Z.y}, {Z.z, Z.w}};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // the oracle's operation order: fadd(fma(v2, z, fma(v1, y, fmul(v0, x))), v3)
                const v2f xc = __builtin_elementwise_fma(v2, pz[h], __builtin_elementwise_fma(v1, py[h], v0 * px[h])) + v3;
                const v2f yc = __builtin_elementwise_fma(v6, pz[h], __builtin_elementwise_fma(v5, py[h], v4 * px[h])) + v7;
                const v2f zc = __builtin_elementwise_fma(v10, pz[h], __builtin_elementwise_fma(v9, py[h], v8 * px[h])) + v11;
                const v2f r = {__builtin_amdgcn_rcpf(-zc.x), __builtin_amdgcn_rcpf(-zc.y)};
                const v2f qx = (foc * xc) * r, qy = (foc * yc) * r;
                mnx = fminf(fminf(mnx, qx.x), qx.y); mxx = fmaxf(fmaxf(mxx, qx.x), qx.y);
                mny = fminf(fminf(mny, qy.x), qy.y); mxy = fmaxf(fmaxf(mxy, qy.x), qy.y);
                bad = __builtin_elementwise_fma(qx, zero, bad);
                bad = __builtin_elementwise_fma(qy, zero, bad);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (cam < c) {
        if (bad.x + bad.y != 0.0f) mnx = mny = mxx = mxy = __builtin_nanf("");        // NaN or (never) a non-zero sum
        *(float4 *)(part + ((size_t)cam * kBoxSlices + slice) * 4) = make_float4(mnx, mny, mxx, mxy);
    }
}

// (b) + (c): one block per camera.  Thread = slice: find the candidate slices; then the four waves take one
// candidate slice each (lane = point, all of a slice's points in flight at once) with IEEE divisions and
// order-preserving keys; the block's minimum keys are the camera's final box.
constexpr int kExactUnroll = 5;
__global__ __launch_bounds__(kPBlock) void project_box_exact_kernel(int n, const float *__restrict__ view, float focal,
                                                                    const float *__restrict__ xyz, const float *__restrict__ part,
                                                                    unsigned *__restrict__ keys)
{
    static_assert(kBoxSlices == kPBlock, "one thread per slice");
    __shared__ int s_list[kBoxSlices];
    __shared__ float s_red[kPBlock / kWave][4];
    __shared__ int s_cnt[kPBlock / kWave];
    __shared__ int s_weird;
    __shared__ unsigned s_keys[4];
    const int cam = blockIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const float4 b = *(const float4 *)(part + ((size_t)cam * kBoxSlices + threadIdx.x) * 4);
    if (threadIdx.x < 4) s_keys[threadIdx.x] = 0xffffffffu;
    if (threadIdx.x == 0) s_weird = 0;
    float q0 = b.x, q1 = b.y, q2 = b.z, q3 = b.w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        q0 = fminf(q0, __shfl_xor(q0, off, kWave)); q1 = fminf(q1, __shfl_xor(q1, off, kWave));
        q2 = fmaxf(q2, __shfl_xor(q2, off, kWave)); q3 = fmaxf(q3, __shfl_xor(q3, off, kWave));
    }
    if (lane == 0) { s_red[wave][0] = q0; s_red[wave][1] = q1; s_red[wave][2] = q2; s_red[wave][3] = q3; }
    __syncthreads();
    if (__ballot(b.x != b.x) != 0ull && lane == 0) s_weird = 1;
#pragma unroll
    for (int w2 = 0; w2 < kPBlock / kWave; w2++) {
        q0 = fminf(q0, s_red[w2][0]); q1 = fminf(q1, s_red[w2][1]); q2 = fmaxf(q2, s_red[w2][2]); q3 = fmaxf(q3, s_red[w2][3]);
    }
    const float big = fmaxf(fmaxf(fabsf(q0), fabsf(q1)), fmaxf(fabsf(q2), fabsf(q3)));
    const float tol = __fmaf_rn(big, 1.9073486328125e-06f /* 2^-19 */, 7.52316384526264e-37f /* 2^-120 */);
    bool cand = b.x <= q0 + tol || b.y <= q1 + tol || b.z >= q2 - tol || b.w >= q3 - tol;
    __syncthreads();
    if (s_weird || !(big < INFINITY)) cand = true;            // a non-finite quotient somewhere: every slice exactly
    const unsigned long long m = __ballot(cand);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0, ncand = 0;
#pragma unroll
    for (int w2 = 0; w2 < kPBlock / kWave; w2++) {
        base += w2 < wave ? s_cnt[w2] : 0;
        ncand += s_cnt[w2];
    }
    if (cand) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = threadIdx.x;
    __syncthreads();
    const int per = slice_len(n);
    const float *V = view + (size_t)cam * 12;
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = V[k];
    unsigned mnx = 0xffffffffu, mny = 0xffffffffu, mxx = 0xffffffffu, mxy = 0xffffffffu;
    for (int r = wave; r < ncand; r += kPBlock / kWave) {
        const int p_begin = min(n, s_list[r] * per), p_end = min(n, p_begin + per);
        for (int p0 = p_begin + lane; p0 < p_end; p0 += kWave * kExactUnroll) {
            float px[kExactUnroll], py[kExactUnroll], pz[kExactUnroll];
#pragma unroll
            for (int u = 0; u < kExactUnroll; u++) {
                const int pp = min(p0 + u * kWave, p_end - 1);          // past the end: the slice's last point again
                px[u] = xyz[(size_t)pp * 3 + 0]; py[u] = xyz[(size_t)pp * 3 + 1]; pz[u] = xyz[(size_t)pp * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < kExactUnroll; u++) {
                const float xc = __fadd_rn(__fmaf_rn(v[2], pz[u], __fmaf_rn(v[1], py[u], __fmul_rn(v[0], px[u]))), v[3]);
                const float yc = __fadd_rn(__fmaf_rn(v[6], pz[u], __fmaf_rn(v[5], py[u], __fmul_rn(v[4], px[u]))), v[7]);
                const float zc = __fadd_rn(__fmaf_rn(v[10], pz[u], __fmaf_rn(v[9], py[u], __fmul_rn(v[8], px[u]))), v[11]);
                const float w = -zc;
                const unsigned kx = f2key(__fmul_rn(focal, xc) / w), ky = f2key(__fmul_rn(focal, yc) / w);
                mnx = min(mnx, kx); mny = min(mny, ky);
                mxx = min(mxx, ~kx); mxy = min(mxy, ~ky);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mnx = min(mnx, (unsigned)__shfl_xor((int)mnx, off, kWave)); mny = min(mny, (unsigned)__shfl_xor((int)mny, off, kWave));
        mxx = min(mxx, (unsigned)__shfl_xor((int)mxx, off, kWave)); mxy = min(mxy, (unsigned)__shfl_xor((int)mxy, off, kWave));
    }
    if (lane == 0) {
        atomicMin(&s_keys[0], mnx); atomicMin(&s_keys[1], mny); atomicMin(&s_keys[2], mxx); atomicMin(&s_keys[3], mxy);
    }
    __syncthreads();
    if (threadIdx.x < 4) keys[(size_t)cam * 4 + threadIdx.x] = s_keys[threadIdx.x];
}

// Pass 2: lane = 4 points (loaded ONCE, kept in registers), loop over the block's 64 cameras whose
// records come from LDS as broadcast reads.  Projects again (recomputing costs nothing next to a
// 28 B/point round trip through HBM), rescales (DepthPrompting.py:246-266) and stores uv, depth and,
// if asked, the NDC point.
// Store shape (tools/ubench_write3.hip, pure stores of this layout at 1024 x 71372): a wave owns 256
// consecutive points, lane l the points {2l, 2l+1, 128+2l, 129+2l}, so every store instruction writes one
// contiguous run (uv 2 x 1 KiB, depth 2 x 512 B per camera row); and consecutive block ids walk the CAMERA
// groups of one point block, not the point blocks of one camera group: rows are N x 8 B apart and N is
// not a multiple of 16, so neighbouring point blocks share a partly written 128-byte line per row --
// launched side by side they cost 172 us against 140 us in this order (plain memset: 129 us).  (Shifting each
// row's points by (row * n) mod 32 so that every store starts on a 128-byte line gains 5 % on pure stores
// (tools/ubench_write3.hip, pattern 3) but needs the points from LDS per row: 221 -> 260 us for the call.)
constexpr int kCamGroup = 64;
constexpr int kWritePer = 4;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ v2f splat2(float a) { return (v2f){a, a}; }

// Arithmetic of the loop: the oracle's operations (same order, same roundings), two per instruction -- (xc, yc)
// of a point and zc of two points as packed fp32, the five correctly rounded divisions per (camera, point) as
// fastdiv.h's shared-reciprocal core (ox, oy, oz share 1/w; the two rescale quotients share the camera's 1/sc,
// refined once per block).  A wave whose operands leave fastdiv's safe range (a point on the camera plane, a
// point exactly at the box centre, ...) redoes that camera with the compiler's divisions.
// (round 2: ~270 VALU + 20 v_rcp per camera and wave, 146 us of issue for 1024 x 71372 -- more than the stores.)
__global__ __launch_bounds__(kPBlock) void project_write_kernel(int c, int n, int cgroups, const float *__restrict__ view,
                                                                const unsigned *__restrict__ keys, float focal,
                                                                float A, float B, const float *__restrict__ xyz, int rescale,
                                                                float padmul, float *__restrict__ transformed,
                                                                float *__restrict__ uv, float *__restrict__ depth,
                                                                float *__restrict__ bbox)
{
    // per camera: (v0 v4 v1 v5) (v2 v6 v3 v7) (v8 v9 v10 v11) (cx cy sc 1/sc): rows x and y interleaved so that a
    // 16-byte broadcast read delivers them as register pairs
    __shared__ __attribute__((aligned(16))) float rec[kCamGroup * 16];
    const int cam0 = (int)(blockIdx.x % (unsigned)cgroups) * kCamGroup;
    const int pblock = (int)(blockIdx.x / (unsigned)cgroups);
    const int ncam = min(kCamGroup, c - cam0);
    for (int i = threadIdx.x; i < ncam * 12; i += kPBlock) {
        const int cam = i / 12, j = i - cam * 12;
        const int pos = j >= 8 ? j : ((j & 3) * 2 + (j >> 2));
        rec[cam * 16 + pos] = view[(size_t)cam0 * 12 + i];
    }
    if (threadIdx.x < ncam) {
        // the camera's box (pass 1) -> centre and extent, DepthPrompting.py:246-262
        float cx = 0.0f, cy = 0.0f, sc = 1.0f;
        if (keys) {
            const unsigned *kq = keys + (size_t)(cam0 + threadIdx.x) * 4;
            const float mnx = key2f(kq[0]), mny = key2f(kq[1]), mxx = key2f(~kq[2]), mxy = key2f(~kq[3]);
            if (bbox && pblock == 0) {
                float *bo = bbox + (size_t)(cam0 + threadIdx.x) * 4;
                bo[0] = mnx; bo[1] = mny; bo[2] = mxx; bo[3] = mxy;
            }
            cx = __fmul_rn(__fadd_rn(mnx, mxx), 0.5f);       // == / 2.0f (exact scaling; a subnormal sum halves with the same rounding)
            cy = __fmul_rn(__fadd_rn(mny, mxy), 0.5f);
            const float sx = __fsub_rn(mxx, mnx), sy = __fsub_rn(mxy, mny);
            sc = sx > sy ? sx : sy;
        }
        *(float4 *)&rec[threadIdx.x * 16 + 12] = make_float4(cx, cy, sc, rcp_refined(sc));
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int wbase = (pblock * (kPBlock / kWave) + wave) * (kWave * kWritePer);
    const int j0 = wbase + lane * 2;                 // points j0, j0+1, j0+128, j0+129
    float p[kWritePer][3];
#pragma unroll
    for (int t = 0; t < kWritePer; t++) {
        const int j = j0 + (t >> 1) * 128 + (t & 1);
        const int jj = j < n ? j : n - 1;
        p[t][0] = xyz[(size_t)jj * 3 + 0]; p[t][1] = xyz[(size_t)jj * 3 + 1]; p[t][2] = xyz[(size_t)jj * 3 + 2];
    }
    const v2f PX[2] = {{p[0][0], p[1][0]}, {p[2][0], p[3][0]}}, PY[2] = {{p[0][1], p[1][1]}, {p[2][1], p[3][1]}},
              PZ[2] = {{p[0][2], p[1][2]}, {p[2][2], p[3][2]}};
    const bool full = wbase + kWave * kWritePer <= n;      // wave-uniform
    __syncthreads();
    if (wbase >= n) return;
    for (int k = 0; k < ncam; k++) {
        const float4 r0 = *(const float4 *)&rec[k * 16 + 0], r1 = *(const float4 *)&rec[k * 16 + 4],
                     r2 = *(const float4 *)&rec[k * 16 + 8], r3 = *(const float4 *)&rec[k * 16 + 12];
        v2f uvv[kWritePer], oxy[kWritePer], oz[2];
        DivRange rng;
        {
            const v2f vxy0 = {r0.x, r0.y}, vxy1 = {r0.z, r0.w}, vxy2 = {r1.x, r1.y}, vxy3 = {r1.z, r1.w};
            v2f w[2], rr[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const v2f zc = __builtin_elementwise_fma(splat2(r2.z), PZ[h], __builtin_elementwise_fma(splat2(r2.y), PY[h], splat2(r2.x) * PX[h])) + splat2(r2.w);
                w[h] = -zc;
                const v2f q = {__builtin_amdgcn_rcpf(w[h].x), __builtin_amdgcn_rcpf(w[h].y)};
                rr[h] = __builtin_elementwise_fma(__builtin_elementwise_fma(-w[h], q, splat2(1.0f)), q, q);
                const v2f nz = __builtin_elementwise_fma(splat2(A), zc, splat2(B));
                oz[h] = div_core2(nz, w[h], rr[h]);
                rng.add(w[h].x, w[h].y);
                rng.add(nz.x, nz.y);
            }
#pragma unroll
            for (int t = 0; t < kWritePer; t++) {
                const v2f xy = __builtin_elementwise_fma(vxy2, splat2(p[t][2]), __builtin_elementwise_fma(vxy1, splat2(p[t][1]), vxy0 * splat2(p[t][0]))) + vxy3;
                const v2f a = splat2(focal) * xy;
                const float wt = (t & 1) ? w[t >> 1].y : w[t >> 1].x, rt = (t & 1) ? rr[t >> 1].y : rr[t >> 1].x;
                oxy[t] = div_core2(a, splat2(wt), splat2(rt));
                rng.add(a.x, a.y);
                if (rescale) {
                    const v2f d = oxy[t] - (v2f){r3.x, r3.y};
                    uvv[t] = div_core2(d, splat2(r3.z), splat2(r3.w)) * splat2(padmul) + splat2(0.5f);
                    rng.add(d.x, d.y);
                } else {
                    uvv[t] = (oxy[t] + splat2(1.0f)) * splat2(0.5f);
                }
            }
            if (rescale) rng.add(r3.z);
        }
        if (__ballot(!rng.ok()) != 0ull) {
            // some operand outside fastdiv's range: this camera again with the compiler's divisions
            const float v[12] = {r0.x, r0.z, r1.x, r1.z, r0.y, r0.w, r1.y, r1.w, r2.x, r2.y, r2.z, r2.w};
#pragma unroll
            for (int t = 0; t < kWritePer; t++) {
                float ox, oy, ozt;
                project_point(v, focal, A, B, p[t][0], p[t][1], p[t][2], ox, oy, ozt);
                oxy[t] = (v2f){ox, oy};
                if (t & 1) oz[t >> 1].y = ozt; else oz[t >> 1].x = ozt;
                if (rescale) {
                    uvv[t].x = __fadd_rn(__fmul_rn(__fsub_rn(ox, r3.x) / r3.z, padmul), 0.5f);
                    uvv[t].y = __fadd_rn(__fmul_rn(__fsub_rn(oy, r3.y) / r3.z, padmul), 0.5f);
                } else {
                    uvv[t].x = __fmul_rn(__fadd_rn(ox, 1.0f), 0.5f);
                    uvv[t].y = __fmul_rn(__fadd_rn(oy, 1.0f), 0.5f);
                }
            }
        }
        const size_t q = (size_t)(cam0 + k) * n + j0;
        if (full) {
            f4u s0 = {uvv[0].x, uvv[0].y, uvv[1].x, uvv[1].y}, s1 = {uvv[2].x, uvv[2].y, uvv[3].x, uvv[3].y};
            f2u d0 = {oz[0].x, oz[0].y}, d1 = {oz[1].x, oz[1].y};
            *(f4u *)(uv + q * 2) = s0;
            *(f4u *)(uv + (q + 128) * 2) = s1;
            *(f2u *)(depth + q) = d0;
            *(f2u *)(depth + q + 128) = d1;
        } else {
#pragma unroll
            for (int t = 0; t < kWritePer; t++) {
                const int off = (t >> 1) * 128 + (t & 1);
                if (j0 + off < n) {
                    uv[(q + off) * 2 + 0] = uvv[t].x;
                    uv[(q + off) * 2 + 1] = uvv[t].y;
                    depth[q + off] = (t & 1) ? oz[t >> 1].y : oz[t >> 1].x;
                }
            }
        }
        if (transformed) {
#pragma unroll
            for (int t = 0; t < kWritePer; t++) {
                const int off = (t >> 1) * 128 + (t & 1);
                if (j0 + off < n) {
                    transformed[(q + off) * 3 + 0] = oxy[t].x; transformed[(q + off) * 3 + 1] = oxy[t].y;
                    transformed[(q + off) * 3 + 2] = (t & 1) ? oz[t >> 1].y : oz[t >> 1].x;
                }
            }
        }
    }
}

// DepthPrompting.py:179-184, ScaleAdapter.py:59-62
__global__ __launch_bounds__(kPBlock) void uv_to_pixels_kernel(int n, const float *__restrict__ uv, float res,
                                                               int clip_max, int *__restrict__ pix)
{
    const int j = blockIdx.x * kPBlock + threadIdx.x;
    if (j >= n) return;
    long long pu = (long long)__fmul_rn(uv[(size_t)j * 2 + 0], res);
    long long pv = (long long)__fmul_rn(uv[(size_t)j * 2 + 1], res);
    pu = pu < 0 ? 0 : (pu > clip_max ? clip_max : pu);
    pv = pv < 0 ? 0 : (pv > clip_max ? clip_max : pv);
    pix[(size_t)j * 2 + 0] = (int)pv;
    pix[(size_t)j * 2 + 1] = (int)pu;
}

// paintPixels pass 1: highest point index covering a pixel owns it (the reference's
// index_put order on the CPU; undefined on its GPU path).
__global__ __launch_bounds__(kPBlock) void splat_owner_kernel(int res, int n, const int *__restrict__ pix,
                                                              int point_size, int *__restrict__ owner)
{
    const int side = 2 * point_size - 1;
    const long long total = (long long)n * side * side;
    for (long long t = (long long)blockIdx.x * kPBlock + threadIdx.x; t < total; t += (long long)gridDim.x * kPBlock) {
        const int j = (int)(t / (side * side));
        const int o = (int)(t % (side * side));
        const int r = pix[(size_t)j * 2 + 0] + o / side - (point_size - 1);
        const int c = pix[(size_t)j * 2 + 1] + o % side - (point_size - 1);
        if (r < 0 || r >= res || c < 0 || c >= res) continue;
        atomicMax(&owner[r * res + c], j);
    }
}

// pass 2: write the owners' colours into img (in place) and the flipped copy
__global__ __launch_bounds__(kPBlock) void splat_write_kernel(int res, const int *__restrict__ owner,
                                                              const float *__restrict__ colors, int ch,
                                                              float *__restrict__ img, float *__restrict__ out)
{
    const int t = blockIdx.x * kPBlock + threadIdx.x;
    if (t >= res * res) return;
    const int r = t / res, c = t % res;
    const int o = owner[t];
    for (int k = 0; k < ch; k++) {
        float v = img[((size_t)k * res + r) * res + c];
        if (o >= 0) {
            v = colors[(size_t)o * ch + k];
            img[((size_t)k * res + r) * res + c] = v;
        }
        out[((size_t)k * res + (res - 1 - r)) * res + c] = v;
    }
}

// ---- paintPixels for MANY points (round 6): the owner election in LDS, per pixel tile ----
// splat_owner_kernel elects with one global atomicMax per (point, pixel): an agent-scope atomic bypasses the L2 -- a 64-byte
// memory transaction per point (2 M points on 1024^2: 355 MB of traffic for 67 MB of algorithmic bytes, 0.068 of the HBM
// roofline).  Here the points are binned by 64 x 64-pixel tile first (a counting sort: count, scan, scatter -- block-level LDS
// histograms, one global atomic per block and tile), then a block per tile elects in a 16 KB LDS copy of its pixels' owners and
// writes owner, img and the flipped out for its tile.  Same result: the highest point index covering a pixel owns it.
constexpr int kPaintTile = 64, kPaintTileShift = 6;
constexpr int kPaintMaxTiles = 4096;          // res <= 4096
constexpr int kPaintChunk = 4096;             // points a block of the count / scatter kernels handles
constexpr int kPaintBlock = 1024;

// the tiles the stamp of pixel (r, c) touches: [tr0, tr1] x [tc0, tc1], false if it lies outside the image
__device__ __forceinline__ bool paint_tiles_of(int r, int c, int ps, int res, int &tr0, int &tr1, int &tc0, int &tc1)
{
    const int r0 = max(r - (ps - 1), 0), r1 = min(r + (ps - 1), res - 1);
    const int c0 = max(c - (ps - 1), 0), c1 = min(c + (ps - 1), res - 1);
    if (r0 > r1 || c0 > c1) return false;
    tr0 = r0 >> kPaintTileShift; tr1 = r1 >> kPaintTileShift;
    tc0 = c0 >> kPaintTileShift; tc1 = c1 >> kPaintTileShift;
    return true;
}

// MODE 0: H[block][tile] = points of the block's chunk whose stamp touches the tile.  MODE 1 (H now holds, per tile, the exclusive
// prefix over the blocks -- paint_colscan_kernel -- and total[] the tiles' sums): the points' entries (index, r + 64 << 16 | c + 64)
// go to entries[offset[tile] + H[block][tile] + rank within the block].  No global atomics: a block-level LDS histogram, plain
// stores (512 blocks x 256 tiles of atomics on 256 addresses cost as much as the old kernel's elections).  grid: blocks of `chunk` points.
template <int MODE>
__global__ __launch_bounds__(kPBlock) void paint_bin_kernel(int res, int n, int chunk, const int *__restrict__ pix, int ps, int tiles_x, int tiles,
                                                            int *__restrict__ H, const int *__restrict__ total, uint2 *__restrict__ entries)
{
    __shared__ int hist[kPaintMaxTiles];
    __shared__ int base[MODE ? kPaintMaxTiles : 1];
    __shared__ int wsum[kPBlock / kWave];
    for (int t = threadIdx.x; t < tiles; t += kPBlock) hist[t] = 0;
    int *Hb = H + (size_t)blockIdx.x * tiles;
    if (MODE == 1) {
        // offset[t] = exclusive prefix of total[]: consecutive tiles per thread, a wave scan, the waves' sums
        const int per = (tiles + kPBlock - 1) / kPBlock;
        const int t0 = threadIdx.x * per;
        int tot = 0;
        for (int q = 0; q < per; q++) if (t0 + q < tiles) tot += total[t0 + q];
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
        int incl = tot;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const int x = __shfl_up(incl, o);
            if (lane >= o) incl += x;
        }
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        int run = incl - tot;
        for (int w = 0; w < wave; w++) run += wsum[w];
        for (int q = 0; q < per; q++)
            if (t0 + q < tiles) { base[t0 + q] = run + Hb[t0 + q]; run += total[t0 + q]; }
    }
    __syncthreads();
    const int j0 = blockIdx.x * chunk, j1 = min(n, j0 + chunk);
    for (int j = j0 + threadIdx.x; j < j1; j += kPBlock) {
        const int r = pix[(size_t)j * 2 + 0], c = pix[(size_t)j * 2 + 1];
        int tr0, tr1, tc0, tc1;
        if (!paint_tiles_of(r, c, ps, res, tr0, tr1, tc0, tc1)) continue;
        // (a stamp that touches the image has its centre within ps - 1 <= 63 pixels of it: both coordinates + 64 fit 16 bits)
        const uint2 ent = make_uint2((unsigned)j, ((unsigned)(r + 64) << 16) | (unsigned)(c + 64));
        for (int tr = tr0; tr <= tr1; tr++)
            for (int tc = tc0; tc <= tc1; tc++) {
                const int t = tr * tiles_x + tc;
                const int k = atomicAdd(&hist[t], 1);
                if (MODE == 1) entries[base[t] + k] = ent;
            }
    }
    if (MODE == 1) return;
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += kPBlock) Hb[t] = hist[t];
}

// one wave per tile: H[block][tile] -> its exclusive prefix over the blocks; total[tile] = the sum
__global__ __launch_bounds__(kWave) void paint_colscan_kernel(int nb, int tiles, int *__restrict__ H, int *__restrict__ total)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    int carry = 0;
    for (int g = 0; g < nb; g += kWave) {
        const int b = g + lane;
        const int c = b < nb ? H[(size_t)b * tiles + t] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const int x = __shfl_up(incl, o);
            if (lane >= o) incl += x;
        }
        if (b < nb) H[(size_t)b * tiles + t] = carry + incl - c;
        carry += __shfl(incl, kWave - 1);
    }
    if (lane == 0) total[t] = carry;
}

// one block per tile: the owners of its 64 x 64 pixels elected in LDS, then owner / img / out written for the tile
__global__ __launch_bounds__(kPaintBlock) void paint_tile_kernel(int res, int ps, int tiles_x, const int *__restrict__ total,
                                                                 const uint2 *__restrict__ entries, const float *__restrict__ colors, int ch,
                                                                 float *__restrict__ img, float *__restrict__ out, int *__restrict__ owner)
{
    __shared__ int own[kPaintTile * kPaintTile];
    __shared__ int s_part[kPaintBlock / kWave];
    const int t = blockIdx.x, tr = t / tiles_x, tc = t % tiles_x;
    const int R0 = tr << kPaintTileShift, C0 = tc << kPaintTileShift;
    for (int p = threadIdx.x; p < kPaintTile * kPaintTile; p += kPaintBlock) own[p] = -1;
    // where the tile's entries start: the sum of the totals of the tiles in front of it
    int before = 0;
    for (int q = threadIdx.x; q < t; q += kPaintBlock) before += total[q];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o);
    if ((threadIdx.x & (kWave - 1)) == 0) s_part[threadIdx.x >> 6] = before;
    __syncthreads();
    int e0 = 0;
    for (int w = 0; w < kPaintBlock / kWave; w++) e0 += s_part[w];
    const int e1 = e0 + total[t];
    for (int e = e0 + threadIdx.x; e < e1; e += kPaintBlock) {
        const uint2 ent = entries[e];
        const int j = (int)ent.x, r = (int)(ent.y >> 16) - 64, c = (int)(ent.y & 0xffffu) - 64;
        const int r0 = max(max(r - (ps - 1), 0), R0), r1 = min(min(r + (ps - 1), res - 1), R0 + kPaintTile - 1);
        const int c0 = max(max(c - (ps - 1), 0), C0), c1 = min(min(c + (ps - 1), res - 1), C0 + kPaintTile - 1);
        for (int rr = r0; rr <= r1; rr++)
            for (int cc = c0; cc <= c1; cc++) atomicMax(&own[((rr - R0) << kPaintTileShift) | (cc - C0)], j);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < kPaintTile * kPaintTile; p += kPaintBlock) {
        const int r = R0 + (p >> kPaintTileShift), c = C0 + (p & (kPaintTile - 1));
        if (r >= res || c >= res) continue;
        const int o = own[p];
        owner[(size_t)r * res + c] = o;
        for (int k = 0; k < ch; k++) {
            float v = img[((size_t)k * res + r) * res + c];
            if (o >= 0) {
                v = colors[(size_t)o * ch + k];
                img[((size_t)k * res + r) * res + c] = v;
            }
            out[((size_t)k * res + (res - 1 - r)) * res + c] = v;
        }
    }
}

// ScaleAdapter.py:57-66
__global__ __launch_bounds__(kPBlock) void gather_colors_kernel(int n, const int *__restrict__ pix,
                                                                const float *__restrict__ img, int ch, int h, int w,
                                                                float *__restrict__ out)
{
    const int j = blockIdx.x * kPBlock + threadIdx.x;
    if (j >= n) return;
    const int r = pix[(size_t)j * 2 + 0], c = pix[(size_t)j * 2 + 1];
    for (int k = 0; k < ch; k++) out[(size_t)j * ch + k] = img[((size_t)k * h + (h - 1 - r)) * w + c];
}

// the image's three planes interleaved: packed[p] = (r, g, b, 0)
__global__ __launch_bounds__(kPBlock) void pack_rgba_kernel(int P, const float *__restrict__ img, float4 *__restrict__ packed)
{
    for (int p = blockIdx.x * kPBlock + threadIdx.x; p < P; p += gridDim.x * kPBlock)
        packed[p] = make_float4(img[p], img[(size_t)P + p], img[2 * (size_t)P + p], 0.0f);
}

// gather_colors_kernel on the interleaved copy: one 16-byte read per point
__global__ __launch_bounds__(kPBlock) void gather_colors_packed_kernel(int n, const int *__restrict__ pix, const float4 *__restrict__ packed,
                                                                       int h, int w, float *__restrict__ out)
{
    const int j = blockIdx.x * kPBlock + threadIdx.x;
    if (j >= n) return;
    const int2 rc = *(const int2 *)(pix + (size_t)j * 2);
    const float4 v = packed[(size_t)(h - 1 - rc.x) * w + rc.y];
    out[(size_t)j * 3 + 0] = v.x;
    out[(size_t)j * 3 + 1] = v.y;
    out[(size_t)j * 3 + 2] = v.z;
}

// Visibility by z-buffer.  The reference asks open3d for
// Katz' hidden-point-removal operator (spherical flipping + convex hull per view,
// DepthPrompting.py:273-290): CPU, qhull, third-party.  This is the GPU counterpart
// with a DIFFERENT, simpler definition: a point is visible from a camera when no
// other point that lands in the same pixel of a res x res image is nearer by more
// than `tol` (NDC depth); a point occludes a (2*point_size-1)^2 stamp of pixels.
// Pass 1: per-pixel minimum depth (atomicMin on order-preserving keys); pass 2:
// compare at the point's own pixel and count.
__global__ __launch_bounds__(kPBlock) void zbuf_min_kernel(int n, const float *__restrict__ uv,
                                                           const float *__restrict__ depth, int res, int point_size,
                                                           unsigned *__restrict__ zbuf)
{
    const int cam = blockIdx.y;
    for (int j = blockIdx.x * kPBlock + threadIdx.x; j < n; j += gridDim.x * kPBlock) {
        const size_t q = (size_t)cam * n + j;
        long long pu = (long long)__fmul_rn(uv[q * 2 + 0], (float)res);
        long long pv = (long long)__fmul_rn(uv[q * 2 + 1], (float)res);
        pu = pu < 0 ? 0 : (pu > res - 1 ? res - 1 : pu);
        pv = pv < 0 ? 0 : (pv > res - 1 ? res - 1 : pv);
        // every point occludes a (2*point_size-1)^2 stamp, like paintPixels, so that a
        // sparse front surface has no pin-holes
        const unsigned key = f2key(depth[q]);
        for (int dy = -point_size + 1; dy < point_size; dy++)
            for (int dx = -point_size + 1; dx < point_size; dx++) {
                const long long r = pv + dy, cc = pu + dx;
                if (r < 0 || r >= res || cc < 0 || cc >= res) continue;
                atomicMin(&zbuf[((size_t)cam * res + r) * res + cc], key);
            }
    }
}

__global__ __launch_bounds__(kPBlock) void zbuf_test_kernel(int n, const float *__restrict__ uv,
                                                            const float *__restrict__ depth, int res, float tol,
                                                            const unsigned *__restrict__ zbuf,
                                                            unsigned char *__restrict__ visible,
                                                            int *__restrict__ counts)
{
    const int cam = blockIdx.y;
    int local = 0;
    for (int j = blockIdx.x * kPBlock + threadIdx.x; j < n; j += gridDim.x * kPBlock) {
        const size_t q = (size_t)cam * n + j;
        long long pu = (long long)__fmul_rn(uv[q * 2 + 0], (float)res);
        long long pv = (long long)__fmul_rn(uv[q * 2 + 1], (float)res);
        pu = pu < 0 ? 0 : (pu > res - 1 ? res - 1 : pu);
        pv = pv < 0 ? 0 : (pv > res - 1 ? res - 1 : pv);
        const float zmin = key2f(zbuf[((size_t)cam * res + pv) * res + pu]);
        const bool vis = depth[q] <= __fadd_rn(zmin, tol);
        visible[q] = vis ? 1 : 0;
        local += vis ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && local) atomicAdd(&counts[cam], local);
}

static int grid_for(long long n, int cap)
{
    long long g = ceil_div64(n, kPBlock);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace genpc

GENPC_API int genpc_get_uvs(int c, int n, const float *view, float focal, float znear, float zfar, const float *xyz,
                            float *transformed, float *uv, float *depth, int rescale, float padmul, float *bbox,
                            void *stream)
{
    using namespace genpc;
    if (c <= 0 || n <= 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    // scratch: per-(camera, slice) approximate boxes, the cameras' final keys
    const size_t part_bytes = (size_t)c * kBoxSlices * 4 * sizeof(float);
    char *ws = (char *)workspace(2, part_bytes + (size_t)c * 4 * sizeof(unsigned), st);
    if (!ws) return 0;
    float *part = (float *)ws;
    unsigned *keys = (unsigned *)(ws + part_bytes);
    const float A = (zfar + znear) / (znear - zfar);
    const float B = (2.0f * zfar * znear) / (znear - zfar);
    const int have_box = (rescale || bbox) ? 1 : 0;
    // Pass 1 is arithmetic only (see above), pass 2 is bound by its stores and five divisions per pair.
    // (Running pass 1 of one camera group on a side stream under pass 2 of the previous one was measured in
    // round 2: 484 us against 365 back to back -- the cross-stream events cost more than the overlap returns.)
    if (have_box) {
        hipLaunchKernelGGL(project_bbox_approx_kernel, dim3(kBoxBlocks, ceil_div(c, kWave)), dim3(kPBlock), 0, st, c, n, view, focal,
                           xyz, part);
        hipLaunchKernelGGL(project_box_exact_kernel, dim3(c), dim3(kPBlock), 0, st, n, view, focal, xyz, (const float *)part, keys);
    }
    const int cgroups = ceil_div(c, kCamGroup);
    const long long wblocks = (long long)ceil_div(n, kPBlock * kWritePer) * cgroups;
    if (wblocks > 0x7fffffffLL) { set_error("get_uvs: cameras x points exceed one launch"); return 0; }
    hipLaunchKernelGGL(project_write_kernel, dim3((unsigned)wblocks), dim3(kPBlock), 0, st, c, n, cgroups, view,
                       have_box ? (const unsigned *)keys : (const unsigned *)nullptr, focal, A, B, xyz, rescale, padmul, transformed, uv,
                       depth, bbox);
    return check(hipGetLastError(), "get_uvs launch") ? 1 : 0;
}

GENPC_API int genpc_uv_to_pixels(int n, const float *uv, float res, int clip_max, int *pix, void *stream)
{
    using namespace genpc;
    if (n <= 0) return 1;
    hipLaunchKernelGGL(uv_to_pixels_kernel, dim3(ceil_div(n, kPBlock)), dim3(kPBlock), 0, (hipStream_t)stream, n, uv,
                       res, clip_max, pix);
    return check(hipGetLastError(), "uv_to_pixels launch") ? 1 : 0;
}

GENPC_API int genpc_paint_pixels(int res, int n, const int *pix, const float *colors, int ch, int point_size,
                                 float *img, float *out, int *owner, void *stream)
{
    using namespace genpc;
    if (res <= 0 || ch <= 0 || point_size < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    // many points: binned by pixel tile, elected in LDS (paint_bin / paint_tile kernels above); few: the two kernels below
    static const int env_bin = tune_env("GENPC_PAINT_BINNED_MIN", 262144, "paintPixels: points from which the owners are elected per 64 x 64-pixel tile in LDS (0 = never)");
    if (env_bin > 0 && n >= env_bin && res <= kPaintTile * 64 && point_size <= 32) {
        const int tiles_x = ceil_div(res, kPaintTile), tiles = tiles_x * tiles_x;
        const size_t per_point = point_size == 1 ? 1 : 4;
        int chunk = ceil_div(ceil_div(n, 512), kPBlock) * kPBlock;        // at most 512 blocks of whole 256-point strides
        if (chunk < kPaintChunk) chunk = kPaintChunk;
        const int gb = ceil_div(n, chunk);
        const size_t h_bytes = ((size_t)gb * tiles * sizeof(int) + 255) / 256 * 256, t_bytes = ((size_t)tiles * sizeof(int) + 255) / 256 * 256;
        char *ws = (char *)workspace(34, h_bytes + t_bytes + (size_t)n * per_point * sizeof(uint2), st);
        if (!ws) return 0;
        int *H = (int *)ws, *total = (int *)(ws + h_bytes);
        uint2 *entries = (uint2 *)(ws + h_bytes + t_bytes);
        hipLaunchKernelGGL(paint_bin_kernel<0>, dim3(gb), dim3(kPBlock), 0, st, res, n, chunk, pix, point_size, tiles_x, tiles, H, (const int *)total, entries);
        hipLaunchKernelGGL(paint_colscan_kernel, dim3(tiles), dim3(kWave), 0, st, gb, tiles, H, total);
        hipLaunchKernelGGL(paint_bin_kernel<1>, dim3(gb), dim3(kPBlock), 0, st, res, n, chunk, pix, point_size, tiles_x, tiles, H, (const int *)total, entries);
        hipLaunchKernelGGL(paint_tile_kernel, dim3(tiles), dim3(kPaintBlock), 0, st, res, point_size, tiles_x, (const int *)total,
                           (const uint2 *)entries, colors, ch, img, out, owner);
        return check(hipGetLastError(), "paint_pixels (binned) launch") ? 1 : 0;
    }
    if (!check(hipMemsetAsync(owner, 0xff, (size_t)res * res * sizeof(int), st), "hipMemsetAsync(owner)")) return 0;
    if (n > 0) {
        const long long side = 2 * point_size - 1;
        hipLaunchKernelGGL(splat_owner_kernel, dim3(grid_for((long long)n * side * side, 4096)), dim3(kPBlock), 0, st,
                           res, n, pix, point_size, owner);
    }
    hipLaunchKernelGGL(splat_write_kernel, dim3(ceil_div(res * res, kPBlock)), dim3(kPBlock), 0, st, res,
                       (const int *)owner, colors, ch, img, out);
    return check(hipGetLastError(), "paint_pixels launch") ? 1 : 0;
}

GENPC_API int genpc_gather_colors(int n, const int *pix, const float *img, int ch, int h, int w, float *out,
                                  void *stream)
{
    using namespace genpc;
    if (n <= 0) return 1;
    // Many points on a three-channel image: one 16-byte gather per point from an interleaved copy of the image instead of three
    // 4-byte gathers from its planes (each a 64-byte line of its own: 9.4 x the algorithmic traffic at 2 M points on 1024^2).
    // The copy costs a pass over the image (28 B per pixel), so only where the points outnumber a quarter of the pixels.
    static const int env_pack = tune_env("GENPC_GATHER_PACK", 1, "colour gather: 1 = many points read one 16-byte word per point from an interleaved copy of the image, 0 = always three planar gathers");
    if (env_pack && ch == 3 && (long long)n * 4 >= (long long)h * w && (long long)h * w <= 0x7fffffffLL) {
        hipStream_t st = (hipStream_t)stream;
        float4 *packed = (float4 *)workspace(33, (size_t)h * w * sizeof(float4), st);
        if (!packed) return 0;
        hipLaunchKernelGGL(pack_rgba_kernel, dim3(grid_for((long long)h * w, 4096)), dim3(kPBlock), 0, st, h * w, img, packed);
        hipLaunchKernelGGL(gather_colors_packed_kernel, dim3(ceil_div(n, kPBlock)), dim3(kPBlock), 0, st, n, pix, (const float4 *)packed, h, w, out);
        return check(hipGetLastError(), "gather_colors (packed) launch") ? 1 : 0;
    }
    hipLaunchKernelGGL(gather_colors_kernel, dim3(ceil_div(n, kPBlock)), dim3(kPBlock), 0, (hipStream_t)stream, n, pix,
                       img, ch, h, w, out);
    return check(hipGetLastError(), "gather_colors launch") ? 1 : 0;
}

GENPC_API int genpc_zbuffer_visibility(int c, int n, const float *uv, const float *depth, int res, int point_size,
                                       float tol, unsigned char *visible, int *counts, void *stream)
{
    using namespace genpc;
    if (c <= 0 || n <= 0) return 1;
    if (res <= 0 || point_size < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    // cameras are processed in groups that keep the z-buffer scratch at <= 64 MiB
    int group = (int)(((size_t)64 << 20) / ((size_t)res * res * sizeof(unsigned)));
    if (group < 1) group = 1;
    if (group > c) group = c;
    unsigned *zbuf = (unsigned *)workspace(8, (size_t)group * res * res * sizeof(unsigned), st);
    if (!zbuf) return 0;
    if (!check(hipMemsetAsync(counts, 0, (size_t)c * sizeof(int), st), "hipMemsetAsync(counts)")) return 0;
    for (int c0 = 0; c0 < c; c0 += group) {
        const int cc = c - c0 < group ? c - c0 : group;
        if (!check(hipMemsetAsync(zbuf, 0xff, (size_t)cc * res * res * sizeof(unsigned), st), "hipMemsetAsync(zbuf)"))
            return 0;
        const int gx = grid_for(n, cc >= 64 ? 16 : 2048 / cc);
        const float *u = uv + (size_t)c0 * n * 2, *d = depth + (size_t)c0 * n;
        hipLaunchKernelGGL(zbuf_min_kernel, dim3(gx, cc), dim3(kPBlock), 0, st, n, u, d, res, point_size, zbuf);
        hipLaunchKernelGGL(zbuf_test_kernel, dim3(gx, cc), dim3(kPBlock), 0, st, n, u, d, res, tol,
                           (const unsigned *)zbuf, visible + (size_t)c0 * n, counts + c0);
    }
    return check(hipGetLastError(), "zbuffer_visibility launch") ? 1 : 0;
}
