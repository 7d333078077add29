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
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
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

// Pass 1: per-camera bounding box of the NDC xy, nothing stored per point.
// Lane = camera (its view matrix in registers), the block's points come from LDS as broadcast
// reads: no per-point global loads (three strided dword loads per (camera, point) kept the first
// version at the texture-address rate: 152 us for 1024 x 71372), no cross-lane reduction inside
// the loop -- a lane's running extremes are its camera's.  A block covers 64 cameras x one of
// kBoxSplits slices of the cloud and stores its partial box; project_box_reduce_kernel folds the
// slices and derives the rescale constants once per camera.
constexpr int kBoxChunk = 2048;        // points staged per LDS round (24 KiB)
constexpr int kBoxSplits = 64;         // slices of the cloud: 16 camera groups x 64 = 1024 blocks at C = 1024

__global__ __launch_bounds__(kPBlock) void project_bbox_kernel(int c, int n, const float *__restrict__ view, float focal,
                                                               const float *__restrict__ xyz, unsigned *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float pts[kBoxChunk * 3 + 16];      // + a padding group: the last group of four is read whole
    __shared__ unsigned s_box[kPBlock / kWave][4][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int cam = blockIdx.y * kWave + lane;
    const int split = blockIdx.x;
    const float *V = view + (size_t)(cam < c ? cam : c - 1) * 12;
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = V[k];
    const int per = (n + kBoxSplits - 1) / kBoxSplits;
    const int p_begin = split * per, p_end = min(n, p_begin + per);
    unsigned mnx = 0xffffffffu, mny = 0xffffffffu, mxx = 0xffffffffu, mxy = 0xffffffffu;
    for (int p0 = p_begin; p0 < p_end; p0 += kBoxChunk) {
        const int cnt = min(kBoxChunk, p_end - p0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 3; i += kPBlock) pts[i] = xyz[(size_t)p0 * 3 + i];
        __syncthreads();
        // the block's four waves take interleaved groups of four points: three 16-byte broadcast reads
        // deliver them, four independent division chains per lane
        const int groups = (cnt + 3) >> 2;
        for (int g = wave; g < groups; g += kPBlock / kWave) {
            const float4 a0 = *(const float4 *)&pts[g * 12 + 0], a1 = *(const float4 *)&pts[g * 12 + 4],
                         a2 = *(const float4 *)&pts[g * 12 + 8];
            const float c4[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float px = c4[t * 3 + 0], py = c4[t * 3 + 1], pz = c4[t * 3 + 2];
                const float xc = __fadd_rn(__fmaf_rn(v[2], pz, __fmaf_rn(v[1], py, __fmul_rn(v[0], px))), v[3]);
                const float yc = __fadd_rn(__fmaf_rn(v[6], pz, __fmaf_rn(v[5], py, __fmul_rn(v[4], px))), v[7]);
                const float zc = __fadd_rn(__fmaf_rn(v[10], pz, __fmaf_rn(v[9], py, __fmul_rn(v[8], px))), v[11]);
                const float w = -zc;
                const unsigned kx = f2key(__fmul_rn(focal, xc) / w), ky = f2key(__fmul_rn(focal, yc) / w);
                if (g * 4 + t < cnt) {
                    mnx = min(mnx, kx); mny = min(mny, ky);
                    mxx = min(mxx, ~kx); mxy = min(mxy, ~ky);
                }
            }
        }
    }
    s_box[wave][0][lane] = mnx; s_box[wave][1][lane] = mny; s_box[wave][2][lane] = mxx; s_box[wave][3][lane] = mxy;
    __syncthreads();
    if (wave == 0 && cam < c) {
#pragma unroll
        for (int w2 = 1; w2 < kPBlock / kWave; w2++) {
            mnx = min(mnx, s_box[w2][0][lane]); mny = min(mny, s_box[w2][1][lane]);
            mxx = min(mxx, s_box[w2][2][lane]); mxy = min(mxy, s_box[w2][3][lane]);
        }
        unsigned *o = part + ((size_t)cam * kBoxSplits + split) * 4;
        o[0] = mnx; o[1] = mny; o[2] = mxx; o[3] = mxy;
    }
}

// camrec[cam] = view[12] | cx, cy, sc, 0 : everything pass 2 needs per camera, 64 bytes.
// One wave per camera: lane s holds slice s's partial box (kBoxSplits == 64), shuffles fold them.
// (A one-thread-per-camera loop over the slices crashes hipcc 7.2's instruction selection.)
__global__ __launch_bounds__(kWave) void project_box_reduce_kernel(int c, const float *__restrict__ view,
                                                                   const unsigned *__restrict__ part, int have_box,
                                                                   float *__restrict__ camrec, float *__restrict__ bbox)
{
    static_assert(kBoxSplits == kWave, "one lane per slice");
    const int cam = blockIdx.x, lane = threadIdx.x;
    float *o = camrec + (size_t)cam * 16;
    if (lane < 12) o[lane] = view[(size_t)cam * 12 + lane];
    unsigned k0 = 0xffffffffu, k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
    if (have_box) {
        const unsigned *p = part + ((size_t)cam * kBoxSplits + lane) * 4;
        k0 = p[0]; k1 = p[1]; k2 = p[2]; k3 = p[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        k0 = min(k0, (unsigned)__shfl_xor((int)k0, off, kWave));
        k1 = min(k1, (unsigned)__shfl_xor((int)k1, off, kWave));
        k2 = min(k2, (unsigned)__shfl_xor((int)k2, off, kWave));
        k3 = min(k3, (unsigned)__shfl_xor((int)k3, off, kWave));
    }
    if (lane != 0) return;
    float cx = 0.0f, cy = 0.0f, sc = 1.0f;
    if (have_box) {
        const float mnx = key2f(k0), mny = key2f(k1), mxx = key2f(~k2), mxy = key2f(~k3);
        if (bbox) { bbox[cam * 4 + 0] = mnx; bbox[cam * 4 + 1] = mny; bbox[cam * 4 + 2] = mxx; bbox[cam * 4 + 3] = mxy; }
        cx = __fadd_rn(mnx, mxx) / 2.0f;
        cy = __fadd_rn(mny, mxy) / 2.0f;
        const float sx = __fsub_rn(mxx, mnx), sy = __fsub_rn(mxy, mny);
        sc = sx > sy ? sx : sy;
    }
    o[12] = cx; o[13] = cy; o[14] = sc; o[15] = 0.0f;
}

// Pass 2: lane = point (loaded ONCE, kept in registers), loop over the block's 64 cameras whose
// records come from LDS as broadcast reads; stores are coalesced per camera row.  Projects again
// (recomputing costs nothing next to a 28 B/point round trip through HBM), rescales
// (DepthPrompting.py:246-266) and stores uv, depth and, if asked, the NDC point.
constexpr int kCamGroup = 64;
constexpr int kWritePer = 4;        // consecutive points per lane: uv leaves as two 16-byte stores, depth as one
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(kPBlock) void project_write_kernel(int c, int n, const float *__restrict__ camrec, float focal,
                                                                float A, float B, const float *__restrict__ xyz, int rescale,
                                                                float padmul, float *__restrict__ transformed,
                                                                float *__restrict__ uv, float *__restrict__ depth)
{
    __shared__ float4 rec[kCamGroup * 4];
    const int cam0 = blockIdx.y * kCamGroup;
    const int ncam = min(kCamGroup, c - cam0);
    for (int i = threadIdx.x; i < ncam * 4; i += kPBlock) rec[i] = ((const float4 *)camrec)[(size_t)cam0 * 4 + i];
    const int j0 = (blockIdx.x * kPBlock + threadIdx.x) * kWritePer;
    float p[kWritePer][3];
#pragma unroll
    for (int t = 0; t < kWritePer; t++) {
        const int jj = j0 + t < n ? j0 + t : n - 1;
        p[t][0] = xyz[(size_t)jj * 3 + 0]; p[t][1] = xyz[(size_t)jj * 3 + 1]; p[t][2] = xyz[(size_t)jj * 3 + 2];
    }
    const bool full = j0 + kWritePer <= n;
    __syncthreads();
    if (j0 >= n) return;
    for (int k = 0; k < ncam; k++) {
        const float4 r0 = rec[k * 4 + 0], r1 = rec[k * 4 + 1], r2 = rec[k * 4 + 2], r3 = rec[k * 4 + 3];
        const float v[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
        float u[kWritePer], vv[kWritePer], oz[kWritePer], ox[kWritePer], oy[kWritePer];
#pragma unroll
        for (int t = 0; t < kWritePer; t++) {
            project_point(v, focal, A, B, p[t][0], p[t][1], p[t][2], ox[t], oy[t], oz[t]);
            if (rescale) {
                u[t] = __fadd_rn(__fmul_rn(__fsub_rn(ox[t], r3.x) / r3.z, padmul), 0.5f);
                vv[t] = __fadd_rn(__fmul_rn(__fsub_rn(oy[t], r3.y) / r3.z, padmul), 0.5f);
            } else {
                u[t] = __fmul_rn(__fadd_rn(ox[t], 1.0f), 0.5f);
                vv[t] = __fmul_rn(__fadd_rn(oy[t], 1.0f), 0.5f);
            }
        }
        const size_t q = (size_t)(cam0 + k) * n + j0;
        if (full) {
            f4u s0 = {u[0], vv[0], u[1], vv[1]}, s1 = {u[2], vv[2], u[3], vv[3]}, s2 = {oz[0], oz[1], oz[2], oz[3]};
            *(f4u *)(uv + q * 2) = s0;
            *(f4u *)(uv + q * 2 + 4) = s1;
            *(f4u *)(depth + q) = s2;
        } else {
            for (int t = 0; t < kWritePer && j0 + t < n; t++) {
                uv[(q + t) * 2 + 0] = u[t];
                uv[(q + t) * 2 + 1] = vv[t];
                depth[q + t] = oz[t];
            }
        }
        if (transformed) {
            for (int t = 0; t < kWritePer && j0 + t < n; t++) {
                transformed[(q + t) * 3 + 0] = ox[t]; transformed[(q + t) * 3 + 1] = oy[t]; transformed[(q + t) * 3 + 2] = oz[t];
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

// one side stream + events per device, created on first use (never destroyed: process lifetime)
struct SideStream {
    std::mutex enqueue;      // one caller at a time records / waits on the events below (host side only)
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, done[4] = {nullptr, nullptr, nullptr, nullptr};
};

static SideStream *side_stream()
{
    static std::mutex mu;
    static SideStream per_dev[64];
    int dev = 0;
    if (!check(hipGetDevice(&dev), "hipGetDevice") || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> l(mu);
    SideStream &s = per_dev[dev];
    if (!s.stream) {
        if (!check(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
        if (!check(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming), "hipEventCreate")) return nullptr;
        for (int i = 0; i < 4; i++)
            if (!check(hipEventCreateWithFlags(&s.done[i], hipEventDisableTiming), "hipEventCreate")) return nullptr;
    }
    return &s;
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
    // scratch: per-camera records (view | cx, cy, sc) and the per-slice partial boxes
    const size_t rec_bytes = ((size_t)c * 16 * sizeof(float) + 255) & ~(size_t)255;
    char *ws = (char *)workspace(2, rec_bytes + (size_t)c * kBoxSplits * 4 * sizeof(unsigned), st);
    if (!ws) return 0;
    float *camrec = (float *)ws;
    unsigned *part = (unsigned *)(ws + rec_bytes);
    const float A = (zfar + znear) / (znear - zfar);
    const float B = (2.0f * zfar * znear) / (znear - zfar);
    const int have_box = (rescale || bbox) ? 1 : 0;
    // Pass 1 is arithmetic only, pass 2 is bound by its stores (877 MB at 1024 x 71372: ~180 us is
    // what a plain store kernel needs on this chip, tools/ubench_write.hip).  With many cameras
    // they are run as a two-stage pipeline over camera groups: the box of group g + 1 is computed on a
    // side stream while group g is written on the caller's stream (fork / join with events; legal
    // under stream capture).  Few cameras: one group, no side stream.
    // (measured: 484 us pipelined vs 365 us back to back at 1024 x 71372 -- the cross-stream events cost more
    // than the overlap returns; the pipeline stays available for experiments)
    static const bool pipe = getenv("GENPC_UVS_PIPELINE") != nullptr;
    const int groups = (pipe && have_box && c >= 256) ? 4 : 1;
    const int per = ceil_div(ceil_div(c, groups), kCamGroup) * kCamGroup;      // cameras per group, whole write blocks
    SideStream *side = groups > 1 ? side_stream() : nullptr;
    if (groups > 1 && !side) return 0;
    std::unique_lock<std::mutex> guard;
    if (side) guard = std::unique_lock<std::mutex>(side->enqueue);
    if (side) {
        if (!check(hipEventRecord(side->fork, st), "hipEventRecord")) return 0;
        if (!check(hipStreamWaitEvent(side->stream, side->fork, 0), "hipStreamWaitEvent")) return 0;
    }
    for (int g = 0; g < groups; g++) {
        const int c0 = g * per, cg = std::min(per, c - c0);
        if (cg <= 0) break;
        const float *vw = view + (size_t)c0 * 12;
        unsigned *pg = part + (size_t)c0 * kBoxSplits * 4;
        if (have_box) {
            hipStream_t bs = side ? side->stream : st;
            hipLaunchKernelGGL(project_bbox_kernel, dim3(kBoxSplits, ceil_div(cg, kWave)), dim3(kPBlock), 0, bs, cg, n, vw, focal, xyz, pg);
            if (side) {
                if (!check(hipEventRecord(side->done[g], side->stream), "hipEventRecord")) return 0;
                if (!check(hipStreamWaitEvent(st, side->done[g], 0), "hipStreamWaitEvent")) return 0;
            }
        }
        hipLaunchKernelGGL(project_box_reduce_kernel, dim3(cg), dim3(kWave), 0, st, cg, vw, (const unsigned *)pg, have_box,
                           camrec + (size_t)c0 * 16, bbox ? bbox + (size_t)c0 * 4 : nullptr);
        hipLaunchKernelGGL(project_write_kernel, dim3(ceil_div(n, kPBlock * kWritePer), ceil_div(cg, kCamGroup)), dim3(kPBlock), 0, st, cg,
                           n, (const float *)(camrec + (size_t)c0 * 16), focal, A, B, xyz, rescale, padmul,
                           transformed ? transformed + (size_t)c0 * n * 3 : nullptr, uv + (size_t)c0 * n * 2, depth + (size_t)c0 * n);
    }
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
