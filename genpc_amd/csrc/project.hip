// project.hip -- point -> image projection, splat and colour gather for gfx950
// (SURVEY.md 8a rows a13-a15): DepthPrompting.getUvs / paintPixels and
// ScaleAdapter.colorPoint of the reference.  All three are O(N) streaming / scatter
// kernels bound by HBM (a13: 12 B read + 12..24 B written per point per camera).
//
// getUvs: the reference materialises cam.transform(points) for all 1024 cameras
// ([1024,N,3] fp32, 0.88 GB at N = 71k) and then consumes two rows
// (DepthPrompting.py:154-165).  Here any subset of cameras is projected in two
// passes over the points: pass 1 only reduces the per-camera bounding box of the
// NDC xy (wave shuffles -> one atomic per wave on order-preserving integer keys),
// pass 2 projects again, rescales and stores uv / depth -- 12 B per (camera, point)
// through HBM instead of 28 with an in-place second pass.  `transformed` is optional.
// Arithmetic (fma order, IEEE division) matches oracle/genpc_oracle_geom.c bit for
// bit; min/max are exact, so uv is bit-exact too.
#include "common.h"
#include "../../include/genpc_hip.h"

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
// keys[C,4]: min_x, min_y as keys, max_x, max_y as ~key, all reduced with atomicMin
// from an all-ones initial state (one memset).
__global__ __launch_bounds__(kPBlock) void project_bbox_kernel(int n, const float *__restrict__ view, float focal,
                                                               float A, float B, const float *__restrict__ xyz,
                                                               unsigned *__restrict__ keys)
{
    const int cam = blockIdx.y;
    const float *V = view + (size_t)cam * 12;
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = V[k];
    unsigned mnx = 0xffffffffu, mny = 0xffffffffu, mxx = 0xffffffffu, mxy = 0xffffffffu;
    for (int j = blockIdx.x * kPBlock + threadIdx.x; j < n; j += gridDim.x * kPBlock) {
        float ox, oy, oz;
        project_point(v, focal, A, B, xyz[(size_t)j * 3 + 0], xyz[(size_t)j * 3 + 1], xyz[(size_t)j * 3 + 2], ox, oy, oz);
        const unsigned kx = f2key(ox), ky = f2key(oy);
        mnx = min(mnx, kx); mny = min(mny, ky);
        mxx = min(mxx, ~kx); mxy = min(mxy, ~ky);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mnx = min(mnx, (unsigned)__shfl_xor((int)mnx, off, kWave));
        mny = min(mny, (unsigned)__shfl_xor((int)mny, off, kWave));
        mxx = min(mxx, (unsigned)__shfl_xor((int)mxx, off, kWave));
        mxy = min(mxy, (unsigned)__shfl_xor((int)mxy, off, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMin(&keys[cam * 4 + 0], mnx);
        atomicMin(&keys[cam * 4 + 1], mny);
        atomicMin(&keys[cam * 4 + 2], mxx);
        atomicMin(&keys[cam * 4 + 3], mxy);
    }
}

// Pass 2: project again (the cloud is a few hundred KB and stays in L2 across the
// cameras; recomputing costs nothing next to a 28 B/point round trip through HBM),
// rescale (DepthPrompting.py:246-266) and store uv, depth and, if asked, the NDC point.
__global__ __launch_bounds__(kPBlock) void project_write_kernel(int n, const float *__restrict__ view, float focal,
                                                                float A, float B, const float *__restrict__ xyz,
                                                                const unsigned *__restrict__ keys, int rescale,
                                                                float padmul, float *__restrict__ transformed,
                                                                float *__restrict__ uv, float *__restrict__ depth,
                                                                float *__restrict__ bbox)
{
    const int cam = blockIdx.y;
    const float *V = view + (size_t)cam * 12;
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = V[k];
    float cx = 0.0f, cy = 0.0f, sc = 1.0f;
    if (rescale || bbox) {
        const float mnx = key2f(keys[cam * 4 + 0]), mny = key2f(keys[cam * 4 + 1]);
        const float mxx = key2f(~keys[cam * 4 + 2]), mxy = key2f(~keys[cam * 4 + 3]);
        if (bbox && blockIdx.x == 0 && threadIdx.x == 0) {
            bbox[cam * 4 + 0] = mnx; bbox[cam * 4 + 1] = mny; bbox[cam * 4 + 2] = mxx; bbox[cam * 4 + 3] = mxy;
        }
        cx = __fadd_rn(mnx, mxx) / 2.0f;
        cy = __fadd_rn(mny, mxy) / 2.0f;
        const float sx = __fsub_rn(mxx, mnx), sy = __fsub_rn(mxy, mny);
        sc = sx > sy ? sx : sy;
    }
    for (int j = blockIdx.x * kPBlock + threadIdx.x; j < n; j += gridDim.x * kPBlock) {
        float ox, oy, oz;
        project_point(v, focal, A, B, xyz[(size_t)j * 3 + 0], xyz[(size_t)j * 3 + 1], xyz[(size_t)j * 3 + 2], ox, oy, oz);
        const size_t q = (size_t)cam * n + j;
        if (transformed) {
            transformed[q * 3 + 0] = ox; transformed[q * 3 + 1] = oy; transformed[q * 3 + 2] = oz;
        }
        float u, vv;
        if (rescale) {
            u = __fadd_rn(__fmul_rn(__fsub_rn(ox, cx) / sc, padmul), 0.5f);
            vv = __fadd_rn(__fmul_rn(__fsub_rn(oy, cy) / sc, padmul), 0.5f);
        } else {
            u = __fmul_rn(__fadd_rn(ox, 1.0f), 0.5f);
            vv = __fmul_rn(__fadd_rn(oy, 1.0f), 0.5f);
        }
        *reinterpret_cast<float2 *>(uv + q * 2) = make_float2(u, vv);
        depth[q] = oz;
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
    unsigned *keys = (unsigned *)workspace(2, (size_t)c * 4 * sizeof(unsigned), st);
    if (!keys) return 0;
    if (!check(hipMemsetAsync(keys, 0xff, (size_t)c * 4 * sizeof(unsigned), st), "hipMemsetAsync(keys)")) return 0;
    const float A = (zfar + znear) / (znear - zfar);
    const float B = (2.0f * zfar * znear) / (znear - zfar);
    // enough blocks per camera to fill the chip when few cameras are projected
    int gx = grid_for(n, c >= 64 ? 16 : 2048 / c);
    if (rescale || bbox)
        hipLaunchKernelGGL(project_bbox_kernel, dim3(gx, c), dim3(kPBlock), 0, st, n, view, focal, A, B, xyz, keys);
    hipLaunchKernelGGL(project_write_kernel, dim3(gx, c), dim3(kPBlock), 0, st, n, view, focal, A, B, xyz,
                       (const unsigned *)keys, rescale, padmul, transformed, uv, depth, bbox);
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
