// voxel.hip -- voxel-grid down-sampling: the counterpart of open3d's
// PointCloud.voxel_down_sample as reg() calls it on both clouds before every ICP / scale search
// (reg_xyz.py:154-155,178-183).  open3d is absent and unpinned; its published definition:
//   voxel_min_bound = min_bound - voxel_size / 2;  index = floor((p - voxel_min_bound) / voxel_size)
//   (double arithmetic on Eigen::Vector3d points);  one output point per occupied voxel = the mean
//   of its points, accumulated in point order; colours, when the cloud has them, are averaged the same way
//   (load_xyz / glb2point down-sample COLOURED clouds: utils/dataUtils.py:174-189,217-250).
// Output order here: ascending (i, j, k) (open3d's is its hash-map iteration order, unspecified).
//
//   voxel_bounds_kernel   min / max of the cloud (order-independent: bitwise reproducible)
//   voxel_key_kernel      64-bit key (i << 42 | j << 21 | k) per point
//   rocprim radix sort    (key, point index) pairs -- ROCm's own primitives library; it is stable, so the points
//                         of a voxel stay in ascending index order
//   voxel_head_kernel     run heads; rocprim inclusive scan gives every run its output slot
//   voxel_mean_kernel     the head of a run sums it in that order, in double: the same additions in
//                         the same order as a CPU loop over the points -> bit-identical means
#include "common.h"
#include "../../include/genpc_hip.h"

#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace genpc {

constexpr int kVBlock = 256;

__device__ __forceinline__ unsigned f2ord(float f)      // order-preserving map float -> uint
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// bounds[0..2] = min, [3..5] = max as ordered uints (initialised to 0xffffffff / 0)
__global__ __launch_bounds__(kVBlock) void voxel_bounds_kernel(int n, const float *__restrict__ xyz, unsigned *bounds)
{
    __shared__ unsigned red[6][kVBlock / kWave];
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * kVBlock + threadIdx.x; i < n; i += gridDim.x * kVBlock) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const unsigned o = f2ord(xyz[(size_t)i * 3 + k]);
            mn[k] = o < mn[k] ? o : mn[k];
            mx[k] = o > mx[k] ? o : mx[k];
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned a = (unsigned)__shfl_xor((int)mn[k], off, kWave), b = (unsigned)__shfl_xor((int)mx[k], off, kWave);
            mn[k] = a < mn[k] ? a : mn[k];
            mx[k] = b > mx[k] ? b : mx[k];
        }
        if (lane == 0) { red[k][wave] = mn[k]; red[3 + k][wave] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned v = red[threadIdx.x][0];
        for (int w = 1; w < kVBlock / kWave; w++) {
            const unsigned o = red[threadIdx.x][w];
            v = threadIdx.x < 3 ? (o < v ? o : v) : (o > v ? o : v);
        }
        if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], v);
        else atomicMax(&bounds[threadIdx.x], v);
    }
}

__global__ __launch_bounds__(kVBlock) void voxel_key_kernel(int n, const float *__restrict__ xyz, double voxel,
                                                            const unsigned *__restrict__ bounds,
                                                            unsigned long long *__restrict__ keys, int *__restrict__ idx,
                                                            int *__restrict__ err)
{
    const int i = blockIdx.x * kVBlock + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = 0ull;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double origin = (double)ord2f(bounds[k]) - voxel * 0.5;
        const double c = floor(((double)xyz[(size_t)i * 3 + k] - origin) / voxel);
        // 21 bits per axis; a non-finite coordinate or a grid finer than 2^21 cells per axis is an error
        if (!(c >= 0.0 && c < 2097152.0)) { *err = 1; key = ~0ull; break; }
        key = (key << 21) | (unsigned long long)c;
    }
    keys[i] = key;
    idx[i] = i;
}

__global__ __launch_bounds__(kVBlock) void voxel_head_kernel(int n, const unsigned long long *__restrict__ keys,
                                                             int *__restrict__ head)
{
    const int i = blockIdx.x * kVBlock + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// rank[i] = inclusive scan of head: the run starting at a head i is output slot rank[i] - 1
__global__ __launch_bounds__(kVBlock) void voxel_mean_kernel(int n, const float *__restrict__ xyz,
                                                             const unsigned long long *__restrict__ keys,
                                                             const float *__restrict__ attr,
                                                             const int *__restrict__ idx, const int *__restrict__ head,
                                                             const int *__restrict__ rank, float *__restrict__ out,
                                                             float *__restrict__ out_attr, int *__restrict__ out_count)
{
    const int i = blockIdx.x * kVBlock + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) *out_count = rank[i];
    if (!head[i]) return;
    const unsigned long long key = keys[i];
    double s[3] = {0.0, 0.0, 0.0}, c[3] = {0.0, 0.0, 0.0};
    int cnt = 0;
    for (int j = i; j < n && keys[j] == key; j++) {
        const int p = idx[j];
        s[0] += (double)xyz[(size_t)p * 3 + 0];
        s[1] += (double)xyz[(size_t)p * 3 + 1];
        s[2] += (double)xyz[(size_t)p * 3 + 2];
        if (attr) {
            c[0] += (double)attr[(size_t)p * 3 + 0];
            c[1] += (double)attr[(size_t)p * 3 + 1];
            c[2] += (double)attr[(size_t)p * 3 + 2];
        }
        cnt++;
    }
    float *o = out + (size_t)(rank[i] - 1) * 3;
    o[0] = (float)(s[0] / cnt);
    o[1] = (float)(s[1] / cnt);
    o[2] = (float)(s[2] / cnt);
    if (attr && out_attr) {
        float *oc = out_attr + (size_t)(rank[i] - 1) * 3;
        oc[0] = (float)(c[0] / cnt);
        oc[1] = (float)(c[1] / cnt);
        oc[2] = (float)(c[2] / cnt);
    }
}

__global__ void voxel_error_kernel(const int *__restrict__ err, int *__restrict__ out_count)
{
    if (*err) *out_count = -1;
}

}  // namespace genpc

GENPC_API int genpc_voxel_down_sample(int n, const float *xyz, const float *colors, double voxel_size, float *out,
                                      float *out_colors, int *out_count, void *stream)
{
    using namespace genpc;
    if (n < 0 || !(voxel_size > 0.0)) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return check(hipMemsetAsync(out_count, 0, sizeof(int), st), "hipMemsetAsync(voxel count)") ? 1 : 0;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const int *)nullptr, (int *)nullptr, (size_t)n, 0u, 63u, st);
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, (const int *)nullptr, (int *)nullptr, (size_t)n, rocprim::plus<int>(), st);
    const size_t tmp_bytes = up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
    size_t off = 0;
    const size_t o_bounds = off; off += 256;
    const size_t o_k0 = off; off += up((size_t)n * 8);
    const size_t o_k1 = off; off += up((size_t)n * 8);
    const size_t o_i0 = off; off += up((size_t)n * 4);
    const size_t o_i1 = off; off += up((size_t)n * 4);
    const size_t o_head = off; off += up((size_t)n * 4);
    const size_t o_rank = off; off += up((size_t)n * 4);
    const size_t o_tmp = off; off += tmp_bytes;
    char *ws = (char *)workspace(16, off, st);
    if (!ws) return 0;
    unsigned *bounds = (unsigned *)(ws + o_bounds);
    int *err = (int *)(ws + o_bounds + 64);
    unsigned long long *k0 = (unsigned long long *)(ws + o_k0), *k1 = (unsigned long long *)(ws + o_k1);
    int *i0 = (int *)(ws + o_i0), *i1 = (int *)(ws + o_i1), *head = (int *)(ws + o_head), *rank = (int *)(ws + o_rank);
    // bounds: min = 0xffffffff x3, max = 0 x3, err = 0
    if (!check(hipMemsetAsync(bounds, 0xff, 12, st), "hipMemsetAsync(voxel)")) return 0;
    if (!check(hipMemsetAsync(bounds + 3, 0, 256 - 12, st), "hipMemsetAsync(voxel)")) return 0;
    const int grid = ceil_div(n, kVBlock);
    hipLaunchKernelGGL(voxel_bounds_kernel, dim3(grid < 1024 ? grid : 1024), dim3(kVBlock), 0, st, n, xyz, bounds);
    hipLaunchKernelGGL(voxel_key_kernel, dim3(grid), dim3(kVBlock), 0, st, n, xyz, voxel_size, (const unsigned *)bounds, k0,
                       i0, err);
    size_t sb = tmp_bytes;
    if (!check(rocprim::radix_sort_pairs(ws + o_tmp, sb, (const unsigned long long *)k0, k1, (const int *)i0, i1, (size_t)n, 0u, 63u, st),
               "voxel radix sort"))
        return 0;
    hipLaunchKernelGGL(voxel_head_kernel, dim3(grid), dim3(kVBlock), 0, st, n, (const unsigned long long *)k1, head);
    sb = tmp_bytes;
    if (!check(rocprim::inclusive_scan(ws + o_tmp, sb, (const int *)head, rank, (size_t)n, rocprim::plus<int>(), st), "voxel scan")) return 0;
    hipLaunchKernelGGL(voxel_mean_kernel, dim3(grid), dim3(kVBlock), 0, st, n, xyz, (const unsigned long long *)k1, colors,
                       (const int *)i1, (const int *)head, (const int *)rank, out, out_colors, out_count);
    if (!check(hipGetLastError(), "voxel_down_sample launch")) return 0;
    // out_count = -1 when a coordinate was not finite or the grid exceeded 2^21 cells per axis
    hipLaunchKernelGGL(voxel_error_kernel, dim3(1), dim3(1), 0, st, (const int *)err, out_count);
    return check(hipGetLastError(), "voxel_down_sample launch") ? 1 : 0;
}
