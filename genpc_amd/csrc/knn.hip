// knn.hip -- mean distance to the k nearest neighbours within one cloud, the
// quantity behind open3d's remove_statistical_outlier, which closes the fusion tail
// of reg() (reg_xyz.py:217 -> utils/dataUtils.py:648-662).
// open3d is absent and unpinned; its published algorithm is restated: for every
// point the k nearest points of the SAME cloud (the point itself included, at
// distance 0), mean of their Euclidean distances; a point survives when its mean is
// below cloud_mean + std_ratio * cloud_std (sample standard deviation).
//
// One lane = one query.  Targets are staged in LDS like chamfer.hip (groups of four,
// broadcast ds_read_b128, 12 B per target).  The k best squared distances live in
// registers as an ascending list; a candidate is compared against the current k-th
// best and only when some lane of the wave improves does the wave run the insertion,
// which is one v_med3 per list slot:  t'[i] = med3(d, t[i-1], t[i]).
// Equal distances at the k-th place carry equal values, so the mean does not depend
// on which of them is kept: bit-exact with the oracle (sum of square roots in
// ascending order, in double).
#include "common.h"
#include "../../include/genpc_hip.h"

namespace genpc {

constexpr int kKTile = 1024;

template <int FMA>
__device__ __forceinline__ float sqdist_k(float dx, float dy, float dz)
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

template <int K, int FMA>
__global__ __launch_bounds__(kWave) void knn_mean_kernel(int n, const float *__restrict__ xyz,
                                                         float *__restrict__ mean_out)
{
    __shared__ float4 tile[kKTile / 4 * 3];
    float *tile_f = (float *)tile;
    const int j = blockIdx.x * kWave + threadIdx.x;
    const int jj = j < n ? j : n - 1;
    const float qx = xyz[(size_t)jj * 3 + 0], qy = xyz[(size_t)jj * 3 + 1], qz = xyz[(size_t)jj * 3 + 2];
    float t[K];
#pragma unroll
    for (int i = 0; i < K; i++) t[i] = __builtin_inff();
    for (int t0 = 0; t0 < n; t0 += kKTile) {
        const int tn = min(kKTile, n - t0);
        const int tn_pad = (tn + 3) / 4 * 4;
        __syncthreads();
        for (int u = threadIdx.x; u < tn_pad; u += kWave) {
            float x, y, z;
            if (u < tn) {
                const float *tp = xyz + (size_t)(t0 + u) * 3;
                x = tp[0]; y = tp[1]; z = tp[2];
            } else {
                x = y = z = __builtin_inff();        // +inf distance: never inserted
            }
            float *g = tile_f + (u >> 2) * 12 + (u & 3);
            g[0] = x; g[4] = y; g[8] = z;
        }
        __syncthreads();
        for (int g = 0; g < tn_pad / 4; g++) {
            const float4 X = tile[g * 3 + 0], Y = tile[g * 3 + 1], Z = tile[g * 3 + 2];
            float d[4];
            d[0] = sqdist_k<FMA>(X.x - qx, Y.x - qy, Z.x - qz);
            d[1] = sqdist_k<FMA>(X.y - qx, Y.y - qy, Z.y - qz);
            d[2] = sqdist_k<FMA>(X.z - qx, Y.z - qy, Z.z - qz);
            d[3] = sqdist_k<FMA>(X.w - qx, Y.w - qy, Z.w - qz);
            const float dmin = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
            if (__any(dmin < t[K - 1])) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (__any(d[c] < t[K - 1])) {
                        // sorted insertion, the largest falls off; a no-op for lanes with d >= t[K-1]
#pragma unroll
                        for (int i = K - 1; i > 0; i--) t[i] = __builtin_amdgcn_fmed3f(d[c], t[i - 1], t[i]);
                        t[0] = fminf(d[c], t[0]);
                    }
                }
            }
        }
    }
    if (j < n) {
        double acc = 0.0;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            if (t[i] < __builtin_inff()) {           // fewer than K points in the cloud
                acc += sqrt((double)t[i]);
                cnt++;
            }
        }
        mean_out[j] = (float)(acc / (double)cnt);
    }
}

template <int K>
static void launch_knn(int n, const float *xyz, float *out, hipStream_t st)
{
    const int blocks = ceil_div(n, kWave);
    if (arith_mode() != 0)
        hipLaunchKernelGGL((knn_mean_kernel<K, 1>), dim3(blocks), dim3(kWave), 0, st, n, xyz, out);
    else
        hipLaunchKernelGGL((knn_mean_kernel<K, 0>), dim3(blocks), dim3(kWave), 0, st, n, xyz, out);
}

}  // namespace genpc

GENPC_API int genpc_knn_mean_distance(int n, const float *xyz, int k, float *mean_out, void *stream)
{
    using namespace genpc;
    if (n <= 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
    case 8: launch_knn<8>(n, xyz, mean_out, st); break;
    case 16: launch_knn<16>(n, xyz, mean_out, st); break;
    case 20: launch_knn<20>(n, xyz, mean_out, st); break;
    case 32: launch_knn<32>(n, xyz, mean_out, st); break;
    default:
        fprintf(stderr, "genpc_knn_mean_distance: k must be 8, 16, 20 or 32\n");
        return -1;
    }
    return check(hipGetLastError(), "knn_mean launch") ? 1 : 0;
}
