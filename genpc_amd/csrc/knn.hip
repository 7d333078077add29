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

// ---------------------------------------------------------------------------------------------------------------------------
// The same quantity through a uniform grid (round 6).  The exhaustive form above costs n^2 distances AND, because the 64 queries
// of a wave are unrelated points, nearly every group of four candidates improves some lane's list and the whole wave walks the
// K-slot insertion: 1.05 ms for the fused cloud's 20000 points, 3 % of a completed scan, chip-wide.  Here the cloud is sorted
// once into <= 8192 cells (one workgroup: bounding box, cell size by bisection on the cell count, LDS histogram, scan, scatter);
// a query -- the queries are walked in CELL order, so a wave's lanes are neighbours -- visits the shells of cells around its own
// in order of Chebyshev distance r = 0, 1, 2 ... (a row of a shell's face is one run of the sorted array) and stops once its
// K-th best squared distance is no larger than what any unvisited cell can hold, (r h - 2 eps)^2 (1 - 1e-5): eps covers the
// rounding of the cell assignment, the factor that of the distance itself.  The K smallest squared distances are the same
// multiset as the exhaustive search's (the same sqdist_k arithmetic), so the ascending list and the mean are the same bits.
struct KnnGrid {
    float lo[3], h, inv_h, eps;
    int c[3];
};
constexpr int kKGThreads = 1024, kKGCells = 8192, kKGAxis = 1024;

__device__ __forceinline__ void knn_cell_coords(const KnnGrid &G, float x, float y, float z, int &ix, int &iy, int &iz)
{
    ix = (int)fminf(fmaxf((x - G.lo[0]) * G.inv_h, 0.0f), (float)(G.c[0] - 1));
    iy = (int)fminf(fmaxf((y - G.lo[1]) * G.inv_h, 0.0f), (float)(G.c[1] - 1));
    iz = (int)fminf(fmaxf((z - G.lo[2]) * G.inv_h, 0.0f), (float)(G.c[2] - 1));
}

__global__ __launch_bounds__(kKGThreads) void knn_grid_build_kernel(int n, const float *__restrict__ xyz, float4 *__restrict__ P,
                                                                    unsigned *__restrict__ cend_out, KnnGrid *__restrict__ Gout, int cells_target)
{
    __shared__ unsigned s_cnt[kKGCells];
    __shared__ float s_bb[6][kKGThreads / kWave];
    __shared__ unsigned s_scan[kKGThreads / kWave];
    __shared__ KnnGrid G;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    {
        float b[6] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        for (int t = tid; t < n; t += kKGThreads)
            for (int a = 0; a < 3; a++) {
                const float v = xyz[(size_t)t * 3 + a];
                b[a] = fminf(b[a], v);
                b[3 + a] = fmaxf(b[3 + a], v);
            }
#pragma unroll
        for (int a = 0; a < 6; a++) {
            float v = b[a];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float o = __shfl_xor(v, off, kWave);
                v = a < 3 ? fminf(v, o) : fmaxf(v, o);
            }
            if (lane == 0) s_bb[a][wave] = v;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float ext[3], emax = 0.0f, mag = 0.0f;
        for (int a = 0; a < 3; a++) {
            float lo = __builtin_inff(), hi = -__builtin_inff();
            for (int w = 0; w < kKGThreads / kWave; w++) {
                lo = fminf(lo, s_bb[a][w]);
                hi = fmaxf(hi, s_bb[3 + a][w]);
            }
            G.lo[a] = lo;
            ext[a] = hi - lo;
            emax = fmaxf(emax, ext[a]);
            mag = fmaxf(mag, fmaxf(fabsf(lo), fabsf(hi)));
        }
        G.c[0] = G.c[1] = G.c[2] = 1;
        G.h = 1.0f;
        G.inv_h = 0.0f;
        G.eps = 0.0f;
        if (emax > 0.0f && emax < 1e30f && mag < 1e30f) {
            // the smallest cell size whose grid has no more than cells_target cells (and no more than kKGAxis along an axis)
            auto cells_of = [&](float h) {
                double p = 1.0;
                for (int a = 0; a < 3; a++) p *= (double)((long long)(ext[a] / h) + 1);
                return p;
            };
            float h_lo = emax / (float)(kKGAxis - 1), h_hi = emax * 1.001f;
            if (cells_of(h_lo) > (double)cells_target) {
                for (int it = 0; it < 40; it++) {
                    const float mid = 0.5f * (h_lo + h_hi);
                    if (cells_of(mid) > (double)cells_target) h_lo = mid; else h_hi = mid;
                }
            } else {
                h_hi = h_lo;
            }
            G.h = h_hi;
            G.inv_h = 1.0f / h_hi;
            for (int a = 0; a < 3; a++) G.c[a] = (int)fminf((float)((long long)(ext[a] * G.inv_h) + 1), (float)kKGAxis);
            while ((long long)G.c[0] * G.c[1] * G.c[2] > kKGCells) {          // (rounding of ext * inv_h against ext / h: never more than a step)
                G.h *= 1.01f;
                G.inv_h = 1.0f / G.h;
                for (int a = 0; a < 3; a++) G.c[a] = (int)fminf((float)((long long)(ext[a] * G.inv_h) + 1), (float)kKGAxis);
            }
            G.eps = 1e-6f * fmaxf(emax, mag);
        }
        *Gout = G;
    }
    __syncthreads();
    const int cells = G.c[0] * G.c[1] * G.c[2];
    for (int q = tid; q < cells; q += kKGThreads) s_cnt[q] = 0;
    __syncthreads();
    for (int t = tid; t < n; t += kKGThreads) {
        int ix, iy, iz;
        knn_cell_coords(G, xyz[(size_t)t * 3 + 0], xyz[(size_t)t * 3 + 1], xyz[(size_t)t * 3 + 2], ix, iy, iz);
        atomicAdd(&s_cnt[(iz * G.c[1] + iy) * G.c[0] + ix], 1u);
    }
    __syncthreads();
    {
        const int per = (cells + kKGThreads - 1) / kKGThreads;
        const int c0 = tid * per, c1 = c0 + per < cells ? c0 + per : cells;
        unsigned sum = 0;
        for (int q = c0; q < c1; q++) sum += s_cnt[q];
        const unsigned incl = (unsigned)wave_scan_incl((int)sum);
        if (lane == kWave - 1) s_scan[wave] = incl;
        __syncthreads();
        unsigned base = incl - sum;
        for (int w = 0; w < wave; w++) base += s_scan[w];
        for (int q = c0; q < c1; q++) {
            const unsigned m = s_cnt[q];
            s_cnt[q] = base;
            base += m;
        }
    }
    __syncthreads();
    for (int t = tid; t < n; t += kKGThreads) {
        const float x = xyz[(size_t)t * 3 + 0], y = xyz[(size_t)t * 3 + 1], z = xyz[(size_t)t * 3 + 2];
        int ix, iy, iz;
        knn_cell_coords(G, x, y, z, ix, iy, iz);
        const unsigned pos = atomicAdd(&s_cnt[(iz * G.c[1] + iy) * G.c[0] + ix], 1u);       // afterwards s_cnt[c] is the END of cell c's run
        P[pos] = make_float4(x, y, z, __int_as_float(t));
    }
    __syncthreads();
    for (int q = tid; q < cells; q += kKGThreads) cend_out[q] = s_cnt[q];
}

constexpr int kKGBlock = 256;
template <int K, int FMA>
__global__ __launch_bounds__(kKGBlock) void knn_grid_kernel(int n, const float4 *__restrict__ P, const unsigned *__restrict__ cend,
                                                            const KnnGrid *__restrict__ Gp, float *__restrict__ mean_out)
{
    const KnnGrid G = *Gp;
    const int q = blockIdx.x * kKGBlock + threadIdx.x;
    if (q >= n) return;
    const float4 me = P[q];
    int cx, cy, cz;
    knn_cell_coords(G, me.x, me.y, me.z, cx, cy, cz);
    float t[K];
#pragma unroll
    for (int i = 0; i < K; i++) t[i] = __builtin_inff();
    const int c0 = G.c[0], c1 = G.c[1], c2 = G.c[2];
    const int rmax = max(max(max(cx, c0 - 1 - cx), max(cy, c1 - 1 - cy)), max(cz, c2 - 1 - cz));
    auto run = [&](int row, int x0, int x1) {          // the points of cells x0 .. x1 (clamped) of one row: one run of P
        x0 = max(x0, 0);
        x1 = min(x1, c0 - 1);
        if (x0 > x1) return;
        const int a = row + x0, z = row + x1;
        const unsigned b = a ? cend[a - 1] : 0u, e = cend[z];
        for (unsigned u = b; u < e; u++) {
            const float4 v = P[u];
            const float d = sqdist_k<FMA>(v.x - me.x, v.y - me.y, v.z - me.z);
            if (d < t[K - 1]) {
#pragma unroll
                for (int i = K - 1; i > 0; i--) t[i] = __builtin_amdgcn_fmed3f(d, t[i - 1], t[i]);
                t[0] = fminf(d, t[0]);
            }
        }
    };
    for (int r = 0;; r++) {
        for (int dz = -r; dz <= r; dz++) {
            const int z = cz + dz;
            if (z < 0 || z >= c2) continue;
            for (int dy = -r; dy <= r; dy++) {
                const int y = cy + dy;
                if (y < 0 || y >= c1) continue;
                const int row = (z * c1 + y) * c0;
                if (dz == -r || dz == r || dy == -r || dy == r) {
                    run(row, cx - r, cx + r);              // a row of the shell's faces
                } else {
                    run(row, cx - r, cx - r);              // the two cells of an inner row
                    run(row, cx + r, cx + r);
                }
            }
        }
        if (r >= rmax) break;                              // the whole grid has been visited
        const float bound = (float)r * G.h - 2.0f * G.eps;
        if (bound > 0.0f && t[K - 1] <= bound * bound * 0.99999f) break;
    }
    double acc = 0.0;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < K; i++) {
        if (t[i] < __builtin_inff()) {           // fewer than K points in the cloud
            acc += sqrt((double)t[i]);
            cnt++;
        }
    }
    mean_out[__float_as_int(me.w)] = (float)(acc / (double)cnt);
}

template <int K>
static bool launch_knn_grid(int n, const float *xyz, float *out, hipStream_t st)
{
    static const int env_grid = tune_env("GENPC_KNN_GRID", 1, "k-NN mean distance: 1 = through a uniform grid (shells of cells around the query), 0 = exhaustive");
    if (!env_grid || n < 256) return false;
    const size_t o_cend = ((size_t)n * sizeof(float4) + 255) / 256 * 256, o_g = o_cend + (size_t)kKGCells * sizeof(unsigned);
    char *ws = (char *)workspace(35, o_g + 256, st);
    if (!ws) return false;
    float4 *P = (float4 *)ws;
    unsigned *cend = (unsigned *)(ws + o_cend);
    KnnGrid *G = (KnnGrid *)(ws + o_g);
    int target = 2 * n;
    target = target < 64 ? 64 : (target > kKGCells ? kKGCells : target);
    hipLaunchKernelGGL(knn_grid_build_kernel, dim3(1), dim3(kKGThreads), 0, st, n, xyz, P, cend, G, target);
    const int blocks = ceil_div(n, kKGBlock);
    if (arith_mode() != 0)
        hipLaunchKernelGGL((knn_grid_kernel<K, 1>), dim3(blocks), dim3(kKGBlock), 0, st, n, (const float4 *)P, (const unsigned *)cend, (const KnnGrid *)G, out);
    else
        hipLaunchKernelGGL((knn_grid_kernel<K, 0>), dim3(blocks), dim3(kKGBlock), 0, st, n, (const float4 *)P, (const unsigned *)cend, (const KnnGrid *)G, out);
    return true;
}

template <int K>
static void launch_knn(int n, const float *xyz, float *out, hipStream_t st)
{
    if (launch_knn_grid<K>(n, xyz, out, st)) return;
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
