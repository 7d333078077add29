// nn_sort.hip -- "sorted mode" of the nearest-neighbour filter (VERDICT r1 item 4 ii): cull whole target slices
// of the MFMA filter instead of evaluating every pair.
//
// Both clouds are put in 3-D Morton order (per batch element: key = batch << 27 | 27-bit code in the box of the
// two clouds; one hipcub sort per cloud).  Then, per block of 512 sorted queries, an UPPER bound D on the
// reference's nearest-neighbour distance of every query in it: each query looks its own key up in the targets'
// sorted keys and takes the exact distance (the reference's arithmetic) to the 16 targets around that place; D
// is the block's maximum.  A target slice (the planner's slice_len consecutive sorted targets: a compact piece
// of the cloud) whose bounding box is farther from the block's bounding box than D cannot hold a nearest
// neighbour of any of the block's queries -- strictly farther than a target that exists -- and the filter
// block for that (query block, slice) pair is never launched.  The finish kernel treats its lists as absent
// (nn_bf16.hip), keys its minimum on (distance, ORIGINAL index) and writes to the original query position; a
// query that needs the exhaustive pass takes it on the caller's arrays.  Everything a block does is unchanged, so
// when nothing can be skipped the cost is today's plus the sort.
//
// Margins: the box distance is rounded down by 1e-5 relative and compared with D (1 + 1e-5); both are fp32
// quantities with errors of a few 2^-24.  NaN anywhere makes a comparison false = "needed".
#include "nn.h"
#include "../../include/genpc_hip.h"

#include <hipcub/hipcub.hpp>

namespace genpc {

constexpr int kSortBox = 512;          // points per bounding box (= queries per filter block)

__device__ __forceinline__ unsigned srt_ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float srt_unord(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// bounds[batch][0..2] = min, [3..5] = max over the finite coordinates of both clouds (ordered uints)
__global__ __launch_bounds__(256) void srt_bounds_kernel(int n0, const float *__restrict__ c0, int n1, const float *__restrict__ c1,
                                                        unsigned *bounds)
{
    const int batch = blockIdx.y;
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n0 + n1; i += gridDim.x * 256) {
        const float *p = i < n0 ? c0 + ((size_t)batch * n0 + i) * 3 : c1 + ((size_t)batch * n1 + (i - n0)) * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = p[k];
            if (!(fabsf(v) < __builtin_inff())) continue;
            const unsigned o = srt_ord(v);
            mn[k] = o < mn[k] ? o : mn[k];
            mx[k] = o > mx[k] ? o : mx[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned a = (unsigned)__shfl_xor((int)mn[k], off, kWave), b = (unsigned)__shfl_xor((int)mx[k], off, kWave);
            mn[k] = a < mn[k] ? a : mn[k];
            mx[k] = b > mx[k] ? b : mx[k];
        }
        if ((threadIdx.x & (kWave - 1)) == 0) {
            atomicMin(&bounds[batch * 6 + k], mn[k]);
            atomicMax(&bounds[batch * 6 + 3 + k], mx[k]);
        }
    }
}

__global__ void srt_init_bounds_kernel(unsigned *bounds) { bounds[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u; }

__device__ __forceinline__ unsigned srt_spread9(unsigned v)      // 9 bits -> every third bit
{
    v &= 0x1ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ unsigned srt_key(const float *p, const unsigned *bounds, int batch)
{
    unsigned key = 0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float v = p[k];
        const float lo = srt_unord(bounds[batch * 6 + k]), hi = srt_unord(bounds[batch * 6 + 3 + k]);
        if (!(fabsf(v) < __builtin_inff())) { ok = false; continue; }
        const float w = hi - lo;
        float q = w > 0.0f ? (v - lo) / w * 511.0f : 0.0f;
        q = q < 0.0f ? 0.0f : (q > 511.0f ? 511.0f : q);
        key |= srt_spread9((unsigned)q) << k;
    }
    return ((unsigned)batch << 27) | (ok ? key : 0x7ffffffu);
}

__global__ __launch_bounds__(256) void srt_key_kernel(int n, const float *__restrict__ pts, const unsigned *__restrict__ bounds,
                                                     unsigned *__restrict__ keys, int *__restrict__ idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x, batch = blockIdx.y;
    if (i >= n) return;
    keys[(size_t)batch * n + i] = srt_key(pts + ((size_t)batch * n + i) * 3, bounds, batch);
    idx[(size_t)batch * n + i] = i;
}

// sorted copy + the bounding box of every kSortBox sorted points
__global__ __launch_bounds__(256) void srt_gather_kernel(int n, const float *__restrict__ pts, const int *__restrict__ perm,
                                                        float *__restrict__ out, float *__restrict__ box)
{
    __shared__ float s_mn[3][4], s_mx[3][4];
    const int batch = blockIdx.y, blk = blockIdx.x;
    const int nbox = (n + kSortBox - 1) / kSortBox;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    bool bad = false;
    for (int r = 0; r < kSortBox / 256; r++) {
        const int pos = blk * kSortBox + r * 256 + threadIdx.x;
        if (pos < n) {
            const int i = perm[(size_t)batch * n + pos];
            const float *p = pts + ((size_t)batch * n + i) * 3;
            float *o = out + ((size_t)batch * n + pos) * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float v = p[k];
                o[k] = v;
                bad |= !(v == v);
                mn[k] = fminf(mn[k], v);
                mx[k] = fmaxf(mx[k], v);
            }
        }
    }
    // a NaN coordinate anywhere in the box makes the box NaN: comparisons with it are false = never skipped
    const unsigned long long anybad = __ballot(bad);
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, kWave));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, kWave));
        }
        if ((threadIdx.x & (kWave - 1)) == 0) {
            s_mn[k][threadIdx.x >> 6] = anybad ? __builtin_nanf("") : mn[k];
            s_mx[k][threadIdx.x >> 6] = anybad ? __builtin_nanf("") : mx[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = s_mn[k][0], b = s_mx[k][0];
        bool nanv = !(a == a);
        for (int w = 1; w < 4; w++) {
            nanv |= !(s_mn[k][w] == s_mn[k][w]);
            a = fminf(a, s_mn[k][w]);
            b = fmaxf(b, s_mx[k][w]);
        }
        float *bx = box + ((size_t)batch * nbox + blk) * 6;
        bx[k] = nanv ? __builtin_nanf("") : a;
        bx[3 + k] = nanv ? __builtin_nanf("") : b;
    }
}

// D[dir][batch][query block] = max over the block's queries of the exact distance to the best of the 16 targets
// around the query's own place in the targets' key order (float bits; atomicMax on non-negative floats)
template <int FMA>
__global__ __launch_bounds__(256) void srt_ub_kernel(int nq, const float *__restrict__ Qs, const unsigned *__restrict__ qkeys, int nt,
                                                    const float *__restrict__ Ts, const unsigned *__restrict__ tkeys, int qper,
                                                    unsigned *__restrict__ D)
{
    const int batch = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int qblocks = (nq + qper - 1) / qper;
    float ub = 0.0f;
    int qb = 0;
    if (j < nq) {
        qb = j / qper;
        const unsigned key = qkeys[(size_t)batch * nq + j];
        const unsigned *tk = tkeys + (size_t)batch * nt;
        int lo = 0, hi = nt;                      // first target key >= key
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tk[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        int first = lo - 8;
        first = first < 0 ? 0 : (first > nt - 16 ? (nt - 16 < 0 ? 0 : nt - 16) : first);
        const float *q = Qs + ((size_t)batch * nq + j) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];
        float best = __builtin_inff();
        for (int c = 0; c < 16 && first + c < nt; c++) {
            const float *t = Ts + ((size_t)batch * nt + first + c) * 3;
            const float dd = sqdist<FMA>(t[0] - qx, t[1] - qy, t[2] - qz);
            best = dd < best ? dd : best;          // (NaN never wins: +inf stays)
        }
        ub = best;
    }
    // one atomic per wave: the 256 threads of a block lie in one query block (qper is a multiple of 256)
    unsigned bits = __float_as_uint(ub);            // ub >= 0 or +inf: the bit patterns order like the values
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)bits, off, kWave);
        bits = o > bits ? o : bits;
    }
    const int qb0 = (blockIdx.x * 256) / qper;
    (void)qb;
    if ((threadIdx.x & (kWave - 1)) == 0 && blockIdx.x * 256 < nq) atomicMax(&D[(size_t)batch * qblocks + qb0], bits);
}

struct SrtPlanArgs {
    const float *qbox[2];     // [B, nqbox, 6] boxes of kSortBox sorted queries (qper == kSortBox)
    const float *tbox[2];     // [B, ntbox, 6]
    const unsigned *D[2];     // [B, qblocks]
    unsigned *need[2];        // [B, qblocks]
    int qblocks[2], slices[2], block_begin[2], ntbox[2], nqbox[2];
    int ndir, b, boxes_per_slice;
    int *work, *work_count;
};

__global__ void srt_store_kernel(NNSortDev v, NNSortDev *dst) { *dst = v; }

__global__ __launch_bounds__(256) void srt_need_kernel(SrtPlanArgs p)
{
    const int d = blockIdx.z;
    if (d >= p.ndir) return;
    const int batch = blockIdx.y;
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int qblocks = p.qblocks[d], slices = p.slices[d];
    if (id >= qblocks * slices) return;
    const int qb = id % qblocks, slice = id / qblocks;
    const float *qx = p.qbox[d] + ((size_t)batch * p.nqbox[d] + qb) * 6;
    float tlo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, thi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    bool nanbox = false;
    for (int k = slice * p.boxes_per_slice; k < (slice + 1) * p.boxes_per_slice && k < p.ntbox[d]; k++) {
        const float *tb = p.tbox[d] + ((size_t)batch * p.ntbox[d] + k) * 6;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            nanbox |= !(tb[a] == tb[a]) || !(tb[3 + a] == tb[3 + a]);
            tlo[a] = fminf(tlo[a], tb[a]);
            thi[a] = fmaxf(thi[a], tb[3 + a]);
        }
    }
    float lb2 = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float g1 = qx[a] - thi[a], g2 = tlo[a] - qx[3 + a];
        float g = g1 > g2 ? g1 : g2;
        nanbox |= !(g == g);
        g = g > 0.0f ? g * 0.99999f : 0.0f;
        lb2 += g * g * 0.99999f;
    }
    const float dmax = __uint_as_float(p.D[d][(size_t)batch * qblocks + qb]);
    // qb == 0 always runs: it is the block that publishes the slice's max |t'|^2 for the finish kernel
    const bool skip = qb != 0 && !nanbox && lb2 > dmax * 1.00001f;
    if (!skip) {
        atomicOr(&p.need[d][(size_t)batch * qblocks + qb], 1u << slice);
        // the filter's block numbering: direction base + (slice * B + batch) * qblocks + qb
        p.work[atomicAdd(p.work_count, 1)] = p.block_begin[d] + (slice * p.b + batch) * qblocks + qb;
    }
}

int nn_sort_prepare(int b, const float *c0, int n0, const float *c1, int n1, hipStream_t st, NNSorted &out)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int ns[2] = {n0, n1};
    const float *cs[2] = {c0, c1};
    size_t sort_bytes[2] = {0, 0};
    for (int k = 0; k < 2; k++)
        if (!check(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes[k], (const unsigned *)nullptr, (unsigned *)nullptr,
                                                      (const int *)nullptr, (int *)nullptr, b * ns[k], 0, 32, st),
                   "nn sort size"))
            return 0;
    size_t off = 1024;
    size_t o_s[2], o_k0[2], o_k1[2], o_i0[2], o_i1[2], o_box[2], o_tmp;
    for (int k = 0; k < 2; k++) {
        const size_t tot = (size_t)b * ns[k];
        o_s[k] = off; off += up(tot * 12);
        o_k0[k] = off; off += up(tot * 4);
        o_k1[k] = off; off += up(tot * 4);
        o_i0[k] = off; off += up(tot * 4);
        o_i1[k] = off; off += up(tot * 4);
        o_box[k] = off; off += up((size_t)b * ceil_div(ns[k], kSortBox) * 24);
    }
    o_tmp = off; off += up(sort_bytes[0] > sort_bytes[1] ? sort_bytes[0] : sort_bytes[1]);
    char *ws = (char *)workspace(19, off, st);
    if (!ws) return 0;
    unsigned *bnd = (unsigned *)ws;              // [b][6] ordered-uint bounds, b <= 32: the first 768 bytes
    hipLaunchKernelGGL(srt_init_bounds_kernel, dim3(1), dim3(192), 0, st, bnd);
    const int gb = ceil_div(n0 + n1, 256);
    hipLaunchKernelGGL(srt_bounds_kernel, dim3(gb < 8 ? gb : 8, b), dim3(256), 0, st, n0, c0, n1, c1, bnd);
    for (int k = 0; k < 2; k++) {
        unsigned *k0 = (unsigned *)(ws + o_k0[k]), *k1 = (unsigned *)(ws + o_k1[k]);
        int *i0 = (int *)(ws + o_i0[k]), *i1 = (int *)(ws + o_i1[k]);
        hipLaunchKernelGGL(srt_key_kernel, dim3(ceil_div(ns[k], 256), b), dim3(256), 0, st, ns[k], cs[k], (const unsigned *)bnd, k0, i0);
        size_t sb = sort_bytes[k];
        if (!check(hipcub::DeviceRadixSort::SortPairs(ws + o_tmp, sb, (const unsigned *)k0, k1, (const int *)i0, i1, b * ns[k], 0, 32, st),
                   "nn radix sort"))
            return 0;
        hipLaunchKernelGGL(srt_gather_kernel, dim3(ceil_div(ns[k], kSortBox), b), dim3(256), 0, st, ns[k], cs[k], (const int *)i1,
                           (float *)(ws + o_s[k]), (float *)(ws + o_box[k]));
        out.s[k] = (const float *)(ws + o_s[k]);
        out.perm[k] = (const int *)i1;
        out.keys[k] = (const unsigned *)k1;
        out.box[k] = (const float *)(ws + o_box[k]);
        out.n[k] = ns[k];
    }
    out.b = b;
    return check(hipGetLastError(), "nn sort launch") ? 1 : 0;
}

int nn_sort_plan(NNArgs &a, const NNSorted &srt, const int cloud_of_q[2], int qper, hipStream_t st)
{
    if (qper != kSortBox) {
        set_error("nn sort: the sorted mode needs 512-query blocks");
        return 0;
    }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 512;
    size_t o_D[2] = {0, 0}, o_need[2] = {0, 0};
    long long total_blocks = 0;
    for (int d = 0; d < a.ndir; d++) {
        o_D[d] = off; off += up((size_t)a.b * a.dir[d].qblocks * 4);
        o_need[d] = off; off += up((size_t)a.b * a.dir[d].qblocks * 4);
        total_blocks += (long long)a.b * a.dir[d].qblocks * a.dir[d].slices;
        if (a.dir[d].slices > 32) {
            set_error("nn sort: more than 32 slices");
            return 0;
        }
    }
    const size_t o_work = off; off += up((size_t)total_blocks * 4);
    char *ws = (char *)workspace(21, off, st);
    if (!ws) return 0;
    // zero: the count, the D maxima (0.0f) and the need masks
    if (!check(hipMemsetAsync(ws, 0, o_work, st), "hipMemsetAsync(nn sort plan)")) return 0;
    SrtPlanArgs p{};
    NNSortDev dev{};
    p.ndir = a.ndir;
    p.b = a.b;
    p.boxes_per_slice = a.slice_len / kSortBox;
    p.work = (int *)(ws + o_work);
    p.work_count = (int *)ws;
    int maxid = 0;
    for (int d = 0; d < a.ndir; d++) {
        const int cq = cloud_of_q[d], ct = 1 - cq;
        NNDir &D = a.dir[d];
        unsigned *Dd = (unsigned *)(ws + o_D[d]);
        if (a.fma)
            hipLaunchKernelGGL((srt_ub_kernel<1>), dim3(ceil_div(D.nq, 256), a.b), dim3(256), 0, st, D.nq, srt.s[cq], srt.keys[cq], D.nt,
                               srt.s[ct], srt.keys[ct], qper, Dd);
        else
            hipLaunchKernelGGL((srt_ub_kernel<0>), dim3(ceil_div(D.nq, 256), a.b), dim3(256), 0, st, D.nq, srt.s[cq], srt.keys[cq], D.nt,
                               srt.s[ct], srt.keys[ct], qper, Dd);
        p.qbox[d] = srt.box[cq];
        p.tbox[d] = srt.box[ct];
        p.D[d] = Dd;
        p.need[d] = (unsigned *)(ws + o_need[d]);
        p.qblocks[d] = D.qblocks;
        p.slices[d] = D.slices;
        p.block_begin[d] = D.block_begin;
        p.nqbox[d] = ceil_div(D.nq, kSortBox);
        p.ntbox[d] = ceil_div(D.nt, kSortBox);
        dev.need[d] = p.need[d];
        dev.perm_q[d] = srt.perm[cq];
        dev.perm_t[d] = srt.perm[ct];
        dev.q_orig[d] = srt.orig[cq];
        dev.t_orig[d] = srt.orig[ct];
        if (D.qblocks * D.slices > maxid) maxid = D.qblocks * D.slices;
    }
    dev.work = p.work;
    dev.work_count = p.work_count;
    NNSortDev *dev_mem = (NNSortDev *)(ws + 256);          // (after the count word; zeroed area ends at o_work)
    hipLaunchKernelGGL(srt_store_kernel, dim3(1), dim3(1), 0, st, dev, dev_mem);
    hipLaunchKernelGGL(srt_need_kernel, dim3(ceil_div(maxid, 256), a.b, a.ndir), dim3(256), 0, st, p);
    a.srt = dev_mem;
    return check(hipGetLastError(), "nn sort plan launch") ? 1 : 0;
}

}  // namespace genpc
