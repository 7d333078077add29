// emd.h -- device helpers shared by the auction kernels (emd.hip: tiled bid, settle / resolve; emd_grid.hip: the
// cell-sorted culled bid).
#pragma once
#include "common.h"

namespace genpc {

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kEBlock = 256;
// objects per LDS tile: template parameter TILE of the bid kernel, 2048 (32 KiB as float4) or 1024 (16 KiB)
constexpr int kZMax = 4;            // object slices per bidder group in the late-round split (measured best of 1..16)
constexpr int kSplitMaxBidders = 4096;   // bidders per batch element the split scratch can hold
constexpr int kArrivePerBatch = 1024;    // arrival counters per batch element (>= kSplitMaxBidders * 64 / 256)

template <int FMA>
__device__ __forceinline__ float sqdist_e(float dx, float dy, float dz)
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

// emd_cuda.cu:142-146
template <int FMA>
__device__ __forceinline__ float bid_value(float x1, float y1, float z1, float x2, float y2, float z2, float price)
{
    float s = sqdist_e<FMA>(x2 - x1, y2 - y1, z2 - z1);
    float r = sqrtf(s);   // correctly rounded (hipcc default); __fsqrt_rn is the ~1 ulp native sqrt
    return (float)((3.0 - (double)r) - (double)price);
}

// float atomic max; increments are >= 0 in every sane call (eps >= 0), where the
// int ordering of the bit patterns equals the float ordering even against the
// -1e9 reset value.  Negative values take the CAS loop of emd_cuda.cu:10-20.
__device__ __forceinline__ void atomic_max_float(float *addr, float val)
{
    if (val >= 0.0f) {
        atomicMax((int *)addr, __float_as_int(val));
    } else {
        int ret = __float_as_int(*addr);
        while (val > __int_as_float(ret)) {
            int old = ret;
            if ((ret = atomicCAS((int *)addr, old, __float_as_int(val))) == old) break;
        }
    }
}

// Folds (ob, obb, oi, obi) into (b, bb, bi, bbi): best / second-best values with the
// index of an object attaining each.  Value-symmetric; on a tie for first place the
// lower index is kept as `bi` (the reference's order is restored by the tie path).
__device__ __forceinline__ void merge_top2(float &b, float &bb, int &bi, int &bbi, float ob, float obb, int oi, int obi)
{
    float nb2;
    int nbi;
    if (b > ob) {
        nb2 = fmaxf(bb, ob);
        nbi = ob > bb ? oi : bbi;
    } else if (ob > b) {
        nb2 = fmaxf(obb, b);
        nbi = b > obb ? bi : obi;
    } else {                       // equal first places: the other one is the second
        nb2 = b;
        nbi = ((unsigned)oi < (unsigned)bi) ? bi : oi;
    }
    const bool take = ob > b || (ob == b && (unsigned)oi < (unsigned)bi);
    bi = take ? oi : bi;
    b = fmaxf(b, ob);
    bb = nb2;
    bbi = nbi;
}

// Threshold of the bid pre-filter for m = max(better, seed): a candidate with squared distance sq
// and price p >= 0 can only matter if fl32((3 - sqrtf(sq)) - p) > m.  With tt = fl(cb - p), the
// test  sq < fl(tt * tt)  must pass whenever that holds.  Roundings: cb and tt (relative u each, on
// magnitudes <= 3 + |m|), the square (u), the correctly rounded sqrtf (u), the fp32 rounding of the
// value itself (2u |m|); with 0 <= p < 3 + |m| (otherwise tt <= 0 and nothing can matter) the test
// is safe iff the slack added to (3 - m) is at least u (21 + 9 |m|).  2e-6 (1 + |m|) = 33.5 u (1 + |m|).
// (The first version used the constant 2e-6: proven only for clouds in the unit cube, |m| <= 3.)
__device__ __forceinline__ float filter_cb(float m)
{
    return __fadd_rn(__fsub_rn(3.0f, m), __fmul_rn(2e-6f, __fadd_rn(1.0f, fabsf(m))));
}

// lanes-per-bidder for U bidders on a grid of G blocks per batch element
__device__ __forceinline__ int pick_p(int U, int G)
{
    int P = 64;
    while (P > 1 && ((long long)U * P + kEBlock - 1) / kEBlock > G) P >>= 1;
    return P;
}

// ---- cell-sorted culled bid (emd_grid.hip) ----
constexpr int kEGMaxCells = 15360;      // LDS counters of the build kernel (60 KiB)
struct EGridHdr {            // one per batch element, written by emd_grid_build_kernel
    float lo[3];
    float inv, h;           // cells per unit length, cell side
    float slack[3];         // 16u (|lo| + (g + 1) h) per axis; the bidder adds 16u |x1|
    int g[3];               // cells per axis
    int cells;
    int bad;                // a non-finite coordinate or a negative / non-finite initial price: search without culling
};
struct EmdGridBid {
    int n, G, nb, cells_max, force_lpb, xcd_pin, lpb_max;      // lpb_max > 0: at most this many lanes per bidder
    float eps;
    unsigned stamp;
    const float *xyz1, *xyz2, *price;
    const int *list, *cnt, *start, *orig_of;
    int *cnt_next, *bid, *second;
    float *bid_increments, *max_increments;
    const float4 *sorted;
    const EGridHdr *hdr;
    unsigned long long *chain_head, *chain_next;
    int *chain_cnt;                // bidders per object this round (emd_settle_kernel)
    int *feedback;                 // round 3 only (else null): pinned host word that receives cloud 0's bidder count (emd_auction.hip: which path suits the data)
    const float *cell_pmin;        // per cloud, cells_max + 1 floats: a lower bound of the prices in every cell, +inf for an empty one (emd_cell_pmin_kernel); null: rows are not culled by price
    unsigned long long *stats;     // hook (genpc_emd_tune): [0] bidders, [1] rows of their boxes, [2] rows kept, [3] objects tested, [4] exact evaluations, [5] first-place ties, [6] unseeded bidders; else null
};
int launch_emd_grid_build(int b, int n, const float *xyz2, const float *price, EGridHdr *hdr, int *start, float4 *sorted, int *pos_of,
                          int *orig_of, int cells_target, int cells_max, hipStream_t st, float *price_sep = nullptr);
// ---- all rounds in one launch, threads own the points (emd_auction.hip) ----
int launch_emd_auction(int b, int n, const float *xyz1, const float *xyz2, float *dist, int *assignment, float *price, int *assignment_inv,
                       int *bid, float *bid_increments, float *max_increments, int *max_idx, float eps, int iters, int fma, hipStream_t st, bool forced);
int *emd_feedback_slot(int b, int n, bool device);
bool persist_reserve(int wgs, int capacity, hipStream_t st);
void persist_commit(int wgs, hipStream_t st);
int launch_emd_bid_grid(const EmdGridBid &a, int fma, hipStream_t st);
int launch_emd_cell_pmin(int b, int cells_max, const EGridHdr *hdr, const int *start, const float4 *sorted, int n, float *pmin, hipStream_t st);

// ---- seeded nearest neighbours of the alignment loop (nn_seeded.hip) ----
struct SeededGrids {
    const EGridHdr *hdr_static, *hdr_rest;
    const int *start_static, *start_rest;
    const float4 *sorted_static, *sorted_rest;
};
size_t seeded_grids_bytes(int b, int nm, int ns);
int build_seeded_grids(int b, int nm, const float *rest_pts, int ns, const float *static_pts, void *ws, SeededGrids &g, hipStream_t st);
int launch_nn_seeded(int b, int nm, const float *moving_pts, int ns, const float *static_pts, const SeededGrids &g, const float *center,
                     int cstride, const float *params, int pstride, float *d1, int *i1, float *d2, int *i2, int fma, hipStream_t st, int sample = 1);

}  // namespace genpc
