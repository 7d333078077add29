// nn.h -- declarations shared by the nearest-neighbour kernels (chamfer.hip: VALU and
// fp32-MFMA paths, nn_f16.hip + nn_finish.hip: split-f16 MFMA filter and its exact finish): launch descriptor, the
// reference's distance arithmetic, candidate lists and the exact re-scan.
#pragma once
#include "common.h"

namespace genpc {

constexpr int kChunk = 32;       // targets per min-only chunk (re-scan granularity)
constexpr int kBlock = 256;      // 4 waves
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
    float *out_d;      // final    [B, nq]
    int *out_i;
    unsigned long long *part;   // per-slice partials [S, B*nq] (S > 1): distance bits << 32 | chunk
    int nq, nt;
    // split-f16 path (nn_f16.hip / nn_finish.hip)
    float *tmaxp;           // partial maxima of |t'|^2 [B][ntmax] (per slice, or per split block)
    const uint4 *arec;      // pre-split targets [B][4 planes][ntp] x 16 B
    int ntp, ntmax;         // nt rounded up to 128; entries of tmaxp per batch element
    int fin_begin;          // first nn_finish_kernel block of this direction
    const unsigned *dupmask;   // [B][ceil(nt / 32)]: targets that are later copies of a bit-identical target (nn_dedupe.hip), or null
    int dup_shared;            // the mask is [1][ceil(nt / 32)] and serves every batch element (scaled / replicated copies of one cloud)
    int qblocks;       // ceil(nq / (256*R))
    int slices;        // S for this direction: ceil(nt / slice_len)
    int block_begin;   // first block id of this direction
    int unit_begin;    // first arrival counter of this direction
};

struct NNArgs {
    NNDir dir[2];
    int ndir;
    int b;
    int slice_len;     // targets per slice (all blocks of a launch do equal work), multiple of kChunk
    int *arrive;       // [sum_d b*qblocks] arrival counters, zero between launches
    int debug;         // experiment switches (GENPC_NN_DEBUG): 1 skip index recovery, 2 skip merge, 4 skip main loop
    int fma;           // arithmetic mode of this call (read once at the entry point; host side only)
    unsigned long long *stats;   // hook 512: [0] queries, [1] exhaustive re-dos, [2] exact pieces; else null
    int hint_stride;   // every hint_stride-th finish block reports to `hint` (scaled up)
    unsigned *hint;    // host-visible word: exhaustive re-dos of the filtered path are added to it (dedupe policy, chamfer.hip), or null
    float radius2;     // grid path only: search limit (squared); +inf = none.  Queries with no target within it get (+inf, -1)
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Top3 {
    float a1, a2, a3;   // three smallest unit minima (a3: bound for everything unlisted)
    int c1, c2;         // first target index of the tiles of a1, a2 (-1: unknown)
};

__device__ __forceinline__ void top3_init(Top3 &s)
{
    s.a1 = s.a2 = s.a3 = __builtin_inff();
    s.c1 = s.c2 = -1;
}

__device__ __forceinline__ void top3_insert(Top3 &s, float m, int id)
{
    // c2 = m < a1 ? c1 : (m < a2 ? id : c2);  c1 = m < a1 ? id : c1;
    // a3 = med3(a2, a3, m);  a2 = med3(a1, a2, m);  a1 = min(a1, m)
    // Spelled out: the compiler turns the selects into exec-mask branches plus register
    // shuffling (14 instructions per insert in the filter's unit loop instead of 8).
    asm("v_cmp_lt_f32 vcc, %[m], %[a2]\n\t"
        "v_cndmask_b32 %[c2], %[c2], %[id], vcc\n\t"
        "v_cmp_lt_f32 vcc, %[m], %[a1]\n\t"
        "v_cndmask_b32 %[c2], %[c2], %[c1], vcc\n\t"
        "v_cndmask_b32 %[c1], %[c1], %[id], vcc\n\t"
        "v_med3_f32 %[a3], %[a2], %[a3], %[m]\n\t"
        "v_med3_f32 %[a2], %[a1], %[a2], %[m]\n\t"
        "v_min_f32 %[a1], %[a1], %[m]"
        : [a1] "+v"(s.a1), [a2] "+v"(s.a2), [a3] "+v"(s.a3), [c1] "+v"(s.c1), [c2] "+v"(s.c2)
        : [m] "v"(m), [id] "v"(id)
        : "vcc");
}

// Hand-off of a list from the filter (nn_f16.hip) to the finish step (nn_finish.hip): two 8-byte words per query
//     (a1, c1)   (code2 : code3, c2)
// a1 exact (the acceptance threshold is derived from it); a2 and a3 travel as 16-bit codes of LOWER bounds --
// a2' = fl(a1 + dec(code2)) <= a2, a3' = fl(a2' + dec(code3)) <= a3 -- which is all the finish step needs of them: a
// unit is evaluated exactly when a2' <= tau (a superset), a query is re-done exhaustively when a3' <= tau (a superset).
// code = the top 16 bits (8 exponent, 8 mantissa, truncated) of (v - base) minus four ulps of the larger magnitude:
// the subtraction, the safety term and the decoder's addition round by at most 0.5 + 0.5 ulp(2 max) + 0.5 ulp(2 max)
// < 3.5 ulp(max) together, so the decoded value never exceeds v.  (Round 2: three words with a3 in a word of its own
// -- 24 bytes per list, a third of the list traffic the finish step is bound by.)
__device__ __forceinline__ unsigned list_enc(float base, float v)
{
    if (!(v < __builtin_inff())) return v != v ? 0xffffu : 0xff00u;          // +inf: nothing there; NaN stays NaN
    const float s = fmaxf(fabsf(base), fabsf(v)) * 4.76837158203125e-07f;    // 2^-21
    const float d = (v - base) - s;
    if (!(d > 0.0f)) return 0u;                                              // (also a non-finite base)
    return (__float_as_uint(d) >> 15) & 0xffffu;
}
__device__ __forceinline__ float list_dec(float base, unsigned code) { return base + __uint_as_float(code << 15); }

template <int FMA, int LEN>
__device__ __forceinline__ void rescan_chunk(const float *__restrict__ T, int nt, int base, float qx, float qy,
                                             float qz, float &bd, int &bi)
{
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    bd = __builtin_inff();
    bi = base;
    if (base + LEN <= nt) {
        const f4u *tp = (const f4u *)(T + (size_t)base * 3);
#pragma unroll
        for (int c8 = LEN - 8; c8 >= 0; c8 -= 8) {
            f4u v[6];
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] = tp[(c8 >> 2) * 3 + i];
            const float f[24] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w,
                                 v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w,
                                 v[4].x, v[4].y, v[4].z, v[4].w, v[5].x, v[5].y, v[5].z, v[5].w};
#pragma unroll
            for (int c = 7; c >= 0; c--) {
                const float dd = sqdist<FMA>(f[c * 3 + 0] - qx, f[c * 3 + 1] - qy, f[c * 3 + 2] - qz);
                const bool le = dd <= bd;
                bd = le ? dd : bd;
                bi = le ? base + c8 + c : bi;
            }
        }
    } else {
        for (int c = LEN - 1; c >= 0; c--) {
            int kk = base + c;
            kk = kk < nt ? kk : nt - 1;
            const float *tp = T + (size_t)kk * 3;
            const float dd = sqdist<FMA>(tp[0] - qx, tp[1] - qy, tp[2] - qz);
            const bool le = dd <= bd;
            bd = le ? dd : bd;
            bi = le ? kk : bi;
        }
    }
}

// MFMA row of target w (0..31) of a 32-target tile in the bf16 / f16 filters.  A lane of half h
// holds output rows 8i + 4h + (0..3), i = 0..3: with this placement those are the 16 CONSECUTIVE
// targets 16h .. 16h + 15, so the finish kernel re-reads one 192-byte run per (tile, half).
__device__ __forceinline__ int tile_row(int w) { return 8 * ((w >> 2) & 3) + 4 * (w >> 4) + (w & 3); }

// The whole block evaluates query j against every target with the reference's
// arithmetic and writes (distance, first index): the last resort of the filtered
// paths (three or more tiles within the error bound) and the ONLY path for non-finite
// input, where the reference's tiling shows (chamfer3D.cu:15-36,126): targets are scanned
// in tiles of 512, the first target of a tile initialises the tile's best unconditionally,
// so a NaN distance there makes every later 'd<best' false and the tile's NaN then fails
// 'result>best' -- the whole tile is dropped; for tile 0 the NaN is the result (index 0)
// and is never replaced.  A NaN elsewhere only drops that target; with nothing below +inf
// the first target stays.  Must be called by all threads of the block.
constexpr int kRefTile = 512;    // chamfer3D.cu:13 `const int batch=512`
template <int FMA>
__device__ __forceinline__ void nn_exhaustive(const float *__restrict__ Qp, const float *__restrict__ T, int nt, int j,
                                              float *__restrict__ od, int *__restrict__ oi, float *s_red, int *s_fi)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const float x = Qp[(size_t)j * 3 + 0], y = Qp[(size_t)j * 3 + 1], z = Qp[(size_t)j * 3 + 2];
    // does any tile start with a NaN distance?
    int dead = 0;
    for (int k = threadIdx.x * kRefTile; k < nt; k += kBlock * kRefTile) {
        const float *tp = T + (size_t)k * 3;
        const float hd = sqdist<FMA>(tp[0] - x, tp[1] - y, tp[2] - z);
        dead |= hd != hd;
    }
    dead = __syncthreads_or(dead);
    float bd = __builtin_inff();
    int bi = 0x7fffffff;
    if (!dead) {
        for (int k = threadIdx.x; k < nt; k += kBlock) {
            const float *tp = T + (size_t)k * 3;
            const float dd = sqdist<FMA>(tp[0] - x, tp[1] - y, tp[2] - z);
            const bool lt = dd < bd;
            bd = lt ? dd : bd;
            bi = lt ? k : bi;
        }
    } else {
        for (int k = threadIdx.x; k < nt; k += kBlock) {
            const float *hp = T + (size_t)(k & ~(kRefTile - 1)) * 3;
            const float hd = sqdist<FMA>(hp[0] - x, hp[1] - y, hp[2] - z);
            const float *tp = T + (size_t)k * 3;
            const float dd = sqdist<FMA>(tp[0] - x, tp[1] - y, tp[2] - z);
            const bool lt = dd < bd && hd == hd;
            bd = lt ? dd : bd;
            bi = lt ? k : bi;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float obd = __shfl_xor(bd, o);
        const int obi = __shfl_xor(bi, o);
        const bool other = obd < bd || (obd == bd && obi < bi);
        bd = other ? obd : bd;
        bi = other ? obi : bi;
    }
    __syncthreads();
    if (lane == 0) { s_red[wave] = bd; s_fi[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; w++) {
            const bool other = s_red[w] < bd || (s_red[w] == bd && s_fi[w] < bi);
            bd = other ? s_red[w] : bd;
            bi = other ? s_fi[w] : bi;
        }
        // tile 0 starts with a NaN: that NaN at index 0 is the result; nothing compared below
        // +inf: the scan's unconditional first assignment (target 0) stands
        const float d0 = sqdist<FMA>(T[0] - x, T[1] - y, T[2] - z);
        if (d0 != d0 || bi == 0x7fffffff) {
            bd = d0;
            bi = 0;
        }
        od[j] = bd;
        oi[j] = bi;
    }
}

// Largest approximate value a target can have and still beat (or tie) the target that
// produced a_best, given |approx + |q'|^2 - |q'-t'|^2| <= u (kQT |q'| T + kTT T^2 + 3 |q'|^2)
// (see the derivation above nn_mfma_kernel in chamfer.hip).  Units whose minimum is above
// the returned value cannot contain the reference's nearest neighbour.  NaN when the
// inputs are not finite (every comparison with it then fails -> exhaustive path).
__device__ __forceinline__ float nn_tau(float a_best, float qq, float tmax2, double kQT, double kTT)
{
    const double u = 1.01 * 5.9604644775390625e-8;
    const double T = sqrt((double)tmax2), qn = sqrt((double)qq);
    // kSub: fp32 results below 2^-126 round to a multiple of 2^-149 instead of relatively
    // (squared distances of clouds ~1e-19 across are subnormal); 64 such roundings of slack
    // keep the bound valid there -- it then admits every tile and the queries go exhaustive.
    const double kSub = 64.0 * 1.401298464324817e-45;
    const double E1 = u * (kQT * qn * T + kTT * T * T + 3.0 * (double)qq) + kSub;
    const double eta = u * (qn + T);
    double up = (double)a_best + (double)qq + E1;
    up = sqrt(up > 0.0 ? up : 0.0) + eta;
    up = up * up * (1.0 + 6.0 * u) + kSub;             // >= reference distance of the best target
    double r = sqrt(up / (1.0 - 6.0 * u)) + eta;       // a target with |q'-t'| above r is out
    const double tau = r * r - (double)qq + E1;
    // round up to float
    float tf = (float)tau;
    if ((double)tf < tau) tf = __uint_as_float(__float_as_uint(tf) + (tf >= 0.0f ? 1 : -1));
    return tf;
}


constexpr int kMaxLists = 16;          // slices x lists per lane, when sliced (planner: chamfer.hip)

// Exact (reference arithmetic) minimum and first index over the 16 targets base + 16h ..
// base + 16h + 15 of a 32-target tile: the rows whose approximate values lane half h of the
// filter held (tile_row()).  Positions past the end are clamped to the last target.
template <int FMA>
__device__ __forceinline__ void rescan_half(const float *__restrict__ T, int nt, int base, int h, float qx, float qy,
                                            float qz, float &bd, int &bi)
{
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int first = base + 16 * h;
    bd = __builtin_inff();
    bi = first < nt ? first : nt - 1;
    if (first + 16 <= nt) {
        const f4u *tp = (const f4u *)(T + (size_t)first * 3);
        f4u v[12];
#pragma unroll
        for (int k = 0; k < 12; k++) v[k] = tp[k];
#pragma unroll
        for (int g = 3; g >= 0; g--) {
            const float f[12] = {v[g * 3].x, v[g * 3].y, v[g * 3].z, v[g * 3].w, v[g * 3 + 1].x, v[g * 3 + 1].y,
                                 v[g * 3 + 1].z, v[g * 3 + 1].w, v[g * 3 + 2].x, v[g * 3 + 2].y, v[g * 3 + 2].z, v[g * 3 + 2].w};
#pragma unroll
            for (int c = 3; c >= 0; c--) {
                const float dd = sqdist<FMA>(f[c * 3 + 0] - qx, f[c * 3 + 1] - qy, f[c * 3 + 2] - qz);
                const bool le = dd <= bd;
                bd = le ? dd : bd;
                bi = le ? first + 4 * g + c : bi;
            }
        }
    } else {
        for (int c = 15; c >= 0; c--) {
            int kk = first + c;
            kk = kk < nt ? kk : nt - 1;
            const float *tp = T + (size_t)kk * 3;
            const float dd = sqdist<FMA>(tp[0] - qx, tp[1] - qy, tp[2] - qz);
            const bool le = dd <= bd;
            bd = le ? dd : bd;
            bi = le ? kk : bi;
        }
    }
}

int launch_nn_f16(NNArgs &a, int q, int u, int nl, int tight, long long total_blocks, hipStream_t st);
int launch_nn_finish(NNArgs &a, int nl, int upieces, float kqt, float ktt, float t2min, hipStream_t st);
int launch_nn_grid(const NNArgs &a, hipStream_t st);
size_t nn_dedupe_mask_words(int b, int n);
int launch_nn_dedupe(int b, int nclouds, const float *const pts[2], const int n[2], unsigned *const masks[2], unsigned *hint,
                     hipStream_t st);
// The bidirectional / one-directional nearest-neighbour launch behind genpc_chamfer_forward / genpc_nm_distance
// (chamfer.hip).  dup0 / dup1: duplicate masks of the TARGET clouds of direction 0 / 1 when the caller keeps them
// across calls (the alignment loop and ICP: both clouds' duplicates are the same at every step), else null;
// dup_shared: one mask row serves all batch elements (candidates that are scaled copies of one cloud).
int nn_forward(int b, int ndir, const float *q0, int n0, const float *t0, int m0, float *d0, int *i0, const float *q1,
               int n1, const float *t1, int m1, float *d1, int *i1, hipStream_t st, float radius2 = __builtin_inff(),
               const unsigned *dup0 = nullptr, const unsigned *dup1 = nullptr, int dup_shared = 0);

}  // namespace genpc
