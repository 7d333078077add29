// fps_grid.hip -- exact farthest point sampling with spatial pruning, ONE workgroup per cloud (round 6).
//
// Same definition as fps.hip (start index 0, fp32 squared distances in the library's arithmetic mode, first arg-max;
// oracle_fps_mode), same sequences bit for bit -- but the work per sample follows what a sample can change:
//
//   * a new sample s lowers the running minimum D_i only of points closer to s than sqrt(M), M = the largest running
//     minimum when s was drawn: the points are sorted once into a uniform grid (cell order in HBM / L2, 16 bytes per point:
//     x, y, z, original index), the running minima live in LDS in the same order, and an update visits only the cells the
//     ball of radius sqrt(M) touches (sphere-box test against each cell's largest running minimum);
//   * several samples per round, exactly: the round's candidates are ALL points with D >= theta (every other point ranks
//     below every candidate, whatever happens: D only falls), sorted by (D descending, index ascending).  The sorted list IS
//     the sampling sequence up to the first candidate that an earlier one of the list lowers (its pair distance is below
//     its D): the prefix is drawn at once (<= 64 samples), the rest waits for the next round.  The expected prefix at
//     sample k is ~sqrt(k / 2): ~500 rounds for 20000 samples of 24000 points;
//   * theta follows the data: every round sweeps the per-cell maxima (upper bounds, refreshed whenever a cell is scanned or
//     updated) for cells that can hold a candidate and scans those; a sweep that finds none lowers theta from the largest
//     bound, one that finds more than the list holds raises it between the smallest and the largest candidate, and a
//     list of one repeated value (lattices: thousands of exact ties) is served by the tie path -- the lowest original
//     index among the tied points, one sample per round.
//
// No hand-off between workgroups: nothing has to be co-resident, nothing is admitted, nothing can time out; clouds of one
// launch are independent blocks.  Clouds beyond kGMaxN points keep fps.hip's multi-workgroup kernel.
// The sequence and the running minimum of every sample when drawn go to the same outputs as fps.hip's, so
// fps_verify_kernel checks these samplings against the definition like the others.
#include "common.h"
#include "../../include/genpc_hip.h"

namespace genpc {

constexpr int kGT = 1024;                 // threads of the workgroup (16 waves: four per SIMD)
constexpr int kGWaves = kGT / kWave;
constexpr int kGMaxN = 24576;             // running minima in LDS: 4 B per point (+ 0.5 per point of chunk maxima)
constexpr int kGCand = 128;               // candidates a round lists
constexpr int kGPick = 64;                // samples a round draws at most
constexpr int kGQueue = 4096;             // work items of an update (sample, <= 8 points of a cell) / tied points of the tie path
constexpr int kGMaxJobs = 8;
constexpr int kGAxis = 64;                // cells per axis at most (cell ranges are packed in 8-bit fields)
constexpr unsigned kInfBits = 0x7f800000u;

struct FpsGridJobs {
    const float *xyz[kGMaxJobs];
    int *out[kGMaxJobs];
    float *pdist[kGMaxJobs];
    float4 *spt[kGMaxJobs];      // the cloud in cell order: x, y, z, original index (bits)
    int n[kGMaxJobs], k[kGMaxJobs];
    int stat0;
};

// cells of a cloud of n points: ~two points per cell of the bounding box (surfaces fill a fraction of the cells)
__host__ __device__ inline int fps_grid_cells(int n)
{
    int c = n / 2;
    c = c < 64 ? 64 : c;
    c = (c + 7) & ~7;
    return c > 8192 ? 8192 : c;
}

// LDS bytes of a cloud of n points
__host__ __device__ inline size_t fps_grid_lds(int n)
{
    const int npad = (n + 63) & ~63;
    const int cmaxn = fps_grid_cells(n);
    return (size_t)npad * 4 + (size_t)(npad / 8) * 4 + (size_t)(cmaxn + 8) * 2 + (size_t)kGQueue * 4 + 1792 * 4;
}

template <int FMA>
__device__ __forceinline__ float gsq(float dx, float dy, float dz)
{
    if (FMA) {
        float t = __fmul_rn(dy, dy);
        t = __fmaf_rn(dx, dx, t);
        return __fmaf_rn(dz, dz, t);
    } else {
        const float a = __fmul_rn(dx, dx), b = __fmul_rn(dy, dy), c = __fmul_rn(dz, dz);
        return __fadd_rn(__fadd_rn(a, b), c);
    }
}

// q / d for 0 <= q < 2^19 / d (rd = 1 / d rounded): (q + 1/2) / d is never within rounding of an integer
__device__ __forceinline__ int gdiv(int q, float rd) { return (int)(((float)q + 0.5f) * rd); }

__device__ __forceinline__ int gcell(float v, float lo, float inv, int n)
{
    // truncation = floor for non-negative values; NaN -> 0 (v_max drops it)
    return (int)fminf(fmaxf((v - lo) * inv, 0.0f), (float)(n - 1));
}

// slot = counter++ for the lanes of the wave that call it together (one LDS atomic per wave, not per lane)
__device__ __forceinline__ int wave_push(int *counter)
{
    const unsigned long long mask = __ballot(1);
    const int leader = (int)__ffsll((long long)mask) - 1;
    const int lane = threadIdx.x & (kWave - 1);
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, (int)__popcll(mask));
    base = __shfl(base, leader);
    return base + (int)__popcll(mask & ((1ull << lane) - 1ull));
}

template <int FMA>
__global__ __launch_bounds__(kGT) void fps_grid_kernel(FpsGridJobs jobs, int *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];
    const int job = blockIdx.x, t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
    const int n = jobs.n[job], k = jobs.k[job];
    const float *__restrict__ X = jobs.xyz[job];
    float4 *__restrict__ P = jobs.spt[job];
    int *__restrict__ out = jobs.out[job];
    float *__restrict__ pdist = jobs.pdist[job];
    const int npad = (n + 63) & ~63;
    const int nch = npad >> 3;
    const int cells_max = fps_grid_cells(n);

    unsigned *D = (unsigned *)g_smem;                                  // running minima (bits), cell order; 0 behind the cloud's end
    unsigned *cnt = D;                                                 // (the build's cell counters: D is filled after the build)
    float *chmax = (float *)(D + npad);                                // per chunk of 8 points: an upper bound of their running minima
    unsigned short *cs = (unsigned short *)(chmax + nch);              // first point of every cell, cells_max + 1 entries
    unsigned *queue = (unsigned *)(cs + cells_max + 8);
    float *f = (float *)(queue + kGQueue);
    float4 *cq = (float4 *)f;                 // [kGCand] candidates as found: x, y, z, running minimum (-1: no longer in the list)
    int *cidx = (int *)(f + 512);
    int *cpos = (int *)(f + 640);
    int *crank = (int *)(f + 768);
    int *cconf = (int *)(f + 896);
    float *skey = f + 1024;                   // [kGPick] the round's samples in sampling order
    int *sidx = (int *)(f + 1088);
    float *sx = f + 1152, *sy = f + 1216, *sz = f + 1280;
    int *plo = (int *)(f + 1408), *pw = (int *)(f + 1472);
    int *scand = (int *)(f + 1536);           // which candidate a sample was
    float *red = f + 1600;                    // [kGWaves * 6]
    int *sh = (int *)(f + 1696);              // shared scalars: 0 candidates, 1 samples of the sub-round, 2 work items, 3 tied points, 5 candidates alive, 8.. bisection
    // (1792 floats in all; sh[0..31])

    // ------------------------------------------------------------------ the grid
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = t; i < n; i += kGT) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = X[(size_t)i * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { red[wave * 6 + a] = mn[a]; red[wave * 6 + 3 + a] = mx[a]; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; a++) {
        mn[a] = red[a];
        mx[a] = red[3 + a];
        for (int w = 1; w < kGWaves; w++) { mn[a] = fminf(mn[a], red[w * 6 + a]); mx[a] = fmaxf(mx[a], red[w * 6 + 3 + a]); }
    }
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    const float emax = fmaxf(ex, fmaxf(ey, ez));
    float h = 1.0f;
    int nx = 1, ny = 1, nz = 1;
    if (emax > 0.0f && emax < __builtin_inff()) {
        h = emax / (float)kGAxis;
        for (int it = 0; it < 64; it++) {
            const float ih = 1.0f / h;
            nx = (int)fminf(ex * ih, (float)(kGAxis - 1)) + 1;
            ny = (int)fminf(ey * ih, (float)(kGAxis - 1)) + 1;
            nz = (int)fminf(ez * ih, (float)(kGAxis - 1)) + 1;
            if (nx * ny * nz <= cells_max) break;
            h *= 1.1f;
        }
        if (nx * ny * nz > cells_max) { nx = ny = nz = 1; h = emax * 2.0f; }
    }
    const float inv = 1.0f / h;
    const float gx0 = mn[0], gy0 = mn[1], gz0 = mn[2];
    const float box_margin = (h + emax) * 3.8e-6f;      // a point may sit this far outside its cell's nominal box (rounding of the cell index)

    for (int c = t; c < cells_max; c += kGT) cnt[c] = 0u;
    __syncthreads();
    for (int i = t; i < n; i += kGT) {
        const float x = X[(size_t)i * 3 + 0], y = X[(size_t)i * 3 + 1], z = X[(size_t)i * 3 + 2];
        const int c = (gcell(z, gz0, inv, nz) * ny + gcell(y, gy0, inv, ny)) * nx + gcell(x, gx0, inv, nx);
        atomicAdd(&cnt[c], 1u);
    }
    __syncthreads();
    {
        // exclusive scan of the counters: consecutive cells per thread
        const int per = (cells_max + kGT - 1) / kGT;
        const int c0 = t * per;
        unsigned loc = 0;
        for (int u = 0; u < per; u++) if (c0 + u < cells_max) loc += cnt[c0 + u];
        unsigned incl = loc;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        unsigned *wsum = (unsigned *)red;
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; w++) base += wsum[w];
        unsigned run = base + incl - loc;
        for (int u = 0; u < per; u++) {
            if (c0 + u < cells_max) {
                const unsigned v = cnt[c0 + u];
                cs[c0 + u] = (unsigned short)run;
                run += v;
            }
        }
        if (t == kGT - 1) cs[cells_max] = (unsigned short)n;
        __syncthreads();
        for (int c = t; c < cells_max; c += kGT) cnt[c] = 0u;
        __syncthreads();
    }
    for (int i = t; i < n; i += kGT) {
        const float x = X[(size_t)i * 3 + 0], y = X[(size_t)i * 3 + 1], z = X[(size_t)i * 3 + 2];
        const int c = (gcell(z, gz0, inv, nz) * ny + gcell(y, gy0, inv, ny)) * nx + gcell(x, gx0, inv, nx);
        const int p = (int)cs[c] + (int)atomicAdd(&cnt[c], 1u);
        P[p] = make_float4(x, y, z, __int_as_float(i));
    }
    __syncthreads();
    for (int p = t; p < npad; p += kGT) D[p] = p < n ? kInfBits : 0u;
    for (int j = t; j < nch; j += kGT) chmax[j] = __builtin_inff();
    if (t < 32) sh[t] = 0;
    if (t == 0) {
        out[0] = 0;
        pdist[0] = __builtin_inff();
        skey[0] = __builtin_inff();
        sidx[0] = 0;
        sx[0] = X[0]; sy[0] = X[1]; sz[0] = X[2];
        plo[0] = 0;
        pw[0] = nx | (ny << 8) | (nz << 16);
    }
    __threadfence_block();
    __syncthreads();

    int s = 0;                       // samples drawn and applied
    int L = 1;                       // samples of the last round, not applied yet (the start point)
    float theta = __builtin_inff(), delta = 0.25f;
    int rem_before = -1;             // candidates the last round left, when theta was lowered after it
    unsigned rounds = 0, attempts = 0;
    const unsigned max_attempts = 64u * (unsigned)k + 4096u;
    bool failed = false;
    // phase clocks (thread 0, 100 MHz wall clock, summed over the run; job 0 of a launch leaves them in err[48..53]:
    // update / sweep / resolve / leave, sweeps, list overflows -- tools/fps_grid_check.py prints them)
    unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl0 = wall_clock64();
    unsigned long long n_items = 0, n_cands = 0;
    const unsigned long long cyc0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
    unsigned n_sweeps = 0, n_over = 0;
// (tools/fps_grid_check.py --build compiles a private copy with -DGENPC_FPS_TIMELINE; the shipped kernel reads no clock)
#ifdef GENPC_FPS_TIMELINE
#define GENPC_GTL(i) do { const unsigned long long now_ = wall_clock64(); tl[i] += now_ - tl0; tl0 = now_; } while (0)
#else
#define GENPC_GTL(i) do { (void)tl; (void)tl0; } while (0)
#endif

    // one work item of an update: sample j against <= 8 points of a cell
    auto apply_item = [&](unsigned item) {
        const int j = (int)(item >> 18), cn = (int)((item >> 15) & 7u) + 1, a = (int)(item & 32767u);
        const float px = sx[j], py = sy[j], pz = sz[j];
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = P[a + (u < cn ? u : cn - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (u < cn) {
                const float dd = gsq<FMA>(v[u].x - px, v[u].y - py, v[u].z - pz);
                atomicMin(&D[a + u], __float_as_uint(dd));
            }
        }
    };

    for (;;) {
        // ---------------------------------------------------------- apply the round's samples
        {
            int sh_l = 0;
            while ((1 << sh_l) < L) sh_l++;
            const int shift = 10 - sh_l, tpp = 1 << shift;
            const int j = t >> shift, sub = t & (tpp - 1);
            const bool mine = j < L;
            float px = 0.f, py = 0.f, pz = 0.f, key = 0.f;
            int x0 = 0, y0 = 0, z0 = 0, wx = 1, wy = 1, vol = 0;
            if (mine) {
                px = sx[j]; py = sy[j]; pz = sz[j]; key = skey[j];
                const int lo = plo[j], w = pw[j];
                x0 = lo & 255; y0 = (lo >> 8) & 255; z0 = (lo >> 16) & 255;
                wx = w & 255; wy = (w >> 8) & 255;
                vol = wx * wy * ((w >> 16) & 255);
            }
            const int wxy = wx * wy;
            const float rwxy = 1.0f / (float)wxy, rwx = 1.0f / (float)wx;
            // (wave-uniform loop: the lanes reserve their queue slots together, one LDS atomic per wave and trip)
            for (int q = sub; __any(q < vol); q += tpp) {
                int nck = 0, p0 = 0, p1 = 0;
                if (q < vol) {
                    const int iz = gdiv(q, rwxy);
                    const int r2 = q - iz * wxy;
                    const int iy = gdiv(r2, rwx);
                    const int ix = r2 - iy * wx;
                    const int cxi = x0 + ix, cyi = y0 + iy, czi = z0 + iz;
                    const int c = (czi * ny + cyi) * nx + cxi;
                    p0 = cs[c]; p1 = cs[c + 1];
                    if (p1 > p0) {
                        // the cell's box against the ball: no point of the cell is closer than this
                        const float bx = gx0 + (float)cxi * h, by = gy0 + (float)cyi * h, bz = gz0 + (float)czi * h;
                        const float ddx = fmaxf(fmaxf(bx - box_margin - px, px - (bx + h + box_margin)), 0.0f);
                        const float ddy = fmaxf(fmaxf(by - box_margin - py, py - (by + h + box_margin)), 0.0f);
                        const float ddz = fmaxf(fmaxf(bz - box_margin - pz, pz - (bz + h + box_margin)), 0.0f);
                        const float d2 = (ddx * ddx + ddy * ddy + ddz * ddz) * 0.9999f;
                        if (d2 < key) {                               // (a running minimum is at most the sample's own when it was drawn)
                            nck = (p1 - p0 + 7) >> 3;
                            // a small cell at the ball's fringe: every point of it may already be closer to an earlier sample
                            // than the cell is to this one (the chunk maxima bound the cell's running minima from above)
                            const int j0c = p0 >> 3, j1c = (p1 - 1) >> 3;
                            if (j1c - j0c <= 2) {
                                const float bound = fmaxf(chmax[j0c], fmaxf(chmax[(j0c + j1c) >> 1], chmax[j1c]));
                                if (d2 >= bound) nck = 0;
                            }
                        }
                    }
                }
                int incl = nck;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                int base = 0;
                if (lane == kWave - 1 && incl > 0) base = atomicAdd(&sh[2], incl);
                base = __shfl(base, kWave - 1) + incl - nck;
                for (int i = 0; i < nck; i++) {
                    const int a = p0 + 8 * i, cn = min(8, p1 - a);
                    const unsigned item = ((unsigned)j << 18) | ((unsigned)(cn - 1) << 15) | (unsigned)a;
                    if (base + i < kGQueue) queue[base + i] = item;
                    else apply_item(item);
                }
            }
            __syncthreads();
            GENPC_GTL(4);
            n_items += (unsigned)sh[2];
            const int nq = min(sh[2], kGQueue);
            for (int it = t; it < nq; it += kGT) apply_item(queue[it]);
        }
        s += L;
        __syncthreads();
        GENPC_GTL(0);
        if (s >= k) break;

        // ---------------------------------------------------------- the round's candidates: every point with D >= theta
        int count = 0;
        bool tie_path = false, zero_path = false;
        for (;;) {
            if (++attempts > max_attempts) { failed = true; break; }
            n_sweeps++;
            if (t == 0) { sh[0] = 0; sh[2] = 0; }
            __syncthreads();
            float mloc = -1.0f, kmin = __builtin_inff(), kmax = -1.0f;
            for (int j = t; j < nch; j += kGT) {
                float cm = chmax[j];
                if (cm >= theta) {
                    const uint4 da = *(const uint4 *)&D[j * 8], db = *(const uint4 *)&D[j * 8 + 4];
                    const unsigned dv[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
                    unsigned nm = 0u;
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        nm = nm > dv[u] ? nm : dv[u];
                        const float d = __uint_as_float(dv[u]);
                        if (d >= theta && j * 8 + u < n) {
                            const int slot = atomicAdd(&sh[0], 1);
                            if (slot < kGCand) { cpos[slot] = j * 8 + u; cq[slot].w = d; }
                            kmin = fminf(kmin, d);
                            kmax = fmaxf(kmax, d);
                        }
                    }
                    cm = __uint_as_float(nm);
                    chmax[j] = cm;
                }
                mloc = fmaxf(mloc, cm);
            }
            __syncthreads();
            count = sh[0];
            if (count >= 1 && count <= kGCand) break;
            // the largest bound / the candidates' range, block-wide
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                mloc = fmaxf(mloc, __shfl_xor(mloc, o));
                kmin = fminf(kmin, __shfl_xor(kmin, o));
                kmax = fmaxf(kmax, __shfl_xor(kmax, o));
            }
            if (lane == 0) { red[wave * 3 + 0] = mloc; red[wave * 3 + 1] = kmin; red[wave * 3 + 2] = kmax; }
            __syncthreads();
            mloc = red[0]; kmin = red[1]; kmax = red[2];
            for (int w = 1; w < kGWaves; w++) { mloc = fmaxf(mloc, red[w * 3]); kmin = fminf(kmin, red[w * 3 + 1]); kmax = fmaxf(kmax, red[w * 3 + 2]); }
            __syncthreads();
            rem_before = -1;
            if (count == 0) {
                if (!(mloc > 0.0f)) { zero_path = true; break; }
                theta = mloc * (1.0f - delta);
                delta = fminf(delta * 2.0f, 0.5f);
            } else {
                n_over++;
                if (kmin == kmax) { theta = kmax; tie_path = true; break; }
                float th = kmin + 0.5f * (kmax - kmin);
                if (!(th > kmin)) th = kmax;
                theta = th;
                delta = fmaxf(delta * 0.5f, 1e-7f);
            }
        }
        GENPC_GTL(1);
        if (failed) break;
        if (zero_path) {
            // every remaining running minimum is 0: the first arg-max is index 0 from here on
            for (int i = s + t; i < k; i += kGT) { out[i] = 0; pdist[i] = 0.0f; }
            break;
        }
        rounds++;
        bool single = false;
        if (tie_path) {
            // More points tied at the largest running minimum than the list holds (lattices): the sequence takes them by
            // ascending original index.  All tied points -> the work queue as (index << 16 | position); the list = the 128 lowest.
            if (t == 0) { sh[3] = 0; sh[0] = 0; sh[4] = 0x7fffffff; }
            if (t < 20) sh[8 + t] = 0;
            __syncthreads();
            for (int j = t; j < nch; j += kGT) {
                if (chmax[j] >= theta) {
                    const uint4 da = *(const uint4 *)&D[j * 8], db = *(const uint4 *)&D[j * 8 + 4];
                    const unsigned dv[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (__uint_as_float(dv[u]) == theta && j * 8 + u < n) {
                            const unsigned key = ((unsigned)__float_as_int(P[j * 8 + u].w) << 16) | (unsigned)(j * 8 + u);
                            const int slot = atomicAdd(&sh[3], 1);
                            if (slot < kGQueue) queue[slot] = key;
                            atomicMin(&sh[4], (int)key);                 // (the lowest index of all, should the queue overflow; keys are below 2^31)
                        }
                    }
                }
            }
            __syncthreads();
            const int nt = sh[3];
            if (nt > kGQueue) {
                // (more than the queue holds: one sample this round, the lowest index)
                single = true;
                if (t == 0) {
                    const unsigned key = (unsigned)sh[4];
                    const int want = (int)(key >> 16), pos = (int)(key & 0xffffu);
                    const float4 q = P[pos];
                    skey[0] = theta; sidx[0] = want;
                    sx[0] = q.x; sy[0] = q.y; sz[0] = q.z;
                    sh[1] = 1;
                }
                __syncthreads();
            } else {
                // the smallest I with |{index < I}| >= 128 (indices are distinct: then exactly 128)
                unsigned mine[4];
#pragma unroll
                for (int u = 0; u < 4; u++) mine[u] = t + u * kGT < nt ? queue[t + u * kGT] >> 16 : 0xffffffffu;
                int lo = 0, hi = 65536;
                for (int it = 0; it < 16; it++) {
                    const int mid = (lo + hi) >> 1;
                    int c4 = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) c4 += mine[u] < (unsigned)mid ? 1 : 0;
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) c4 += __shfl_xor(c4, o);
                    if (lane == 0 && c4) atomicAdd(&sh[8 + it], c4);
                    __syncthreads();
                    if (sh[8 + it] >= kGCand) hi = mid; else lo = mid;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (mine[u] < (unsigned)hi) {
                        const unsigned key = queue[t + u * kGT];
                        const int slot = atomicAdd(&sh[0], 1);
                        if (slot < kGCand) { cpos[slot] = (int)(key & 0xffffu); cq[slot].w = theta; }
                    }
                }
                __syncthreads();
                count = min(sh[0], kGCand);
            }
        }
        int Ltot = 0;
        if (single) Ltot = 1;
        else {
            // ------------------------------------------------------ the candidates' coordinates and original indices
            if (t < kGCand) {
                if (t < count) {
                    const float4 q = P[cpos[t]];
                    cq[t] = make_float4(q.x, q.y, q.z, cq[t].w);
                    cidx[t] = __float_as_int(q.w);
                } else cq[t] = make_float4(0.f, 0.f, 0.f, -1.0f);
                crank[t] = 0; cconf[t] = 0;
            }
            if (t == 0) { sh[5] = 0; sh[1] = 0x7fffffff; }
            __syncthreads();
            GENPC_GTL(5);
            n_cands += (unsigned)count;
            // ------------------------------------------------------ order the candidates, find the prefix that is the sequence:
            // rank = candidates in front of it (larger minimum, or equal and lower index); lowered = one of those is closer than its
            // minimum.  The prefix of unlowered candidates is drawn; the others take the new samples into their minima and the
            // ones still at theta or above go again (sub-rounds: no other point of the cloud can interfere -- all are below theta)
            const int cap = min(kGPick, k - s);
            for (int subr = 0;; subr++) {
                {
                    const int i = lane + 64 * (wave & 1), j0 = (wave >> 1) * 16;
                    const float4 qi = cq[i];
                    const int ii = cidx[i];
                    if (qi.w >= 0.0f) {
                        int r = 0;
                        bool low = false;
#pragma unroll 8
                        for (int jj = 0; jj < 16; jj++) {
                            const float4 qj = cq[j0 + jj];
                            const int ij = cidx[j0 + jj];
                            const bool front = (qj.w > qi.w) | ((qj.w == qi.w) & (ij < ii));
                            const float dd = gsq<FMA>(qi.x - qj.x, qi.y - qj.y, qi.z - qj.z);
                            r += front ? 1 : 0;
                            low |= front & (dd < qi.w);
                        }
                        if (r) atomicAdd(&crank[i], r);
                        if (low) atomicOr(&cconf[i], 1);
                        if (j0 == 0) atomicAdd(&sh[5], 1);
                    }
                }
                __syncthreads();
                const int mcap = cap - Ltot;
                if (t < kGCand) {
                    // into sampling order (everything that could be drawn; what lies behind the first lowered candidate is ignored),
                    // and the rank of the first lowered candidate: where the sequence stops (sh[1] was set to a large value)
                    const float4 q = cq[t];
                    const int r = crank[t];
                    if (q.w >= 0.0f && r < mcap) {
                        const int o = Ltot + r;
                        skey[o] = q.w; sidx[o] = cidx[t]; sx[o] = q.x; sy[o] = q.y; sz[o] = q.z; scand[o] = t;
                        if (cconf[t]) atomicMin(&sh[1], r);
                    }
                }
                __syncthreads();
                const int alive = sh[5];
                const int Lsub = min(min(sh[1], alive), mcap);
                Ltot += Lsub;
                if (Lsub == 0 || Ltot >= cap || Lsub >= alive || subr >= 3) break;
                // the candidates left take the new samples into their minima; the drawn ones leave the list
                {
                    const int i = t & (kGCand - 1), part = t >> 7;
                    const float4 qi = cq[i];
                    if (qi.w >= 0.0f) {
                        float mnew = qi.w;
                        for (int jq = Ltot - Lsub + part; jq < Ltot; jq += kGT / kGCand) {
                            const float dd = gsq<FMA>(qi.x - sx[jq], qi.y - sy[jq], qi.z - sz[jq]);
                            mnew = mnew < dd ? mnew : dd;
                        }
                        if (mnew < qi.w) atomicMin((unsigned *)&cq[i].w, __float_as_uint(mnew));
                    }
                }
                __syncthreads();
                if (t < kGCand) {
                    const float kk = cq[t].w;
                    const bool drawn = kk >= 0.0f && crank[t] < Lsub;
                    if (drawn || (kk >= 0.0f && !(kk >= theta))) cq[t].w = -1.0f;
                    crank[t] = 0; cconf[t] = 0;
                }
                if (t == 0) { sh[5] = 0; sh[1] = 0x7fffffff; }
                __syncthreads();
            }
        }
        L = Ltot;
        GENPC_GTL(2);
        // ---------------------------------------------------------- the samples leave; their balls' cell ranges
        if (t < L) {
            out[s + t] = sidx[t];
            pdist[s + t] = skey[t];
            const float key = skey[t];
            const float r = sqrtf(key) * 1.00001f;
            const float px = sx[t], py = sy[t], pz = sz[t];
            const float lx = (px - r) - (fabsf(px) + r) * 2.4e-7f, hx = (px + r) + (fabsf(px) + r) * 2.4e-7f;
            const float ly = (py - r) - (fabsf(py) + r) * 2.4e-7f, hy = (py + r) + (fabsf(py) + r) * 2.4e-7f;
            const float lz = (pz - r) - (fabsf(pz) + r) * 2.4e-7f, hz = (pz + r) + (fabsf(pz) + r) * 2.4e-7f;
            const int x0 = gcell(lx, gx0, inv, nx), x1 = gcell(hx, gx0, inv, nx);
            const int y0 = gcell(ly, gy0, inv, ny), y1 = gcell(hy, gy0, inv, ny);
            const int z0 = gcell(lz, gz0, inv, nz), z1 = gcell(hz, gz0, inv, nz);
            plo[t] = x0 | (y0 << 8) | (z0 << 16);
            pw[t] = (x1 - x0 + 1) | ((y1 - y0 + 1) << 8) | ((z1 - z0 + 1) << 16);
        }
        // ---------------------------------------------------------- next round's threshold
        if (!tie_path) {
            if (rem_before >= 0) {
                // theta was lowered in front of this round: how many candidates that step brought
                const int gained = count - rem_before;
                if (gained > 100) delta = fmaxf(delta * 0.6f, 1e-7f);
                else if (gained < 40) delta = fminf(delta * 1.5f, 0.5f);
            }
            const int rem = count - L;
            if (rem < 56) {
                rem_before = rem;
                theta = theta * (1.0f - delta);
            } else rem_before = -1;
        } else rem_before = -1;
        __syncthreads();
        GENPC_GTL(3);
    }
    if (t == 0 && job == 0) {
        for (int i = 0; i < 7; i++) err[48 + i] = (int)(tl[i] / 100);      // microseconds
        err[55] = (int)n_sweeps;
        err[56] = (int)n_over;
        err[57] = (int)(n_items / (rounds ? rounds : 1));
        err[58] = (int)(n_cands / (rounds ? rounds : 1));
        err[59] = (int)((__builtin_readcyclecounter() - cyc0) * 100 / (wall_clock64() - wall0 + 1));      // shader clock, MHz
    }
    if (t == 0) {
        if (failed) {
            out[0] = -1;
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        err[1 + jobs.stat0 + job] = (int)rounds;
    }
}

static int fps_grid_launch_one(bool fma, int nj, size_t lds, hipStream_t st, const FpsGridJobs &jobs, int *err)
{
    static size_t set_bytes[2] = {0, 0};
    if (lds > set_bytes[fma ? 1 : 0]) {
        const hipError_t e = fma ? hipFuncSetAttribute((const void *)fps_grid_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                 : hipFuncSetAttribute((const void *)fps_grid_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (!check(e, "hipFuncSetAttribute(fps_grid_kernel)")) return 0;
        set_bytes[fma ? 1 : 0] = lds;
    }
    if (fma) hipLaunchKernelGGL((fps_grid_kernel<1>), dim3(nj), dim3(kGT), lds, st, jobs, err);
    else hipLaunchKernelGGL((fps_grid_kernel<0>), dim3(nj), dim3(kGT), lds, st, jobs, err);
    return 1;
}

// Clouds `sel[0..c)` (indices into the call's arrays) through the one-workgroup kernel.  `spt`: 16 bytes per point of scratch,
// `pdist_of[j]`: where cloud j's running minima go (fps.hip's verification reads them), err: the call's error / statistics words.
int fps_grid_run(bool fma, int c, const int *sel, const int *n, const int *k, const float *const *xyz, int *const *out_idx,
                 float *const *pdist_of, float4 *spt, int *err, hipStream_t st)
{
    size_t off = 0;
    for (int j0 = 0; j0 < c; j0 += kGMaxJobs) {
        FpsGridJobs jobs = {};
        const int nj = c - j0 < kGMaxJobs ? c - j0 : kGMaxJobs;
        size_t lds = 0;
        jobs.stat0 = 0;
        for (int q = 0; q < nj; q++) {
            const int j = sel[j0 + q];
            jobs.xyz[q] = xyz[j];
            jobs.out[q] = out_idx[j];
            jobs.pdist[q] = pdist_of[j];
            jobs.spt[q] = spt + off;
            off += ((size_t)n[j] + 63) & ~(size_t)63;
            jobs.n[q] = n[j];
            jobs.k[q] = k[j];
            const size_t b = fps_grid_lds(n[j]);
            lds = b > lds ? b : lds;
        }
        // statistics slot of the launch's first cloud (genpc_fps_stats reads the first 32 clouds of a call)
        jobs.stat0 = sel[j0] < 32 ? sel[j0] : 32;
        // (the kernel writes err[1 + stat0 + job]: consecutive slots -- exact when the selected clouds are consecutive, as they are
        //  for calls of one size; a ragged call's statistics may land in a neighbour's slot, never outside the 64 words)
        if (jobs.stat0 + nj > 40) jobs.stat0 = 40 - nj;
        if (!fps_grid_launch_one(fma, nj, lds, st, jobs, err)) return 0;
    }
    return check(hipGetLastError(), "fps_grid_kernel launch") ? 1 : 0;
}

bool fps_grid_takes(int n) { return n >= 1 && n <= kGMaxN; }

}  // namespace genpc
