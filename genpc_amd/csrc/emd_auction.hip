// emd_auction.hip -- ALL rounds of the auction (emd_cuda.cu:256-269: Bid, GetMax, Assign x iters, then CalcDist) as ONE
// launch whose threads OWN the points.
//
// Why.  With the culled bid (emd_grid.hip) a late round's work is a few microseconds, but a round was two dependent
// launches whose bodies are chains of dependent memory round trips: count -> list -> point -> last bid / second ->
// their prices -> cell table -> entries -> entries again (exact values) -> publish, then the settle launch's list ->
// bid -> head / owner / price (profiles/r04_emd_B1_16384.json: bid 12.1 us, settle 7.0 us per round, SQ_WAIT_ANY 0.61 /
// 0.96 -- 50 x 19 us = 0.95 of the 0.99 ms call).  Here thread t of the cloud's n / 256 workgroups
// IS point t for the whole call: its coordinates and the sorted positions of the two objects of its last bid live in
// registers, "who bids this round" is one load of assignment[t] -- no lists, no compaction, no appends --, and what
// Settle needs (the object, the increment, the displaced chain record) never leaves the thread.  A round is
//   assignment[t] -> prices of the two seeds -> cell table -> entries (+ prices) -> publish | barrier |
//   head / count / owner / price of the object -> winner's stores | barrier
// i.e. about seven round trips and two barriers among the cloud's OWN workgroups (the clouds of a call never wait for
// each other), against ~fifteen and two launch boundaries.
//
// Who does the work.  The bidders of a wave (ballot of assignment == -1) are served by the wave's lanes in groups of
// LPB = 64 / 32 / 16 / 8 lanes for 1 / 2 / 3-4 / 5+ bidders, several passes if more than 8 bid: the group's lanes get
// the owner's registers by shuffle, run exactly the culled search of emd_bid_grid_kernel (same boxes, same row test,
// same pre-filter, same fp64 value, same proxy scan for stale seeds, same tie sweep), and hand the result back
// through LDS.  So the bits are the reference's for the reasons given in emd_grid.hip.
//
// What other workgroups may read or write is touched ONLY through agent-scope relaxed atomics (L1-bypassing loads,
// write-through stores: assignment, assignment_inv, price in both orders, bid_increments, max_increments, max_idx, the
// chain words, counters) -- the per-XCD L2s are not coherent for plain accesses and a CU's L1 is never refreshed
// (MI355X guide, inter-workgroup visibility: "8-B / sc1 agent atomics both sides" is a valid hand-off without fences).
// Immutable data (both clouds, the sorted copy (x, y, z, object index), the cell table, pos_of) is read with plain
// loads and stays in L1.  The prices of the sorted order are therefore a SEPARATE array here (emd_grid.hip keeps them
// in the entry's .w).
//
// Barrier: one monotone 64-bit word per cloud (low half arrivals, high half abort flag); a workgroup drains its own
// stores (s_waitcnt vmcnt(0) per wave), thread 0 adds 1 and polls.  The launch needs all workgroups of a CLOUD resident
// at once; the host admits a call only while the whole launch fits the chip beside the other persistent launches in
// flight on other streams (persist_reserve), else the call takes the launch-per-round path.  Every spin is bounded: a
// workgroup that gives up raises the abort flag, everybody leaves, dist is poisoned with NaN and genpc_emd_status()
// reports it.
#include "emd.h"
#include "../../include/genpc_hip.h"

#include <mutex>
#include <thread>
#include <vector>

namespace genpc {

constexpr int kABlock = 256;
constexpr int kAWaves = kABlock / kWave;
constexpr float kAU16 = 9.5367431640625e-7f;      // 16 u
constexpr int kATwoPassRows = 25;
constexpr unsigned kAStampBits = 14, kAWhoBits = 18;   // chain record: inc << 32 | stamp << 18 | bidder
constexpr unsigned kAStampPeriod = (1u << kAStampBits) - 1u;
// per-cloud control block, 16 words (one 128-byte line) per item: lines 0..7 the arrival counters of the eight workgroup
// classes (bx % 8: an XCD each when the runtime deals workgroups round-robin), lines 8..15 their release words
// (epoch | abort << 32), line 16 the top counter, line 17 the number of bidders left
constexpr int kCtrlWords = 19 * 16;      // (line 18: the one-word barrier of small clouds)

template <class T> __device__ __forceinline__ T ald(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void ast(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// merge_top2 (emd.h) for indices that carry a payload: position << 32 | object index; ties keep the lower POSITION (any
// choice does: an exact tie for first place is settled by the tie path, the second's identity only seeds the next search)
__device__ __forceinline__ void merge_top2_64(float &b, float &bb, long long &bi, long long &bbi, float ob, float obb, long long oi, long long obi)
{
    float nb2;
    long long nbi;
    if (b > ob) {
        nb2 = fmaxf(bb, ob);
        nbi = ob > bb ? oi : bbi;
    } else if (ob > b) {
        nb2 = fmaxf(obb, b);
        nbi = b > obb ? bi : obi;
    } else {
        nb2 = b;
        nbi = ((unsigned long long)oi < (unsigned long long)bi) ? bi : oi;
    }
    const bool take = ob > b || (ob == b && (unsigned long long)oi < (unsigned long long)bi);
    bi = take ? oi : bi;
    b = fmaxf(b, ob);
    bb = nb2;
    bbi = nbi;
}

struct __attribute__((aligned(16))) F3A { float x, y, z; };      // the coordinates of a sorted entry as one 12-byte load

__device__ __forceinline__ int acell1(float p, float lo, float inv, int g)
{
    const float t = __fmul_rn(__fsub_rn(p, lo), inv);
    int c = (int)floorf(t);
    c = c < 0 ? 0 : c;
    return c > g - 1 ? g - 1 : c;
}

struct EmdAuction {
    int n, nb, cells_max, iters, force_lpb, xcd_pin;
    int K;                                  // lanes that own a point (a power of two <= 64): the cloud has n K / 256 workgroups
    int *feedback;                          // pinned host word (may be null): cloud 0's bidders left after round 2
    float eps;
    const float *xyz1, *xyz2;
    float *price, *price_s;                 // object order (the ABI's array) | sorted order
    const float4 *sorted;                   // (x, y, z, object index)
    const int *start, *pos_of;
    const EGridHdr *hdr;
    int *assignment, *assignment_inv, *bid, *max_idx;
    float *bid_increments, *max_increments, *dist;
    unsigned long long *chain_head, *chain_next;
    int *chain_cnt, *arrived;
    unsigned long long *ctrl;               // per cloud kCtrlWords words on 128-byte lines: see cloud_barrier
    int *status;                            // sticky: != 0 once a call gave up (genpc_emd_status)
    unsigned spin_limit;
    int flat_max;                           // clouds of up to this many workgroups take the one-word barrier
    unsigned long long *timeline;           // debug (GENPC_EMD_TIMELINE=1): 100 MHz stamps of workgroup 0, 8 per round, 64 rounds
};

__global__ __launch_bounds__(kABlock) void emd_auction_init_kernel(int b, int n, unsigned long long *__restrict__ ctrl,
                                                                  unsigned long long *__restrict__ chain_head,
                                                                  int *__restrict__ chain_cnt, int *__restrict__ arrived)
{
    const int t = blockIdx.x * kABlock + threadIdx.x;
    if (t < b * n) { chain_head[t] = 0ull; chain_cnt[t] = 0; arrived[t] = 0; }
    if (t < b * kCtrlWords) ctrl[t] = (t % kCtrlWords) == 17 * 16 ? (unsigned long long)n : 0ull;
}

// Barrier among the G workgroups of one cloud; all threads of the workgroup call it; false once the call is being abandoned.
// Two levels, no polling on a word that is also arrived at: a workgroup arrives at its class's counter (class = bx % 8;
// monotone counters, `epoch` = the barrier's ordinal from 1); the last of a class arrives at the top counter; the last
// there stores the epoch to the eight release words; everybody polls ITS class's release word.  (One word for arrivals and
// polls cost 35 ns per arrival: 512 workgroups, 18 us per barrier -- the pollers' loads queue in front of the atomics.)
__device__ __forceinline__ bool cloud_barrier(unsigned long long *ctrl, int bx, int G, unsigned epoch, unsigned spin_limit, int *s_flag, int flat_max)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's stores and atomics have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0 && G <= flat_max) {
        // A/B form (GENPC_EMD_AUCTION_FLAT, off): ONE word, arrived at without waiting for the old value and polled at a
        // leisurely pace.  Slower than the two-level form below even at 64 workgroups: a word that is arrived at is a bad word
        // to poll on this chip.
        unsigned long long *word = ctrl + 18 * 16;
        (void)__hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        unsigned long long r = ald(word);
        while ((unsigned)r < (unsigned)G * epoch && (r >> 32) == 0ull) {
            __builtin_amdgcn_s_sleep(4);
            r = ald(word);
            if (++spins > spin_limit) {
                __hip_atomic_fetch_or(word, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r |= 1ull << 32;
            }
        }
        *s_flag = (r >> 32) != 0ull ? 1 : 0;
    } else if (threadIdx.x == 0) {
        const int cls = bx & 7, ncls = G < 8 ? G : 8;
        const unsigned size = (unsigned)((G - cls + 7) >> 3);
        unsigned long long *rel = ctrl + (8 + cls) * 16;
        const unsigned long long v = __hip_atomic_fetch_add(ctrl + cls * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
        if (v == (unsigned long long)size * epoch) {
            const unsigned long long t = __hip_atomic_fetch_add(ctrl + 16 * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
            if (t == (unsigned long long)ncls * epoch)
                for (int c = 0; c < ncls; c++) __hip_atomic_fetch_max(ctrl + (8 + c) * 16, (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        unsigned long long r = ald(rel);
        while ((unsigned)r < epoch && (r >> 32) == 0ull) {
            __builtin_amdgcn_s_sleep(2);
            r = ald(rel);
            if (++spins > spin_limit) {
                for (int c = 0; c < ncls; c++) __hip_atomic_fetch_or(ctrl + (8 + c) * 16, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r |= 1ull << 32;
            }
        }
        *s_flag = (r >> 32) != 0ull ? 1 : 0;
    }
    __syncthreads();
    const int f = *s_flag;
    __syncthreads();
    return f == 0;
}

template <int FMA>
__global__ __launch_bounds__(kABlock, 4) void emd_auction_kernel(EmdAuction a)
{
    __shared__ int s_pre[kAWaves][72], s_p0[kAWaves][72];
    __shared__ int s_que[kAWaves][512];
    __shared__ float s_qpr[kAWaves][512];
    __shared__ int s_list[kABlock];             // the workgroup's bidders of the round (owner thread ids), dealt out to its waves in turn
    __shared__ int s_wcnt[kAWaves];
    __shared__ int s_out[kABlock][6];           // per owner thread: object, its position, second's position, increment, displaced record (2 words)
    __shared__ int s_flag;
    const int n = a.n, nb = a.nb, K = a.K, G = (int)(((long long)n * K) / kABlock);
    int batch, bx;
    {
        const int lin = blockIdx.x, nb8 = nb & ~7;
        if (a.xcd_pin && lin < G * nb8) {            // a cloud's workgroups on one XCD (blocks go to the XCDs round-robin): speed only
            const int k = lin >> 3;
            batch = 8 * (k / G) + (lin & 7);
            bx = k % G;
        } else if (a.xcd_pin) {
            batch = nb8 + (lin - G * nb8) / G;
            bx = (lin - G * nb8) % G;
        } else {
            batch = lin % nb;
            bx = lin / nb;
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const size_t base = (size_t)batch * n;
    const int j = (bx * kABlock + (int)threadIdx.x) / K;      // the point this thread owns (with its K - 1 neighbours: K lanes per point)
    const bool rep = (threadIdx.x & (K - 1)) == 0;            // the lane that acts for the point
    const float *__restrict__ X1 = a.xyz1 + base * 3;
    const float *__restrict__ X2 = a.xyz2 + base * 3;
    const float4 *__restrict__ S = a.sorted + base;
    float *PS = a.price_s + base;
    const int *__restrict__ ST = a.start + (size_t)batch * (a.cells_max + 1);
    const int *__restrict__ PO = a.pos_of + base;
    unsigned long long *bar = a.ctrl + (size_t)batch * kCtrlWords;
    int *ucnt = (int *)(bar + 17 * 16);
    const EGridHdr H = a.hdr[batch];
    const int gx = H.g[0], gy = H.g[1], gz = H.g[2];
    const float h = H.h, inf = __builtin_inff();
    const float kShrink = 0.99999905f;      // 1 - 2^-20
    const int block_cnt = n / 256;
    // per point, in LDS (slot of the owner lane): [1], [2] the sorted positions of the objects ranked first and second at its
    // last bid (the seeds of its next one), and what Settle needs of this round's bid
    s_out[threadIdx.x][1] = -1; s_out[threadIdx.x][2] = -1;
    int my_asg = -1;
    unsigned nbar = 0;
    bool ok = true;

    for (int it = 0; it < a.iters; it++) {
        const int last = it == a.iters - 1;
        const unsigned stamp = (unsigned)(it % (int)kAStampPeriod) + 1u;
        if (it > 0 && stamp == 1u) {
            // the stamp wraps: hand the head words back clean (object j's words by thread j), once per 16383 rounds
            ast(&a.chain_head[base + j], 0ull);
            ok = cloud_barrier(bar, bx, G, ++nbar, a.spin_limit, &s_flag, a.flat_max);
            if (!ok) break;
        }
        const bool tl = a.timeline != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && it < 64;
        if (tl) a.timeline[it * 16 + 0] = wall_clock64();
        my_asg = ald(&a.assignment[base + j]);
        const int U = ald(ucnt);
        if (U <= 0) break;                                    // everybody is assigned: the remaining rounds are empty (uniform over the cloud)
        if (a.feedback != nullptr && it == 3 && blockIdx.x == 0 && threadIdx.x == 0)      // (cloud 0 speaks for the call)
            __hip_atomic_store(a.feedback, U, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const bool bidder = rep && my_asg == -1;
        const unsigned long long mask = __ballot(bidder);
        // the workgroup's bidders are pooled and dealt out to its four waves in turn (a wave serving only its own points
        // waited for the wave that happened to hold seven of them: 26 us per late round at 1 x 16384 for ~5 us of search)
        if (lane == 0) s_wcnt[wave] = __popcll(mask);
        __syncthreads();
        int w0 = 0, T = 0;
#pragma unroll
        for (int w = 0; w < kAWaves; w++) { const int c = s_wcnt[w]; w0 += w < wave ? c : 0; T += c; }
        if (bidder) s_list[w0 + __popcll(mask & ((1ull << lane) - 1ull))] = threadIdx.x;
        __syncthreads();
        const int nbid = T > wave ? (T - wave + kAWaves - 1) / kAWaves : 0;      // this wave serves bidders wave, wave + 4, ... of the pool
        if (tl) { a.timeline[it * 16 + 1] = wall_clock64(); a.timeline[it * 16 + 6] = (unsigned long long)U; a.timeline[it * 16 + 7] = (unsigned long long)nbid; }
        if (nbid > 0) {
            // ---------------- Bid (emd_cuda.cu:95-179) ----------------
            int LPB = a.force_lpb > 0 ? a.force_lpb : (nbid <= 1 ? 64 : (nbid == 2 ? 32 : (nbid <= 4 ? 16 : 8)));
            LPB = LPB < 8 ? 8 : (LPB > 64 ? 64 : LPB);
            const int per_wave = kWave / LPB;
            const int sub = lane & (LPB - 1), grp = lane / LPB;
            const int unass_per_block = (U + block_cnt - 1) / block_cnt;
            const int thread_per_unass = 256 / unass_per_block;
            for (int k0 = 0; k0 < nbid; k0 += per_wave) {
                const int kq = k0 + grp;
                const bool active = kq < nbid;
                const int owner = s_list[wave + kAWaves * (active ? kq : nbid - 1)];      // (a thread of this workgroup)
                const int jj = (bx * kABlock + owner) / K;
                const float x1 = X1[(size_t)jj * 3 + 0], y1 = X1[(size_t)jj * 3 + 1], z1 = X1[(size_t)jj * 3 + 2];
                const int pa = s_out[owner][1], pc = s_out[owner][2];
                float best = -1e9f, better = -1e9f;
                long long best_p = -1, better_p = -1;      // position << 32 | object index (the index rides along: no dependent load after the merge)
                float seed = -1e9f;
                bool seeded = false;
                int mode = 0;                           // 0 bid, 1 collect the objects tied for first place, 2 proxy scan
                unsigned long long tie_key = ~0ull;
                float k1 = inf, k2 = inf;
                int q1 = -1, q2 = -1;
                if (pc >= 0 && pa != pc && pa >= 0) {
                    const float4 ea = S[pa], ec = S[pc];
                    const float wa = ald(&PS[pa]), wc = ald(&PS[pc]);
                    const float da = bid_value<FMA>(x1, y1, z1, ea.x, ea.y, ea.z, wa);
                    const float dc = bid_value<FMA>(x1, y1, z1, ec.x, ec.y, ec.z, wc);
                    seed = fminf(da, dc);
                    seeded = true;
                }
                if (tl && k0 == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.timeline[it * 16 + 8] = wall_clock64(); }
                float cb = filter_cb(fmaxf(better, seed));
                const float sx = H.slack[0] + kAU16 * fabsf(x1), sy = H.slack[1] + kAU16 * fabsf(y1), sz = H.slack[2] + kAU16 * fabsf(z1);
                auto gap1 = [&](int c, int g, float lo, float q, float s) {
                    const float wl = c > 0 ? __fadd_rn(lo, __fmul_rn((float)c, h)) : -inf;
                    const float wh = c + 1 < g ? __fadd_rn(lo, __fmul_rn((float)(c + 1), h)) : inf;
                    return fmaxf(0.0f, fmaxf((wl - s) - q, (q - s) - wh));
                };
                int *pre = s_pre[wave] + grp * (LPB + 1), *pp0 = s_p0[wave] + grp * (LPB + 1);
                int *que = s_que[wave] + grp * (8 * LPB);
                float *qpr = s_qpr[wave] + grp * (8 * LPB);
                int qn = 0;
                const unsigned long long gmask = LPB == 64 ? ~0ull : (((1ull << LPB) - 1ull) << (lane & ~(LPB - 1)));
                auto flush = [&]() {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    for (int e = sub; e < qn; e += LPB) {
                        const int ps = que[e];
                        const float w = qpr[e];
                        const float4 o = S[ps];
                        const float d = bid_value<FMA>(x1, y1, z1, o.x, o.y, o.z, w);
                        if (mode == 0) {
                            const bool gt = d > best;
                            const bool gt2 = !gt && d > better;
                            const long long pk = ((long long)ps << 32) | (unsigned)__float_as_int(o.w);
                            better_p = gt ? best_p : (gt2 ? pk : better_p);
                            better = __builtin_amdgcn_fmed3f(d, best, better);
                            best = fmaxf(best, d);
                            best_p = gt ? pk : best_p;
                        } else if (d == best) {
                            // an object that ties for first place: its key in the reference's thread-major scan order
                            // (emd_cuda.cu:108-118,136-139,165-173: the candidate the scan meets first is reported)
                            const int k = __float_as_int(o.w);
                            const int kt = k & 2047;
                            const int tile0 = k - kt;
                            const int end_k = min(n, tile0 + 2048) - tile0;
                            const int delta = (end_k + thread_per_unass - 1) / thread_per_unass;
                            const unsigned long long kk = ((unsigned long long)(kt / delta) << 32) | (unsigned)k;
                            tie_key = kk < tie_key ? kk : tie_key;
                        }
                    }
                    qn = 0;
                    if (mode == 0) {
                        float gb = better;
                        for (int off = 1; off < LPB; off <<= 1) gb = fmaxf(gb, __shfl_xor(gb, off, kWave));
                        seed = fmaxf(seed, gb);
                        cb = filter_cb(fmaxf(better, seed));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                };
                auto batch_eval = [&](int pA, int lA) {
                    // run by run, as emd_bid_grid_kernel (emd_grid.hip): the batch's non-empty runs compacted into the group's
                    // list, the group's lanes stride a run together, four runs' first strides in flight
                    const unsigned long long mA = __ballot(lA > 0) & gmask;
                    const int E = __popcll(mA);
                    if (lA > 0) { const int e = __popcll(mA & ((1ull << lane) - 1ull)); pre[e] = lA; pp0[e] = pA; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    auto test = [&](const F3A &o, float w, int pos, bool in) -> bool {
                        if (!in) return false;
                        const float sq = sqdist_e<FMA>(o.x - x1, o.y - y1, o.z - z1);
                        if (mode == 2) {
                            const float key = sqrtf(sq) + w;
                            if (key < k1) { k2 = k1; q2 = q1; k1 = key; q1 = pos; }
                            else if (key < k2) { k2 = key; q2 = pos; }
                            return false;
                        }
                        const float tt = cb - w;
                        return sq < tt * tt;
                    };
                    auto enqueue = [&](bool pass, int pos, float w) {
                        const unsigned long long m = __ballot(pass) & gmask;
                        if (pass) {
                            const int at = qn + __popcll(m & ((1ull << lane) - 1ull));
                            que[at] = pos;
                            qpr[at] = w;
                        }
                        qn += __popcll(m);
                    };
                    int longest = 0;
                    for (int e0 = 0; e0 < E; e0 += 4) {            // group-uniform trip counts throughout
                        F3A o[4];
                        float w[4];
                        int ps[4];
                        bool in[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const bool have = e0 + i < E;
                            const int p0 = have ? pp0[e0 + i] : pp0[0], ln = have ? pre[e0 + i] : 0;
                            longest = max(longest, ln);
                            in[i] = sub < ln;
                            ps[i] = p0 + (in[i] ? sub : 0);
                            o[i] = *(const F3A *)&S[ps[i]];
                            w[i] = ald(&PS[ps[i]]);
                        }
                        bool pass[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) pass[i] = test(o[i], w[i], ps[i], in[i]);
                        if (mode != 2 && (__ballot(pass[0] | pass[1] | pass[2] | pass[3]) & gmask) != 0ull) {
#pragma unroll
                            for (int i = 0; i < 4; i++) enqueue(pass[i], ps[i], w[i]);
                            if (qn > 4 * LPB) flush();
                        }
                    }
                    if (longest > LPB) {
                        for (int e = 0; e < E; e++) {
                            const int ln = pre[e];
                            if (ln <= LPB) continue;
                            const int p0 = pp0[e];
                            for (int off0 = LPB; off0 < ln; off0 += 2 * LPB) {
                                const int a0 = off0 + sub, a1 = off0 + LPB + sub;
                                const bool i0 = a0 < ln, i1 = a1 < ln;
                                const int pa_i = p0 + (i0 ? a0 : 0), pb_i = p0 + (i1 ? a1 : 0);
                                const F3A oa = *(const F3A *)&S[pa_i], ob = *(const F3A *)&S[pb_i];
                                const float wa = ald(&PS[pa_i]), wb = ald(&PS[pb_i]);
                                const bool pa_ = test(oa, wa, pa_i, i0), pb_ = test(ob, wb, pb_i, i1);
                                if (mode != 2 && (__ballot(pa_ | pb_) & gmask) != 0ull) {
                                    enqueue(pa_, pa_i, wa);
                                    enqueue(pb_, pb_i, wb);
                                    if (qn > 4 * LPB) flush();
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                };
                const int cqx = acell1(x1, H.lo[0], H.inv, gx), cqy = acell1(y1, H.lo[1], H.inv, gy), cqz = acell1(z1, H.lo[2], H.inv, gz);
                const bool cull = !H.bad && (fabsf(x1) + fabsf(y1)) + fabsf(z1) < inf;
                int bx0 = 0, bx1 = gx - 1, by0 = 0, by1 = gy - 1, bz0 = 0, bz1 = gz - 1;
                auto set_box = [&](float R) {
                    bx0 = acell1((x1 - R) - sx, H.lo[0], H.inv, gx); bx1 = acell1((x1 + R) + sx, H.lo[0], H.inv, gx);
                    by0 = acell1((y1 - R) - sy, H.lo[1], H.inv, gy); by1 = acell1((y1 + R) + sy, H.lo[1], H.inv, gy);
                    bz0 = acell1((z1 - R) - sz, H.lo[2], H.inv, gz); bz1 = acell1((z1 + R) + sz, H.lo[2], H.inv, gz);
                };
                if (cull) set_box(cb);
                if (cull && (!seeded || (by1 - by0 + 1) * (bz1 - bz0 + 1) > kATwoPassRows)) {
                    mode = 2;
                    const int ex0 = max(0, cqx - 1), ex1 = min(gx - 1, cqx + 1);
                    for (int r0 = 0; r0 < 9; r0 += LPB) {
                        const int r = r0 + sub;
                        const int cy = cqy + r % 3 - 1, cz = cqz + r / 3 - 1;
                        int pA = 0, lA = 0;
                        if (r < 9 && cy >= 0 && cy < gy && cz >= 0 && cz < gz) {
                            const int row = (cz * gy + cy) * gx;
                            pA = ST[row + ex0];
                            lA = ST[row + ex1 + 1] - pA;
                        }
                        batch_eval(pA, lA);
                    }
                    mode = 0;
                    for (int off = 1; off < LPB; off <<= 1) {
                        const float o1 = __shfl_xor(k1, off, kWave), o2 = __shfl_xor(k2, off, kWave);
                        const int p1 = __shfl_xor(q1, off, kWave), p2 = __shfl_xor(q2, off, kWave);
                        if (o1 < k1) { k2 = fminf(k1, o2) == k1 ? k1 : o2; q2 = (k1 <= o2) ? q1 : p2; k1 = o1; q1 = p1; }
                        else { const bool t = o1 < k2; k2 = t ? o1 : k2; q2 = t ? p1 : q2; }
                    }
                    if (q2 >= 0) {
                        const float4 oa = S[q1], ob = S[q2];
                        const float wa = ald(&PS[q1]), wb = ald(&PS[q2]);
                        seed = fmaxf(seed, fminf(bid_value<FMA>(x1, y1, z1, oa.x, oa.y, oa.z, wa), bid_value<FMA>(x1, y1, z1, ob.x, ob.y, ob.z, wb)));
                        cb = filter_cb(seed);
                        set_box(cb);
                    }
                }
                if (tl && k0 == 0) a.timeline[it * 16 + 9] = wall_clock64();
                auto sweep = [&]() {
                    const int wy = by1 - by0 + 1, nrows = wy * (bz1 - bz0 + 1);
                    for (int r0 = 0; r0 < nrows; r0 += LPB) {         // group-uniform trip count
                        const int r = r0 + sub;
                        int pA = 0, lA = 0;
                        if (r < nrows) {
                            const int rz = r / wy;
                            const int cy = by0 + (r - rz * wy), cz = bz0 + rz;
                            int cx0 = bx0, cx1 = bx1;
                            bool keep = true;
                            if (cull) {
                                const float gyv = gap1(cy, gy, H.lo[1], y1, sy), gzv = gap1(cz, gz, H.lo[2], z1, sz);
                                const float lb = __fmaf_rn(gyv, gyv, __fmul_rn(gzv, gzv)) * kShrink;
                                const float c2 = __fmul_rn(cb, cb);
                                keep = lb < c2;
                                if (keep) {
                                    const float W = sqrtf(fmaxf(0.0f, __fmul_rn(c2, 1.000001f) - lb)) * 1.000001f;
                                    cx0 = max(bx0, acell1((x1 - W) - sx, H.lo[0], H.inv, gx));
                                    cx1 = min(bx1, acell1((x1 + W) + sx, H.lo[0], H.inv, gx));
                                }
                            }
                            if (keep && cx0 <= cx1) {
                                const int row = (cz * gy + cy) * gx;
                                pA = ST[row + cx0];
                                lA = ST[row + cx1 + 1] - pA;
                            }
                        }
                        batch_eval(pA, lA);
                    }
                };
                sweep();
                flush();
                if (tl && k0 == 0) a.timeline[it * 16 + 10] = wall_clock64();
                for (int off = 1; off < LPB; off <<= 1) {
                    const float ob = __shfl_xor(best, off, kWave), obb = __shfl_xor(better, off, kWave);
                    const long long oi = __shfl_xor(best_p, off, kWave), obi = __shfl_xor(better_p, off, kWave);
                    merge_top2_64(best, better, best_p, better_p, ob, obb, oi, obi);
                }
                int best_i = best_p >= 0 ? (int)(best_p & 0xffffffffll) : -1;
                if (tl && k0 == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.timeline[it * 16 + 11] = wall_clock64(); }
                const bool tie = active && (best == better);
                if (__any(tie)) {
                    if (tie) {
                        mode = 1;
                        cb = filter_cb(__uint_as_float(__float_as_uint(best) + (best > 0.0f ? -1 : (best < 0.0f ? 1 : 0))));
                        if (best == 0.0f) cb = filter_cb(-1e-30f);
                        bx0 = 0; bx1 = gx - 1; by0 = 0; by1 = gy - 1; bz0 = 0; bz1 = gz - 1;
                        if (cull) set_box(cb);
                        sweep();
                        flush();
                    }
                    for (int off = 1; off < LPB; off <<= 1) {
                        const unsigned long long o = __shfl_xor(tie_key, off, kWave);
                        tie_key = o < tie_key ? o : tie_key;
                    }
                    if (tie) {
                        best_i = (int)(tie_key & 0xffffffffu);
                        best_p = ((long long)PO[best_i] << 32) | (unsigned)best_i;
                    }
                }
                if (active && sub == 0) {
                    const float inc = __fadd_rn(__fsub_rn(best, better), a.eps);
                    ast(&a.bid[base + jj], best_i);
                    ast(&a.bid_increments[base + jj], inc);
                    atomic_max_float(&a.max_increments[base + best_i], inc);
                    const unsigned long long mine = ((unsigned long long)(unsigned)__float_as_int(inc) << 32) | ((unsigned long long)stamp << kAWhoBits) | (unsigned)jj;
                    const unsigned long long old = atomicExch(&a.chain_head[base + best_i], mine);
                    ast(&a.chain_next[base + jj], old);
                    atomicAdd(&a.chain_cnt[base + best_i], 1);
                    int *out = s_out[owner];
                    out[0] = best_i; out[1] = (int)(best_p >> 32); out[2] = (int)(better_p >> 32); out[3] = __float_as_int(inc);
                    out[4] = (int)(unsigned)(old & 0xffffffffull); out[5] = (int)(unsigned)(old >> 32);
                }
                if (tl && k0 == 0) a.timeline[it * 16 + 12] = wall_clock64();
            }
        }
        if (tl) a.timeline[it * 16 + 2] = wall_clock64();
        ok = cloud_barrier(bar, bx, G, ++nbar, a.spin_limit, &s_flag, a.flat_max);
        if (!ok) break;
        if (tl) a.timeline[it * 16 + 3] = wall_clock64();
        // ---------------- GetMax + Assign (emd_cuda.cu:181-215) ----------------
        if (bidder) {
            auto live = [&](unsigned long long r) { return (unsigned)((r >> kAWhoBits) & kAStampPeriod) == stamp; };
            auto who = [](unsigned long long r) { return (int)(r & ((1u << kAWhoBits) - 1u)); };
            auto inc_of = [](unsigned long long r) { return __int_as_float((int)(r >> 32)); };
            auto take = [&](int o, int w, float inc_w, int pos_s, int prev, float old_price) {
                if (prev != -1) ast(&a.assignment[base + prev], -1);
                else atomicAdd(ucnt, -1);                          // one bidder fewer next round
                ast(&a.assignment_inv[base + o], w);
                ast(&a.assignment[base + w], o);
                const float np_ = __fadd_rn(old_price, inc_w);
                ast(&a.price[base + o], np_);
                ast(&PS[pos_s], np_);
            };
            const int *out = s_out[threadIdx.x];     // (written by a lane of this workgroup before the barrier)
            const int bid_id = out[0], r_pos = out[1];
            const float r_inc = __int_as_float(out[3]);
            const unsigned long long r_next = ((unsigned long long)(unsigned)out[5] << 32) | (unsigned)out[4];
            const unsigned long long head = ald(&a.chain_head[base + bid_id]);
            const int C = ald(&a.chain_cnt[base + bid_id]);
            const int owner_ = ald(&a.assignment_inv[base + bid_id]);
            const float old_price = ald(&a.price[base + bid_id]);
            if (C > 4) {
                const float my_inc = r_inc;
                const double bid_inc = (double)my_inc, max_inc = (double)ald(&a.max_increments[base + bid_id]);
                const bool inwin = last || (bid_inc - 1e-6 <= max_inc && max_inc <= bid_inc + 1e-6);
                if (last) {
                    ast(&a.assignment[base + j], bid_id);
                    atomicAdd(&a.price[base + bid_id], my_inc);
                }
                if (inwin) atomicMax(&a.max_idx[base + bid_id], j);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my election has landed before my ticket is drawn
                const int ticket = __hip_atomic_fetch_add(&a.arrived[base + bid_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ticket == C - 1) {
                    const int w = ald(&a.max_idx[base + bid_id]);
                    if (!last) take(bid_id, w, ald(&a.bid_increments[base + w]), r_pos, owner_, old_price);
                    else ast(&a.assignment_inv[base + bid_id], w);
                    ast(&a.max_increments[base + bid_id], -1e9f);
                    ast(&a.max_idx[base + bid_id], -1);
                    ast(&a.chain_cnt[base + bid_id], 0);
                    ast(&a.arrived[base + bid_id], 0);
                }
            } else {
                int winner = who(head);
                float my_inc = inc_of(head);
                if (winner != j || live(r_next)) {
                    float mx = -1e9f;
                    for (unsigned long long r = head; live(r); r = ald(&a.chain_next[base + who(r)])) {
                        const float v = inc_of(r);
                        mx = v > mx ? v : mx;
                        if (who(r) == j) my_inc = v;
                    }
                    winner = -1;
                    for (unsigned long long r = head; live(r); r = ald(&a.chain_next[base + who(r)])) {
                        const double bid_inc = (double)inc_of(r), max_inc = (double)mx;
                        if ((last || (bid_inc - 1e-6 <= max_inc && max_inc <= bid_inc + 1e-6)) && who(r) > winner) winner = who(r);
                    }
                }
                const bool elected = winner == j;
                if (last) {
                    ast(&a.assignment[base + j], bid_id);
                    atomicAdd(&a.price[base + bid_id], my_inc);
                    my_asg = bid_id;
                    if (elected) {
                        ast(&a.assignment_inv[base + bid_id], j);
                        ast(&a.max_increments[base + bid_id], -1e9f);
                        ast(&a.max_idx[base + bid_id], -1);
                        ast(&a.chain_cnt[base + bid_id], 0);
                    }
                } else if (elected) {
                    take(bid_id, j, my_inc, r_pos, owner_, old_price);
                    ast(&a.max_increments[base + bid_id], -1e9f);
                    ast(&a.max_idx[base + bid_id], -1);
                    ast(&a.chain_cnt[base + bid_id], 0);
                }
            }
            if (last) my_asg = bid_id;
        }
        if (tl) a.timeline[it * 16 + 4] = wall_clock64();
        if (last) break;
        ok = cloud_barrier(bar, bx, G, ++nbar, a.spin_limit, &s_flag, a.flat_max);
        if (!ok) break;
        if (tl) a.timeline[it * 16 + 5] = wall_clock64();
    }
    // ---------------- CalcDist (emd_cuda.cu:217-226) ----------------
    if (!ok) {
        if (threadIdx.x == 0 && bx == 0) atomicExch(a.status, 1);
        if (rep) a.dist[base + j] = __builtin_nanf("");
        return;
    }
    if (!rep) return;
    // (a point assigned in an earlier round may have been evicted since this thread last looked; the forced last round
    // evicts nobody, so what a thread saw at the start of the last round it took part in, or took there, stands --
    // except when the loop ended early because nobody was left to bid: then nobody was evicted after that look either)
    if (my_asg < 0 && a.iters > 0) my_asg = ald(&a.assignment[base + j]);
    if (my_asg < 0) {
        a.dist[base + j] = 0.0f;            // iters == 0: the reference would read out of bounds
    } else {
        const float *p2 = X2 + (size_t)my_asg * 3;
        a.dist[base + j] = sqdist_e<FMA>(X1[(size_t)j * 3 + 0] - p2[0], X1[(size_t)j * 3 + 1] - p2[1], X1[(size_t)j * 3 + 2] - p2[2]);
    }
}

// ---- admission of persistent launches (all workgroups of a launch must be resident together) ----
// One transaction: persist_reserve() leaves a PENDING entry (no event yet) that already counts against the
// budget, so two host threads cannot both pass the check in the gap between their reservation and their launch;
// persist_commit() attaches the event behind the launch, persist_cancel() withdraws the entry when the launch did not
// happen.  A thread that has to wait does so OUTSIDE the lock (on a copy of the oldest event's handle, or by yielding while
// the blocker is still pending), so other lanes' commits are not held up behind it.
struct PersistEntry { hipEvent_t ev; int wgs; hipStream_t stream; int dev; };
static std::mutex g_persist_mu;
static std::vector<PersistEntry> g_persist;
static std::vector<hipEvent_t> g_persist_free;
// the calling thread's last admitted launch was let in BESIDE other streams' persistent launches: only such a launch can fail to
// become resident as a whole (genpc_emd_contended(): the Python layer then reads the status word and, if need be, repeats the call)
static thread_local int t_persist_contended = 0;

// Blocks until `wgs` workgroups fit beside the persistent launches in flight (or reserved) on OTHER streams of the device
// (launches of one stream run one after the other anyway); false if the request alone exceeds `capacity`.
bool persist_reserve(int wgs, int capacity, hipStream_t st)
{
    if (wgs > capacity) return false;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::unique_lock<std::mutex> l(g_persist_mu);
    for (;;) {
        int used = 0, oldest = -1;
        for (size_t i = 0; i < g_persist.size();) {
            if (g_persist[i].dev == dev && g_persist[i].ev && hipEventQuery(g_persist[i].ev) == hipSuccess) {
                g_persist_free.push_back(g_persist[i].ev);
                g_persist.erase(g_persist.begin() + i);
                continue;
            }
            if (g_persist[i].dev == dev && g_persist[i].stream != st) {
                used += g_persist[i].wgs;
                if (oldest < 0) oldest = (int)i;
            }
            i++;
        }
        if (used + wgs <= capacity || oldest < 0) {
            g_persist.push_back(PersistEntry{nullptr, wgs, st, dev});      // pending: counted from now on
            t_persist_contended = used > 0 ? 1 : 0;
            return true;
        }
        const hipEvent_t ev = g_persist[oldest].ev;      // (null: that launch is reserved, not enqueued yet)
        l.unlock();
        if (ev) (void)hipEventSynchronize(ev);           // (a recycled handle waits for something else or nothing: the scan above decides)
        else std::this_thread::yield();
        l.lock();
    }
}

// the launch reserved on `st` has been enqueued: its entry gets the event that tells when it is over
void persist_commit(int wgs, hipStream_t st)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> l(g_persist_mu);
    size_t at = g_persist.size();
    for (size_t i = 0; i < g_persist.size(); i++)
        if (g_persist[i].dev == dev && g_persist[i].stream == st && !g_persist[i].ev && g_persist[i].wgs == wgs) { at = i; break; }
    if (at == g_persist.size()) return;                  // (nothing reserved: nothing to commit)
    hipEvent_t ev = nullptr;
    if (!g_persist_free.empty()) { ev = g_persist_free.back(); g_persist_free.pop_back(); }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    if (!ev || hipEventRecord(ev, st) != hipSuccess) {
        if (ev) g_persist_free.push_back(ev);
        g_persist.erase(g_persist.begin() + at);         // (cannot be tracked: rather uncounted than counted for ever)
        return;
    }
    g_persist[at].ev = ev;
}

// the launch reserved on `st` did not happen
void persist_cancel(int wgs, hipStream_t st)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> l(g_persist_mu);
    for (size_t i = 0; i < g_persist.size(); i++)
        if (g_persist[i].dev == dev && g_persist[i].stream == st && !g_persist[i].ev && g_persist[i].wgs == wgs) {
            g_persist.erase(g_persist.begin() + i);
            return;
        }
}

int emd_auction_capacity()
{
    // resident 256-thread workgroups the admission counts on: __launch_bounds__(256, 4) = four per CU; one in eight is
    // kept back (the occupancy the API reports can be one block per CU high: MI355X guide, residency)
    return num_cus() * 7 / 2;
}

size_t emd_auction_bytes(int b, int n)
{
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t total = (size_t)b * n;
    return 256 /* status */ + al((size_t)b * kCtrlWords * 8) + al((size_t)b * sizeof(EGridHdr)) + al((size_t)b * (kEGMaxCells + 1) * sizeof(int)) +
           al(total * sizeof(float4)) + al(total * sizeof(float)) + 2 * al(total * sizeof(int)) + 2 * al(total * 8) + 2 * al(total * sizeof(int));
}

// Which path suits the data is known only on the device: how many points still bid after the first rounds (uniform clouds:
// a fifth; a partial scan against its ground truth: two thirds, for all 50 rounds -- there a round is throughput, and the
// launch-per-round path, which spreads the bidders over the whole chip whatever the cloud count, is the faster one).  Both
// paths give the same bits, so the choice is free: every call leaves the number of bidders after round 2 (of its
// first cloud) in a pinned host word per (b, n) class -- a system-scope store, no copy, no synchronisation --, and the
// next call of that class reads whatever has arrived by then.
static std::mutex g_fb_mu;
static int *g_fb_host = nullptr, *g_fb_dev = nullptr;
constexpr int kFbSlots = 256;
int *emd_feedback_slot(int b, int n, bool device)
{
    std::lock_guard<std::mutex> l(g_fb_mu);
    if (!g_fb_host) {
        if (hipHostMalloc((void **)&g_fb_host, kFbSlots * sizeof(int), hipHostMallocMapped) != hipSuccess) { g_fb_host = nullptr; return nullptr; }
        for (int i = 0; i < kFbSlots; i++) g_fb_host[i] = 0;
        if (hipHostGetDevicePointer((void **)&g_fb_dev, g_fb_host, 0) != hipSuccess) g_fb_dev = nullptr;
    }
    const unsigned h = ((unsigned)n * 2654435761u ^ (unsigned)b * 40503u) % kFbSlots;
    return device ? (g_fb_dev ? g_fb_dev + h : nullptr) : g_fb_host + h;
}

// Returns 1 launched, 0 error, -1 not admitted (the caller takes the launch-per-round path).
int launch_emd_auction(int b, int n, const float *xyz1, const float *xyz2, float *dist, int *assignment, float *price, int *assignment_inv,
                       int *bid, float *bid_increments, float *max_increments, int *max_idx, float eps, int iters, int fma, hipStream_t st, bool forced)
{
    static const int env_cap = tune_env("GENPC_EMD_AUCTION_CAP", 0, "one-launch EMD: workgroups admitted at once (0 = 3.5 per CU)");
    const int cap = env_cap > 0 ? env_cap : emd_auction_capacity();
    if (n > (1 << kAWhoBits) || n < kABlock || (long long)b * (n / kABlock) > cap) return -1;
    {
        static const int env_heavy = tune_env("GENPC_EMD_AUCTION_HEAVY_PCT", 35, "one-launch EMD: bidders left after round 2 (percent of n, last call of the same shape) above which the launch-per-round path is taken (0 = never)");
        volatile int *fb = emd_feedback_slot(b, n, false);
        if (!forced && env_heavy > 0 && fb && iters > 3 && (long long)*fb * 100 > (long long)n * env_heavy) return -1;
    }
    // lanes per point: as many as keep the launch within `fill` workgroups (the bidders of a round are served by the lanes
    // of their own workgroup: more lanes per point = fewer bidders per wave, and round 0 -- everybody bids -- on the whole chip)
    static const int env_k = tune_env("GENPC_EMD_AUCTION_K", 0, "one-launch EMD: lanes that own a point (1..64, 0 = pick)");
    static const int env_fill = tune_env("GENPC_EMD_AUCTION_FILL", 512, "one-launch EMD: workgroups up to which points get more lanes");
    int K = 1;
    if (env_k > 0) { while (K < env_k && K < 64) K <<= 1; }
    else { while (K < 64 && (long long)b * n * (2 * K) / kABlock <= env_fill) K <<= 1; }
    while (K > 1 && (long long)b * n * K / kABlock > cap) K >>= 1;
    const int G = (int)((long long)n * K / kABlock), wgs = b * G;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t total = (size_t)b * n;
    char *ws = (char *)workspace(31, emd_auction_bytes(b, n), st, nullptr, 256);
    if (!ws) return 0;
    int *status = (int *)ws;
    char *p = ws + 256;
    unsigned long long *ctrl = (unsigned long long *)p; p += al((size_t)b * kCtrlWords * 8);
    EGridHdr *hdr = (EGridHdr *)p; p += al((size_t)b * sizeof(EGridHdr));
    int *start = (int *)p; p += al((size_t)b * (kEGMaxCells + 1) * sizeof(int));
    float4 *sorted = (float4 *)p; p += al(total * sizeof(float4));
    float *price_s = (float *)p; p += al(total * sizeof(float));
    int *pos_of = (int *)p; p += al(total * sizeof(int));
    int *orig_of = (int *)p; p += al(total * sizeof(int));
    unsigned long long *chain_head = (unsigned long long *)p; p += al(total * 8);
    unsigned long long *chain_next = (unsigned long long *)p; p += al(total * 8);
    int *chain_cnt = (int *)p; p += al(total * sizeof(int));
    int *arrived = (int *)p;
    if (!persist_reserve(wgs, cap, st)) return -1;
    {
        const size_t items = total > (size_t)b * kCtrlWords ? total : (size_t)b * kCtrlWords;      // (a 256-point cloud has fewer points than control words)
        hipLaunchKernelGGL(emd_auction_init_kernel, dim3(ceil_div((int)items, kABlock)), dim3(kABlock), 0, st, b, n, ctrl, chain_head, chain_cnt, arrived);
    }
    // (six objects per cell for the one-launch kernel since round 6 -- tools/emd_env_sweep.py, 1 x 16384 / 13 x 16384 uniform: 20 -> 1.04 /
    //  1.70 ms, 40 -> 0.95 / 1.68, 60 -> 0.89 / 1.67, 80 -> 0.89 / 1.69, 120 -> 0.93 / 1.74: a bidder's ball is a handful of
    //  longer runs instead of dozens of short ones, each a dependent read of the cell table; the launch-per-round path keeps its own)
    static const int env_ppc = tune_env("GENPC_EMD_AUCTION_PPC_X10", 60, "one-launch EMD: target objects per cell x 10 of the culled bid's grid");
    int target = (int)((long long)n * 10 / (env_ppc > 0 ? env_ppc : 60));
    target = target < 8 ? 8 : (target > kEGMaxCells * 3 / 4 ? kEGMaxCells * 3 / 4 : target);
    if (!launch_emd_grid_build(b, n, xyz2, price, hdr, start, sorted, pos_of, orig_of, target, kEGMaxCells, st, price_s)) {
        persist_cancel(wgs, st);
        return 0;
    }
    EmdAuction a{};
    a.n = n; a.nb = b; a.cells_max = kEGMaxCells; a.iters = iters; a.eps = eps; a.K = K;
    a.feedback = emd_feedback_slot(b, n, true);
    static const int env_lpb = tune_env("GENPC_EMD_LPB", 0, "culled EMD bid: lanes per bidder (8..64, 0 = pick)");
    a.force_lpb = env_lpb;
    static const int env_xcd = tune_env("GENPC_EMD_AUCTION_XCD", 0, "one-launch EMD: 1 = a cloud's workgroups on one XCD");
    a.xcd_pin = env_xcd;
    a.xyz1 = xyz1; a.xyz2 = xyz2; a.price = price; a.price_s = price_s; a.sorted = sorted; a.start = start; a.pos_of = pos_of; a.hdr = hdr;
    a.assignment = assignment; a.assignment_inv = assignment_inv; a.bid = bid; a.max_idx = max_idx;
    a.bid_increments = bid_increments; a.max_increments = max_increments; a.dist = dist;
    a.chain_head = chain_head; a.chain_next = chain_next; a.chain_cnt = chain_cnt; a.arrived = arrived;
    a.ctrl = ctrl; a.status = status;
    static const int env_spin = tune_env("GENPC_EMD_AUCTION_SPIN", 1 << 21, "one-launch EMD: polls of a barrier before the call is abandoned");
    a.spin_limit = (unsigned)env_spin;
    // (measured and OFF: 13 x 16384 -- 64 workgroups per cloud -- 1.68 ms with the two-level barrier, 2.60 with one word per cloud
    // that is both arrived at and polled, however leisurely the polls; 1 x 16384 at 4 lanes per point 1.05 = 1.05)
    static const int env_flat = tune_env("GENPC_EMD_AUCTION_FLAT", 0, "one-launch EMD: clouds of up to this many workgroups use a one-word barrier (0 = never: measured slower)");
    a.flat_max = env_flat;
    static const int env_tl = tune_env("GENPC_EMD_TIMELINE", 0, "one-launch EMD: 1 = workgroup 0 stamps the phases of the first 64 rounds (genpc_debug_emd_timeline)");
    a.timeline = env_tl ? (unsigned long long *)workspace(32, 64 * 16 * 8, nullptr, nullptr, 64 * 16 * 8) : nullptr;
    if (fma) hipLaunchKernelGGL((emd_auction_kernel<1>), dim3(wgs), dim3(kABlock), 0, st, a);
    else hipLaunchKernelGGL((emd_auction_kernel<0>), dim3(wgs), dim3(kABlock), 0, st, a);
    persist_commit(wgs, st);
    return check(hipGetLastError(), "emd_auction_kernel launch") ? 1 : 0;
}

}  // namespace genpc

/* 1 = the calling thread's last one-launch EMD call was admitted beside persistent launches of other streams (the only
 * situation in which it can be abandoned: genpc_emd_status then tells); reading clears the flag.  No device access. */
GENPC_API int genpc_emd_contended(void)
{
    const int v = genpc::t_persist_contended;
    genpc::t_persist_contended = 0;
    return v;
}

/* 0 = no one-launch EMD call on this stream's workspace has been abandoned since the last query (synchronises the stream);
 * 1 = one was (its dist is NaN): the launch did not become resident as a whole within the spin bound.  reset != 0 clears. */
GENPC_API int genpc_emd_status(int reset, void *stream)
{
    using namespace genpc;
    int *dev = (int *)workspace(31, 256, (hipStream_t)stream, nullptr, 256);
    if (!dev) return -1;
    if (!check(hipStreamSynchronize((hipStream_t)stream), "genpc_emd_status sync")) return -1;
    int v = 0;
    if (!check(hipMemcpy(&v, dev, sizeof v, hipMemcpyDeviceToHost), "genpc_emd_status copy")) return -1;
    if (reset && v && !check(hipMemset(dev, 0, sizeof v), "genpc_emd_status reset")) return -1;
    return v;
}

// Diagnostic (not part of the public header): the stamps of GENPC_EMD_TIMELINE=1, 64 rounds x 8 words.
extern "C" __attribute__((visibility("default"))) int genpc_debug_emd_timeline(unsigned long long *out)
{
    using namespace genpc;
    unsigned long long *dev = (unsigned long long *)workspace(32, 64 * 16 * 8, nullptr, nullptr, 64 * 16 * 8);
    if (!dev || hipDeviceSynchronize() != hipSuccess) return 0;
    return hipMemcpy(out, dev, 64 * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 1 : 0;
}
