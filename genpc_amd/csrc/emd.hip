// emd.hip -- auction-algorithm EMD (forward + backward) for gfx950.
// Replaces loss_functions/emd/emd_cuda.cu of the reference.
//
// Round structure.  The reference runs 7 launches per round (clear, count,
// prefix-sum, compact, Bid, GetMax, Assign: emd_cuda.cu:256-268).  Here a round is
// three launches with launch shapes that do not depend on device-side state, so
// the host never synchronises:
//   bid      every unassigned point j finds its best and second-best object
//            (value 3 - |x2_k - x1_j| - price_k, evaluated in double exactly as
//            emd_cuda.cu:146 does) and atomically raises max_increments[best];
//   getmax   among the bidders of an object, the highest j whose increment is
//            within 1e-6 of the maximum is elected (atomicMax on max_idx; the
//            reference's plain store is racy -- last writer wins -- and the oracle
//            resolves it the same way, in favour of the highest j);
//   assign   winners take the object, evict the previous owner, raise the price;
//            the list of NEXT round's bidders (losers + evicted owners) is built
//            here with wave-aggregated appends, which replaces the reference's
//            four compaction kernels.
// getmax + assign are ONE launch where they were two list-walking ones (emd_settle_kernel: the bid
// kernel chains the bidders of an object, every bidder finds the winner by itself), or one
// single-block launch per cloud (emd_resolve_kernel: small clouds / many clouds, late rounds).
// Bid layout.  P lanes cooperate on one bidder (P = 1..64, a power of two picked
// per round from the number of bidders so that the grid stays full when few
// points are left); lane p of a bidder visits objects k = p (mod P) of a tile of
// (x,y,z,price) float4s staged in LDS -- consecutive lanes read consecutive 16 B
// slots (conflict-free ds_read_b128), lanes of different bidders read the same
// slot (broadcast).  Per-lane top-2 is v_max + v_med3; the P partial results are
// merged with xor-shuffles.
// Index ties.  With exactly equal top values the reference reports the
// candidate that comes first in ITS thread-major scan order (emd_cuda.cu:108-118,
// 136-139,165-173).  The fast path only needs the two top VALUES; when they are
// equal (a tie for first place) the bidder's lanes re-scan and select the tied
// object with the smallest (reference-thread, index) key, so assignments agree
// with the oracle even on clouds with duplicated points.
#include "emd.h"
#include "../../include/genpc_hip.h"

#include <stdlib.h>
#include <type_traits>

namespace genpc {

// First launch of a call: the list of round 0 (everybody bids), and the library's own per-call state -- no seed yet
// (second = -1), empty bidder chains, no tickets drawn -- which used to be four memsets in front of the first round
// (a stream operation each: ~4 us apiece of a 1 ms call).
__global__ __launch_bounds__(kEBlock) void emd_init_kernel(int b, int n, int *__restrict__ list, int *__restrict__ cnt_a,
                                                           int *__restrict__ cnt_b, int *__restrict__ second,
                                                           unsigned long long *__restrict__ chain_head,
                                                           unsigned long long *__restrict__ whead, int *__restrict__ chain_cnt,
                                                           int *__restrict__ arrived)
{
    int t = blockIdx.x * kEBlock + threadIdx.x;
    if (t < b * n) {
        list[t] = t % n;
        if (second) second[t] = -1;
        if (chain_head) { chain_head[t] = 0ull; whead[t] = 0ull; chain_cnt[t] = 0; arrived[t] = 0; }
    }
    if (t < b) {
        cnt_a[t] = n;
        cnt_b[t] = 0;
    }
}

template <int FMA, int FILTER, int TILE, int AHEAD>
__global__ __launch_bounds__(kEBlock) void emd_bid_kernel(int n, const float *__restrict__ xyz1,
                                                          const float *__restrict__ xyz2,
                                                          const float *__restrict__ price, float eps,
                                                          const int *__restrict__ list, const int *__restrict__ cnt,
                                                          int *__restrict__ cnt_next, int *__restrict__ bid,
                                                          float *__restrict__ bid_increments,
                                                          float *__restrict__ max_increments, int force_p,
                                                          float4 *__restrict__ parts, int *__restrict__ arrive,
                                                          int *__restrict__ second, int zmax,
                                                          unsigned long long *__restrict__ chain_head,
                                                          unsigned long long *__restrict__ chain_next, unsigned stamp, int G, int nb,
                                                          int *__restrict__ chain_cnt)
{
    constexpr int kTile = TILE, kLoadsPerThread = TILE / 256;
    // one tile of objects as four planes (x, y, z, price): a 16-byte read delivers four consecutive objects' x as two
    // register pairs for the packed fp32 filter below
    __shared__ __attribute__((aligned(16))) float sX[kTile], sY[kTile], sZ[kTile], sP[kTile];
    // 1-D grid of G * nb blocks.  The blocks of a cloud run on ONE XCD (blocks go to the XCDs round-robin by linear id), for
    // the largest multiple of 8 clouds: every block of a cloud re-reads the cloud's objects and prices tile by tile, and with
    // its blocks on all eight XCDs every 4 MB L2 held every cloud (8 x 32768: 4.2 MB; 14.9 -> 13.9 ms, 64 x 2048 1.74 -> 1.54).
    int batch, bx;
    {
        const int lin = blockIdx.x, nb8 = nb & ~7;      // (the clouds beyond a multiple of 8: the (G, nb) order)
        if (lin < G * nb8) {
            const int k = lin >> 3;
            batch = 8 * (k / G) + (lin & 7);
            bx = k % G;
        } else {
            batch = nb8 + (lin - G * nb8) / G;
            bx = (lin - G * nb8) % G;
        }
    }
    const int U = cnt[batch];
    if (bx == 0 && threadIdx.x == 0) cnt_next[batch] = 0;   // filled by this round's assign
    if (U == 0) return;
    const int P = force_p > 0 ? force_p : pick_p(U, G);
    const int per_wave = kWave / P;
    const int per_block = kEBlock / P;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int p = lane & (P - 1);
    const int g = lane / P;

    const float *__restrict__ X1 = xyz1 + (size_t)batch * n * 3;
    const float *__restrict__ X2 = xyz2 + (size_t)batch * n * 3;
    const float *__restrict__ PR = price + (size_t)batch * n;
    const int *__restrict__ L = list + (size_t)batch * n;

    // reference partition, needed only to order exactly tied candidates
    const int block_cnt = n / 256;
    const int unass_per_block = (U + block_cnt - 1) / block_cnt;
    const int thread_per_unass = 256 / unass_per_block;

    // Work units.  NB blocks cover the U bidders at P lanes each.  When even the
    // finest split (P = 64) leaves most of the grid idle -- the late rounds, where a
    // few hundred bidders remain -- the OBJECTS are split as well: Z slices of tiles,
    // unit = (bidder group, slice); a unit writes its partial top-2 and the last unit
    // of a group to arrive merges them (agent-scope release -> ticket -> acquire, as
    // in chamfer.hip).  Without this a late round is one block per CU walking all
    // n/1024 tiles back to back: pure load latency (measured 50 us per round at
    // n = 16384 for ~10 us of work).
    const int NB = (int)(((long long)U * P + kEBlock - 1) / kEBlock);
    const int ntiles = (n + kTile - 1) / kTile;
    int Z = 1;
    if (P == 64 && parts != nullptr) {
        Z = G / NB;
        Z = Z > ntiles ? ntiles : Z;
        Z = Z > zmax ? zmax : Z;
        Z = Z < 1 ? 1 : Z;
        if ((long long)U > kSplitMaxBidders) Z = 1;     // parts[] holds kSplitMaxBidders bidders per batch
    }
    const int tps = (ntiles + Z - 1) / Z;               // tiles per slice
    const int units = NB * Z;

    for (int unit = bx; unit < units; unit += G) {
        const int grp = unit % NB;
        const int zs = unit / NB;
        const int k_lo = zs * tps * kTile;
        const int k_hi = min(n, (zs + 1) * tps * kTile);
        // Tiles are staged through registers one tile ahead: the global loads of tile
        // t+1 are in flight while tile t is scanned (a round with few bidders is otherwise
        // a chain of load -> barrier -> short scan -> barrier per tile: pure latency).  The first
        // tile is requested before the bidder's own chain of loads (list -> point -> seed objects),
        // on which it does not depend and behind whose waits it would otherwise queue.
        // AHEAD tiles are in flight (registers) at any time: 1 where many blocks are resident (the other blocks hide the
        // fetch), 4 in the late rounds of a few clouds, where a unit is a chain -- fetch, stage, a four-trip scan, fetch
        // ... -- and each tile's fetch latency was exposed (tools: ~2.9 k ticks per 1024-object tile of a unit's 27 k).
        float4 pre[AHEAD][kLoadsPerThread];
        auto fetch = [&](auto slot_c, int k2) {
            constexpr int slot = decltype(slot_c)::value;
#pragma unroll
            for (int i = 0; i < kLoadsPerThread; i++) {
                const int k = k2 + threadIdx.x + i * kEBlock;
                const int kk = k < n ? k : n - 1;
                pre[slot][i] = make_float4(X2[(size_t)kk * 3 + 0], X2[(size_t)kk * 3 + 1], X2[(size_t)kk * 3 + 2], PR[kk]);
            }
        };
        if (k_lo < k_hi) fetch(std::integral_constant<int, 0>{}, k_lo);
        if (AHEAD > 1 && k_lo + kTile < k_hi) fetch(std::integral_constant<int, (AHEAD > 1 ? 1 : 0)>{}, k_lo + kTile);
        if (AHEAD > 2 && k_lo + 2 * kTile < k_hi) fetch(std::integral_constant<int, (AHEAD > 2 ? 2 : 0)>{}, k_lo + 2 * kTile);
        if (AHEAD > 3 && k_lo + 3 * kTile < k_hi) fetch(std::integral_constant<int, (AHEAD > 3 ? 3 : 0)>{}, k_lo + 3 * kTile);
        const int u = grp * per_block + wave * per_wave + g;
        const bool active = u < U;
        const int j = L[active ? u : U - 1];
        const float x1 = X1[(size_t)j * 3 + 0], y1 = X1[(size_t)j * 3 + 1], z1 = X1[(size_t)j * 3 + 2];
        float best = -1e9f, better = -1e9f;
        int best_i = -1, better_i = -1;
        // Seed of the pre-filter: the objects this point ranked first and second the
        // last time it bid, re-valued at today's prices.  The second-best value over ALL
        // objects is at least the smaller of any two distinct objects' values, so a
        // candidate provably not above `seed` can be neither best nor strictly second.
        // The filter is thereby selective from the first object on, whatever the length
        // of a lane's sequence (without it, late rounds with 64 lanes per bidder
        // evaluated every pair exactly).
        float seed = -1e9f;
        if (FILTER && second != nullptr) {
            const int sa = bid[(size_t)batch * n + j], sc = second[(size_t)batch * n + j];
            if (sc >= 0 && sa != sc && (unsigned)sa < (unsigned)n) {
                const float da = bid_value<FMA>(x1, y1, z1, X2[(size_t)sa * 3 + 0], X2[(size_t)sa * 3 + 1],
                                                X2[(size_t)sa * 3 + 2], PR[sa]);
                const float dc = bid_value<FMA>(x1, y1, z1, X2[(size_t)sc * 3 + 0], X2[(size_t)sc * 3 + 1],
                                                X2[(size_t)sc * 3 + 2], PR[sc]);
                seed = fminf(da, dc);
            }
        }
        float cb = filter_cb(fmaxf(better, seed));

        auto tile_step = [&](auto slot_c, int k2) {
            constexpr int slot = decltype(slot_c)::value;
            const int end_k = min(n, k2 + kTile) - k2;
            __syncthreads();                               // the previous tile has been scanned
#pragma unroll
            for (int i = 0; i < kLoadsPerThread; i++) {
                const int t = threadIdx.x + i * kEBlock;
                if (t < end_k) { sX[t] = pre[slot][i].x; sY[t] = pre[slot][i].y; sZ[t] = pre[slot][i].z; sP[t] = pre[slot][i].w; }
            }
            __syncthreads();
            if (k2 + AHEAD * kTile < k_hi) fetch(slot_c, k2 + AHEAD * kTile);
            // Pre-filter.  A candidate can change this lane's (best, better) only if its
            // value exceeds `better`, i.e. only if sqrt(s) < 3 - price - better.  That is
            // tested conservatively in squared space with fp32 and no sqrt / fp64:
            // cb = (3 - better) + 2e-6 (1 + |better|) absorbs every rounding of the test itself at
            // ANY magnitude of the clouds (filter_cb: the slack needed is u (21 + 9 |better|),
            // u = 2^-24, the slack given 33 u (1 + |better|)), so a candidate that fails the
            // test provably evaluates to d <= better and the exact update would be a
            // no-op.  The exact path (double-precision expression of emd_cuda.cu:146)
            // runs for the whole wave when any lane passes; for lanes that did not pass
            // it is that same no-op.  Results are therefore bit-identical with and
            // without the filter (GENPC_EMD_NOFILTER=1 disables it for A/B).
            // Four CONSECUTIVE objects per lane and iteration (4*P divides every tile length: n % 256 == 0,
            // P <= 64), evaluated two per instruction (packed fp32: same operations, same roundings as
            // sqdist_e and filter_cb's test); a lane's verdicts stay in SGPRs (one ballot per object), so a
            // group none of whose members can matter costs 16 packed operations, 4 compares and one scalar branch
            // (28 instead of 51 instructions per four objects; in-run A/B with the 1024-object tiles: 13 x 16384 7.68 ->
            // 7.31 ms, 8 x 32768 16.05 -> 15.05 -- with 2048-object tiles and half the resident waves it LOST 6 %;
            // issuing the next group's LDS reads early costs registers and residency: 7.23 -> 8.6 ms).
            const v2f x1v = {x1, x1}, y1v = {y1, y1}, z1v = {z1, z1};
            for (int t = 4 * p; t < end_k; t += 4 * P) {
                const float4 X4 = *(const float4 *)&sX[t], Y4 = *(const float4 *)&sY[t], Z4 = *(const float4 *)&sZ[t],
                             W4 = *(const float4 *)&sP[t];
                const v2f cbv = {cb, cb};
                v2f sqv[2];
                unsigned long long pass[4];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const v2f dx = (h ? (v2f){X4.z, X4.w} : (v2f){X4.x, X4.y}) - x1v;
                    const v2f dy = (h ? (v2f){Y4.z, Y4.w} : (v2f){Y4.x, Y4.y}) - y1v;
                    const v2f dz = (h ? (v2f){Z4.z, Z4.w} : (v2f){Z4.x, Z4.y}) - z1v;
                    if (FMA) sqv[h] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
                    else sqv[h] = (dx * dx + dy * dy) + dz * dz;
                    const v2f tt = cbv - (h ? (v2f){W4.z, W4.w} : (v2f){W4.x, W4.y});
                    const v2f t2 = tt * tt;
                    pass[2 * h] = __ballot(!FILTER || sqv[h].x < t2.x);
                    pass[2 * h + 1] = __ballot(!FILTER || sqv[h].y < t2.y);
                }
                if ((pass[0] | pass[1] | pass[2] | pass[3]) != 0ull) {
                    const float sq[4] = {sqv[0].x, sqv[0].y, sqv[1].x, sqv[1].y};
                    const float pr[4] = {W4.x, W4.y, W4.z, W4.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (pass[i] != 0ull) {
                            const float r = sqrtf(sq[i]);
                            const float d = (float)((3.0 - (double)r) - (double)pr[i]);
                            const bool gt = d > best;
                            const bool gt2 = !gt && d > better;
                            const int kk = k2 + t + i;
                            better_i = gt ? best_i : (gt2 ? kk : better_i);
                            better = __builtin_amdgcn_fmed3f(d, best, better);
                            best = fmaxf(best, d);
                            best_i = gt ? kk : best_i;
                        }
                    }
                    cb = filter_cb(fmaxf(better, seed));
                }
            }
        };
        for (int k2 = k_lo; k2 < k_hi; k2 += AHEAD * kTile) {          // block-uniform trip tests
            tile_step(std::integral_constant<int, 0>{}, k2);
            if (AHEAD > 1 && k2 + kTile < k_hi) tile_step(std::integral_constant<int, (AHEAD > 1 ? 1 : 0)>{}, k2 + kTile);
            if (AHEAD > 2 && k2 + 2 * kTile < k_hi) tile_step(std::integral_constant<int, (AHEAD > 2 ? 2 : 0)>{}, k2 + 2 * kTile);
            if (AHEAD > 3 && k2 + 3 * kTile < k_hi) tile_step(std::integral_constant<int, (AHEAD > 3 ? 3 : 0)>{}, k2 + 3 * kTile);
        }
        // merge the P partial top-2s of a bidder (value-symmetric)
        for (int off = 1; off < P; off <<= 1) {
            const float ob = __shfl_xor(best, off, kWave);
            const float obb = __shfl_xor(better, off, kWave);
            const int oi = __shfl_xor(best_i, off, kWave);
            const int obi = __shfl_xor(better_i, off, kWave);
            merge_top2(best, better, best_i, better_i, ob, obb, oi, obi);
        }
        if (Z > 1) {
            // publish this slice's top-2 per bidder, then let the last slice to arrive
            // fold all Z of them (top-2 merging is associative and symmetric)
            // Partials are two 8-byte words per (bidder, slice), stored and loaded with
            // agent-scope atomics (write-through stores, cache-bypassing loads): no release /
            // acquire fence is needed around the ticket -- a release would write back every
            // dirty line of the XCD's L2 (measured: late rounds 20 -> 15 us).
            unsigned long long *slot =
                (unsigned long long *)(parts + ((size_t)batch * kSplitMaxBidders + (active ? u : 0)) * kZMax);
            if (active && p == 0) {
                __hip_atomic_store(slot + 2 * zs, ((unsigned long long)__float_as_uint(best) << 32) | __float_as_uint(better),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(slot + 2 * zs + 1, ((unsigned long long)(unsigned)best_i << 32) | (unsigned)better_i,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int *cntp = arrive + (size_t)batch * kArrivePerBatch + grp;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                       // also: every lane is done with the tile
            int *s_ticket = (int *)sX;
            if (threadIdx.x == 0)
                *s_ticket = __hip_atomic_fetch_add(cntp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int ticket = *s_ticket;
            __syncthreads();                       // the ticket slot is tile memory: read before reuse
            if (ticket != Z - 1) continue;         // block-uniform
            if (threadIdx.x == 0) __hip_atomic_store(cntp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float mb = -1e9f, mbb = -1e9f;
            int mi = -1, mbi = -1;
            if (active && p == 0) {
                for (int q = 0; q < Z; q++) {
                    const unsigned long long w0 = __hip_atomic_load(slot + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long w1 = __hip_atomic_load(slot + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    merge_top2(mb, mbb, mi, mbi, __uint_as_float((unsigned)(w0 >> 32)), __uint_as_float((unsigned)w0),
                               (int)(unsigned)(w1 >> 32), (int)(unsigned)w1);
                }
            }
            // hand the merged result to all P lanes of the bidder (the tie path below is cooperative)
            const int src_lane = lane & ~(P - 1);
            best = __shfl(mb, src_lane, kWave);
            better = __shfl(mbb, src_lane, kWave);
            best_i = __shfl(mi, src_lane, kWave);
            better_i = __shfl(mbi, src_lane, kWave);
        }
        // exact tie for first place: pick the candidate the reference's scan meets first
        const bool tie = active && (best == better);
        if (__any(tie)) {
            unsigned long long key = ~0ull;
            if (tie) {
                for (int k = p; k < n; k += P) {
                    const float d = bid_value<FMA>(x1, y1, z1, X2[(size_t)k * 3 + 0], X2[(size_t)k * 3 + 1],
                                                   X2[(size_t)k * 3 + 2], PR[k]);
                    if (d == best) {
                        const int kt = k & 2047;                       // position in the reference's 2048-tile
                        const int tile0 = k - kt;
                        const int end_k = min(n, tile0 + 2048) - tile0;
                        const int delta = (end_k + thread_per_unass - 1) / thread_per_unass;
                        const unsigned long long kk = ((unsigned long long)(kt / delta) << 32) | (unsigned)k;
                        key = kk < key ? kk : key;
                    }
                }
            }
            for (int off = 1; off < P; off <<= 1) {
                const unsigned long long o = __shfl_xor(key, off, kWave);
                key = o < key ? o : key;
            }
            if (tie) best_i = (int)(key & 0xffffffffu);
        }
        if (active && p == 0) {
            const float inc = __fadd_rn(__fsub_rn(best, better), eps);
            bid[(size_t)batch * n + j] = best_i;
            if (second != nullptr) second[(size_t)batch * n + j] = better_i;
            bid_increments[(size_t)batch * n + j] = inc;
            atomic_max_float(&max_increments[(size_t)batch * n + best_i], inc);
            // the round's bidders of an object as a chain through the object's head word (emd_settle_kernel): a record is
            // (increment bits << 32) | (stamp << 24) | bidder, the head holds the latest bidder's, next[j] what j displaced.
            // The bidder field is 24 bits: the chain is only built for n <= 2^24 (`settle` in genpc_emd_forward)
            if (chain_head != nullptr) {
                const unsigned long long mine = ((unsigned long long)(unsigned)__float_as_int(inc) << 32) | (stamp << 24) | (unsigned)j;
                chain_next[(size_t)batch * n + j] = atomicExch(&chain_head[(size_t)batch * n + best_i], mine);
                atomicAdd(&chain_cnt[(size_t)batch * n + best_i], 1);       // how many bid for it this round (no value returned: nothing waits)
            }
        }
    }
}

// emd_cuda.cu:181-194.  In the forced last round every bidder is "assigned", and
// the owner recorded in assignment_inv is the last writer: elect the highest j.
__global__ __launch_bounds__(kEBlock) void emd_getmax_kernel(int n, const int *__restrict__ list,
                                                             const int *__restrict__ cnt, const int *__restrict__ bid,
                                                             const float *__restrict__ bid_increments,
                                                             const float *__restrict__ max_increments,
                                                             int *__restrict__ max_idx, int last)
{
    const int batch = blockIdx.y;
    const int U = cnt[batch];
    for (int u = blockIdx.x * kEBlock + threadIdx.x; u < U; u += gridDim.x * kEBlock) {
        const int j = list[(size_t)batch * n + u];
        const int bid_id = bid[(size_t)batch * n + j];
        const double bid_inc = (double)bid_increments[(size_t)batch * n + j];
        const double max_inc = (double)max_increments[(size_t)batch * n + bid_id];
        if (last || (bid_inc - 1e-6 <= max_inc && max_inc <= bid_inc + 1e-6))
            atomicMax(&max_idx[(size_t)batch * n + bid_id], j);
    }
}

// emd_cuda.cu:196-215 plus construction of the next round's bidder list.
__global__ __launch_bounds__(kEBlock) void emd_assign_kernel(int n, const int *__restrict__ list,
                                                             const int *__restrict__ cnt, int *__restrict__ list_next,
                                                             int *__restrict__ cnt_next, int *__restrict__ assignment,
                                                             int *__restrict__ assignment_inv, float *__restrict__ price,
                                                             const int *__restrict__ bid,
                                                             const float *__restrict__ bid_increments,
                                                             float *__restrict__ max_increments,
                                                             int *__restrict__ max_idx, int last,
                                                             const int *__restrict__ pos_of, float *__restrict__ price_s)
{
    const int batch = blockIdx.y;
    const int U = cnt[batch];
    const size_t base = (size_t)batch * n;
    for (int u = blockIdx.x * kEBlock + threadIdx.x; u < U; u += gridDim.x * kEBlock) {
        const int j = list[base + u];
        const int bid_id = bid[base + j];
        const bool elected = max_idx[base + bid_id] == j;
        if (last) {
            // every remaining bidder takes its object (not a bijection, :201)
            assignment[base + j] = bid_id;
            atomicAdd(&price[base + bid_id], bid_increments[base + j]);
            if (elected) {
                assignment_inv[base + bid_id] = j;
                max_increments[base + bid_id] = -1e9f;
                max_idx[base + bid_id] = -1;
            }
        } else if (elected) {
            const int prev = assignment_inv[base + bid_id];
            if (prev != -1) {
                assignment[base + prev] = -1;
                const int pos = atomicAdd(&cnt_next[batch], 1);
                list_next[base + pos] = prev;
            }
            assignment_inv[base + bid_id] = j;
            assignment[base + j] = bid_id;
            const float np_ = __fadd_rn(price[base + bid_id], bid_increments[base + j]);
            price[base + bid_id] = np_;
            if (price_s) price_s[4 * (base + pos_of[base + bid_id])] = np_;
            max_increments[base + bid_id] = -1e9f;
            // elections are per round; only this thread's own comparison above
            // needed the value, every other bidder of the object compares != j
            max_idx[base + bid_id] = -1;
        } else {
            const int pos = atomicAdd(&cnt_next[batch], 1);
            list_next[base + pos] = j;
        }
    }
}

// GetMax and Assign of one round in one MULTI-block launch (emd_cuda.cu:181-215).  GetMax needs every bid of the round (a
// grid-wide dependency, hence the reference's two launches); here every bidder walks the chain of its object's bidders that
// the bid kernel left (chain_head / chain_next) and finds the winner itself -- the bidder with the highest index among those
// whose increment lies within 1e-6 of the largest (all of them in the forced last round) -- so all bidders of an object
// agree without exchanging anything, and the winner does Assign's work.  The largest increment of the chain IS
// max_increments[object] (increments are >= 0 when eps >= 0, the only case this kernel is launched for; the winner resets
// that word, so it is not read here).  A chain of one -- the usual case once few bidders are left -- is read off the head
// word alone: the launch is as deep as Assign was, and GetMax's 4.7 us per round are gone.
// LONG chains (round 4).  With L bidders on one object the walks cost L^2 dependent loads: a partial scan whose ground truth
// is mis-framed (bundled scan 06830: thousands of points bid for the same few boundary objects, every round) took 79 ms
// where its neighbours take 2-4, and the first rounds of 13 scans 2 ms each.  The bid kernel therefore also counts the
// bidders of an object (chain_cnt), and an object with more than kChainWalkMax of them is settled without any walk:
// every bidder tests itself against max_increments[object] (complete: the bid kernel is over; nobody resets it before the
// end), a bidder outside the window is a loser and re-lists itself at once, one inside raises max_idx (atomicMax) and links
// itself into a second, short chain of the in-window bidders; then it draws a ticket.  The bidder with the LAST ticket
// knows every other one is done: it reads the winner, re-lists the other in-window bidders, does Assign's work for the
// winner and hands the four words back clean.  O(1) per bidder, nobody waits for anybody, one launch.
constexpr int kChainWalkMax = 4;
__global__ __launch_bounds__(kEBlock) void emd_settle_kernel(int n, const int *__restrict__ list, const int *__restrict__ cnt,
                                                             int *__restrict__ list_next, int *__restrict__ cnt_next,
                                                             int *__restrict__ assignment, int *__restrict__ assignment_inv,
                                                             float *__restrict__ price, const int *__restrict__ bid,
                                                             const float *__restrict__ bid_increments,
                                                             float *__restrict__ max_increments, int *__restrict__ max_idx,
                                                             const unsigned long long *__restrict__ chain_head,
                                                             const unsigned long long *__restrict__ chain_next, unsigned stamp,
                                                             int last, const int *__restrict__ pos_of, float *__restrict__ price_s,
                                                             int *__restrict__ chain_cnt, int *__restrict__ arrived,
                                                             unsigned long long *__restrict__ whead, unsigned long long *__restrict__ wnext)
{
    const int batch = blockIdx.y;
    const int U = cnt[batch];
    const size_t base = (size_t)batch * n;
    auto live = [&](unsigned long long r) { return (unsigned)((r >> 24) & 0xffu) == stamp; };
    auto who = [](unsigned long long r) { return (int)(r & 0xffffffu); };
    auto inc_of = [](unsigned long long r) { return __int_as_float((int)(r >> 32)); };
    auto relist = [&](int jj) {
        const int pos = atomicAdd(&cnt_next[batch], 1);
        list_next[base + pos] = jj;
    };
    // Assign's work for the winner w of object o (not the forced last round); prev / old price: the object's owner and
    // price, requested by every bidder beside the chain head (a winner's path would otherwise be two round trips longer)
    auto take = [&](int o, int w, float inc_w, int pos_s, int prev, float old_price) {
        if (prev != -1) {
            assignment[base + prev] = -1;
            relist(prev);
        }
        assignment_inv[base + o] = w;
        assignment[base + w] = o;
        const float np_ = __fadd_rn(old_price, inc_w);
        price[base + o] = np_;
        if (price_s) price_s[4 * (base + pos_s)] = np_;
    };
    for (int u = blockIdx.x * kEBlock + threadIdx.x; u < U; u += gridDim.x * kEBlock) {
        const int j = list[base + u];
        const int bid_id = bid[base + j];
        const unsigned long long mynext = chain_next[base + j];      // (what j displaced: read beside bid[j], not behind the head)
        const unsigned long long head = chain_head[base + bid_id];
        const int C = chain_cnt[base + bid_id];                      // (0 once a short chain's winner has cleaned up: same path)
        const int pos = pos_of ? pos_of[base + bid_id] : 0;          // the object's place in the cell-sorted copy (emd_grid.hip)
        const int owner = assignment_inv[base + bid_id];             // (only this object's winner writes these two, and only at the end)
        const float old_price = price[base + bid_id];
        if (C > kChainWalkMax) {
            const float my_inc = bid_increments[base + j];
            const double bid_inc = (double)my_inc, max_inc = (double)max_increments[base + bid_id];
            const bool inwin = last || (bid_inc - 1e-6 <= max_inc && max_inc <= bid_inc + 1e-6);
            if (last) {
                assignment[base + j] = bid_id;                       // every remaining bidder takes its object (:201)
                atomicAdd(&price[base + bid_id], my_inc);
            }
            if (inwin) {
                atomicMax(&max_idx[base + bid_id], j);
                if (!last) {
                    const unsigned long long old = atomicExch(&whead[base + bid_id], ((unsigned long long)stamp << 24) | (unsigned)j);
                    __hip_atomic_store(&wnext[base + j], old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                relist(j);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my atomics and the link have landed before my ticket is drawn
            const int ticket = __hip_atomic_fetch_add(&arrived[base + bid_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == C - 1) {
                const int w = __hip_atomic_load(&max_idx[base + bid_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!last) {
                    for (unsigned long long r = __hip_atomic_load(&whead[base + bid_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); live(r);
                         r = __hip_atomic_load(&wnext[base + who(r)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        if (who(r) != w) relist(who(r));
                    take(bid_id, w, bid_increments[base + w], pos, owner, old_price);
                } else {
                    assignment_inv[base + bid_id] = w;
                }
                max_increments[base + bid_id] = -1e9f;
                max_idx[base + bid_id] = -1;
                chain_cnt[base + bid_id] = 0;
                __hip_atomic_store(&arrived[base + bid_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        int winner = who(head);
        float my_inc = inc_of(head);
        if (winner != j || live(mynext)) {
            // several bidders: the largest increment first, then the highest index inside its window
            float mx = -1e9f;
            for (unsigned long long r = head; live(r); r = chain_next[base + who(r)]) {
                const float v = inc_of(r);
                mx = v > mx ? v : mx;
                if (who(r) == j) my_inc = v;
            }
            winner = -1;
            for (unsigned long long r = head; live(r); r = chain_next[base + who(r)]) {
                const double bid_inc = (double)inc_of(r), max_inc = (double)mx;
                if ((last || (bid_inc - 1e-6 <= max_inc && max_inc <= bid_inc + 1e-6)) && who(r) > winner) winner = who(r);
            }
        }
        const bool elected = winner == j;
        if (last) {
            // every remaining bidder takes its object (not a bijection, :201)
            assignment[base + j] = bid_id;
            atomicAdd(&price[base + bid_id], my_inc);
            if (elected) {
                assignment_inv[base + bid_id] = j;
                max_increments[base + bid_id] = -1e9f;
                max_idx[base + bid_id] = -1;
                chain_cnt[base + bid_id] = 0;
            }
        } else if (elected) {
            take(bid_id, j, my_inc, pos, owner, old_price);
            max_increments[base + bid_id] = -1e9f;
            max_idx[base + bid_id] = -1;
            chain_cnt[base + bid_id] = 0;
        } else {
            relist(j);
        }
    }
}

// GetMax and Assign of one round in ONE launch, one 1024-thread block per batch element: used from
// the round on in which few bidders are left (the two list-walking launches are then pure latency:
// 4.7 + 5.1 us plus a launch boundary per round at 1 x 16384).  Phase 1 elects (atomicMax, performed in
// L2), the block barrier separates it from phase 2, which reads the elections with L1-bypassing loads.
constexpr int kResolveBlock = 1024;
__global__ __launch_bounds__(kResolveBlock) void emd_resolve_kernel(int n, const int *__restrict__ list,
                                                                    const int *__restrict__ cnt, int *__restrict__ list_next,
                                                                    int *__restrict__ cnt_next, int *__restrict__ assignment,
                                                                    int *__restrict__ assignment_inv, float *__restrict__ price,
                                                                    const int *__restrict__ bid,
                                                                    const float *__restrict__ bid_increments,
                                                                    float *__restrict__ max_increments,
                                                                    int *__restrict__ max_idx, int last,
                                                                    const int *__restrict__ pos_of, float *__restrict__ price_s)
{
    const int batch = blockIdx.x;
    const int U = cnt[batch];
    const size_t base = (size_t)batch * n;
    for (int u = threadIdx.x; u < U; u += kResolveBlock) {
        const int j = list[base + u];
        const int bid_id = bid[base + j];
        const double bid_inc = (double)bid_increments[base + j];
        const double max_inc = (double)max_increments[base + bid_id];
        if (last || (bid_inc - 1e-6 <= max_inc && max_inc <= bid_inc + 1e-6)) atomicMax(&max_idx[base + bid_id], j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int u = threadIdx.x; u < U; u += kResolveBlock) {
        const int j = list[base + u];
        const int bid_id = bid[base + j];
        const bool elected = __hip_atomic_load(&max_idx[base + bid_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == j;
        if (last) {
            assignment[base + j] = bid_id;
            atomicAdd(&price[base + bid_id], bid_increments[base + j]);
            if (elected) {
                assignment_inv[base + bid_id] = j;
                max_increments[base + bid_id] = -1e9f;
                max_idx[base + bid_id] = -1;
            }
        } else if (elected) {
            const int prev = assignment_inv[base + bid_id];
            if (prev != -1) {
                assignment[base + prev] = -1;
                const int pos = atomicAdd(&cnt_next[batch], 1);
                list_next[base + pos] = prev;
            }
            assignment_inv[base + bid_id] = j;
            assignment[base + j] = bid_id;
            const float np_ = __fadd_rn(price[base + bid_id], bid_increments[base + j]);
            price[base + bid_id] = np_;
            if (price_s) price_s[4 * (base + pos_of[base + bid_id])] = np_;
            max_increments[base + bid_id] = -1e9f;
            max_idx[base + bid_id] = -1;
        } else {
            const int pos = atomicAdd(&cnt_next[batch], 1);
            list_next[base + pos] = j;
        }
    }
}

// emd_cuda.cu:217-226
template <int FMA>
__global__ __launch_bounds__(kEBlock) void emd_calc_dist_kernel(long long total, int n, const float *__restrict__ xyz1,
                                                                const float *__restrict__ xyz2,
                                                                float *__restrict__ dist,
                                                                const int *__restrict__ assignment)
{
    const long long t = (long long)blockIdx.x * kEBlock + threadIdx.x;
    if (t >= total) return;
    const long long i = t / n;
    const int k = assignment[t];
    if (k < 0) {            // iters == 0: the reference would read out of bounds
        dist[t] = 0.0f;
        return;
    }
    const float *p1 = xyz1 + t * 3;
    const float *p2 = xyz2 + (i * n + k) * 3;
    dist[t] = sqdist_e<FMA>(p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]);
}

// emd_cuda.cu:284-300
__global__ __launch_bounds__(kEBlock) void emd_grad_kernel(long long total, int n, const float *__restrict__ xyz1,
                                                           const float *__restrict__ xyz2,
                                                           const float *__restrict__ grad_dist,
                                                           const int *__restrict__ idx, float *__restrict__ grad_xyz)
{
    const long long t = (long long)blockIdx.x * kEBlock + threadIdx.x;
    if (t >= total) return;
    const long long i = t / n;
    const float *p1 = xyz1 + t * 3;
    const float *p2 = xyz2 + (i * n + idx[t]) * 3;
    const float g = __fmul_rn(grad_dist[t], 2.0f);
    float *o = grad_xyz + t * 3;
    o[0] = __fadd_rn(o[0], __fmul_rn(g, p1[0] - p2[0]));
    o[1] = __fadd_rn(o[1], __fmul_rn(g, p1[1] - p2[1]));
    o[2] = __fadd_rn(o[2], __fmul_rn(g, p1[2] - p2[2]));
}

}  // namespace genpc

namespace genpc { thread_local int t_emd_grid = -1, t_emd_hooks = 0; }

/* Bid kernel selection for tests and A/B (thread-local like genpc_nn_tune): 1 the cell-sorted culled bid (emd_grid.hip),
 * 0 the tiled bid over all objects (emd_bid_kernel), < 0 the default (culled when eps >= 0 and n >= 4096 or B n >= 65536:
 * a single small cloud is bound by the round's dependent loads, of which the culled bid has more).  Returns
 * the previous setting.  Every choice yields the same bits.  hooks (>= 0 to set): 1 = count what the culled bid does
 * (genpc_emd_stats). */
GENPC_API int genpc_emd_tune(int grid, int hooks)
{
    const int prev = genpc::t_emd_grid;
    genpc::t_emd_grid = grid < 0 ? -1 : (grid > 2 ? 2 : grid);
    if (hooks >= 0) genpc::t_emd_hooks = hooks;
    return prev;
}

GENPC_API int genpc_emd_stats(unsigned long long out[8], int reset, void *stream)
{
    using namespace genpc;
    unsigned long long *dev = (unsigned long long *)workspace(28, 256, nullptr, nullptr, 256);
    if (!dev) return 0;
    if (!check(hipStreamSynchronize((hipStream_t)stream), "genpc_emd_stats sync")) return 0;
    if (!check(hipMemcpy(out, dev, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost), "genpc_emd_stats copy")) return 0;
    if (reset && !check(hipMemset(dev, 0, 256), "genpc_emd_stats reset")) return 0;
    return 1;
}

GENPC_API int genpc_emd_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                                int *assignment, float *price, int *assignment_inv, int *bid,
                                float *bid_increments, float *max_increments, int *unass_idx, int *unass_cnt,
                                int *unass_cnt_sum, int *cnt_tmp, int *max_idx, float eps, int iters, void *stream)
{
    using namespace genpc;
    (void)unass_cnt_sum;
    if (n != m) {                      // emd_cuda.cu:236-239
        fprintf(stderr, "Input Error! The two point clouds should have the same size.\n");
        return -1;
    }
    if (b > 512) {                     // :241-244
        fprintf(stderr, "Input Error! The batch size should be less than 512.\n");
        return -1;
    }
    if (n % 256 != 0) {                // :246-249
        fprintf(stderr, "Input Error! The size of the point clouds should be a multiple of 256.\n");
        return -1;
    }
    if (b <= 0 || n <= 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)b * n;
    if (total > 0x7fffffffLL) {
        set_error("emd: B*n too large");
        return 0;
    }
    // all rounds in ONE launch whose threads own the points (emd_auction.hip): whenever the whole launch can be resident
    {
        static const int env_auction = tune_env("GENPC_EMD_AUCTION", -1, "EMD: 1 all rounds in one launch (threads own the points) / 0 a launch per round step (-1 = pick)");
        static const bool noseed_a = tune_env("GENPC_EMD_NOSEED", 0, "EMD: 1 = no seeds from the previous bid (tiled bid only; disables the culled bid)") != 0;
        // (default: from 8192 points per call on -- measured in one process against the launch-per-round path, 50 rounds:
        // 13 x 16384 2.50 -> 1.67 ms, 4 x 16384 1.30 -> 1.14, 64 x 2048 1.52 -> 1.33, 4 x 4096 0.88 -> 0.82, 1 x 16384 0.995 ->
        // 0.96, 1 x 8192 0.90 -> 0.89; a single small cloud loses: 1 x 2048 0.72 -> 0.77, 1 x 512 0.62 -> 0.66)
        const bool want = t_emd_grid >= 0 ? t_emd_grid == 2 : (env_auction >= 0 ? env_auction != 0 : total >= 8192);
        if (want && eps >= 0.0f && !noseed_a && !(t_emd_hooks & 1)) {
            const int rc = launch_emd_auction(b, n, xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments,
                                              max_idx, eps, iters, arith_mode() != 0 ? 1 : 0, st, t_emd_grid == 2);
            if (rc >= 0) return rc;
        }
    }
    // [second bidder list | arrival counters (zeroed on allocation, restored by the
    //  merging block) | per-slice partial top-2s]
    const size_t list_bytes = ((size_t)total * sizeof(int) + 255) / 256 * 256;
    // fixed size (32 batch elements) so that the zeroed prefix never moves between calls
    const size_t arrive_bytes = (size_t)32 * kArrivePerBatch * sizeof(int);
    // the late-round object split only matters when few batch elements are in flight
    const bool want_split = b <= 32;
    const size_t parts_bytes = want_split ? (size_t)b * kSplitMaxBidders * kZMax * sizeof(float4) : 0;
    const size_t second_bytes = ((size_t)total * sizeof(int) + 255) / 256 * 256;
    // bidder chains per object (emd_settle_kernel): head word per object, link word per bidder
    static const bool no_settle = tune_env("GENPC_EMD_SETTLE", 1, "EMD: 0 = GetMax and Assign as two launches instead of the one-launch settle") == 0;
    const bool settle = !no_settle && eps >= 0.0f && n <= (1 << 24);
    // chain_head | chain_next | whead | wnext (8-byte words per object / bidder), chain_cnt | arrived (ints per object)
    const size_t chain_bytes = settle ? (4 * (size_t)total * sizeof(unsigned long long) + 2 * (size_t)total * sizeof(int) + 255) / 256 * 256 : 0;
    // cell-sorted copy of the objects for the culled bid (emd_grid.hip): needs prices >= 0 (eps >= 0) and the seeds
    static const int env_grid = tune_env("GENPC_EMD_GRID", -1, "EMD: 1 culled bid / 0 tiled bid whatever the size (-1 = pick)");
    static const bool noseed_env = tune_env("GENPC_EMD_NOSEED", 0, "EMD: 1 = no seeds from the previous bid (tiled bid only; disables the culled bid)") != 0;
    const bool grid = (t_emd_grid >= 0 ? t_emd_grid != 0 : (env_grid >= 0 ? env_grid != 0 : (n >= 4096 || (long long)b * n >= 65536))) && eps >= 0.0f && !noseed_env;
    auto al256 = [](size_t v) { return (v + 255) / 256 * 256; };
    const int cells_max = kEGMaxCells;
    const size_t g_hdr = grid ? al256((size_t)b * sizeof(EGridHdr)) : 0, g_start = grid ? al256((size_t)b * (cells_max + 1) * sizeof(int)) : 0;
    const size_t g_sorted = grid ? al256((size_t)total * sizeof(float4)) : 0, g_pos = grid ? al256((size_t)total * sizeof(int)) : 0;
    const size_t g_ps = grid ? al256((size_t)total * sizeof(float)) : 0;
    // per-cell lower bounds of the prices (emd_grid.hip: the bid culls cell by cell with them), refreshed in front of a round's bid
    static const int env_cc = tune_env("GENPC_EMD_CELLCULL", 2, "culled EMD bid: refresh the cells' smallest prices every this many rounds and cull cells by them (0 = rows are culled by distance only)");
    const size_t g_pm = grid && env_cc > 0 ? al256((size_t)b * (cells_max + 1) * sizeof(float)) : 0;
    const size_t grid_off = arrive_bytes + list_bytes + second_bytes + parts_bytes + chain_bytes;
    char *ws = (char *)workspace(1, grid_off + g_hdr + g_start + g_sorted + g_pos + g_ps + g_pm, st, nullptr, arrive_bytes);
    if (!ws) return 0;
    EGridHdr *g_hdr_p = (EGridHdr *)(ws + grid_off);
    int *g_start_p = (int *)(ws + grid_off + g_hdr);
    float4 *g_sorted_p = (float4 *)(ws + grid_off + g_hdr + g_start);
    int *g_pos_p = grid ? (int *)(ws + grid_off + g_hdr + g_start + g_sorted) : nullptr;
    int *g_of_p = grid ? (int *)(ws + grid_off + g_hdr + g_start + g_sorted + g_pos) : nullptr;
    float *g_pm_p = g_pm ? (float *)(ws + grid_off + g_hdr + g_start + g_sorted + g_pos + g_ps) : nullptr;
    float *g_ps_p = grid ? (float *)g_sorted_p + 3 : nullptr;      // the price of sorted position p: g_ps_p[4 p] (the .w of its entry)
    unsigned long long *chain_head = settle ? (unsigned long long *)(ws + arrive_bytes + list_bytes + second_bytes + parts_bytes) : nullptr;
    unsigned long long *chain_next = settle ? chain_head + total : nullptr;
    unsigned long long *whead = settle ? chain_head + 2 * total : nullptr, *wnext = settle ? chain_head + 3 * total : nullptr;
    int *chain_cnt = settle ? (int *)(chain_head + 4 * total) : nullptr, *arrived = settle ? chain_cnt + total : nullptr;
    // (bidder counts and tickets are zero between rounds by construction -- the settling thread hands them back --; they
    // are cleared once per call with everything else, by emd_init_kernel)
    int *arrive = (int *)ws;
    int *list_b = (int *)(ws + arrive_bytes);
    int *second = (int *)(ws + arrive_bytes + list_bytes);
    float4 *parts = (float4 *)(ws + arrive_bytes + list_bytes + second_bytes);
    // second-best object of each point's last bid (-1: has not bid yet; set by emd_init_kernel)
    static const bool noseed = tune_env("GENPC_EMD_NOSEED", 0, "EMD: 1 = no seeds from the previous bid (tiled bid only; disables the culled bid)") != 0;
    if (noseed) second = nullptr;
    static const bool nosplit = tune_env("GENPC_EMD_NOSPLIT", 0, "tiled EMD bid: 1 = no object slices in late rounds") != 0;
    if (nosplit || !want_split) parts = nullptr;
    int *lists[2] = {unass_idx, list_b};
    int *cnts[2] = {unass_cnt, cnt_tmp};
    const bool fma = arith_mode() != 0;

    const int lin_blocks = ceil_div((int)total, kEBlock);
    hipLaunchKernelGGL(emd_init_kernel, dim3(lin_blocks), dim3(kEBlock), 0, st, b, n, lists[0], cnts[0], cnts[1], second, chain_head, whead,
                       chain_cnt, arrived);
    // most points of this shape's last call kept bidding (the feedback word, emd_auction.hip)
    bool heavy = false;
    if (const volatile int *fb = emd_feedback_slot(b, n, false)) heavy = (long long)*fb * 100 > (long long)n * 35;
    if (grid) {
        // about two objects per cell if the cloud filled its box (surfaces fill far fewer cells, with more objects each)
        static const int env_ppc = tune_env("GENPC_EMD_GRID_PPC_X10", 20, "culled EMD bid: target objects per cell x 10");
        // ... twice that where most points keep bidding (the feedback word of this shape's last call, emd_auction.hip: a partial scan
        // against its ground truth): the balls hold hundreds of objects there and a row of coarser cells is one run instead of
        // several (13 bundled scans 18.3 -> 17.4 ms; 80: 18.0, 10: 19.8)
        static const int env_ppc_heavy = tune_env("GENPC_EMD_GRID_PPC_HEAVY_X10", 40, "culled EMD bid: target objects per cell x 10 for clouds whose last call kept more than a third of their points bidding");
        int ppc = env_ppc > 0 ? env_ppc : 20;
        if (heavy && env_ppc_heavy > 0) ppc = env_ppc_heavy;
        int target = (int)((long long)n * 10 / ppc);
        target = target < 8 ? 8 : (target > cells_max * 3 / 4 ? cells_max * 3 / 4 : target);
        if (!launch_emd_grid_build(b, n, xyz2, price, g_hdr_p, g_start_p, g_sorted_p, g_pos_p, g_of_p, target, cells_max, st)) return 0;
    }

    // Blocks per batch element for the bid kernel: ~16 blocks per CU overall (the bid
    // loop is latency-bound per wave -- LDS read, compare, branch -- and wants >= 8
    // waves per SIMD: measured 13x16384, round 0: 3.9 ms at 4/CU, 1.8 ms at 16/CU),
    // never more than the finest split (64 lanes per bidder, all n bidding).
    // (heavy clouds -- thousands of bidders per cloud in every round: 36 per CU; 13 bundled scans, blocks per cloud 316 / 512 /
    //  640 / 768 / 1024: 17.4 / 16.9 / 16.7 / 16.7 / 17.1 ms)
    int G = ceil_div(num_cus() * (heavy && grid ? 36 : 16), b);
    if (G > 1024) G = 1024;     // a single cloud: more blocks only add dispatch + hand-off latency (measured)
    const int g_max = ceil_div(n * 64, kEBlock);
    if (G > g_max) G = g_max;
    if (G < 1) G = 1;
    static const int env_g = tune_env("GENPC_EMD_G", 0, "EMD: bid blocks per cloud (0 = pick)");
    if (env_g > 0) G = env_g;
    // rounds from which GetMax + Assign run as one single-block launch per cloud (few bidders left:
    // ~n/7 after four rounds).  The forced last round takes the same kernel: its bidders are that round's
    // unassigned points like any other round's (every one of them is assigned, none evicted) -- the numbers below
    // were measured that way
    // (measured, 50 rounds: 1 x 2048 0.75 -> 0.68 ms, 64 x 2048 2.18 -> 1.98, 13 x 16384 8.45 -> 8.19; but
    // 1 x 16384 1.61 -> 1.80: with ~1000-2000 bidders left per round one block walking the list is
    // slower than 64 -- so only for small clouds or many of them)
    static const int env_rf = tune_env("GENPC_EMD_RESOLVE_FROM", -1, "EMD: round from which a single block per cloud resolves (-1 = pick)");
    // (the single block also where most points keep bidding -- a partial scan against its ground truth, thousands of bidders
    // per cloud in every round: measured in one process on the 13 bundled scans, 19.1 ms against 20.9 with the multi-block
    // settle, whose per-object tickets and chain walks cost more there than one block's two passes)
    const int resolve_from = env_rf >= 0 ? env_rf : ((n <= 4096 || b >= 8) ? 4 : 0x7fffffff);
    int GL = ceil_div(n, kEBlock);          // list-walking kernels
    if (GL > 64) GL = 64;

    for (int it = 0; it < iters; it++) {
        const int cur = it & 1, nxt = cur ^ 1;
        const int last = (it == iters - 1);
        // rounds that took the two list-walking launches take the settle kernel (the single-block resolve keeps its rounds)
        const bool use_chain = settle && it < resolve_from;
        const unsigned stamp = (unsigned)(it % 255) + 1u;      // 8 bits in a record: the heads are cleared every 255 rounds
        if (use_chain && it > 0 && it % 255 == 0 &&
            (!check(hipMemsetAsync(chain_head, 0, (size_t)total * sizeof(unsigned long long), st), "hipMemsetAsync(chain heads)") ||
             !check(hipMemsetAsync(whead, 0, (size_t)total * sizeof(unsigned long long), st), "hipMemsetAsync(window heads)")))
            return 0;
        if (grid) {
            EmdGridBid ga{};
            ga.n = n; ga.G = G; ga.nb = b; ga.cells_max = cells_max; ga.eps = eps; ga.stamp = stamp;
            static const int env_lpb = tune_env("GENPC_EMD_LPB", 0, "culled EMD bid: lanes per bidder (8..64, 0 = pick)");
            ga.force_lpb = env_lpb > 0 ? env_lpb : (heavy ? 16 : 0);      // (thousands of bidders per cloud in every round: sixteen lanes each -- 13 bundled scans 16.7 -> 16.0 ms; 8: 17.4)
            ga.lpb_max = 0;
            static const int env_xcd = tune_env("GENPC_EMD_XCD", 0, "culled EMD bid: 1 = a cloud's blocks on one XCD (the tiled bid's order), 0 = clouds interleaved");
            ga.xcd_pin = env_xcd;
            ga.xyz1 = xyz1; ga.xyz2 = xyz2; ga.price = price; ga.orig_of = g_of_p;
            ga.list = lists[cur]; ga.cnt = cnts[cur]; ga.start = g_start_p; ga.cnt_next = cnts[nxt];
            ga.bid = bid; ga.second = second; ga.bid_increments = bid_increments; ga.max_increments = max_increments;
            ga.sorted = g_sorted_p; ga.hdr = g_hdr_p;
            // (round 0: every price is the caller's initial one -- zero in the reference's use --, nothing to cull by)
            if (g_pm_p != nullptr && it >= 1) {
                if ((it - 1) % env_cc == 0 && !launch_emd_cell_pmin(b, cells_max, g_hdr_p, g_start_p, g_sorted_p, n, g_pm_p, st)) return 0;
                ga.cell_pmin = g_pm_p;
            }
            ga.chain_head = use_chain ? chain_head : nullptr; ga.chain_next = chain_next; ga.chain_cnt = chain_cnt;
            ga.feedback = it == 3 ? emd_feedback_slot(b, n, true) : nullptr;
            ga.stats = (t_emd_hooks & 1) ? (unsigned long long *)workspace(28, 256, nullptr, nullptr, 256) : nullptr;
            launch_emd_bid_grid(ga, fma ? 1 : 0, st);
        } else {
            typedef void (*bid_fn)(int, const float *, const float *, const float *, float, const int *, const int *,
                                   int *, int *, float *, float *, int, float4 *, int *, int *, int, unsigned long long *,
                                   unsigned long long *, unsigned, int, int, int *);
            static const int zmax_env = tune_env("GENPC_EMD_ZMAX", kZMax, "tiled EMD bid: object slices per bidder group in late rounds (1..4)");
            const int zmax = zmax_env < 1 ? 1 : (zmax_env > kZMax ? kZMax : zmax_env);
            // Lanes per bidder.  Round 0 has no filter seeds: the fewer lanes share a bidder, the sooner a
            // lane's own second-best makes the filter selective (16 lanes: 186 us, 64: 298 us at
            // n = 16384).  From round 1 on the seeds do that, and one bidder per wave (64 lanes)
            // keeps one bidder's rare exact evaluations from stalling another's lanes (round 1:
            // 95 us against 166 us at 32 lanes), even when that needs more units than blocks.  With
            // many clouds in flight (b >= 32) fewer bidders per staged tile cost more than that (+2 %).
            static const int env_p = tune_env("GENPC_EMD_P", 0, "tiled EMD bid: lanes per bidder (0 = pick)");
            static const int env_p0 = tune_env("GENPC_EMD_P0", 0, "tiled EMD bid: lanes per bidder in round 0 (0 = pick)");
            const int force_p = env_p > 0 ? env_p : (it == 0 ? env_p0 : (second != nullptr && b < 32 ? 64 : 0));
            static const bool nofilter = tune_env("GENPC_EMD_NOFILTER", 0, "tiled EMD bid: 1 = no squared-distance pre-filter (A/B)") != 0;
            // 1024-object tiles (16 KiB of LDS; the default filter variant is 78 VGPRs = six waves per SIMD, the unfiltered one
            // 72 = eight: tools/kmeta.sh emd) wherever
            // the bid is throughput-bound; a single small cloud is latency-bound and pays for the extra barrier pairs
            // (in-run A/B: 13 x 16384 8.28 -> 7.65 ms, 64 x 2048 1.99 -> 1.80, 1 x 16384 =, 1 x 2048 0.69 -> 0.76;
            // 512-object tiles: 7.85 / 1.84 / 1.69 / 0.70)
            static const int env_tile = tune_env("GENPC_EMD_TILE", 0, "tiled EMD bid: objects per LDS tile (1024 | 2048, 0 = pick)");
            const bool small_tile = env_tile ? env_tile == 1024 : (long long)b * n > 8192;
            // four tiles in flight from round 2 on for a single cloud (see the kernel)
            static const int env_ahead = tune_env("GENPC_EMD_AHEAD", 0, "tiled EMD bid: object tiles in flight (1 | 4, 0 = pick)");
            const bool deep = env_ahead ? env_ahead == 4 : (small_tile && it >= 2 && b == 1);      // in-run A/B: 1 x 8192 1.22 -> 1.07 ms, 1 x 16384 -2 %, 2 x 16384 +7 %, 4 x 16384 +15 %
            bid_fn f = deep ? (fma ? (nofilter ? emd_bid_kernel<1, 0, 1024, 4> : emd_bid_kernel<1, 1, 1024, 4>)
                                   : (nofilter ? emd_bid_kernel<0, 0, 1024, 4> : emd_bid_kernel<0, 1, 1024, 4>))
                     : small_tile ? (fma ? (nofilter ? emd_bid_kernel<1, 0, 1024, 1> : emd_bid_kernel<1, 1, 1024, 1>)
                                         : (nofilter ? emd_bid_kernel<0, 0, 1024, 1> : emd_bid_kernel<0, 1, 1024, 1>))
                                  : (fma ? (nofilter ? emd_bid_kernel<1, 0, 2048, 1> : emd_bid_kernel<1, 1, 2048, 1>)
                                         : (nofilter ? emd_bid_kernel<0, 0, 2048, 1> : emd_bid_kernel<0, 1, 2048, 1>));
            hipLaunchKernelGGL(f, dim3(G * b), dim3(kEBlock), 0, st, n, xyz1, xyz2, (const float *)price, eps,
                               (const int *)lists[cur], (const int *)cnts[cur], cnts[nxt], bid, bid_increments,
                               max_increments, force_p, parts, arrive, second, zmax, use_chain ? chain_head : (unsigned long long *)nullptr,
                               chain_next, stamp, G, b, chain_cnt);
        }
        if (use_chain) {
            hipLaunchKernelGGL(emd_settle_kernel, dim3(GL, b), dim3(kEBlock), 0, st, n, (const int *)lists[cur],
                               (const int *)cnts[cur], lists[nxt], cnts[nxt], assignment, assignment_inv, price,
                               (const int *)bid, (const float *)bid_increments, max_increments, max_idx,
                               (const unsigned long long *)chain_head, (const unsigned long long *)chain_next, stamp, last,
                               (const int *)g_pos_p, g_ps_p, chain_cnt, arrived, whead, wnext);
        } else if (it >= resolve_from) {
            hipLaunchKernelGGL(emd_resolve_kernel, dim3(b), dim3(kResolveBlock), 0, st, n, (const int *)lists[cur],
                               (const int *)cnts[cur], lists[nxt], cnts[nxt], assignment, assignment_inv, price,
                               (const int *)bid, (const float *)bid_increments, max_increments, max_idx, last, (const int *)g_pos_p, g_ps_p);
        } else {
            hipLaunchKernelGGL(emd_getmax_kernel, dim3(GL, b), dim3(kEBlock), 0, st, n, (const int *)lists[cur],
                               (const int *)cnts[cur], (const int *)bid, (const float *)bid_increments,
                               (const float *)max_increments, max_idx, last);
            hipLaunchKernelGGL(emd_assign_kernel, dim3(GL, b), dim3(kEBlock), 0, st, n, (const int *)lists[cur],
                               (const int *)cnts[cur], lists[nxt], cnts[nxt], assignment, assignment_inv, price,
                               (const int *)bid, (const float *)bid_increments, max_increments, max_idx, last, (const int *)g_pos_p, g_ps_p);
        }
    }
    if (fma)
        hipLaunchKernelGGL((emd_calc_dist_kernel<1>), dim3(lin_blocks), dim3(kEBlock), 0, st, total, n, xyz1, xyz2, dist,
                           (const int *)assignment);
    else
        hipLaunchKernelGGL((emd_calc_dist_kernel<0>), dim3(lin_blocks), dim3(kEBlock), 0, st, total, n, xyz1, xyz2, dist,
                           (const int *)assignment);
    return check(hipGetLastError(), "emd forward launch") ? 1 : 0;
}

// Diagnostic (not part of the public header): resident blocks per CU the runtime
// reports for the bid kernel.
extern "C" __attribute__((visibility("default"))) int genpc_debug_emd_bid_occupancy(void)
{
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, genpc::emd_bid_kernel<1, 1, 1024, 1>, genpc::kEBlock, 0);
    return nb;
}

/* CalcDist alone (emd_cuda.cu:217-226): dist[B,n] = |xyz1[j] - xyz2[assignment[j]]|^2 for a caller that holds an
 * assignment (the one-launch forward folds this step into its last round; this entry is what bench.py measures row a8 with). */
GENPC_API int genpc_emd_calc_dist(int b, int n, const float *xyz1, const float *xyz2, const int *assignment, float *dist, void *stream)
{
    using namespace genpc;
    if (b <= 0 || n <= 0) return 1;
    const long long total = (long long)b * n;
    const unsigned blocks = (unsigned)ceil_div64(total, kEBlock);
    if (arith_mode() != 0)
        hipLaunchKernelGGL((emd_calc_dist_kernel<1>), dim3(blocks), dim3(kEBlock), 0, (hipStream_t)stream, total, n, xyz1, xyz2, dist, assignment);
    else
        hipLaunchKernelGGL((emd_calc_dist_kernel<0>), dim3(blocks), dim3(kEBlock), 0, (hipStream_t)stream, total, n, xyz1, xyz2, dist, assignment);
    return check(hipGetLastError(), "emd_calc_dist_kernel launch") ? 1 : 0;
}

GENPC_API int genpc_emd_backward(int b, int n, const float *xyz1, const float *xyz2, float *gradxyz,
                                 const float *graddist, const int *idx, void *stream)
{
    using namespace genpc;
    if (b <= 0 || n <= 0) return 1;
    const long long total = (long long)b * n;
    hipLaunchKernelGGL(emd_grad_kernel, dim3((unsigned)ceil_div64(total, kEBlock)), dim3(kEBlock), 0,
                       (hipStream_t)stream, total, n, xyz1, xyz2, graddist, idx, gradxyz);
    return check(hipGetLastError(), "emd_grad_kernel launch") ? 1 : 0;
}
