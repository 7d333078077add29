// fps.hip -- farthest point sampling for gfx950 (the metric of
// main.py:21-24 subsamples both clouds to 16384 points before CD/EMD, reg_xyz.py:215
// samples 20000 fused points).  The reference calls fpsample.fps_sampling, a
// third-party CPU extension with a RANDOM start index (not reproducible, unpinned);
// this is the deterministic counterpart used by the fixtures: start index 0, fp32
// squared distances, first arg-max.  Bit-exact with oracle_fps.
//
// FPS is k strictly sequential steps of (update N running minima, arg-max): what a step
// costs is the LATENCY of its dependency chain, not bandwidth (a step touches no memory but
// the hand-off).  Round 3 layout, built around that chain (round 2: 2.8 us per step):
//   * a cloud is owned by W <= 64 workgroups of 192 worker threads; every thread keeps R <= 24 points and their
//     running minima in REGISTERS for the whole run -- 192 to 4608 points per workgroup, one wave per
//     SIMD (round 2: 16 points on each of 16 waves of one CU, 1.2 us of VALU issue per step);
//   * a thread carries the coordinates of its best point along, the wave's best is found with DPP row
//     rotations + v_readlane (no LDS round trips), the workgroup's best with one LDS exchange;
//   * the hand-off carries the candidates' COORDINATES: four 8-byte write-through stores per candidate,
//     {dist, gen | idx} {x, gen} {y, gen} {z, gen}, each self-tagged with the exchange's generation (the
//     8-byte granule written by one store is the unit whose atomicity the hardware documents; round 2 published
//     the index alone and every workgroup then fetched the pivot from memory: one more dependent miss);
//   * one wave per workgroup polls the cloud's W slots (a lane per slot), picks the global winner
//     and hands the pivot to its workgroup through LDS.  Slots are double-buffered by step parity: no
//     reset, no fences, no grid barrier.
// Several clouds -- of DIFFERENT sizes and sample counts -- run side by side in one launch (grid.y):
// the two subsamplings of the metric and the fused cloud's can share one pass (genpc_fps_multi).
// All workgroups of a cloud must be co-resident; launches are sized from the device's CU count at
// four 256-thread workgroups per CU at most (the hardware admits eight).  The polls are bounded all the same:
// a hand-off that times out aborts the cloud's run (every workgroup leaves its step loop), raises the error
// word and writes -1 to out[0]; genpc_amd/fps.py raises on it.
#include "common.h"
#include "../../include/genpc_hip.h"
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

namespace genpc {

// admission of launches whose workgroups wait for each other (csrc/emd_auction.hip): quarter-CU units
bool persist_reserve(int wgs, int capacity, hipStream_t st);
void persist_commit(int wgs, hipStream_t st);
int emd_auction_capacity();
// fps_grid.hip: the one-workgroup sampling with spatial pruning
bool fps_grid_takes(int n);
int fps_grid_run(bool fma, int c, const int *sel, const int *n, const int *k, const float *const *xyz, int *const *out_idx,
                 float *const *pdist_of, float4 *spt, int *err, hipStream_t st);

constexpr int kFThreads = 192;           // worker threads of a workgroup: three waves, the fourth wave coordinates -- one wave per SIMD
constexpr int kFWaves = kFThreads / kWave;
constexpr int kFMaxR = 24;
constexpr int kFMaxW = 64;
constexpr int kFMaxJobs = 8;
constexpr int kFPointsPerWg = 192;       // target: one point per worker thread.  What a round yields grows with the NUMBER of lists
                                         // (64 lists x 4 candidates: ~45 samples per exchange; 8 lists: 13), so small clouds are spread
                                         // over many workgroups too; the cap of 64 workgroups raises the share for clouds beyond 16384 points

struct FpsJobs {
    const float *xyz[kFMaxJobs];
    int *out[kFMaxJobs];
    int n[kFMaxJobs], k[kFMaxJobs], W[kFMaxJobs];
    int slot0[kFMaxJobs];        // first slot of the job in the slot array (slots are per (job, parity, workgroup))
    int stat0;                   // index of the launch's first cloud in the call (statistics)
    float *pdist[kFMaxJobs];     // the running minimum of every sample when it was drawn (fps_verify_kernel)
    int *verr[kFMaxJobs];        // != 0: the verification found a step whose sample is not the first arg-max
    int segoff[kFMaxJobs];       // first point of the job in the verification's per-(point, segment) minima
    int legacy_pivot;            // test hook (genpc_fps_tune): the workers read the pivot as per-lane LDS broadcasts again -- the
                                 // form that drew wrong samples next to f16 MFMAs on another stream (tests/test_gpu_concurrency.py)
};

template <int FMA>
__device__ __forceinline__ float sqdist_f(float dx, float dy, float dz)
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

// Wave-wide integer max / min, uniform result, in seven instructions: four row rotations (each one v_max_i32 with a
// DPP operand -- written in assembly: the compiler emits v_mov_b32 + v_mov_b32_dpp + v_max_i32 per step) leave every
// row's result in all of its lanes, row_bcast:15 / row_bcast:31 fold the rows into lane 63, one v_readlane fetches it.
// (s_nop 1: a VGPR written by a VALU instruction needs two wait states before a DPP read.)
#define GENPC_WAVE_REDUCE_I32(OP)                                                                         \
    int r;                                                                                                \
    asm volatile("s_nop 1\n\t"                                                                            \
                 OP " %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                                 \
                 "s_nop 1\n\t"                                                                            \
                 OP " %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"                                 \
                 "s_nop 1\n\t"                                                                            \
                 OP " %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"                                 \
                 "s_nop 1\n\t"                                                                            \
                 OP " %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                                 \
                 "s_nop 1\n\t"                                                                            \
                 OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                              \
                 "s_nop 1\n\t"                                                                            \
                 OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                              \
                 "s_nop 1"                                                                                \
                 : "=&v"(r)                                                                               \
                 : "v"(v));                                                                               \
    return __builtin_amdgcn_readlane(r, 63);

__device__ __forceinline__ int wave_min_i32(int v) { GENPC_WAVE_REDUCE_I32("v_min_i32_dpp") }
__device__ __forceinline__ int wave_max_i32_dpp(int v) { GENPC_WAVE_REDUCE_I32("v_max_i32_dpp") }

__device__ __forceinline__ float lane_f32(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// 8-byte write-through stores and L1-bypassing 16-byte loads (sc0 sc1: what agent-scope relaxed atomics lower to
// for 8 bytes).  A 16-byte load may see its two granules at different times: each carries its own tag.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_sc_b64(void *p, uint2 v)
{
    const u32x2 w = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}

constexpr int kFT = 4;          // candidates a workgroup publishes per exchange
constexpr int kFBatch = 64;     // most samples drawn from one exchange
constexpr int kFBlock = kFThreads + kWave;     // three waves own the points, a fourth coordinates: each has a SIMD to itself (a fifth
                                               // wave shared its SIMD with a worker, and the replay -- a chain of dependent
                                               // cross-lane steps -- got every other issue slot: 1200 cycles per pick)

struct FpsCand {          // 32 bytes = four self-tagged 8-byte granules, each written by ONE write-through store (the
    uint4 a;              //   documented single-copy-atomic unit): {dist bits, gen << 20 | idx} {x bits, gen}
    uint4 b;              //   {y bits, gen} {z bits, gen}.  Read back as two 16-byte loads; every half carries its tag.
};
struct FpsSlot {          // one 128-byte line per (job, parity, workgroup)
    FpsCand c[kFT];
};

// the eight loads of one slot in flight together, one wait
__device__ __forceinline__ void load_slot(const FpsSlot *p, u32x4 (&a)[kFT], u32x4 (&b)[kFT])
{
    static_assert(kFT == 4 && sizeof(FpsCand) == 32, "offsets below");
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %8, off offset:16 sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %8, off offset:32 sc0 sc1\n\t"
                 "global_load_dwordx4 %5, %8, off offset:48 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:64 sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %8, off offset:80 sc0 sc1\n\t"
                 "global_load_dwordx4 %3, %8, off offset:96 sc0 sc1\n\t"
                 "global_load_dwordx4 %7, %8, off offset:112 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                 : "v"(p)
                 : "memory");
}

// One exchange among a cloud's workgroups yields SEVERAL samples, exactly:
//   every workgroup lists its kFT best points in the order (distance desc, index asc) with their coordinates;
//   every workgroup reads all lists and replays the sequential algorithm on the listed points alone -- pick the
//   best, lower the listed distances by the pick, pick again ...  A pick is the true next sample as long as its
//   key is not below tau = the best of the workgroups' LAST listed keys (as listed): every unlisted point of a
//   workgroup ranks below that workgroup's last listed point, and distances only fall.  The first pick of an
//   exchange always qualifies, so a round never stalls; with points dealt to workgroups round-robin the top of
//   the global order is spread over many lists and a round draws tens of samples (measured: 30 at 172000
//   points, 13 at 16384, 8.5 at 8192).
// Inside a workgroup: waves 0..2 own the points (registers); wave 3, the coordinator, owns none.  It merges the
// waves' lists, publishes, polls, replays -- and hands every pick to the workers through LDS (s_piv + a progress
// word) the moment it is made, so the workers lower their running minima WHILE the replay goes on (the replay is a
// chain of dependent cross-lane steps, ~400 cycles per pick, and would otherwise leave three SIMDs idle).  One
// barrier per round: workers' lists ready -> coordinator.
constexpr unsigned kProgDone = 0x8000u, kProgFinal = 0x4000u, kProgAbort = 0x2000u, kProgCount = 0x0fffu;

// The progress word and the pivots it announces are both LDS, written by ONE wave and read by the others: the LDS unit
// serves a wave's requests in issue order, so "pivot, then word" on the writer and "word, then pivot" on the readers
// need no hardware fence -- only the compiler must keep the order (a release / acquire pair at workgroup scope also
// waits for the wave's outstanding GLOBAL stores: ~500 cycles per pick next to the 16-byte publishes).
__device__ __forceinline__ void prog_store(unsigned *p, unsigned v)
{
    // The pivots are 16-byte LDS writes, the word a 4-byte one: the wave's earlier LDS writes are PERFORMED before the word is
    // issued (lgkmcnt counts LDS operations only: ~60 cycles, no global store is waited for).  Belt and braces -- the
    // failure that prompted it (round 4: a sampling next to the f16 nearest-neighbour filter on another stream drew a sample
    // too early, silently) turned out to sit on the readers' side: see the workers' pivot reads.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned prog_load(unsigned *p)
{
    asm volatile("" ::: "memory");
    const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}

// The lane with the wave's best key.  Distances are >= 0 or one of the negative sentinels (-1 padding, -2 dead),
// never NaN: the signed-integer order of their bit patterns ranks every live value above every sentinel and agrees
// with the float order among live values, so the maximum is ONE integer reduction (fmaxf would canonicalise every
// operand first).  Ties (equal distances) take the index reduction.
__device__ __forceinline__ int wave_argbest(float v, int idx, float &m)
{
    const int vb = __float_as_int(v);
    const int mb = wave_max_i32_dpp(vb);
    m = __int_as_float(mb);
    const unsigned long long mask = __ballot(vb == mb);
    if (__popcll(mask) == 1) return __ffsll((long long)mask) - 1;
    const int cand = vb == mb ? idx : 0x7fffffff;
    const int best = wave_min_i32(cand);
    return __ffsll((long long)__ballot(cand == best)) - 1;
}

// The body of the sampling.  HOOK = 1 compiles the bisection's variants (genpc_fps_tune bits, packed fp32 written out) into the
// workers' update: that instantiation is inlined into fps_kernel_hook ONLY, the one kernel of the library that carries the
// packed-fp32-ops target attribute and is launched when a test asks for a variant; the shipped fps_kernel has no such
// attribute and no such code (the attribute used to sit on the shipped kernel, its update kept free of packed
// instructions by opaque statements alone; tests/test_abi.py now exempts fps_kernel_hook by name and nothing else).
template <int FMA, int R, int HOOK>
static __device__ __forceinline__ void fps_body(const FpsJobs &jobs, FpsSlot *slots, int *__restrict__ err)
{
    __shared__ float s_c[kFWaves][kFT][5];     // per worker wave: dist, idx (bits), x, y, z of its kFT best
    __shared__ float s_piv[kFBatch][4];        // the round's pivots: x, y, z, idx (bits)
    // ... and the same pivots as three SELF-TAGGED 8-byte granules {x, tag} {y, tag} {z, tag}, tag = round * 64 + slot + 1: what
    // the workers read.  A worker takes a pivot only when all three tags of ITS OWN copy are the expected one and re-reads
    // otherwise, so a lane that is handed the previous occupant of the slot (round 4: lanes 48-63 next to another stream's f16
    // MFMAs, mechanism unknown; round 5: a wrong sample again with the pivot taken through scalar registers, next to the
    // one-launch auction) cannot lower its minima by it -- the hand-off no longer depends on what caused that.
    __shared__ uint2 s_pivt[kFBatch][3];
    __shared__ float s_pd[kFBatch];            // the round's picks: their running minimum when drawn
    __shared__ unsigned s_prog;                // round << 16 | flags | pivots of that round published so far
    const int job = blockIdx.y, wg = blockIdx.x, t = threadIdx.x;
    const int W = jobs.W[job];
    if (wg >= W) return;
    const int n = jobs.n[job], k = jobs.k[job];
    const int lane = t & (kWave - 1), wave = t >> 6;
    const float *__restrict__ X = jobs.xyz[job];
    if (t == 0) s_prog = 0xffff0000u;          // no round carries this tag
    __syncthreads();

    if (wave < kFWaves) {
        // ------------------------------------------------------------------ workers
        // points are dealt round-robin: point i belongs to workgroup i % W (a file's scan order would otherwise
        // give every workgroup one compact patch, and a patch's best points fall together)
        float px[R], py[R], pz[R], d[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const long long i = ((long long)r * kFThreads + t) * W + wg;
            const bool ok = i < n;
            const size_t ii = ok ? (size_t)i : (size_t)(n - 1);
            px[r] = X[ii * 3 + 0];
            py[r] = X[ii * 3 + 1];
            pz[r] = X[ii * 3 + 2];
            d[r] = ok ? __builtin_inff() : -1.0f;          // -1: never selected, never updated upward
        }
        for (unsigned round = 0;; round++) {
            // ---- lower the running minima by the round's pivots as the coordinator hands them over
            unsigned applied = 0, pr;
            for (;;) {
                pr = prog_load(&s_prog);
                if ((pr >> 16) != (round & 0xffffu)) { __builtin_amdgcn_s_sleep(1); continue; }
                const unsigned avail = pr & kProgCount;
                // (the bisection's variants -- genpc_fps_tune bits, a test hook -- are compiled into a second copy of the loop body: with
                //  their tests inside the shipped loop a pick cost 0.55 us instead of 0.53)
                auto apply_pivot = [&](auto hook_tag) {
                    constexpr bool kHook = decltype(hook_tag)::value;
                    const int hook_bits = kHook ? jobs.legacy_pivot : 0;
                    // The pivot travels through SCALAR registers (first lane's copy).  As plain per-lane reads of the one LDS
                    // address (ds_read_b96 into VGPRs, consumed by packed fp32 ops right behind the wait) the lanes 48-63 of a
                    // worker wave were seen to use the PREVIOUS pivot now and then while another stream's kernel issued
                    // v_mfma_f32_32x32x16_f16 on the same SIMD -- one stale running minimum, one sample drawn a step early, the
                    // rest of the sequence shifted by one, no error (tools/stress_concurrent.py: every run with the f16
                    // filter or tools/burn.hip's bare MFMA loop next to it, never alone, never next to fp32 MFMA / LDS / VALU
                    // loads; all extra samples were points held by lanes 48-63).  The mechanism is not established (an isolated
                    // probe of broadcast reads + packed adds under the same load, tools/lds_probe.hip, shows nothing); with the
                    // value in SGPRs nine of nine stress runs are clean.  tests/test_gpu_concurrency.py keeps watch.
                    float cx, cy, cz;
                    if (hook_bits & 1) {   // the pre-fix form, kept reachable so that the trigger stays reproducible
                        cx = s_piv[applied][0]; cy = s_piv[applied][1]; cz = s_piv[applied][2];
                    } else {
                        const unsigned want = (round & 0x3ffffffu) * 64u + applied + 1u;
                        uint2 g0, g1, g2;
                        int tries = 0;
                        bool good;
                        do {
                            asm volatile("" ::: "memory");      // (re-read from LDS on every trip)
                            const unsigned long long *gp = (const unsigned long long *)&s_pivt[applied][0];
                            const unsigned long long w0 = __hip_atomic_load(gp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned long long w1 = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned long long w2 = __hip_atomic_load(gp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            g0 = make_uint2((unsigned)w0, (unsigned)(w0 >> 32));
                            g1 = make_uint2((unsigned)w1, (unsigned)(w1 >> 32));
                            g2 = make_uint2((unsigned)w2, (unsigned)(w2 >> 32));
                            good = g0.y == want && g1.y == want && g2.y == want;
                        } while (!__all(good) && ++tries < 4096);
                        // ... and the VALUE every lane uses is lane 0's copy, through scalar registers (tools/fps_reject_probe.py, round 5:
                        // with six scans in flight 8 % of the samplings failed the device-side check, and every wrong sample -- 44 of 44 --
                        // was a point held in REGISTER 0 by lanes 48-63 of a worker wave, its running minimum not lowered by one pivot:
                        // the first arithmetic behind the wait used the previous occupant of the value register in the last sixteen
                        // lanes although the tag registers of the same loads, compared later, were current)
                        cx = __uint_as_float(__builtin_amdgcn_readfirstlane(g0.x));
                        cy = __uint_as_float(__builtin_amdgcn_readfirstlane(g1.x));
                        cz = __uint_as_float(__builtin_amdgcn_readfirstlane(g2.x));
                    }
                    // What the wrong samples were (round 5, tools/fps_reject_probe.py; six scans in flight): written plainly, the compiler
                    // pairs registers r, r + 1 into PACKED fp32 instructions whose subtracts take ONE half of a source pair for both
                    // lanes (v_pk_add_f32 ... op_sel_hi:[1,0] / op_sel:[0,1]), and beside other streams' kernels 8 % of the samplings
                    // then failed the device-side check -- every first wrong sample, 55 of 55, a point held in the LOW lane of a pair
                    // (register 0) by lanes 48-63 of a worker wave whose running minimum had missed one pivot.  The bits of
                    // genpc_fps_tune select the variants of the bisection (DESIGN.md 6a has the table): the packed form WITH half
                    // selection fails beside the f16 filter however the pivot arrives and however many wait states surround it (written
                    // out, bits 2 | 32 | 128: every sampling), the same packed arithmetic on {c, c} pairs WITHOUT half selection never
                    // (bits 2 | 32), and neither does anything alone on the GPU.  It reproduces in a plain HIP program: tools/opsel_probe.hip,
                    // v_pk_add_f32 ... op_sel:[0,1] on known data beside tools/burn.hip's MFMA kernel on another stream -- 2.5e9 wrong
                    // results in 5 s, all in lanes 48-63, all the low lane's, as if the selected half were 0 (op_sel_hi / plain: none).
                    // Shipped: one register at a time (the opaque statements keep the compiler from pairing), and the whole
                    // library is built without packed fp32 instructions (genpc_amd/build.py).
                    if (hook_bits & 4) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                    if (hook_bits & 8) {
                        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\ts_nop 4" : "+v"(cx), "+v"(cy), "+v"(cz));
                    }
                    if (hook_bits & 2) {       // the pre-fix form (test hook): registers r, r + 1 as two-element vectors -> v_pk_*_f32
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        static_assert(R % 2 == 0, "pairs of registers");
#pragma unroll
                        for (int r = 0; r < R; r += 2) {
                            const f32x2 dx = (f32x2){px[r], px[r + 1]} - cx, dy = (f32x2){py[r], py[r + 1]} - cy, dz = (f32x2){pz[r], pz[r + 1]} - cz;
                            f32x2 dd;
                            if (FMA && (hook_bits & 32)) {
                                // (bisect: the same six packed instructions written out, four wait states behind each)
                                const f32x2 pxx = {px[r], px[r + 1]}, pyy = {py[r], py[r + 1]}, pzz = {pz[r], pz[r + 1]};
                                const f32x2 cxx = {cx, cx}, cyy = {cy, cy}, czz = {cz, cz};
                                f32x2 ex, ey, ez;
                                if (hook_bits & 128) {    // (... and with the operand forms the compiler chose where it failed: the pivot
                                    // as pairs (x, y) and (y, z), a packed subtract taking ONE half of a pair for both of its lanes --
                                    // op_sel_hi:[1,0] / op_sel:[0,1] --, four wait states behind each instruction)
                                    const f32x2 cxy = {cx, cy}, cyz = {cy, cz};
                                    asm volatile("s_nop 3\n\t"
                                                 "v_pk_add_f32 %0, %4, %7 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3\n\t"
                                                 "v_pk_add_f32 %1, %5, %8 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3\n\t"
                                                 "v_pk_add_f32 %2, %6, %8 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3\n\t"
                                                 "v_pk_mul_f32 %3, %1, %1\n\ts_nop 3\n\t"
                                                 "v_pk_fma_f32 %3, %0, %0, %3\n\ts_nop 3\n\t"
                                                 "v_pk_fma_f32 %3, %2, %2, %3\n\ts_nop 3"
                                                 : "=&v"(ex), "=&v"(ey), "=&v"(ez), "=&v"(dd)
                                                 : "v"(pxx), "v"(pyy), "v"(pzz), "v"(cxy), "v"(cyz));
                                } else
                                if (hook_bits & 64)       // (... and with ONE wait state behind each, what the compiler leaves between dependent ones)
                                asm volatile("s_nop 0\n\t"
                                             "v_pk_add_f32 %0, %4, %7 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n\t"
                                             "v_pk_add_f32 %1, %5, %8 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n\t"
                                             "v_pk_add_f32 %2, %6, %9 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n\t"
                                             "v_pk_mul_f32 %3, %1, %1\n\ts_nop 0\n\t"
                                             "v_pk_fma_f32 %3, %0, %0, %3\n\ts_nop 0\n\t"
                                             "v_pk_fma_f32 %3, %2, %2, %3\n\ts_nop 0"
                                             : "=&v"(ex), "=&v"(ey), "=&v"(ez), "=&v"(dd)
                                             : "v"(pxx), "v"(pyy), "v"(pzz), "v"(cxx), "v"(cyy), "v"(czz));
                                else
                                asm volatile("s_nop 3\n\t"
                                             "v_pk_add_f32 %0, %4, %7 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3\n\t"
                                             "v_pk_add_f32 %1, %5, %8 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3\n\t"
                                             "v_pk_add_f32 %2, %6, %9 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3\n\t"
                                             "v_pk_mul_f32 %3, %1, %1\n\ts_nop 3\n\t"
                                             "v_pk_fma_f32 %3, %0, %0, %3\n\ts_nop 3\n\t"
                                             "v_pk_fma_f32 %3, %2, %2, %3\n\ts_nop 3"
                                             : "=&v"(ex), "=&v"(ey), "=&v"(ez), "=&v"(dd)
                                             : "v"(pxx), "v"(pyy), "v"(pzz), "v"(cxx), "v"(cyy), "v"(czz));
                            } else if (FMA) dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
                            else dd = (dx * dx + dy * dy) + dz * dz;
                            if (hook_bits & 16) {      // (bisect: the two results leave the pair through separate 32-bit registers)
                                float e0 = dd.x, e1 = dd.y;
                                asm volatile("" : "+v"(e0));
                                asm volatile("" : "+v"(e1));
                                d[r] = d[r] < e0 ? d[r] : e0;
                                d[r + 1] = d[r + 1] < e1 ? d[r + 1] : e1;
                                asm volatile("" : "+v"(d[r]));
                                asm volatile("" : "+v"(d[r + 1]));
                            } else {
                                d[r] = d[r] < dd.x ? d[r] : dd.x;
                                d[r + 1] = d[r + 1] < dd.y ? d[r + 1] : dd.y;
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            float dx = px[r] - cx, dy = py[r] - cy, dz = pz[r] - cz;
                            asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dz));
                            float dd = sqdist_f<FMA>(dx, dy, dz);
                            asm volatile("" : "+v"(dd));
                            d[r] = d[r] < dd ? d[r] : dd;       // padding slots stay at -1
                        }
                    }
                                };
                for (; applied < avail; applied++) apply_pivot(std::integral_constant<bool, HOOK != 0>{});
                if (pr & kProgDone) break;
            }
            if (pr & (kProgFinal | kProgAbort)) break;
            // ---- the wave's kFT best, best first
            unsigned taken = 0;
#pragma unroll
            for (int c = 0; c < kFT; c++) {
                float bv = -2.0f, bx = 0.0f, by = 0.0f, bz = 0.0f;
                int br = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool gt = !((taken >> r) & 1u) && d[r] > bv;    // ascending index within a thread: first max
                    bv = gt ? d[r] : bv;
                    br = gt ? r : br;
                    bx = gt ? px[r] : bx;
                    by = gt ? py[r] : by;
                    bz = gt ? pz[r] : bz;
                }
                const int bi = bv > -2.0f ? (int)((((long long)br * kFThreads + t) * W + wg)) : 0x7fffffff;
                float mx;
                const int src = wave_argbest(bv, bi, mx);
                // (cross-lane reads in wave-uniform code: inside a lane-0 branch the compiler may copy a value with
                // an exec-masked move, and v_readlane would then fetch another lane's stale register)
                const float wi = __int_as_float(__builtin_amdgcn_readlane(bi, src));
                const float w0 = lane_f32(bx, src), w1 = lane_f32(by, src), w2 = lane_f32(bz, src);
                if (lane == 0) {
                    s_c[wave][c][0] = mx;
                    s_c[wave][c][1] = wi;
                    s_c[wave][c][2] = w0;
                    s_c[wave][c][3] = w1;
                    s_c[wave][c][4] = w2;
                }
                if (lane == src) taken |= 1u << br;
            }
            __syncthreads();                   // lists ready -> coordinator
        }
        return;
    }

    // ---------------------------------------------------------------------- coordinator (the last wave)
    int *__restrict__ out = jobs.out[job];
    float *__restrict__ pdist = jobs.pdist[job];
    FpsSlot *S = slots + jobs.slot0[job];
    int s = 1;                                 // samples drawn so far
    if (lane == 0) {
        s_piv[0][0] = X[0]; s_piv[0][1] = X[1]; s_piv[0][2] = X[2];
        s_piv[0][3] = __int_as_float(0);
        s_pivt[0][0] = make_uint2(__float_as_uint(X[0]), 1u);      // round 0, slot 0
        s_pivt[0][1] = make_uint2(__float_as_uint(X[1]), 1u);
        s_pivt[0][2] = make_uint2(__float_as_uint(X[2]), 1u);
        if (wg == 0) { out[0] = 0; pdist[0] = __builtin_inff(); }
        prog_store(&s_prog, (0u << 16) | kProgDone | (k <= 1 ? kProgFinal : 0u) | 1u);
    }
    bool timed_out = false;
    unsigned round = 0;
    while (s < k) {
        __syncthreads();                       // the workers' lists of this round
        round++;
        // ---- the workgroup's kFT best of the waves' lists (lanes 0..15 hold one entry each)
        float ev = -2.0f, ex = 0.0f, ey = 0.0f, ez = 0.0f;
        int ei = 0x7fffffff;
        if (lane < kFWaves * kFT) {
            const float *e = &s_c[lane / kFT][lane % kFT][0];
            ev = e[0]; ei = __float_as_int(e[1]); ex = e[2]; ey = e[3]; ez = e[4];
        }
        float cd[kFT], cxs[kFT], cys[kFT], czs[kFT];
        int ci[kFT];
#pragma unroll
        for (int c = 0; c < kFT; c++) {
            float mx;
            const int src = wave_argbest(ev, ei, mx);
            cd[c] = mx;
            ci[c] = __builtin_amdgcn_readlane(ei, src);
            cxs[c] = lane_f32(ex, src); cys[c] = lane_f32(ey, src); czs[c] = lane_f32(ez, src);
            if (lane == src) ev = -2.0f;
        }
        if (W > 1) {
            const unsigned gen = (round % 4095u) + 1u;
            FpsSlot *slot = S + (size_t)(round & 1u) * W;
            if (lane < kFT) {
                // lane c publishes candidate c; an entry without a live point (dist < 0) carries index 0xFFFFF
                float pd = cd[0], pxx = cxs[0], pyy = cys[0], pzz = czs[0];
                int pi = ci[0];
#pragma unroll
                for (int c = 1; c < kFT; c++)
                    if (lane == c) { pd = cd[c]; pi = ci[c]; pxx = cxs[c]; pyy = cys[c]; pzz = czs[c]; }
                const unsigned pidx = pd >= 0.0f ? (unsigned)pi : 0xFFFFFu;
                uint2 *g = (uint2 *)&slot[wg].c[lane];
                store_sc_b64(g + 0, make_uint2(__float_as_uint(pd), (gen << 20) | pidx));
                store_sc_b64(g + 1, make_uint2(__float_as_uint(pxx), gen));
                store_sc_b64(g + 2, make_uint2(__float_as_uint(pyy), gen));
                store_sc_b64(g + 3, make_uint2(__float_as_uint(pzz), gen));
            }
            // every lane < W reads workgroup `lane`'s list, bounded
            u32x4 a[kFT];
            u32x4 b[kFT];
#pragma unroll
            for (int c = 0; c < kFT; c++) { a[c] = (u32x4){0, 0, 0, 0}; b[c] = (u32x4){0, 0, 0, 0}; }
            int spins = 0;
            for (;;) {
                bool ready = true;
                if (lane < W) {
                    load_slot(&slot[lane], a, b);
#pragma unroll
                    for (int c = 0; c < kFT; c++) ready = ready && (a[c].y >> 20) == gen && a[c].w == gen && b[c].y == gen && b[c].w == gen;
                }
                if (__all(ready)) break;
                if (++spins > (1 << 21)) {
                    if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    timed_out = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (timed_out) break;
#pragma unroll
            for (int c = 0; c < kFT; c++) {
                const bool live = lane < W && (a[c].y & 0xFFFFFu) != 0xFFFFFu;
                cd[c] = live ? __uint_as_float(a[c].x) : -2.0f;
                ci[c] = live ? (int)(a[c].y & 0xFFFFFu) : 0x7fffffff;
                cxs[c] = __uint_as_float(a[c].z);
                cys[c] = __uint_as_float(b[c].x);
                czs[c] = __uint_as_float(b[c].z);
            }
        } else if (lane != 0) {
#pragma unroll
            for (int c = 0; c < kFT; c++) { cd[c] = -2.0f; ci[c] = 0x7fffffff; }
        }
        // ---- replay the sequential algorithm on the listed points (a lane per workgroup's list)
        // tau: the best of the lists' last entries, as listed (a list that is not full bounds nothing)
        float td;
        const int ti = __builtin_amdgcn_readlane(ci[kFT - 1], wave_argbest(cd[kFT - 1], ci[kFT - 1], td));
        int mm = 0;
        for (;;) {
            float ld = cd[0], lx = cxs[0], ly = cys[0], lz = czs[0];
            int li = ci[0];
#pragma unroll
            for (int c = 1; c < kFT; c++) {
                const bool better = (cd[c] > ld) | ((cd[c] == ld) & (ci[c] < li));
                ld = better ? cd[c] : ld;
                li = better ? ci[c] : li;
                lx = better ? cxs[c] : lx;
                ly = better ? cys[c] : ly;
                lz = better ? czs[c] : lz;
            }
            float md;
            const int src = wave_argbest(ld, li, md);
            const int mi = __builtin_amdgcn_readlane(li, src);
            const float qx = lane_f32(lx, src), qy = lane_f32(ly, src), qz = lane_f32(lz, src);
            if ((mm > 0) & !((md > td) | ((md == td) & (mi <= ti)))) break;
            if (lane == 0) {
                s_piv[mm][0] = qx; s_piv[mm][1] = qy; s_piv[mm][2] = qz;
                s_piv[mm][3] = __int_as_float(mi);
                s_pd[mm] = md;
                const unsigned tag = (round & 0x3ffffffu) * 64u + (unsigned)mm + 1u;
                s_pivt[mm][0] = make_uint2(__float_as_uint(qx), tag);
                s_pivt[mm][1] = make_uint2(__float_as_uint(qy), tag);
                s_pivt[mm][2] = make_uint2(__float_as_uint(qz), tag);
            }
            mm++;
            if (s + mm >= k || mm == kFBatch) break;
            if (lane == 0) prog_store(&s_prog, (round << 16) | (unsigned)mm);       // the workers may take it
#pragma unroll
            for (int c = 0; c < kFT; c++) {
                // (one candidate at a time, like the workers' update: no packed fp32 instructions)
                float dx = cxs[c] - qx, dy = cys[c] - qy, dz = czs[c] - qz;
                asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dz));
                float dd = sqdist_f<FMA>(dx, dy, dz);
                asm volatile("" : "+v"(dd));
                cd[c] = cd[c] < dd ? cd[c] : dd;        // dead entries stay at -2
            }
        }
        if (lane == 0) prog_store(&s_prog, (round << 16) | kProgDone | (s + mm >= k ? kProgFinal : 0u) | (unsigned)mm);
        // the samples of the round, after the workers have been released (no global store inside the pick loop)
        if (wg == 0 && lane < mm) { out[s + lane] = __float_as_int(s_piv[lane][3]); pdist[s + lane] = s_pd[lane]; }
        s += mm;
    }
    if (timed_out && lane == 0) prog_store(&s_prog, (round << 16) | kProgDone | kProgAbort);
    // a hand-off that timed out (workgroups of a cloud not co-resident) poisons the result visibly: index 0 is
    // always 0 in a good run.  The other workgroups' coordinators time out once on the missing list and leave too.
    if (wg == 0 && lane == 0) {
        if (timed_out || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) out[0] = -1;
        err[1 + jobs.stat0 + job] = (int)round;          // exchanges this cloud took (genpc_fps_stats)
    }
}

template <int FMA, int R>
__global__ __launch_bounds__(kFBlock) void fps_kernel(FpsJobs jobs, FpsSlot *slots, int *__restrict__ err)
{
    fps_body<FMA, R, 0>(jobs, slots, err);
}

template <int FMA, int R>
__global__ __launch_bounds__(kFBlock) __attribute__((target("packed-fp32-ops"))) void fps_kernel_hook(FpsJobs jobs, FpsSlot *slots, int *__restrict__ err)
{
    fps_body<FMA, R, 1>(jobs, slots, err);
}

// Verification of a sampling, on the device, against the DEFINITION (round 5).  The k steps of a sampling are sequential, but
// checking a finished sequence is not: sample j (drawn with running minimum M_j, as the sampling itself recorded it) is right
// iff, after the first j samples have been applied, every point i has running minimum D_i < M_j, or D_i == M_j and i > s_j,
// and the sample itself has D == M_j exactly -- "s_j is the FIRST arg-max" -- with the sampling's own arithmetic.  One thread
// per point walks the sample list once (n k distance evaluations, all independent across points: ~0.4 ms for 16384 of 24000,
// beside the sampling's 7).  Why: twice now (round 4 next to another stream's f16 MFMAs, round 5 again with the pivot taken
// through scalar registers and with self-tagged pivot granules) a sampling running BESIDE other kernels drew a sample a step
// early -- silently, mechanism not established.  A violation poisons out[0] with -1, which genpc_amd/fps.py
// already treats as "this cloud again": a wrong sequence can no longer leave the library unnoticed, whatever causes it.
constexpr int kFVBlock = 128, kFVTile = 1024, kFVMaxSeg = 16;
// The sample list is cut into `nseg` segments (grid.z) so that the whole chip takes part (one thread per point alone
// leaves a 24000-point cloud on 188 two-wave blocks: 1.8 ms beside a 10 ms sampling).  CHECK = 0: the minimum over the
// segment's samples of the point's distances -> segmin[point][segment]; CHECK = 1: the running minimum enters the segment
// with the minimum of the earlier segments' results and the segment's steps are checked.  Twice the distance
// evaluations, nseg times the threads.
template <int FMA, int CHECK>
__global__ __launch_bounds__(kFVBlock) void fps_verify_kernel(FpsJobs jobs, float *__restrict__ segmin, int nseg)
{
    __shared__ float4 s_s[kFVTile];          // x, y, z of sample l; M of sample l + 1
    __shared__ int s_i[kFVTile];             // index of sample l + 1
    __shared__ unsigned s_hit[kFVTile / 256];      // per trip of eight samples: one of them is a point of this block
    const int job = blockIdx.y, seg = blockIdx.z;
    const int n = jobs.n[job], k = jobs.k[job];
    const float *__restrict__ X = jobs.xyz[job];
    const int *__restrict__ out = jobs.out[job];
    const float *__restrict__ pd = jobs.pdist[job];
    float *__restrict__ sm = segmin + (size_t)jobs.segoff[job] * nseg;
    // samples l = 0 .. k - 2 are applied (sample l decides step l + 1); the segment's share, a multiple of 8 long
    const int per = (((k - 1) + nseg - 1) / nseg + 7) & ~7;
    const int l_lo = min(seg * per, k - 1), l_hi = min(l_lo + per, k - 1);
    bool bad = false;
    for (int i0 = blockIdx.x * kFVBlock; i0 < n; i0 += gridDim.x * kFVBlock) {      // (block-uniform: the tiles are staged together)
        const int i = min(i0 + (int)threadIdx.x, n - 1);      // (lanes past the end repeat the last point)
        const float px = X[(size_t)i * 3 + 0], py = X[(size_t)i * 3 + 1], pz = X[(size_t)i * 3 + 2];
        float D = __builtin_inff();
        if (CHECK) {
            for (int g = 0; g < seg; g++) D = fminf(D, sm[(size_t)i * nseg + g]);
            if (i == 0 && seg == 0 && out[0] != 0) bad = true;      // (-1: the hand-off gave up; anything else: not the start point)
        }
        for (int l0 = l_lo; l0 < l_hi; l0 += kFVTile) {
            __syncthreads();
            if (CHECK && threadIdx.x < kFVTile / 256) s_hit[threadIdx.x] = 0u;
            if (CHECK) __syncthreads();
            for (int t = threadIdx.x; t < kFVTile && l0 + t < l_hi; t += kFVBlock) {
                int sl = out[l0 + t];
                if (CHECK && ((unsigned)sl >= (unsigned)n || (unsigned)out[l0 + t + 1] >= (unsigned)n)) bad = true;      // (not an index of the cloud)
                sl = (unsigned)sl < (unsigned)n ? sl : 0;
                s_s[t] = make_float4(X[(size_t)sl * 3 + 0], X[(size_t)sl * 3 + 1], X[(size_t)sl * 3 + 2], pd[l0 + t + 1]);
                const int sn = out[l0 + t + 1];
                s_i[t] = sn;
                if (CHECK && (unsigned)(sn - i0) < (unsigned)kFVBlock) atomicOr(&s_hit[t >> 8], 1u << ((t >> 3) & 31));
            }
            __syncthreads();
            const int cnt = min(kFVTile, l_hi - l0);
            // eight samples per trip: the LDS reads (broadcasts) and the eight independent distances first, then the
            // running minimum's short dependent chain
            int t = 0;
            for (; t + 8 <= cnt; t += 8) {
                float4 q[8];
                float dd[8];
#pragma unroll
                for (int u = 0; u < 8; u++) q[u] = s_s[t + u];
#pragma unroll
                for (int u = 0; u < 8; u++) dd[u] = sqdist_f<FMA>(px - q[u].x, py - q[u].y, pz - q[u].z);
                if (!CHECK) {
#pragma unroll
                    for (int u = 0; u < 8; u++) D = D < dd[u] ? D : dd[u];
                    continue;
                }
                // the common case costs one comparison per step: D < M.  A step with D >= M -- the sample itself (D == M), an exact
                // tie, or a violation -- is rare and looked at again below with the index rule
                const float D0 = D;
                bool ge = false;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    D = D < dd[u] ? D : dd[u];
                    ge |= D >= q[u].w;
                }
                // (a sample whose recorded minimum is too HIGH has D < M at its own step: the trips in which a point of THIS block
                // is sampled are marked while the tile is staged and take the full rule whatever the comparison says)
                const bool hit = CHECK && ((s_hit[t >> 8] >> ((t >> 3) & 31)) & 1u) != 0u;
                if (hit || __any(ge)) {
                    float E = D0;
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        E = E < dd[u] ? E : dd[u];
                        const int sj = s_i[t + u];
                        // E < M, or the tie goes to the lower index s_j; and s_j itself has exactly its recorded minimum
                        bad |= !(E < q[u].w || (E == q[u].w && i >= sj)) || (i == sj && E != q[u].w);
                    }
                }
            }
            for (; t < cnt; t++) {
                const float4 q = s_s[t];
                const float dd = sqdist_f<FMA>(px - q.x, py - q.y, pz - q.z);
                D = D < dd ? D : dd;
                const int sj = s_i[t];
                if (CHECK) bad |= !(D < q.w || (D == q.w && i >= sj)) || (i == sj && D != q.w);
            }
        }
        if (!CHECK && i0 + (int)threadIdx.x < n) sm[(size_t)i * nseg + seg] = D;
    }
    if (CHECK && __any(bad) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(jobs.verr[job], 1);      // (padding lanes repeat point n - 1: same verdict)
}

__global__ void fps_poison_kernel(FpsJobs jobs, int nj)
{
    const int j = threadIdx.x;
    // (-2: the sequence failed the check; a sampling whose hand-off timed out has written -1 itself, and fails the check too)
    if (j < nj && *jobs.verr[j] != 0) { if (jobs.out[j][0] != -1) jobs.out[j][0] = -2; *jobs.verr[j] = 0; }
}

// deferred verification (below): a failed check is counted instead of poisoning a result that has left already
__global__ void fps_count_kernel(FpsJobs jobs, int nj, int *__restrict__ violations)
{
    const int j = threadIdx.x;
    if (j < nj && *jobs.verr[j] != 0) { atomicAdd(violations, 1); *jobs.verr[j] = 0; }
}

template <int FMA>
static void launch_fps(int R, dim3 grid, hipStream_t st, const FpsJobs &jobs, FpsSlot *slots, int *err)
{
    if (jobs.legacy_pivot != 0) {          // a test asked for one of the bisection's variants
        switch (R) {
        case 1: case 2: hipLaunchKernelGGL((fps_kernel_hook<FMA, 2>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
        case 3: case 4: hipLaunchKernelGGL((fps_kernel_hook<FMA, 4>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
        case 5: case 6: case 7: case 8: hipLaunchKernelGGL((fps_kernel_hook<FMA, 8>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
        case 9: case 10: case 11: case 12: hipLaunchKernelGGL((fps_kernel_hook<FMA, 12>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
        case 13: case 14: case 15: case 16: hipLaunchKernelGGL((fps_kernel_hook<FMA, 16>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
        default: hipLaunchKernelGGL((fps_kernel_hook<FMA, 24>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
        }
        return;
    }
    switch (R) {
    case 1: case 2: hipLaunchKernelGGL((fps_kernel<FMA, 2>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
    case 3: case 4: hipLaunchKernelGGL((fps_kernel<FMA, 4>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
    case 5: case 6: case 7: case 8: hipLaunchKernelGGL((fps_kernel<FMA, 8>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
    case 9: case 10: case 11: case 12: hipLaunchKernelGGL((fps_kernel<FMA, 12>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
    case 13: case 14: case 15: case 16: hipLaunchKernelGGL((fps_kernel<FMA, 16>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
    default: hipLaunchKernelGGL((fps_kernel<FMA, 24>), grid, dim3(kFBlock), 0, st, jobs, slots, err); break;
    }
}

static int fps_workgroups(int n)
{
    int W = ceil_div(n, kFPointsPerWg);
    if (W > kFMaxW) W = kFMaxW;
    return W < 1 ? 1 : W;
}

// the instantiation launch_fps takes for R points per thread, as an index; its points per thread
static int fps_class(int R) { return R <= 2 ? 0 : (R <= 4 ? 1 : (R <= 8 ? 2 : (R <= 12 ? 3 : (R <= 16 ? 4 : 5)))); }

// Workgroups of an instantiation one CU holds at a time, as the runtime reports it (registers: fps_kernel<*, 16> is 152
// VGPRs = 3 per CU, <*, 24> 224 VGPRs = 2; the smaller ones 4+).  The hand-off between a cloud's workgroups needs ALL of
// the launch resident together, so launches are sized from this, not from a constant (a budget of four per CU
// admitted grids of the two large instantiations that an idle full chip just holds and a busy or partitioned one does not).
template <int FMA>
static int fps_blocks_per_cu(int cls)
{
    // (the hook kernel -- a test's variant of the update -- may need more registers than the shipped one: asked separately)
    const int hook = (t_fps_legacy & 255) != 0 ? 1 : 0;
    static int cache[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
    if (cache[hook][cls] > 0) return cache[hook][cls];
    int nb = 0;
    hipError_t e = hipSuccess;
    if (hook) {
        switch (cls) {
        case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel_hook<FMA, 2>, kFBlock, 0); break;
        case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel_hook<FMA, 4>, kFBlock, 0); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel_hook<FMA, 8>, kFBlock, 0); break;
        case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel_hook<FMA, 12>, kFBlock, 0); break;
        case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel_hook<FMA, 16>, kFBlock, 0); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel_hook<FMA, 24>, kFBlock, 0); break;
        }
    } else {
        switch (cls) {
        case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel<FMA, 2>, kFBlock, 0); break;
        case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel<FMA, 4>, kFBlock, 0); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel<FMA, 8>, kFBlock, 0); break;
        case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel<FMA, 12>, kFBlock, 0); break;
        case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel<FMA, 16>, kFBlock, 0); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fps_kernel<FMA, 24>, kFBlock, 0); break;
        }
    }
    if (e != hipSuccess || nb < 1) nb = 1;
    cache[hook][cls] = nb;
    return nb;
}

thread_local int t_fps_legacy = 0;

// Verification OFF the caller's critical path (round 6; genpc_fps_defer).  The check of a finished sequence reads the cloud, the
// samples and their recorded minima: ~1 ms of chip-wide kernels behind a 4-5 ms sampling, twice per completed scan.  Deferred,
// the three arrays are copied (stream-ordered, ~0.5 MB) into a buffer of the library's own and the check runs on a side stream
// of the lowest priority class beside whatever the caller enqueues next; a failed check is COUNTED (the indices have left by
// then) and genpc_fps_deferred_check() hands the count to the caller, who samples again with the check in line
// (genpc_amd/pipeline.py does, per completed scan).  Only samplings of the one-workgroup kernel (csrc/fps_grid.hip) are
// deferred: the failures that made the check necessary were hand-offs between workgroups, which it does not have.
thread_local int t_fps_defer = 0;
struct FpsDeferred {
    hipStream_t side = nullptr;
    hipEvent_t fork[4] = {nullptr, nullptr, nullptr, nullptr}, done[4] = {nullptr, nullptr, nullptr, nullptr};
    char *buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    bool used[4] = {false, false, false, false};
    int *violations = nullptr;      // device word
    unsigned next = 0;
    bool ok = false;
};
static std::mutex g_fps_def_mu;
static std::map<std::pair<int, hipStream_t>, FpsDeferred *> g_fps_def;
static FpsDeferred *fps_deferred_of(hipStream_t st, bool create)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(g_fps_def_mu);
    auto it = g_fps_def.find({dev, st});
    if (it != g_fps_def.end()) return it->second->ok ? it->second : nullptr;
    if (!create) return nullptr;
    FpsDeferred *p = new FpsDeferred();
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    p->ok = hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio_lo) == hipSuccess;
    for (int i = 0; i < 4 && p->ok; i++)
        p->ok = hipEventCreateWithFlags(&p->fork[i], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming) == hipSuccess;
    p->ok = p->ok && hipMalloc((void **)&p->violations, sizeof(int)) == hipSuccess && hipMemset(p->violations, 0, sizeof(int)) == hipSuccess;
    g_fps_def[{dev, st}] = p;
    return p->ok ? p : nullptr;
}

}  // namespace genpc

/* Test hook (applies to the calling host thread; returns the previous setting): 1 = the sampling's workers read the
 * round's pivots as per-lane LDS broadcasts -- the round-4 form that silently drew a sample a step early while another
 * stream ran v_mfma_f32_32x32x16_f16 --, 0 = through scalar registers (the shipped form). */
GENPC_API int genpc_fps_tune(int legacy_pivot)
{
    const int prev = genpc::t_fps_legacy;
    // bits (for bisecting the trigger, tools/fps_reject_probe.py): 1 pivots read as per-lane LDS broadcasts, 2 packed update,
    // 4 sixteen wait states in front of the update, 8 the update's operands copied through fresh VGPRs first, 16 the packed results
    // leave their pair through separate 32-bit registers, 32 (with 2) the packed instructions written out with four wait states behind
    // each; 1 alone = 3, the pre-fix form
    genpc::t_fps_legacy = legacy_pivot == 1 ? 3 : (legacy_pivot & 511);      // bit 256: never the one-workgroup kernel of fps_grid.hip
    return prev;
}

GENPC_API int genpc_fps_multi(int c, const int *n, const int *k, const float *const *xyz, int *const *out_idx, void *stream)
{
    using namespace genpc;
    if (c <= 0) return 1;
    for (int j = 0; j < c; j++) {
        static_assert(kFMaxW * kFMaxR * kFThreads >= 262144, "the documented limit");
        if (n[j] <= 0 || k[j] <= 0 || k[j] > n[j] || n[j] > 262144) {
            fprintf(stderr, "genpc_fps: need 0 < k <= n <= 262144\n");
            return -1;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    // co-residency budget: what THIS device holds of the instantiation a launch takes (a partitioned or smaller part
    // has fewer than the 256 CUs of a full MI355X; another stream's kernels are not counted -- a hand-off that times
    // out because of them is retried one cloud at a time by genpc_amd/fps.py)
    int dev = 0, cus = 0;
    if (!check(hipGetDevice(&dev), "hipGetDevice")) return 0;
    if (!check(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return 0;
    if (cus < 1) cus = 1;
    const bool fma = arith_mode() != 0;
    auto budget_of = [&](int R) { return (fma ? fps_blocks_per_cu<1>(fps_class(R)) : fps_blocks_per_cu<0>(fps_class(R))) * cus; };
    // a cloud's workgroups: as many as its size asks for, fewer (more points per thread) if the device does not hold them
    auto plan = [&](int npts, int &W, int &R) {
        W = fps_workgroups(npts);
        for (;;) {
            R = ceil_div(ceil_div(npts, W), kFThreads);
            if (R > kFMaxR) return false;
            if (W <= budget_of(R) || W == 1) return W <= budget_of(R);
            W = W > 2 * budget_of(R) ? budget_of(R) : W - 1;
        }
    };
    // Which kernel: clouds that fit one workgroup's LDS take the pruned sampling of fps_grid.hip (no hand-off, several samples
    // per round, updates confined to the ball a sample can change), the others the multi-workgroup kernel above.  Same
    // sequences (tests/test_gpu_fps.py runs both on every case); genpc_fps_tune bit 256 / GENPC_FPS_GRID=0: never the former.
    static const int env_grid = tune_env("GENPC_FPS_GRID", 1, "farthest point sampling: 1 = clouds of up to 32768 points are sampled by one workgroup with spatial pruning (csrc/fps_grid.hip), 0 = always the multi-workgroup kernel");
    const bool grid_on = env_grid != 0 && t_fps_legacy == 0;
    std::vector<int> small, big;
    for (int j = 0; j < c; j++) (grid_on && fps_grid_takes(n[j]) ? small : big).push_back(j);
    size_t total_slots = 0, total_k = 0, spt_pts = 0;
    for (int j : big) total_slots += 2 * (size_t)fps_workgroups(n[j]);
    for (int j : small) spt_pts += ((size_t)n[j] + 63) & ~(size_t)63;
    for (int j = 0; j < c; j++) total_k += ((size_t)k[j] + 63) / 64 * 64;
    static const int env_verify = tune_env("GENPC_FPS_VERIFY", 1, "farthest point sampling: 1 = every sequence is checked on the device against the definition (a violation poisons out[0] = -1), 0 = no check");
    const size_t head = 256 + total_slots * sizeof(FpsSlot), verr_bytes = ((size_t)c * sizeof(int) + 255) / 256 * 256;
    size_t total_n = 0;
    for (int j = 0; j < c; j++) total_n += (size_t)n[j];
    const size_t pd_bytes = (total_k * sizeof(float) + 255) / 256 * 256;
    const size_t seg_bytes = env_verify ? (total_n * kFVMaxSeg * sizeof(float) + 255) / 256 * 256 : 0;
    char *ws = (char *)workspace(7, head + verr_bytes + pd_bytes + seg_bytes + spt_pts * sizeof(float4), st);
    if (!ws) return 0;
    int *err = (int *)ws;
    FpsSlot *slots = (FpsSlot *)(ws + 256);
    int *verr = (int *)(ws + head);
    float *pdist = (float *)(ws + head + verr_bytes);
    float *segmin = (float *)(ws + head + verr_bytes + pd_bytes);
    float4 *spt = (float4 *)(ws + head + verr_bytes + pd_bytes + seg_bytes);
    if (!check(hipMemsetAsync(ws, 0, head + verr_bytes, st), "hipMemsetAsync(fps)")) return 0;
    std::vector<float *> pd_of(c);
    std::vector<int> seg_of(c);
    {
        size_t pd_off = 0, seg_off = 0;
        for (int j = 0; j < c; j++) {
            pd_of[j] = pdist + pd_off;
            seg_of[j] = (int)seg_off;
            pd_off += ((size_t)k[j] + 63) / 64 * 64;
            seg_off += (size_t)n[j];
        }
    }
    // the verification of a group of clouds (fps_verify_kernel above), whichever kernel sampled them
    auto verify = [&](const FpsJobs &jobs, int nj, hipStream_t st, float *segmin, int *violations) {
        if (!env_verify) return;
        int nmax = 1;
        for (int q = 0; q < nj; q++) nmax = jobs.n[q] > nmax ? jobs.n[q] : nmax;
        const int gxv = ceil_div(nmax, kFVBlock);
        int nseg = (4 * cus) / (gxv * nj > 0 ? gxv * nj : 1);      // about four blocks per CU
        nseg = nseg < 1 ? 1 : (nseg > kFVMaxSeg ? kFVMaxSeg : nseg);
        const dim3 vg(gxv, nj, nseg);
        if (nseg > 1) {
            if (fma) hipLaunchKernelGGL((fps_verify_kernel<1, 0>), vg, dim3(kFVBlock), 0, st, jobs, segmin, nseg);
            else hipLaunchKernelGGL((fps_verify_kernel<0, 0>), vg, dim3(kFVBlock), 0, st, jobs, segmin, nseg);
        }
        if (fma) hipLaunchKernelGGL((fps_verify_kernel<1, 1>), vg, dim3(kFVBlock), 0, st, jobs, segmin, nseg);
        else hipLaunchKernelGGL((fps_verify_kernel<0, 1>), vg, dim3(kFVBlock), 0, st, jobs, segmin, nseg);
        if (violations) hipLaunchKernelGGL(fps_count_kernel, dim3(1), dim3(kFMaxJobs), 0, st, jobs, nj, violations);
        else hipLaunchKernelGGL(fps_poison_kernel, dim3(1), dim3(kFMaxJobs), 0, st, jobs, nj);
    };
    // deferred form: the group's arrays copied behind the sampling, the check on the side stream
    auto verify_deferred = [&](const FpsJobs &jobs, int nj, FpsDeferred *d) -> bool {
        auto up = [](size_t x) { return (x + 255) / 256 * 256; };
        size_t bytes = 256, pts = 0;
        for (int q = 0; q < nj; q++) {
            bytes += up((size_t)jobs.n[q] * 12) + 2 * up((size_t)jobs.k[q] * 4);
            pts += (size_t)jobs.n[q];
        }
        bytes += up(pts * kFVMaxSeg * sizeof(float));
        const int sl = (int)(d->next++ & 3u);
        if (d->used[sl] && hipStreamWaitEvent(st, d->done[sl], 0) != hipSuccess) return false;       // (the slot's previous check: long finished)
        if (d->cap[sl] == 0 && !d->used[0] && !d->used[1] && !d->used[2] && !d->used[3]) {
            // first use: all four buffers at once (an allocation synchronises the device: not inside the caller's second and third call)
            for (int q = 0; q < 4; q++) {
                if (hipMalloc((void **)&d->buf[q], bytes + bytes / 4) != hipSuccess) return false;
                d->cap[q] = bytes + bytes / 4;
            }
        }
        if (d->cap[sl] < bytes) {
            if (d->buf[sl]) {
                if (hipStreamSynchronize(d->side) != hipSuccess || hipFree(d->buf[sl]) != hipSuccess) return false;
                d->buf[sl] = nullptr;
                d->cap[sl] = 0;
            }
            if (hipMalloc((void **)&d->buf[sl], bytes + bytes / 4) != hipSuccess) return false;
            d->cap[sl] = bytes + bytes / 4;
        }
        char *b = d->buf[sl];
        FpsJobs cp = jobs;
        size_t off = 256, seg = 0;
        if (hipMemsetAsync(b, 0, 256, st) != hipSuccess) return false;
        for (int q = 0; q < nj; q++) {
            const size_t xb = (size_t)jobs.n[q] * 12, kb = (size_t)jobs.k[q] * 4;
            if (hipMemcpyAsync(b + off, jobs.xyz[q], xb, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
            cp.xyz[q] = (const float *)(b + off); off += up(xb);
            if (hipMemcpyAsync(b + off, jobs.out[q], kb, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
            cp.out[q] = (int *)(b + off); off += up(kb);
            if (hipMemcpyAsync(b + off, jobs.pdist[q], kb, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
            cp.pdist[q] = (float *)(b + off); off += up(kb);
            // (test hook, genpc_fps_defer(2): the check's copy of one recorded minimum is zeroed -- the check must fail)
            if (t_fps_defer == 2 && jobs.k[q] > 8 && hipMemsetAsync(cp.pdist[q] + 8, 0, sizeof(float), st) != hipSuccess) return false;
            cp.verr[q] = (int *)b + q;
            cp.segoff[q] = (int)seg;
            seg += (size_t)jobs.n[q];
        }
        if (hipEventRecord(d->fork[sl], st) != hipSuccess || hipStreamWaitEvent(d->side, d->fork[sl], 0) != hipSuccess) return false;
        verify(cp, nj, d->side, (float *)(b + off), d->violations);
        if (hipEventRecord(d->done[sl], d->side) != hipSuccess) return false;
        d->used[sl] = true;
        return true;
    };
    FpsDeferred *defer = t_fps_defer && env_verify ? fps_deferred_of(st, true) : nullptr;
    if (!small.empty()) {
        if (fps_grid_run(fma, (int)small.size(), small.data(), n, k, xyz, out_idx, pd_of.data(), spt, err, st) != 1) return 0;
        for (size_t q0 = 0; q0 < small.size(); q0 += kFMaxJobs) {
            FpsJobs jobs = {};
            int nj = 0;
            for (; nj < kFMaxJobs && q0 + nj < small.size(); nj++) {
                const int j = small[q0 + nj];
                jobs.xyz[nj] = xyz[j];
                jobs.out[nj] = out_idx[j];
                jobs.pdist[nj] = pd_of[j];
                jobs.verr[nj] = verr + j;
                jobs.segoff[nj] = seg_of[j];
                jobs.n[nj] = n[j];
                jobs.k[nj] = k[j];
            }
            if (!defer || !verify_deferred(jobs, nj, defer)) verify(jobs, nj, st, segmin, nullptr);
        }
    }
    int slot0 = 0;
    const int cb = (int)big.size();
    for (int j0 = 0; j0 < cb;) {
        // a launch takes up to kFMaxJobs clouds; its grid is (largest W) x (clouds), all of it resident together
        FpsJobs jobs = {};
        jobs.stat0 = big[j0] < 32 ? big[j0] : 32;
        jobs.legacy_pivot = t_fps_legacy & 255;
        int nj = 0, wmax = 0, rmax = 1;
        while (j0 + nj < cb && nj < kFMaxJobs) {
            const int j = big[j0 + nj];
            int W = 0, R = 0;
            if (!plan(n[j], W, R)) {
                set_error("genpc_fps: the cloud needs more co-resident workgroups than the device admits");
                return 0;
            }
            const int wm = W > wmax ? W : wmax, rm = R > rmax ? R : rmax;
            if (nj > 0 && (long long)wm * (nj + 1) > budget_of(rm)) break;
            jobs.xyz[nj] = xyz[j];
            jobs.out[nj] = out_idx[j];
            jobs.pdist[nj] = pd_of[j];
            jobs.verr[nj] = verr + j;
            jobs.segoff[nj] = seg_of[j];
            jobs.n[nj] = n[j];
            jobs.k[nj] = k[j];
            jobs.W[nj] = W;
            jobs.slot0[nj] = slot0;
            slot0 += 2 * W;
            wmax = wm;
            rmax = rm;
            nj++;
        }
        // The hand-off needs the launch resident as a whole, like the one-launch auction: both take their share of the chip
        // from one budget (quarter-CU units; a workgroup of an instantiation the CU holds k of costs ceil(4 / k)), so two such
        // launches of different streams never sit half-resident waiting for workgroups the other one keeps out (ADVICE r4:
        // two samplings of different lanes could spin against each other until the timeout).
        const int per_cu = fma ? fps_blocks_per_cu<1>(fps_class(rmax)) : fps_blocks_per_cu<0>(fps_class(rmax));
        const int units = wmax * nj * ((4 + per_cu - 1) / per_cu);
        const bool admitted = persist_reserve(units, emd_auction_capacity(), st);
        if (fma) launch_fps<1>(rmax, dim3(wmax, nj), st, jobs, slots, err);
        else launch_fps<0>(rmax, dim3(wmax, nj), st, jobs, slots, err);
        if (admitted) persist_commit(units, st);
        verify(jobs, nj, st, segmin, nullptr);
        j0 += nj;
    }
    if (!check(hipGetLastError(), "fps launch")) return 0;
    return 1;
}

namespace genpc {
void fps_deferred_prepare(hipStream_t st)
{
    FpsDeferred *d = fps_deferred_of(st, true);
    if (d) {
        (void)hipEventRecord(d->fork[0], d->side);
        (void)hipStreamSynchronize(d->side);
    }
}
}  // namespace genpc

GENPC_API int genpc_fps_defer(int on)
{
    const int prev = genpc::t_fps_defer;
    genpc::t_fps_defer = on == 2 ? 2 : (on ? 1 : 0);
    return prev;
}

GENPC_API int genpc_fps_deferred_check(void *stream)
{
    using namespace genpc;
    FpsDeferred *d = fps_deferred_of((hipStream_t)stream, false);
    if (!d) return 0;
    int v = 0;
    if (!check(hipStreamSynchronize(d->side), "hipStreamSynchronize(fps check)")) return -1;
    if (!check(hipMemcpy(&v, d->violations, sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy(fps check)")) return -1;
    if (v != 0 && !check(hipMemset(d->violations, 0, sizeof(int)), "hipMemset(fps check)")) return -1;
    return v;
}

GENPC_API int genpc_fps_stats(int c, int *rounds, void *stream)
{
    // exchanges (hand-off rounds) each of the first c <= 32 clouds of the last genpc_fps_multi call on this
    // stream took; synchronises the stream
    using namespace genpc;
    if (c < 0 || c > 60) return -1;
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace(7, 256, st);
    if (!ws) return 0;
    if (!check(hipMemcpyAsync(rounds, ws + 4, (size_t)c * sizeof(int), hipMemcpyDeviceToHost, st), "fps stats copy")) return 0;
    return check(hipStreamSynchronize(st), "fps stats sync") ? 1 : 0;
}

GENPC_API int genpc_fps(int c, int n, const float *xyz, int k, int *out_idx, void *stream)
{
    if (c <= 0 || k <= 0) return 1;
    // C clouds of one size: chunks of kFMaxJobs through the ragged entry
    for (int c0 = 0; c0 < c; c0 += genpc::kFMaxJobs) {
        const int cc = c - c0 < genpc::kFMaxJobs ? c - c0 : genpc::kFMaxJobs;
        int ns[genpc::kFMaxJobs], ks[genpc::kFMaxJobs];
        const float *xs[genpc::kFMaxJobs];
        int *os[genpc::kFMaxJobs];
        for (int j = 0; j < cc; j++) {
            ns[j] = n; ks[j] = k;
            xs[j] = xyz + (size_t)(c0 + j) * n * 3;
            os[j] = out_idx + (size_t)(c0 + j) * k;
        }
        const int rc = genpc_fps_multi(cc, ns, ks, xs, os, stream);
        if (rc != 1) return rc;
    }
    return 1;
}
