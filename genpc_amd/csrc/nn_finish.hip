// nn_finish.hip -- second launch of the filtered nearest-neighbour paths (the f16 MFMA filter of nn_f16.hip):
// the exact step.  A block takes 64 queries of one (direction, batch element): it gathers each query's candidate
// lists, evaluates the best-looking unit at once (one piece per thread of the query), derives the acceptance
// threshold tau from the smallest approximate value, evaluates every OTHER listed tile that is not provably out with
// the reference's exact arithmetic (work items of 16 targets, spread over the block), and writes
// (distance, first index).  A query with a list whose THIRD minimum is not provably out
// (or with non-finite values) is re-done exhaustively by the block (nn_exhaustive, nn.h: the reference's
// 512-target tile semantics).  The proof obligation is spelled out in nn_f16.hip and DESIGN.md section 4.1.
//
// (Round 3, built and measured against this structure inside one gpurun call, not kept: the threshold computed by all
// four threads of a query and (i) each list's holder evaluating its candidates itself -- the wave runs the 64-target scan
// once per list ENTRY in which any lane has a candidate: 1 x 16384^2 32.3 -> 34.9 us; (ii) per-wave work lists built with
// one prefix sum, no block barrier between threshold and results: 32.5 -> 32.1 us at 1 x 16384^2 but 57.6 -> 60.0 at
// 4 x 16384 x 8192; (iii) two items per lane and trip: 33.7 / 66.3.  tools/nn_timeline.py shows the phases.)
//
// (Round 1-2 history: this kernel was written for a three-piece bf16 filter -- 27 products in two chained
// v_mfma_f32_32x32x16_bf16 -- which the two-piece f16 filter superseded at half the matrix work; the bf16 kernel,
// its pre-split / LDS-DMA staging variant and the Morton-sorted culling mode were removed in round 3.)
#include "nn.h"

namespace genpc {

#ifdef GENPC_NN_TIMELINE       // tools/nn_timeline.py (see nn_f16.hip)
__device__ unsigned long long g_timeline_fin[4096 * 8];
#define GENPC_TLF(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_timeline_fin[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define GENPC_TLF(k) do {} while (0)
#endif

constexpr int kFQ = 64;                // queries per finish block
constexpr int kFWork = 2048;           // work-item capacity (64 queries x 32 pieces)

template <int FMA>
__global__ __launch_bounds__(kBlock) void nn_finish_kernel(NNArgs a, int nl, int upieces, float kqt, float ktt, float t2min)
{
    __shared__ unsigned long long s_best[kFQ];   // (distance bits << 32 | index): atomic min == (distance, first index)
    __shared__ float4 s_q[kFQ];
    __shared__ float s_a[4][kFQ];
    __shared__ int s_c[4][kFQ];
    __shared__ int s_qflag[kFQ];
    __shared__ int s_flagged[kFQ];
    __shared__ unsigned s_work[kFWork];          // query slot << 22 | tile (first target / 32) << 1 | lane half
    __shared__ float s_red[kWavesPerBlock];
    __shared__ int s_fi[kWavesPerBlock];
    __shared__ int s_misc[3];                    // work items, flagged queries, pieces evaluated (stats hook)
    GENPC_TLF(0);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    const int d = (a.ndir > 1 && bid >= a.dir[1].fin_begin) ? 1 : 0;
    const NNDir &D = a.dir[d];
    bid -= D.fin_begin;
    const int nq = D.nq, nt = D.nt;
    const int fblocks = (nq + kFQ - 1) / kFQ;
    const int batch = bid / fblocks, fb = bid % fblocks;
    const float *__restrict__ Qp = D.q + (size_t)batch * nq * 3;
    const float *__restrict__ T = D.t + (size_t)batch * nt * 3;
    float *__restrict__ od = D.out_d + (size_t)batch * nq;
    int *__restrict__ oi = D.out_i + (size_t)batch * nq;
    const size_t bnq = (size_t)a.b * nq;
    const unsigned long long *P = D.part + (size_t)batch * nq;
    const int nlists = D.slices * nl;

    const int ql = threadIdx.x & (kFQ - 1), part = threadIdx.x >> 6;     // part == wave
    int j = fb * kFQ + ql;
    const bool live = j < nq;
    j = live ? j : nq - 1;

    const float *cptr = a.dir[0].t + (size_t)batch * a.dir[0].nt * 3;      // common centre of the filter
    const float ccx = cptr[0], ccy = cptr[1], ccz = cptr[2];
    // this thread's lists: li = part + 4k
    unsigned long long w0[kMaxLists / 4], w1[kMaxLists / 4];      // (a1, c1), (codes of a2 | a3, c2): list_enc in nn.h
    float amin = __builtin_inff();
#pragma unroll
    for (int k = 0; k < kMaxLists / 4; k++) {
        const int li = part + 4 * k;
        w0[k] = 0x7f800000ull << 32;      // (+inf, 0)
        w1[k] = 0xff00ff00ull << 32;      // a2, a3 = +inf
        if (li < nlists) {
            const unsigned long long *p = P + (size_t)li * 2 * bnq + j;
            w0[k] = p[0];
            w1[k] = p[bnq];
        }
    }
    int cmin = -1, kmin = -1;               // unit of this thread's smallest first minimum
#pragma unroll
    for (int k = 0; k < kMaxLists / 4; k++) {
        const float a1 = __uint_as_float((unsigned)(w0[k] >> 32));
        if (a1 < amin) { amin = a1; cmin = (int)(unsigned)w0[k]; kmin = k; }
    }
    s_a[part][ql] = amin;
    s_c[part][ql] = cmin;
    // max |t'|^2 over the whole target cloud (per-slice maxima of the filter kernel)
    float tmax2 = 0.0f;
    {
        const float *tp = D.tmaxp + (size_t)batch * D.ntmax;
        for (int i = threadIdx.x; i < D.ntmax; i += kBlock) tmax2 = fmaxf(tmax2, tp[i]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tmax2 = fmaxf(tmax2, __shfl_xor(tmax2, o));
        if (lane == 0) s_red[wave] = tmax2;
    }
    if (part == 0) {
        s_q[ql] = make_float4(Qp[(size_t)j * 3 + 0], Qp[(size_t)j * 3 + 1], Qp[(size_t)j * 3 + 2], 0.0f);
        s_best[ql] = ~0ull;
        s_qflag[ql] = 0;
    }
    if (threadIdx.x == 0) { s_misc[0] = 0; s_misc[1] = 0; s_misc[2] = 0; }
    __syncthreads();
    GENPC_TLF(1);
    tmax2 = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    // The query's best unit (smallest first minimum over all its lists; lowest part on ties) is a candidate whatever the
    // threshold turns out to be -- tau >= its approximate value -- so its pieces are requested NOW, one per thread of
    // the query (a unit has at most four tiles, a query four threads), and evaluated while nothing else is known yet:
    // one candidate per query is the rule (4.0-4.2 pieces per query on uniform clouds), so the second memory round trip
    // starts right behind the first barrier instead of behind the threshold, the work list and two more barriers
    // (tools/nn_timeline.py: 17.9 k ticks per block before).  Further candidates take the work list below.
    int bp = 0;
    float abest = s_a[0][ql];
#pragma unroll
    for (int q2 = 1; q2 < 4; q2++) {
        const float v = s_a[q2][ql];
        if (v < abest) { abest = v; bp = q2; }
    }
    const int cb = s_c[bp][ql];
    const float4 qv = s_q[ql];
    unsigned long long mine = ~0ull;
    int pieces = 0;
    if (cb >= 0 && live && !(a.debug & 1)) {
        const int h = cb & 1, c0 = cb & ~1;
        const int left = (nt - c0 + 31) >> 5;
        const int n2 = left < upieces ? left : upieces;
        if (part < n2) {
            float dd;
            int ii;
            rescan_half<FMA>(T, nt, c0 + 32 * part, h, qv.x, qv.y, qv.z, dd, ii);
            mine = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)ii;
            pieces = 1;
        }
    }
    // the threshold is fp64 arithmetic (three square roots); every thread of the query computes it -- the four waves
    // run side by side, and no barrier stands between it and the lists
    float tau, qq;
    {
        const float x = qv.x - ccx, y = qv.y - ccy, z = qv.z - ccz;
        qq = __fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x)));
        tau = nn_tau(abest, qq, tmax2, (double)kqt, (double)ktt);
        if (!(tmax2 >= t2min)) tau = __builtin_nanf("");      // below the magnitudes the filter's bound covers: exhaustive
        // non-finite targets anywhere in the cloud (the filter publishes +inf) or a non-finite query:
        // the reference's result depends on its 512-target tiling, only nn_exhaustive reproduces it
        if (!(tmax2 < __builtin_inff()) || !(qq < __builtin_inff())) tau = __builtin_nanf("");
        if (a.debug & 16) tau = __builtin_inff();          // test hook: every listed tile is evaluated
    }
    if (mine != ~0ull) atomicMin(&s_best[ql], mine);
    GENPC_TLF(2);
    bool flag = (a.debug & 8) != 0 || !(tau == tau);   // test hook / non-finite input: exhaustive pass
    int ncand = 0;
    // any OTHER listed unit whose minimum is not provably out becomes work items of 16 targets for the block
    auto consider = [&](float av, int c, bool is_best) {
        if (av <= tau) {
            ncand++;
            if (c < 0) {
                flag = true;
            } else if (live && !is_best) {
                // c = first target of the unit | lane half: rows 8i + 4h + (0..3) of each tile
                const int h = c & 1, c0 = c & ~1;
                const int left = (nt - c0 + 31) >> 5;
                const int n2 = left < upieces ? left : upieces;
                const int w = atomicAdd(&s_misc[0], n2);
                if (w + n2 <= kFWork) {
                    for (int k = 0; k < n2; k++) s_work[w + k] = ((unsigned)ql << 22) | (unsigned)((((c0 >> 5) + k) << 1) | h);
                } else {
                    flag = true;
                }
            }
        }
    };
#pragma unroll
    for (int k = 0; k < kMaxLists / 4; k++) {
        const float a1 = __uint_as_float((unsigned)(w0[k] >> 32));
        const unsigned codes = (unsigned)(w1[k] >> 32);
        const float a2 = list_dec(a1, codes >> 16), a3 = list_dec(a2, codes & 0xffffu);      // lower bounds of a2, a3
        if (!(a3 > tau)) flag = true;
        consider(a1, (int)(unsigned)w0[k], part == bp && k == kmin);
        consider(a2, (int)(unsigned)w1[k], false);
    }
    if (flag && live) s_qflag[ql] = 1;
    if (a.debug & 32) {      // diagnostics: approximate minimum and candidate count instead of the result
        __syncthreads();
        if (part == 0) s_a[0][ql] = 0.0f;
        __syncthreads();
        atomicAdd(&s_a[0][ql], (float)ncand);
        __syncthreads();
        if (part == 0 && live) { od[j] = abest + qq; oi[j] = (int)s_a[0][ql] | (s_qflag[ql] ? 1 << 16 : 0); }
        return;
    }
    __syncthreads();
    GENPC_TLF(3);
    const int nwork = min(s_misc[0], kFWork);
    if (nwork > 0 && !(a.debug & 1)) {               // block-uniform
        for (int w = threadIdx.x; w < nwork; w += kBlock) {
            const unsigned it = s_work[w];
            const int slot = (int)(it >> 22);
            const float4 qs = s_q[slot];
            float dd;
            int ii;
            rescan_half<FMA>(T, nt, (int)((it & 0x3fffffu) >> 1) << 5, (int)(it & 1u), qs.x, qs.y, qs.z, dd, ii);
            atomicMin(&s_best[slot], ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)ii);
            pieces++;
        }
        __syncthreads();
    }
    if (a.stats && pieces) atomicAdd(&s_misc[2], pieces);
    __syncthreads();
    GENPC_TLF(4);
    if (part == 0 && live) {
        if (s_qflag[ql]) {
            s_flagged[atomicAdd(&s_misc[1], 1)] = j;
        } else {
            const unsigned long long v = s_best[ql];
            od[j] = __uint_as_float((unsigned)(v >> 32));
            oi[j] = (int)(unsigned)v;
        }
    }
    __syncthreads();
    const int nflag = s_misc[1];
    // queries the filter could not settle: the next call's duplicate pre-pass policy reads this (chamfer.hip); blocks
    // without such queries -- all of them on ordinary input -- do nothing
    // (sampled: every hint_stride-th block, scaled up -- at most 16 atomics on host memory per launch; they cost ~0.3 us
    // each and the kernel does not end before the last one has landed)
    if (nflag && a.hint && threadIdx.x == 0 && blockIdx.x % a.hint_stride == 0 && !(tmax2 != tmax2) && tmax2 < __builtin_inff())
        __hip_atomic_fetch_add(a.hint, (unsigned)(nflag * a.hint_stride), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.stats && threadIdx.x == 0) {
        atomicAdd(&a.stats[0], (unsigned long long)min(kFQ, nq - fb * kFQ));
        atomicAdd(&a.stats[1], (unsigned long long)nflag);
        atomicAdd(&a.stats[2], (unsigned long long)s_misc[2]);
    }
    for (int fidx = 0; fidx < nflag; fidx++) nn_exhaustive<FMA>(Qp, T, nt, s_flagged[fidx], od, oi, s_red, s_fi);
    GENPC_TLF(5);
}

// Second launch of the filtered paths.  kqt / ktt: coefficients of the filter's error bound.
int launch_nn_finish(NNArgs &a, int nl, int upieces, float kqt, float ktt, float t2min, hipStream_t st)
{
    long long fb = 0;
    for (int d = 0; d < a.ndir; d++) {
        a.dir[d].fin_begin = (int)fb;
        fb += (long long)a.b * ceil_div(a.dir[d].nq, kFQ);
    }
    if (fb > 0x7fffffffLL) {
        set_error("chamfer: problem too large for one launch");
        return 0;
    }
    a.hint_stride = (int)ceil_div64(fb, 16);
    if (a.fma)
        hipLaunchKernelGGL((nn_finish_kernel<1>), dim3((unsigned)fb), dim3(kBlock), 0, st, a, nl, upieces, kqt, ktt, t2min);
    else
        hipLaunchKernelGGL((nn_finish_kernel<0>), dim3((unsigned)fb), dim3(kBlock), 0, st, a, nl, upieces, kqt, ktt, t2min);
    return check(hipGetLastError(), "nn_finish_kernel launch") ? 1 : 0;
}

}  // namespace genpc

#ifdef GENPC_NN_TIMELINE
extern "C" __attribute__((visibility("default"))) int genpc_nn_timeline_read_finish(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(genpc::g_timeline_fin), sizeof(unsigned long long) * 4096 * 8) == hipSuccess;
}
#endif
