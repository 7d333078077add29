// ubench_dispatch.hip -- where does the hardware put workgroups?  Launches G blocks
// of 256 threads with L bytes of LDS that each spin for ~T microseconds and record
// (XCC, SE, CU) and start/end times; prints the histogram of blocks per CU and the
// number of CUs used.  Informs grid sizing for latency-bound kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dispatch.hip -o tools/ubench_dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LDS>
__global__ __launch_bounds__(256) void spin(unsigned *ids, unsigned long long *t0, unsigned long long *t1, long long ticks, int heavy_mod, int heavy_lt)
{
    __shared__ char lds[LDS];
    if (threadIdx.x == 0) lds[0] = 1;
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID bits [3:0]
    unsigned long long s = wall_clock64();
    bool heavy = (int)(blockIdx.x % heavy_mod) < heavy_lt;
    if (heavy) while ((long long)(wall_clock64() - s) < ticks) { __builtin_amdgcn_s_sleep(8); }
    unsigned long long e = wall_clock64();
    if (threadIdx.x == 0) { ids[blockIdx.x] = (xcc << 16) | (hw & 0xffff) | (lds[0] ? 0u : 1u << 31); t0[blockIdx.x] = s; t1[blockIdx.x] = e; }
}

int main(int argc, char **argv)
{
    int G = argc > 1 ? atoi(argv[1]) : 832;
    int heavy_mod = argc > 2 ? atoi(argv[2]) : 1, heavy_lt = argc > 3 ? atoi(argv[3]) : 1;
    unsigned *ids; unsigned long long *t0, *t1;
    CHECK(hipMalloc(&ids, G * 4)); CHECK(hipMalloc(&t0, G * 8)); CHECK(hipMalloc(&t1, G * 8));
    long long ticks = 100 * 100;   // wall_clock64 = 100 MHz -> 100 us
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(spin<16384>, dim3(G), dim3(256), 0, 0, ids, t0, t1, ticks, heavy_mod, heavy_lt);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h(G); std::vector<unsigned long long> a(G), b(G);
    CHECK(hipMemcpy(h.data(), ids, G * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(a.data(), t0, G * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), t1, G * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_cu;
    unsigned long long mn = ~0ull, mx = 0;
    int nheavy = 0;
    for (int i = 0; i < G; i++) {
        bool heavy = (i % heavy_mod) < heavy_lt;
        if (!heavy) continue;
        nheavy++;
        unsigned key = (h[i] >> 16 << 16) | (h[i] & 0xff00);   // xcc | se/sh/cu bits [15:8]
        per_cu[key]++;
        mn = std::min(mn, a[i]); mx = std::max(mx, b[i]);
    }
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf("G=%d heavy=%d (x %% %d < %d): distinct CUs used %zu, total span %.1f us\n  blocks-per-CU histogram:", G, nheavy, heavy_mod, heavy_lt, per_cu.size(), (mx - mn) / 100.0);
    for (auto &kv : hist) printf(" %dx%d", kv.first, kv.second);
    printf("\n  first 16 blocks (xcc,hw_id[15:8]):");
    for (int i = 0; i < 16 && i < G; i++) printf(" (%u,%02x)", h[i] >> 16, (h[i] >> 8) & 0xff);
    printf("\n");
    return 0;
}
