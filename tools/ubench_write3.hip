// Store patterns for a [C, N] row layout (uv 8 B + depth 4 B per (camera, point)), pure stores, 1024 x 71372.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_write3.hip -o tools/ubench_write3 && tools/ubench_write3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
// PAT 0: lane = 4 consecutive points (uv 2 x 16 B at stride 32, depth 16 B)      [the current write pass]
// PAT 1: lane = points {2l, 2l+1, 128+2l, 129+2l} of the wave's 256: uv 2 x 16 B contiguous, depth 2 x 8 B contiguous
// PAT 2: as 1, 8 points per lane (wave = 512 points)
// PAT 3: as 1, but each camera row shifts the wave's 256 points by (row * n) mod 32 so that every store starts on a 128-byte line
// CAMS: cameras per block; XFAST: 1 = blockIdx.x walks points (fastest), 0 = blockIdx.x walks camera groups
template <int PAT, int CAMS, int XFAST, int WSPLIT>
__global__ __launch_bounds__(256) void wr_rows(float *uv, float *depth, int n, int cams)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = XFAST ? blockIdx.x : blockIdx.y, by = XFAST ? blockIdx.y : blockIdx.x;
    const int ppl = PAT == 2 ? 8 : 4;
    const int wbase = (bx * 4 + wave) * 64 * ppl;
    if (PAT != 3 && wbase + 64 * ppl > n) return;
    if (PAT == 3 && wbase >= n + 32) return;
    const int kb = 0, ks = 1;
    for (int k = kb; k < CAMS; k += ks) {
        const size_t row = (size_t)(by * CAMS + k) * n;
        if (PAT == 0) {
            const size_t q = row + wbase + lane * 4;
            f4 a = {1.f, 2.f, 3.f, (float)k};
            *(f4 *)(uv + q * 2) = a; *(f4 *)(uv + q * 2 + 4) = a; *(f4 *)(depth + q) = a;
        } else if (PAT == 3) {
            const int sh = (int)((size_t)(by * CAMS + k) * n & 31);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const long long j = (long long)wbase + h * 128 + lane * 2 - sh;
                if (j >= 0 && j + 2 <= n) {
                    const size_t q = row + j;
                    f4 a = {1.f, 2.f, 3.f, (float)k}; f2 b = {1.f, (float)k};
                    *(f4 *)(uv + q * 2) = a; *(f2 *)(depth + q) = b;
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < ppl / 2; h++) {
                const size_t q = row + wbase + h * 128 + lane * 2;
                f4 a = {1.f, 2.f, 3.f, (float)k}; f2 b = {1.f, (float)k};
                *(f4 *)(uv + q * 2) = a; *(f2 *)(depth + q) = b;
            }
        }
    }
}
#include <stdlib.h>
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 71372, cams = 1024;
    printf("n = %d\n", n);
    const size_t bytes = (size_t)cams * n * 12;
    float *uv; hipMalloc(&uv, bytes + (64 << 20));
    float *depth = uv + (size_t)cams * n * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto fn) {
        fn(); fn();
        hipEventRecord(e0);
        for (int r = 0; r < 10; r++) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-56s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, (double)bytes / (ms * 1e-3) / 1e12);
    };
#define RUN(PAT, CAMS, XFAST, WS) { const int ppl = PAT == 2 ? 8 : 4; const int bxn = (n / (256 * ppl)) + 1, byn = cams / CAMS; \
        dim3 g(XFAST ? bxn : byn, XFAST ? byn : bxn); char nm[128]; snprintf(nm, 128, "pat %d cams/block %4d %s %s", PAT, CAMS, XFAST ? "points-fastest" : "cams-fastest  ", WS ? "wave=cam" : ""); \
        run(nm, [&] { wr_rows<PAT, CAMS, XFAST, WS><<<g, 256>>>(uv, depth, n, cams); }); }
    RUN(3, 64, 0, 0) RUN(3, 16, 0, 0) RUN(0, 64, 0, 0) RUN(1, 64, 0, 0) RUN(2, 64, 0, 0) RUN(1, 4, 0, 0) RUN(0, 64, 1, 0)
    run("hipMemsetAsync", [&] { (void)hipMemsetAsync(uv, 1, bytes, 0); });
    return 0;
}
