// How many small dependent kernels per second the device runs, from T host threads with a stream each (the alignment loops of
// several scans in flight are exactly this: ~2000 launches of 5-30 us per scan).
//   hipcc --offload-arch=gfx950 -O2 tools/launch_rate.hip -o tools/_launch_rate/launch_rate -lpthread && tools/_launch_rate/launch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>

__global__ void spin_kernel(float *p, int iters)
{
    float x = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 20000;
    for (int blocks : {1, 64, 512}) {
        for (int iters : {16, 4000}) {
            for (int T : {1, 2, 4, 6, 8, 12}) {
                std::vector<hipStream_t> st(T);
                std::vector<float *> buf(T);
                for (int t = 0; t < T; t++) {
                    hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking);
                    hipMalloc((void **)&buf[t], (size_t)blocks * 256 * 4);
                    hipMemset(buf[t], 0, (size_t)blocks * 256 * 4);
                }
                hipDeviceSynchronize();
                auto run = [&](int t, int n) {
                    for (int i = 0; i < n; i++) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, st[t], buf[t], iters);
                    hipStreamSynchronize(st[t]);
                };
                { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(run, t, 200); for (auto &x : th) x.join(); }
                const auto t0 = std::chrono::steady_clock::now();
                { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(run, t, launches / T); for (auto &x : th) x.join(); }
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                printf("blocks %4d iters %5d threads/streams %2d: %8.0f kernels/s (%.2f us per kernel per stream)\n", blocks, iters, T,
                       (launches / T) * T / dt, dt / (launches / T) * 1e6);
                for (int t = 0; t < T; t++) { hipStreamDestroy(st[t]); hipFree(buf[t]); }
            }
        }
    }
    return 0;
}
