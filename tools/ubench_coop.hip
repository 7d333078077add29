// Launch cost of a 512-block kernel: plain launch vs hipLaunchCooperativeKernel (chain of 200, per-kernel average).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_coop.hip -o /tmp/ubench_coop && /tmp/ubench_coop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void step(int *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main()
{
    int *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    const int chain = 200, blocks = 512;
    int n = blocks * 256;
    void *args[] = {&d, &n};
    for (int coop = 0; coop < 2; coop++) {
        auto run = [&]() {
            for (int k = 0; k < chain; k++) {
                if (coop) hipLaunchCooperativeKernel((const void *)step, dim3(blocks), dim3(256), args, 0, s);
                else hipLaunchKernelGGL(step, dim3(blocks), dim3(256), 0, s, d, n);
            }
        };
        run(); hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 10; r++) run();
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10 / chain;
        printf("%s: %.2f us per kernel\n", coop ? "cooperative" : "plain", us);
    }
    return 0;
}
