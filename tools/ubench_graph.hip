// Per-launch cost of a chain of dependent small kernels: plain stream launches vs one hipGraph replay.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_graph.hip -o /tmp/ubench_graph && /tmp/ubench_graph
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void step(int *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main()
{
    int *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    const int chain = 150;
    for (int blocks : {1, 64, 512}) {
        auto run_stream = [&]() { for (int k = 0; k < chain; k++) hipLaunchKernelGGL(step, dim3(blocks), dim3(256), 0, s, d, blocks * 256); };
        run_stream(); hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; r++) run_stream();
        hipStreamSynchronize(s);
        double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20 / chain;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        run_stream();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; r++) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20 / chain;
        printf("blocks %4d: stream %.2f us per kernel, graph %.2f us per kernel\n", blocks, us_stream, us_graph);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
