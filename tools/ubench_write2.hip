// Store-bandwidth variants (context for get_uvs' write pass): plain / nontemporal / block-contiguous spans /
// hipMemsetAsync, 877 MB each.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_write2.hip -o tools/ubench_write2 && tools/ubench_write2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT, int SPAN>
__global__ __launch_bounds__(256) void wr(f4 *o, size_t n)
{
    if (SPAN) {
        const size_t per = (n + gridDim.x - 1) / gridDim.x;
        const size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
        for (size_t i = b + threadIdx.x; i < e; i += 256) {
            f4 v = {1.f, 2.f, 3.f, (float)i};
            if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            f4 v = {1.f, 2.f, 3.f, (float)i};
            if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
        }
    }
}
// rows: what the write pass does -- each block writes 64 rows (cameras) x 1024 points, row pitch n_pts
template <int NT>
__global__ __launch_bounds__(256) void wr_rows(float *uv, float *depth, int n_pts, int cams)
{
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j0 + 4 > n_pts) return;
    for (int k = 0; k < 64; k++) {
        const size_t q = (size_t)(blockIdx.y * 64 + k) * n_pts + j0;
        f4 a = {1.f, 2.f, 3.f, (float)k}, b = {2.f, 2.f, 3.f, (float)k}, c = {3.f, 2.f, 3.f, (float)k};
        if (NT) { __builtin_nontemporal_store(a, (f4 *)(uv + q * 2)); __builtin_nontemporal_store(b, (f4 *)(uv + q * 2 + 4)); __builtin_nontemporal_store(c, (f4 *)(depth + q)); }
        else { *(f4 *)(uv + q * 2) = a; *(f4 *)(uv + q * 2 + 4) = b; *(f4 *)(depth + q) = c; }
    }
}
int main()
{
    const size_t bytes = 877ull << 20;
    f4 *a; hipMalloc(&a, bytes + (64 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto fn) {
        fn(); fn();
        hipEventRecord(e0);
        for (int r = 0; r < 10; r++) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-44s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, (double)bytes / (ms * 1e-3) / 1e12);
    };
    char nm[128];
    for (int blocks : {1024, 2048, 4096, 16384}) {
        snprintf(nm, 128, "plain grid-stride        blocks %5d", blocks); run(nm, [&] { wr<0, 0><<<blocks, 256>>>(a, bytes / 16); });
        snprintf(nm, 128, "nontemporal grid-stride  blocks %5d", blocks); run(nm, [&] { wr<1, 0><<<blocks, 256>>>(a, bytes / 16); });
        snprintf(nm, 128, "plain block-span         blocks %5d", blocks); run(nm, [&] { wr<0, 1><<<blocks, 256>>>(a, bytes / 16); });
        snprintf(nm, 128, "nontemporal block-span   blocks %5d", blocks); run(nm, [&] { wr<1, 1><<<blocks, 256>>>(a, bytes / 16); });
    }
    run("hipMemsetAsync", [&] { hipMemsetAsync(a, 1, bytes, 0); });
    const int n_pts = 71372, cams = 1024;         // 1024 x 71372 x 12 B = 877 MB
    float *uv = (float *)a, *depth = uv + (size_t)cams * n_pts * 2;
    dim3 g((n_pts / 4 + 255) / 256, cams / 64);
    run("rows 64 cams x 1024 pts / block, plain", [&] { wr_rows<0><<<g, 256>>>(uv, depth, n_pts, cams); });
    run("rows 64 cams x 1024 pts / block, nontemporal", [&] { wr_rows<1><<<g, 256>>>(uv, depth, n_pts, cams); });
    return 0;
}
