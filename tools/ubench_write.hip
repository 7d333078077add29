// Write-only / copy bandwidth ceilings on this chip (context for get_uvs, which is all writes):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_write.hip -o tools/ubench_write && tools/ubench_write
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void wr16(float4 *o, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = make_float4(1.f, 2.f, 3.f, (float)i); }
__global__ void wr8_4(float2 *a, float *b, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { a[i] = make_float2(1.f, (float)i); b[i] = 3.f; } }
__global__ void cp16(const float4 *in, float4 *o, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = in[i]; }
int main() {
    const size_t bytes = 877ull << 20;
    float4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1024, 4096, 16384}) {
        for (int k = 0; k < 3; k++) {
            for (int w = 0; w < 2; w++) { if (k == 0) wr16<<<blocks, 256>>>(a, bytes / 16); else if (k == 1) wr8_4<<<blocks, 256>>>((float2 *)a, (float *)b, bytes / 12); else cp16<<<blocks, 256>>>(a, b, bytes / 32); }
            hipEventRecord(e0);
            for (int r = 0; r < 10; r++) { if (k == 0) wr16<<<blocks, 256>>>(a, bytes / 16); else if (k == 1) wr8_4<<<blocks, 256>>>((float2 *)a, (float *)b, bytes / 12); else cp16<<<blocks, 256>>>(a, b, bytes / 32); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%-28s blocks %5d: %7.1f us  %6.2f TB/s\n", k == 0 ? "write 16 B/lane (877 MB)" : (k == 1 ? "write 8+4 B/lane (877 MB)" : "copy 16 B/lane (438+438 MB)"), blocks, ms * 1e3, (double)bytes / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
