// HBM streaming rates on MI355X with plain kernels: write-only (16-byte stores), read-only (16-byte loads, one
// accumulator per thread) and copy, 336 MB each (the size of the F = 64 EdgeConv feature tensor).  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o tools/ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void wr(float4 *o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = float4{1.f, 2.f, 3.f, (float)i};
}
__global__ __launch_bounds__(256) void rd(const float4 *x, size_t n, float *sink) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = x[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 123.456f) *sink = a;
}
__global__ __launch_bounds__(256) void cp(const float4 *x, float4 *o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = x[i];
}
int main() {
    const size_t bytes = (size_t)336 << 20, n = bytes / 16;
    float4 *a, *b; float *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        float ms;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, a, n);
        hipEventRecord(e0); for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, a, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("grid %6d  write %.2f TB/s", grid, bytes * 10 / (ms * 1e-3) / 1e12);
        hipEventRecord(e0); for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, n, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("  read %.2f TB/s", bytes * 10 / (ms * 1e-3) / 1e12);
        hipEventRecord(e0); for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("  copy %.2f TB/s (read + write)\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
