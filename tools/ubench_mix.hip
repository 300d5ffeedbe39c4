// Can the fp32 matrix pipe and the fp32 VALU run concurrently?  Per wave: NM independent MFMAs
// (16x16x4 f32) interleaved with NV independent v_fmac_f32, 4 waves/SIMD.  Reports effective TFLOP/s
// (MFMA flops = 2048, VALU fma flops = 128 per wave-instruction) and the s_memtime tick rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ __launch_bounds__(512, 4) void k(float *out, int iters, float seed, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    float a = seed * (lane + 1), b = seed * 0.37f * (lane + 3);
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{a, b, a + b, a - b};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a * (i + 1);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NM > 0) {
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(u * NM + m) & 3]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int w = 0; w < NV; ++w) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[w & 7]) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 1234.5f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NM, int NV>
void run(float seed) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 8 * 512);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NM, NV>), dim3(512), dim3(512), 0, 0, out, 10, seed, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV>), dim3(512), dim3(512), 0, 0, out, iters, seed, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512]; hipMemcpy(h, cyc, 8 * 512, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 512; ++i) avg += h[i]; avg /= 512;
    const double waves = 512.0 * 8;
    const double mf = waves * iters * 4 * NM * 2048.0, vf = waves * iters * 4 * NV * 128.0;
    printf("NM=%d NV=%2d seed=%g : %.3f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TF   tick rate %.2f GHz\n", NM, NV, seed, ms,
           mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9, avg / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (float seed : {0.0f, 1.37f}) {
        run<1, 0>(seed); run<0, 8>(seed); run<1, 4>(seed); run<1, 8>(seed); run<1, 12>(seed); run<1, 16>(seed); run<2, 8>(seed);
    }
    return 0;
}
