// Rate of the candidate bf16 filter loop: one v_mfma_f32_32x32x16_bf16 (1024 filter values) + 8 v_min3_f32
// folding its 16 accumulators (+ optional extra VALU standing in for tile tracking), 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int NV, int F32>
__global__ __launch_bounds__(512, 4) void k(float *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane + i); b[i] = (short)(0x3f00 + 3 * lane + i); }
    float fa = seed * lane, fb = seed + lane;
    f32x16 acc[2];
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0; acc[1][i] = 0; }
    float tm = 1e30f, v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x16 z; for (int i = 0; i < 16; ++i) z[i] = 0;
            if (F32) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                // 4 x (16x16x4 f32) = same 1024 outputs
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 r = __builtin_amdgcn_mfma_f32_16x16x4f32(fa + q, fb, f32x4{0, 0, 0, 0}, 0, 0, 0);
                    acc[u][q * 4 + 0] = r[0]; acc[u][q * 4 + 1] = r[1]; acc[u][q * 4 + 2] = r[2]; acc[u][q * 4 + 3] = r[3];
                }
            } else {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
            }
            const f32x16 c = acc[u ^ 1];
#pragma unroll
            for (int i = 0; i < 16; i += 2) tm = __builtin_fminf(__builtin_fminf(tm, c[i]), c[i + 1]);
#pragma unroll
            for (int w = 0; w < NV; ++w) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[w & 7]) : "v"(tm));
            a[0] += 1;
        }
    }
    float s = tm;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 1234.5f) out[0] = s;
}

template <int NV, int F32>
void run() {
    float *out; hipMalloc(&out, 64);
    const int iters = 4000;
    hipLaunchKernelGGL((k<NV, F32>), dim3(512), dim3(512), 0, 0, out, 10, 1.3f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, F32>), dim3(512), dim3(512), 0, 0, out, iters, 1.3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = 4.0 * iters * 2;  // 4 waves/SIMD
    printf("%s NV=%2d: %.3f ms => %.1f cyc@2.4GHz per 1024 filter values per SIMD ; C2 main-loop time %.1f us\n",
           F32 ? "f32 4x16x16x4 " : "bf16 32x32x16 ", NV, ms, ms * 1e-3 * 2.4e9 / tiles_per_simd,
           (2.0 * 32 * 4096 * 4096 / 1024.0) / 1024.0 * (ms * 1e-3 / tiles_per_simd) * 1e6);
    hipFree(out);
}

int main() {
    run<0, 0>(); run<4, 0>(); run<8, 0>(); run<16, 0>();
    run<0, 1>(); run<8, 1>();
    return 0;
}
