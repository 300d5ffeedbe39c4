// Does VALU work overlap with the matrix pipe on one SIMD?  4 waves/SIMD, per iteration one v_mfma_f32_32x32x16_f16 and NV
// v_min3_f32, in four flavours: MODE 0 = MFMA only, 1 = VALU only, 2 = MFMA + min3 on registers the MFMA does not touch,
// 3 = MFMA + min3 folding the PREVIOUS MFMA's accumulators (the nn1 loop's shape, software-pipelined by one),
// 4 = as 3 but pipelined by two (three accumulator sets).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>
__global__ __launch_bounds__(512, 4) void k(float *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.0f + lane + i); b[i] = (_Float16)(0.5f + 3 * lane + i); }
    f32x16 acc[3], z;
    for (int i = 0; i < 16; ++i) { acc[0][i] = seed * i; acc[1][i] = seed + i; acc[2][i] = seed - i; z[i] = 0; }
    float v[16], tm = 1e30f;
    for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1) + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (MODE != 1) acc[u % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int w = 0; w < NV; ++w) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[w & 7]) : "v"(v[8 + (w & 7)]), "v"(v[(w + 3) & 7]));
            }
            if (MODE == 3 || MODE == 4) {
                const f32x16 c = acc[(u + (MODE == 3 ? 2 : 1)) % 3];   // mode 3: the previous one, mode 4: two back
                const float t0 = __builtin_fminf(__builtin_fminf(c[0], c[1]), c[2]), t1 = __builtin_fminf(__builtin_fminf(c[3], c[4]), c[5]);
                const float t2 = __builtin_fminf(__builtin_fminf(c[6], c[7]), c[8]), t3 = __builtin_fminf(__builtin_fminf(c[9], c[10]), c[11]);
                const float t4 = __builtin_fminf(__builtin_fminf(c[12], c[13]), c[14]);
                const float t5 = __builtin_fminf(__builtin_fminf(t0, t1), t2), t6 = __builtin_fminf(__builtin_fminf(t3, t4), c[15]);
                tm = __builtin_fminf(__builtin_fminf(tm, t5), t6);
#pragma unroll
                for (int w = 8; w < NV; ++w) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[w & 7]) : "v"(tm), "v"(v[(w + 3) & 7]));
            }
            a[0] += (_Float16)1.0f;
        }
    }
    float s = tm;
    for (int i = 0; i < 16; ++i) s += v[i] + acc[0][i] + acc[1][i] + acc[2][i];
    if (s == 1234.5f) out[0] = s;
}

template <int MODE, int NV>
void run(const char *name) {
    float *out; (void)hipMalloc(&out, 64);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, NV>), dim3(512), dim3(512), 0, 0, out, 2000, 1.3f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(512), dim3(512), 0, 0, out, iters, 1.3f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double tiles_per_simd = 4.0 * iters * 6;
    printf("%-44s NV=%2d: %.3f ms => %.1f cyc@2.4GHz per tile per SIMD\n", name, NV, ms, ms * 1e-3 * 2.4e9 / tiles_per_simd);
    (void)hipFree(out);
}

int main() {
    run<0, 0>("MFMA only");
    run<1, 8>("min3 only"); run<1, 10>("min3 only"); run<1, 12>("min3 only");
    run<2, 4>("MFMA + unrelated min3"); run<2, 8>("MFMA + unrelated min3"); run<2, 10>("MFMA + unrelated min3"); run<2, 12>("MFMA + unrelated min3");
    run<3, 8>("MFMA + fold of the previous"); run<3, 10>("MFMA + fold of the previous"); run<3, 12>("MFMA + fold of the previous");
    run<4, 8>("MFMA + fold of the one before"); run<4, 10>("MFMA + fold of the one before"); run<4, 12>("MFMA + fold of the one before");
    return 0;
}
