// Micro-benchmark: issue rate of the fp32 VALU instructions the all-pairs kernels are built from.
// Prints cycles per wave-instruction per SIMD (assuming the measured clock) for 1/2/4 waves/SIMD.
// Usage: hipcc --offload-arch=gfx950 -O2 ubench_valu.hip -o ubench_valu && ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void k(float *out, int iters, float b, float c) {
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    f2 bb = f2{b, b + 1.f}, cc = f2{c, c * 0.5f};
    for (int it = 0; it < iters; ++it) {
#define ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MIN(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define MIN3(i) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(bb));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(bb));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(bb), "v"(cc));
#define CMPSEL(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define SUBS(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(b));
#define MAXU(i) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MED3(i) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKMOV(i) asm volatile("v_pk_mov_b32 %0, %1, %0" : "+v"(p[i]) : "v"(bb));
#define MINU(i) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define MIN3U(i) asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 0) { REP8(ADD) REP8(ADD) }
        if (OP == 1) { REP8(MUL) REP8(MUL) }
        if (OP == 2) { REP8(FMA) REP8(FMA) }
        if (OP == 3) { REP8(MIN) REP8(MIN) }
        if (OP == 4) { REP8(MIN3) REP8(MIN3) }
        if (OP == 5) { REP8(PKADD) REP8(PKADD) }
        if (OP == 6) { REP8(PKMUL) REP8(PKMUL) }
        if (OP == 7) { REP8(PKFMA) REP8(PKFMA) }
        if (OP == 8) { REP8(CMPSEL) }
        if (OP == 9) { REP8(MOV) REP8(MOV) }
        if (OP == 10) { REP8(SUBS) REP8(SUBS) }
        if (OP == 11) { REP8(FMAC) REP8(FMAC) }
        if (OP == 12) { REP8(MAXU) REP8(MAXU) }
        if (OP == 13) { REP8(ANDOR) REP8(ANDOR) }
        if (OP == 14) { REP8(PKFMA) REP8(MIN3) }   // mix: pk_fma + min3
        if (OP == 15) { REP8(PKMOV) REP8(PKMOV) }
        if (OP == 16) { REP8(MIN3U) REP8(MIN3U) }
        if (OP == 17) { REP8(MED3) REP8(MED3) }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int OP>
void run(const char *name, float *d, int ipi) {
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        dim3 grid(256 * 4), block(64 * wps);  // 4 blocks per CU, wps waves each => wps waves/SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 100, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves each issuing iters*ipi instrs
        double instr_per_simd = (double)wps * iters * ipi;
        double cyc = ms * 1e-3 * 2.4e9;
        printf("%-22s waves/SIMD=%d  %.3f ms  cycles@2.4GHz per wave-instr per SIMD: %.2f\n", name, wps, ms,
               cyc / instr_per_simd);
    }
}

int main() {
    float *d; hipMalloc(&d, 1024);
    run<0>("v_add_f32", d, 16); run<1>("v_mul_f32", d, 16); run<2>("v_fma_f32", d, 16);
    run<11>("v_fmac_f32", d, 16);
    run<3>("v_min_f32", d, 16); run<4>("v_min3_f32", d, 16);
    run<5>("v_pk_add_f32", d, 16); run<6>("v_pk_mul_f32", d, 16); run<7>("v_pk_fma_f32", d, 16);
    run<8>("v_cmp+v_cndmask (pair)", d, 16); run<9>("v_mov_b32", d, 16); run<10>("v_sub_f32 sgpr", d, 16);
    run<12>("v_max_u32", d, 16); run<13>("v_and_or_b32", d, 16);
    run<14>("pk_fma + min3 mix", d, 16); run<15>("v_pk_mov_b32", d, 16); run<16>("v_min3_u32", d, 16);
    run<17>("v_med3_f32", d, 16);
    return 0;
}
