// Does VALU work run under the matrix pipe on ONE SIMD of gfx950, and under which schedule?  (round 6: the round-3 version of
// this file lost half of its MFMAs to dead-code elimination -- acc[u % 3] was overwritten without a reader -- so its table proved
// nothing.  Here every MFMA result is read by a later instruction, the instruction counts of every kernel's loop are taken from
// the ISA by tools/ubench_overlap_isa.py and printed next to the timings, and the cycle counts come from s_memtime inside the
// kernel, not from an assumed clock.)
//
// One block per CU (96 KB of dynamic LDS keeps a second one out), W = blockDim / 256 waves per SIMD.  A "tile" is one
// v_mfma_f32_32x32x16_f16 (C = 0, as in nn1_f16_kernel) and/or NV VALU instructions.  Kernels, per loop iteration of 6 tiles:
//   mfma          : the MFMAs alone; every result is consumed two tiles later by ONE v_min3 (so NV = 1)
//   valu<OP>      : NV independent OP (OP = v_min3_f32: three VGPR sources; v_min_f32: two)
//   both<OP>      : MFMA, then NV OP on registers the MFMAs never touch (+ the one consuming v_min3), in ONE wave's stream
//   fold          : MFMA, then the nn1 fold of the block issued two tiles earlier (8 v_min3) + the lane-tile tracking every other
//                   tile (v_and_or, 3 v_med3, v_min) + the operand's ds_read_b128: the loop of csrc/chamfer.hip, PRIO = its s_setprio
//   fold_before   : the same with the fold in FRONT of the MFMA issue
//   roles         : waves are MFMA-only or VALU-only (NV OP per tile) by wave id: ROLE_BY = 1 alternates inside a SIMD
//                   (waves w and w + 4 share a SIMD), ROLE_BY = 0 puts all MFMA waves on SIMDs 0, 2 and all VALU waves on 1, 3
//   grp4          : ONE wave carries four query groups: one operand read, four MFMAs (four B operands), four folds -- the
//                   structure that needs only one wave per SIMD to keep LDS traffic and latency covered
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum { K_MFMA, K_VALU, K_BOTH, K_FOLD, K_FOLD_BEFORE, K_ROLES, K_GRP4 };
enum { OP_MIN3, OP_MIN2 };

#define VOP(OP, D, A, B)                                                                                      \
    do {                                                                                                      \
        if (OP == OP_MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(D) : "v"(A), "v"(B));             \
        else asm volatile("v_min_f32 %0, %0, %1" : "+v"(D) : "v"(A));                                        \
    } while (0)

__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float vmin(float a, float b) {   // csrc/chamfer.hip's: plain v_min_f32, no canonicalising v_max
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int KIND, int NV, int OP, int PRIO, int ROLE_BY>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float seed) {
    extern __shared__ h8 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
        h8 t;
        for (int e = 0; e < 8; ++e) t[e] = (_Float16)(seed * (float)((i * 8 + e) % 97) - 3.0f);
        lds[i] = t;
    }
    __syncthreads();
    h8 a = lds[lane], b, b1, b2, b3;
    for (int i = 0; i < 8; ++i) {
        b[i] = (_Float16)(0.5f + seed * (float)(3 * lane + i));
        b1[i] = b[i] + (_Float16)1.0f; b2[i] = b[i] + (_Float16)2.0f; b3[i] = b[i] + (_Float16)3.0f;
    }
    f32x16 zero;
    for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
    f32x16 acc0 = zero, acc1 = zero, acc2 = zero, acc3 = zero;
    float v[16], tm = 1e30f, fk[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    float tmg[4] = {1e30f, 1e30f, 1e30f, 1e30f};
    for (int i = 0; i < 16; ++i) v[i] = seed * (float)(i + 1) + (float)lane;
    const unsigned int keymask = ~63u;
    int lt = 0;
    const h8 *pa = lds + lane;
    const bool vrole = KIND == K_ROLES && (ROLE_BY ? ((wave >> 2) & 1) : (wave & 1));
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();

#define FOLD8(ACC, FIRST)                                                                                     \
    {                                                                                                         \
        const float t0_ = min3f(ACC[0], ACC[1], ACC[2]), t1_ = min3f(ACC[3], ACC[4], ACC[5]);                 \
        const float t2_ = min3f(ACC[6], ACC[7], ACC[8]), t3_ = min3f(ACC[9], ACC[10], ACC[11]);               \
        const float t4_ = min3f(ACC[12], ACC[13], ACC[14]);                                                   \
        const float t5_ = min3f(t0_, t1_, t2_), t6_ = min3f(t3_, t4_, ACC[15]);                               \
        tm = (FIRST) ? vmin(t5_, t6_) : min3f(tm, t5_, t6_);                                       \
    }
#define TRACK()                                                                                               \
    {                                                                                                         \
        float key;                                                                                            \
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(tm), "v"(keymask), "s"(lt));                      \
        fk[3] = __builtin_amdgcn_fmed3f(fk[2], fk[3], key);                                                   \
        fk[2] = __builtin_amdgcn_fmed3f(fk[1], fk[2], key);                                                   \
        fk[1] = __builtin_amdgcn_fmed3f(fk[0], fk[1], key);                                                   \
        fk[0] = vmin(fk[0], key);                                                                  \
        ++lt;                                                                                                 \
    }
#define OPAQUE(X) asm volatile("" : "+v"(X));   /* identical MFMAs are merged by the compiler: every issue gets an operand it cannot see through */
/* one real reader + a whole-tuple use (no instruction): without it the allocator overlaps the dead lanes of the accumulator sets */
#define CONSUME(ACC) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(tm) : "v"(ACC[0]), "v"(ACC[9])); asm volatile("" :: "v"(ACC));
#define FILL(FIRSTW)                                                                                          \
    _Pragma("unroll") for (int w = FIRSTW; w < NV; ++w) VOP(OP, v[w & 7], v[8 + (w & 7)], v[(w + 3) & 7]);

    if (KIND == K_MFMA || KIND == K_BOTH) {
        for (int it = 0; it < iters; ++it) {
#define STEP(ISSUE, OLD)                                                                                      \
            OPAQUE(a)                                                                                         \
            ISSUE = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);                              \
            CONSUME(OLD)                                                                                      \
            if (KIND == K_BOTH) { FILL(1) }
            STEP(acc2, acc0) STEP(acc0, acc1) STEP(acc1, acc2) STEP(acc2, acc0) STEP(acc0, acc1) STEP(acc1, acc2)
#undef STEP
        }
    } else if (KIND == K_VALU) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u) { FILL(0) }
        }
    } else if (KIND == K_ROLES) {
        if (vrole) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 6; ++u) { FILL(0) }
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#define STEP(ISSUE, OLD)                                                                                      \
                OPAQUE(a)                                                                                     \
                ISSUE = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);                          \
                CONSUME(OLD)
                STEP(acc2, acc0) STEP(acc0, acc1) STEP(acc1, acc2) STEP(acc2, acc0) STEP(acc0, acc1) STEP(acc1, acc2)
#undef STEP
            }
        }
    } else if (KIND == K_FOLD || KIND == K_FOLD_BEFORE) {
        h8 an = pa[0];
        for (int it = 0; it < iters; ++it) {
#define STEP(ISSUE, FOLD, ODD, OFF)                                                                           \
            if (KIND == K_FOLD_BEFORE) {                                                                      \
                if (PRIO) __builtin_amdgcn_s_setprio(1);                                                      \
                FOLD8(FOLD, !(ODD))                                                                           \
                if (ODD) TRACK()                                                                              \
                if (PRIO) __builtin_amdgcn_s_setprio(0);                                                      \
                ISSUE = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, b, zero, 0, 0, 0);                         \
                an = pa[(OFF) * 64];                                                                          \
            } else {                                                                                          \
                if (PRIO) __builtin_amdgcn_s_setprio(0);                                                      \
                ISSUE = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, b, zero, 0, 0, 0);                         \
                an = pa[(OFF) * 64];                                                                          \
                if (PRIO) __builtin_amdgcn_s_setprio(1);                                                      \
                FOLD8(FOLD, !(ODD))                                                                           \
                if (ODD) TRACK()                                                                              \
            }                                                                                                 \
            if (!PRIO) __builtin_amdgcn_sched_barrier(0);
            STEP(acc2, acc0, 0, 1) STEP(acc0, acc1, 1, 2) STEP(acc1, acc2, 0, 3)
            STEP(acc2, acc0, 1, 4) STEP(acc0, acc1, 0, 5) STEP(acc1, acc2, 1, 6)
#undef STEP
            pa = lds + lane + (((it + 1) * 6 * 64) & 2047);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    } else if (KIND == K_GRP4) {
        // one wave, four query groups (B operands b, b1, b2, b3): per candidate block ONE ds_read_b128 and four MFMAs, each folded
        // into its group's running minimum four MFMAs later; the tracking of a group every other block.  6 "tiles" per
        // iteration would not divide by four: 12 tiles (3 candidate blocks) per iteration, reported per tile all the same.
        float fkg[4][4];
        for (int g = 0; g < 4; ++g) for (int s = 0; s < 4; ++s) fkg[g][s] = 1e30f;
        h8 an = pa[0];
        for (int it = 0; it < iters; it += 2) {
#define FOLDG(ACC, G, FIRST)                                                                                  \
            {                                                                                                 \
                const float t0_ = min3f(ACC[0], ACC[1], ACC[2]), t1_ = min3f(ACC[3], ACC[4], ACC[5]);         \
                const float t2_ = min3f(ACC[6], ACC[7], ACC[8]), t3_ = min3f(ACC[9], ACC[10], ACC[11]);       \
                const float t4_ = min3f(ACC[12], ACC[13], ACC[14]);                                           \
                const float t5_ = min3f(t0_, t1_, t2_), t6_ = min3f(t3_, t4_, ACC[15]);                       \
                tmg[G] = (FIRST) ? vmin(t5_, t6_) : min3f(tmg[G], t5_, t6_);                       \
            }
#define TRACKG(G)                                                                                             \
            {                                                                                                 \
                float key;                                                                                    \
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(tmg[G]), "v"(keymask), "s"(lt));          \
                fkg[G][3] = __builtin_amdgcn_fmed3f(fkg[G][2], fkg[G][3], key);                               \
                fkg[G][2] = __builtin_amdgcn_fmed3f(fkg[G][1], fkg[G][2], key);                               \
                fkg[G][1] = __builtin_amdgcn_fmed3f(fkg[G][0], fkg[G][1], key);                               \
                fkg[G][0] = vmin(fkg[G][0], key);                                                  \
            }
            // block A (even): first of its lane tile; block B (odd): closes it
#define BLK(ODD, OFF)                                                                                         \
            {                                                                                                 \
                const h8 ac = an;                                                                             \
                an = pa[(OFF) * 64];                                                                          \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac, b, zero, 0, 0, 0);                          \
                FOLDG(acc2, 2, !(ODD)) if (ODD) TRACKG(2)                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                            \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac, b1, zero, 0, 0, 0);                         \
                FOLDG(acc3, 3, !(ODD)) if (ODD) { TRACKG(3) ++lt; }                                           \
                __builtin_amdgcn_sched_barrier(0);                                                            \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac, b2, zero, 0, 0, 0);                         \
                FOLDG(acc0, 0, (ODD)) if (!(ODD)) TRACKG(0)                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                            \
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac, b3, zero, 0, 0, 0);                         \
                FOLDG(acc1, 1, (ODD)) if (!(ODD)) TRACKG(1)                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                            \
            }
            BLK(0, 1) BLK(1, 2) BLK(0, 3)
            pa = lds + lane + (((it + 2) * 3 * 64) & 2047);
#undef BLK
        }
        for (int g = 0; g < 4; ++g) for (int s = 0; s < 4; ++s) tm = __builtin_fminf(tm, fkg[g][s]);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = tm + fk[0] + fk[1] + fk[2] + fk[3] + tmg[0] + tmg[1] + tmg[2] + tmg[3];
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i] + acc2[i] + acc3[i];
    if (s == 1234.5f) out[0] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

static float *g_out;
static unsigned long long *g_cyc;

template <int KIND, int NV, int OP, int PRIO, int ROLE_BY>
void run(const char *name, int W) {
    const int iters = 4000, nblk = 256;
    const size_t shmem = 96 * 1024;
    auto kern = k<KIND, NV, OP, PRIO, ROLE_BY>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipMemset(g_cyc, 0, 8 * 16 * nblk);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256 * W), shmem, 0, g_out, g_cyc, 64, 0.37f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256 * W), shmem, 0, g_out, g_cyc, iters, 0.37f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(16 * nblk);
    (void)hipMemcpy(h.data(), g_cyc, 8 * 16 * nblk, hipMemcpyDeviceToHost);
    // per wave cycles; for the role kernels MFMA waves and VALU waves are reported apart
    double sum[2] = {0, 0}, mx[2] = {0, 0}; int n[2] = {0, 0};
    for (int bI = 0; bI < nblk; ++bI)
        for (int w = 0; w < 4 * W; ++w) {
            const int r = KIND == K_ROLES ? (ROLE_BY ? ((w >> 2) & 1) : (w & 1)) : 0;
            const double c = (double)h[bI * 16 + w];
            sum[r] += c; mx[r] = std::max(mx[r], c); ++n[r];
        }
    const double tiles = (double)iters * 6;
    // tiles a SIMD retires per wave-duration: every wave of the SIMD does `tiles` (role kernels: half of the waves each kind)
    const double wps = KIND == K_ROLES ? (ROLE_BY ? W / 2.0 : (double)W) : (double)W;
    printf("%-34s W=%d NV=%2d prio=%d | wall %.3f ms | cyc/tile/SIMD %6.1f (avg wave) %6.1f (slowest wave)", name, W, NV, PRIO, ms,
           sum[0] / n[0] / tiles / wps, mx[0] / tiles / wps);
    if (KIND == K_ROLES) printf(" | VALU waves: %6.1f / %6.1f", sum[1] / n[1] / tiles / wps, mx[1] / tiles / wps);
    printf(" | tick rate %.2f GHz\n", mx[0] > mx[1] ? mx[0] / (ms * 1e6) : mx[1] / (ms * 1e6));
    fflush(stdout);
}

int main() {
    (void)hipMalloc(&g_out, 64); (void)hipMalloc(&g_cyc, 8 * 16 * 256);
    for (int W : {1, 2, 4}) {
        run<K_MFMA, 1, OP_MIN3, 0, 0>("mfma only (+1 consuming min3)", W);
        run<K_VALU, 8, OP_MIN3, 0, 0>("valu only, v_min3", W);
        run<K_VALU, 11, OP_MIN3, 0, 0>("valu only, v_min3", W);
        run<K_VALU, 8, OP_MIN2, 0, 0>("valu only, v_min", W);
        run<K_BOTH, 3, OP_MIN3, 0, 0>("mfma + unrelated v_min3", W);
        run<K_BOTH, 5, OP_MIN3, 0, 0>("mfma + unrelated v_min3", W);
        run<K_BOTH, 7, OP_MIN3, 0, 0>("mfma + unrelated v_min3", W);
        run<K_BOTH, 9, OP_MIN3, 0, 0>("mfma + unrelated v_min3", W);
        run<K_BOTH, 11, OP_MIN3, 0, 0>("mfma + unrelated v_min3", W);
        run<K_BOTH, 7, OP_MIN2, 0, 0>("mfma + unrelated v_min", W);
        run<K_BOTH, 11, OP_MIN2, 0, 0>("mfma + unrelated v_min", W);
        run<K_FOLD, 0, OP_MIN3, 0, 0>("nn1 loop (fold after issue)", W);
        run<K_FOLD, 0, OP_MIN3, 1, 0>("nn1 loop (fold after issue)", W);
        run<K_FOLD_BEFORE, 0, OP_MIN3, 0, 0>("nn1 loop (fold before issue)", W);
        run<K_FOLD_BEFORE, 0, OP_MIN3, 1, 0>("nn1 loop (fold before issue)", W);
        run<K_GRP4, 0, OP_MIN3, 0, 0>("four query groups per wave", W);
        if (W >= 2) {
            run<K_ROLES, 8, OP_MIN3, 0, 1>("roles inside a SIMD, v_min3", W);
            run<K_ROLES, 11, OP_MIN3, 0, 1>("roles inside a SIMD, v_min3", W);
            run<K_ROLES, 8, OP_MIN3, 0, 0>("roles on separate SIMDs, v_min3", W);
        }
    }
    return 0;
}
