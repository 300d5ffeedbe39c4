// Shared by the translation units of the k-NN kernels (knn.hip: wave / selection / gather / EdgeConv-feature kernels, the
// dispatch and the C entry points; knn_d3.hip: knn_f16_d3_kernel; knn_mfma.hip: the feature-space pre-pass + knn_mfma_kernel):
// constants, the wave-level sorting / selection helpers, knn_exact_bruteforce, knn_rank_ties / knn_rank_ties4, knn_tau_8of16, and the
// declarations of what the units call across.  (File-local helpers live in an anonymous namespace: every unit has its copy.)
#pragma once
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "fx3d_common.h"

using namespace fx3d;

// Phase timestamps for tools/knn_probe.hip (compiled out of the product build).
#ifdef FX3D_PROBE
__device__ unsigned long long g_kprobe[4096 * 32];
#define KNN_PROBE_MARK(k)                                                                              \
    do {                                                                                               \
        const int pb__ = blockIdx.x + gridDim.x * blockIdx.y;                                          \
        if (threadIdx.x == 0 && pb__ < 4096) g_kprobe[pb__ * 32 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#ifdef FX3D_PROBE_STATS  // (same-address atomics: distorts the timings)
#define KNN_PROBE_STAT(i, v) atomicAdd(&g_kprobe[4095 * 32 + (i)], (unsigned long long)(v))
#else
#define KNN_PROBE_STAT(i, v) do { } while (0)
#endif
#else
#define KNN_PROBE_MARK(k) do { } while (0)
#define KNN_PROBE_STAT(i, v) do { } while (0)
#endif

namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------
// knn_wave_d3_kernel: one WAVE per query (D = 3).  The 64 lanes split the candidates (16 per lane and
// 1024-candidate chunk, held in registers and reused for every query of the wave), so a cloud of
// 1024 points keeps 4 waves per SIMD busy where the thread-per-query kernel had half a wave.
//   per query and chunk:
//     1. 16 exact distances per lane (the oracle's unfused Float32 form)
//     2. threshold tau: the current kk-th best; for the first chunk the kk-th smallest of the 64
//        lane minima (an upper bound of the kk-th smallest overall, typically admitting ~1.2 kk points)
//     3. candidates with d <= tau are compacted (ballot + mbcnt) into a per-wave LDS list
//     4. list + current best list (<= 64 keys, one per lane) are sorted by a 64-lane bitonic network
//        on the key (distance, index) -- exactly the reference ordering -- and the first kk survive.
// Output is bit-identical to fx3d_oracle_knn (same arithmetic, same (distance, index) order).
constexpr int kWQ = 8;           // queries per wave
constexpr int kWThreads = 256;   // 4 waves

// Order of the exact selection paths = the oracle's (oracle/flux3d_oracle.c: fless): Julia's isless on the Float32
// squared distance -- ascending, every NaN after +Inf, all NaNs equal -- then the lower index.  A squared distance is
// >= +0 or NaN, so with NaNs made canonical this is the UNSIGNED order of the bit patterns: the wave-per-query kernels
// and the brute-force fallback below keep distances as such keys (kNoKey = "no candidate", above every real key) and
// compare them as integers -- the same instructions as the float compares, and non-finite data needs no special case.
constexpr unsigned int kNoKey = 0xffffffffu;
__device__ __forceinline__ unsigned int dist_key(float d) { return d != d ? 0x7fc00000u : __builtin_bit_cast(unsigned int, d); }
__device__ __forceinline__ float key_dist(unsigned int k) { return __builtin_bit_cast(float, k); }
__device__ __forceinline__ bool key_less(unsigned int d, int j, unsigned int od, int oj) { return d < od || (d == od && j < oj); }

// ascending bitonic sort of one (key, j) pair per lane
// lane ^ S exchange for the sorting networks: DPP quad permutes for S = 1, 2 (VALU speed), ds_swizzle in bit mode for S = 4, 8, 16
// (no address register, half the latency of ds_bpermute), ds_bpermute for S = 32.  The networks below are 21 dependent stages,
// eleven of them at S <= 2: the medium / slow paths of clustered or tied data spend most of their time here.
template <int S>
__device__ __forceinline__ int xor_lane(int v) {
    if (S == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    if (S == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    if (S == 4 || S == 8 || S == 16) return __builtin_amdgcn_ds_swizzle(v, (S << 10) | 0x1F);  // and 0x1f, or 0, xor S
    return __shfl_xor(v, S, 64);
}
__device__ __forceinline__ void bitonic64(unsigned int &d, int &j, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int s = k >> 1; s > 0; s >>= 1) {
            const unsigned int od = (unsigned int)(s == 1 ? xor_lane<1>((int)d) : s == 2 ? xor_lane<2>((int)d) : s == 4 ? xor_lane<4>((int)d) :
                                                   s == 8 ? xor_lane<8>((int)d) : s == 16 ? xor_lane<16>((int)d) : xor_lane<32>((int)d));
            const int oj = s == 1 ? xor_lane<1>(j) : s == 2 ? xor_lane<2>(j) : s == 4 ? xor_lane<4>(j) : s == 8 ? xor_lane<8>(j) :
                           s == 16 ? xor_lane<16>(j) : xor_lane<32>(j);
            const bool up = (lane & k) == 0 || k == 64;   // final merge: ascending everywhere
            const bool lower = (lane & s) == 0;
            const bool take_min = lower == up;
            const bool o_less = key_less(od, oj, d, j);
            const bool swap = take_min ? o_less : !o_less && !(od == d && oj == j);
            d = swap ? od : d;
            j = swap ? oj : j;
        }
    }
}
__device__ __forceinline__ void bitonic64u(unsigned int &v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int s = k >> 1; s > 0; s >>= 1) {
            const unsigned int o = (unsigned int)(s == 1 ? xor_lane<1>((int)v) : s == 2 ? xor_lane<2>((int)v) : s == 4 ? xor_lane<4>((int)v) :
                                                  s == 8 ? xor_lane<8>((int)v) : s == 16 ? xor_lane<16>((int)v) : xor_lane<32>((int)v));
            const bool up = (lane & k) == 0 || k == 64;
            const bool lower = (lane & s) == 0;
            v = (lower == up) ? (v < o ? v : o) : (v > o ? v : o);
        }
    }
}
__device__ __forceinline__ unsigned int readlane_u(unsigned int v, int l) { return (unsigned int)__builtin_amdgcn_readlane((int)v, l); }

// ---- shared by the matrix-core kNN kernels ------------------------------------------------------------
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kMWaves = 4;                       // consumer waves
constexpr int kMThreads = 2 * kMWaves * 64;      // + as many producer waves
constexpr int kMProd = kMWaves * 64;             // producer threads

// plain v_min_f32 (fminf() adds canonicalising v_max ops; a NaN filter value only sends the query down
// the exact path through its non-finite threshold)
__device__ __forceinline__ float vmin_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The filter loops consume MFMA results with inline asm, which the compiler's hazard recogniser does not look
// into (an 8-pass MFMA's result may be read 11 issue slots after its issue at the earliest).  mfma_settle()
// marks the point where the listed accumulators have been issued and spends four slots; the consumers are
// `asm volatile`, so they stay behind it and in program order, and each loop reads the accumulator issued last
// only after sixteen other consumers.
#define KNN_MFMA_SETTLE2(a, b) asm volatile("s_nop 3" : "+v"(a), "+v"(b))
#define KNN_MFMA_SETTLE4(a, b, c, d) asm volatile("s_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
__device__ __forceinline__ float vmin_acc(float a, float b) {  // v_min_f32 on an MFMA result (ordered)
    float r;
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// branch-free (distance, index) comparison, 0 / 1
__device__ __forceinline__ int key_less_bf(float d, int j, float od, int oj) {
    return (int)(d < od) | ((int)(d == od) & (int)(j < oj));
}

// Exact top-kk of ONE query by the whole wave -- the fallback of the matrix-core kernels (queries whose filter is
// unusable or whose survivor lists overflow: heavy ties, degenerate clouds).  Same scheme as knn_wave_d3_kernel:
// per 1024-candidate chunk 16 exact distances per lane, threshold = kk-th smallest lane minimum (first chunk) or the
// current kk-th best, candidates <= threshold compacted into a 64-entry LDS list (flushed into the best list
// whenever it is full), one 64-lane bitonic sort on (distance, index) per merge.  kk <= 63; FULL64 instantiations also take
// kk = 64 (no room for a pending list: 64 candidates at a time are sorted and merged into the best list).
// lst_d / lst_j: 64 floats / ints of wave-private LDS.  Result: lanes 0..kk-1 hold the answer in order.
// ids == nullptr: all M candidates; else the M candidates ids[0..M) (LDS): a query's own survivors when they exceed the
// fast path's key capacity.
template <bool FULL64 = false>
__device__ __forceinline__ void knn_exact_bruteforce(const float *__restrict__ q, const float *__restrict__ yb, int M,
                                                      int D, int kk, int lane, float *lst_f, int *lst_j, float &bd_out,
                                                      int &bj, const int *ids = nullptr) {
    // distances as canonical unsigned keys: the isless order of the oracle, non-finite data included (see dist_key)
    unsigned int *lst_d = reinterpret_cast<unsigned int *>(lst_f);
    unsigned int bd = kNoKey;
    bj = 0x7fffffff;
    const int cap = 64 - kk;
    if (FULL64 && cap == 0) {
#pragma unroll 1
        for (int j0 = 0; j0 < M; j0 += 64) {
            const int j = j0 + lane;
            unsigned int nd = kNoKey;
            int nj = 0x7fffffff;
            if (j < M) {
                nj = ids ? ids[j] : j;
                const float *c = yb + (size_t)nj * D;
                float s = 0.0f;
                for (int dd = 0; dd < D; ++dd) {
                    const float t = q[dd] - c[dd];
                    s = s + t * t;
                }
                nd = dist_key(s);
            }
            bitonic64(nd, nj, lane);                                             // ascending new batch
            const unsigned int rd = (unsigned int)__shfl((int)nd, 63 - lane, 64);  // reversed
            const int rj = __shfl(nj, 63 - lane, 64);
            const bool o_less = key_less(rd, rj, bd, bj);
            bd = o_less ? rd : bd;                                               // lower half of the union (bitonic)
            bj = o_less ? rj : bj;
            bitonic64(bd, bj, lane);
        }
        bd_out = key_dist(bd);
        return;
    }
    const bool vec4 = (D & 3) == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(yb)) & 15) == 0;
    for (int j0 = 0; j0 < M; j0 += 1024) {
        unsigned int d[16];
        unsigned int lmin = kNoKey;
        // (a short list -- a query's own survivors -- fills only the first sweeps: the others are skipped, wave-uniformly)
        const int nsw = (M - j0 + 63) / 64 < 16 ? (M - j0 + 63) / 64 : 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = kNoKey;
        if (vec4 && nsw <= 2) {
            // a short list (a query's own survivors, <= 128): two candidates per lane, eight 16-byte steps of both rows in
            // flight -- with one step at a time the D / 4 steps were a chain of L2 round trips (~7 us per query at D = 64)
            const float *c[2];
            float sacc[2] = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + lane + 64 * u;
                const int jc = j < M ? j : M - 1;
                c[u] = yb + (size_t)(ids ? ids[jc] : jc) * D;
            }
#pragma unroll 8
            for (int dd = 0; dd < D; dd += 4) {
                const float4 qv = *reinterpret_cast<const float4 *>(q + dd);
                float4 cv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) cv[u] = *reinterpret_cast<const float4 *>(c[u] + dd);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float t0 = qv.x - cv[u].x, t1 = qv.y - cv[u].y, t2 = qv.z - cv[u].z, t3 = qv.w - cv[u].w;
                    sacc[u] = sacc[u] + t0 * t0;
                    sacc[u] = sacc[u] + t1 * t1;
                    sacc[u] = sacc[u] + t2 * t2;
                    sacc[u] = sacc[u] + t3 * t3;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                d[u] = j0 + lane + 64 * u < M ? dist_key(sacc[u]) : kNoKey;
                lmin = lmin < d[u] ? lmin : d[u];
            }
        } else if (vec4) {
            // rows as 16-byte pieces, four candidates in flight (dimension order kept: x, y, z, w of every piece)
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += 4) {
                if (i0 >= nsw) continue;
                const float *c[4];
                float sacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + lane + 64 * (i0 + u);
                    const int jc = j < M ? j : M - 1;
                    c[u] = yb + (size_t)(ids ? ids[jc] : jc) * D;
                }
#pragma unroll 2
                for (int dd = 0; dd < D; dd += 4) {  // (two steps' loads in flight: the loop is a chain of L2 round trips otherwise)
                    const float4 qv = *reinterpret_cast<const float4 *>(q + dd);
                    float4 cv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) cv[u] = *reinterpret_cast<const float4 *>(c[u] + dd);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float t0 = qv.x - cv[u].x, t1 = qv.y - cv[u].y, t2 = qv.z - cv[u].z, t3 = qv.w - cv[u].w;
                        sacc[u] = sacc[u] + t0 * t0;
                        sacc[u] = sacc[u] + t1 * t1;
                        sacc[u] = sacc[u] + t2 * t2;
                        sacc[u] = sacc[u] + t3 * t3;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    d[i0 + u] = j0 + lane + 64 * (i0 + u) < M ? dist_key(sacc[u]) : kNoKey;
                    lmin = lmin < d[i0 + u] ? lmin : d[i0 + u];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = j0 + lane + 64 * i;
                if (i < nsw && j < M) {
                    const float *c = yb + (size_t)(ids ? ids[j] : j) * D;
                    float s = 0.0f;
                    for (int dd = 0; dd < D; ++dd) {
                        const float t = q[dd] - c[dd];
                        s = s + t * t;
                    }
                    d[i] = dist_key(s);
                }
                lmin = lmin < d[i] ? lmin : d[i];
            }
        }
        unsigned int tau = readlane_u(bd, kk - 1);
        if (j0 == 0) {  // kk-th smallest lane minimum bounds the kk-th smallest distance
            unsigned int v = lmin;
            bitonic64u(v, lane);
            tau = readlane_u(v, kk - 1);
        }
        int cnt = 0;
#pragma unroll  // (unrolled: d[] stays in registers -- indexed dynamically it lives in scratch memory, ~10 us per query)
        for (int i = 0; i < 16; ++i) {
            if (i >= nsw) break;
            bool pred = d[i] <= tau && d[i] != kNoKey;
            unsigned long long bal = __ballot(pred);
            while (bal) {  // usually one pass; more only when > cap candidates qualify
                const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32),
                                      __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0));
                const bool put = pred && pos < cap;
                if (put) { lst_d[pos] = d[i]; lst_j[pos] = ids ? ids[j0 + lane + 64 * i] : j0 + lane + 64 * i; }
                const int np = __builtin_popcountll(bal);
                const bool overflow = cnt + np > cap;
                cnt = overflow ? cap : cnt + np;
                pred = pred && !put;
                if (overflow) {  // flush: merge the full list into the best list, tighten tau
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    unsigned int sd = lane < kk ? bd : (lane - kk < cnt ? lst_d[lane - kk] : kNoKey);
                    int sj = lane < kk ? bj : (lane - kk < cnt ? lst_j[lane - kk] : 0x7fffffff);
                    bitonic64(sd, sj, lane);
                    bd = lane < kk ? sd : kNoKey;
                    bj = lane < kk ? sj : 0x7fffffff;
                    tau = readlane_u(sd, kk - 1);
                    cnt = 0;
                    pred = pred && d[i] <= tau;
                }
                bal = __ballot(pred);
            }
        }
        if (cnt > 0) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            unsigned int sd = lane < kk ? bd : (lane - kk < cnt ? lst_d[lane - kk] : kNoKey);
            int sj = lane < kk ? bj : (lane - kk < cnt ? lst_j[lane - kk] : 0x7fffffff);
            bitonic64(sd, sj, lane);
            bd = lane < kk ? sd : kNoKey;
            bj = lane < kk ? sj : 0x7fffffff;
        }
    }
    bd_out = key_dist(bd);
}

// LDS image of a chunk: rows of PPR = DP/4 16-byte pieces, no padding; piece c of row r sits at position
// (c + r) mod PPR of its row.  The rotation makes the consumers' b128 operand fetches (32 consecutive rows, one
// column) conflict-free, and it is applied on the SOURCE side of the direct-to-LDS loads
// (global_load_lds_dwordx4 writes lane-linear: wave-uniform base + lane*16), so staging costs one
// instruction per KiB and no VGPR round trip -- the producers share their SIMD's issue slots with the
// consumers' MFMAs, every VALU instruction they do not execute is matrix-core time.
// knn_gather_kernel with 16-byte elements (F4 = F/4 float4 per row)
// NT: streaming (non-temporal) stores for tensors beyond the caches (round 4: the F = 64 feature build gained 28 % from them).
// Only for outputs larger than 3/4 of the 256 MB Infinity Cache: a consumer kernel may still find a smaller tensor there (the
// 168 MB gather of C4' gains 3 % from streaming stores -- not worth taking that away from its reader).
template <int DK>
__device__ __forceinline__ int knn_piece_off(int row, int c) {  // float offset of piece c of row `row`
    constexpr int PPR = DK * 8;
    return (row * PPR + ((c + row) & (PPR - 1))) * 4;
}

// producer wave pw stages rows [pw*RW, (pw+1)*RW) of the chunk [j0, j0+cn) and their norms
template <int DK>
__device__ __forceinline__ void knn_stage_chunk(const float *__restrict__ yb, int D, int j0, int cn, int CH, float *img,
                                                float *cnorm, unsigned int *cmax, bool want_cmax, bool do_norms,
                                                bool vec4, int pw, int lane) {
    constexpr int PPR = DK * 8;
    const int RW = CH / kMWaves;          // rows per producer wave (CH is a multiple of 64)
    const int row_lo = pw * RW;
    const int rq = D / 4;
    if (vec4) {
        const int ninstr = RW * PPR / 64;
        for (int i = 0; i < ninstr; ++i) {
            const int S0 = row_lo * PPR + i * 64;  // first 16-byte slot of this wave-instruction
            const int S = S0 + lane;
            const int row = S / PPR, pos = S & (PPR - 1);
            const int c = (pos - row) & (PPR - 1);
            if (row < cn && c < rq)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(yb + (size_t)(j0 + row) * D + 4 * c),
                    (__attribute__((address_space(3))) void *)(img + (size_t)S0 * 4), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces have landed
    } else {
        for (int e = lane; e < RW * D; e += 64) {
            const int row = row_lo + e / D, d = e % D;
            if (row < cn) img[knn_piece_off<DK>(row, d >> 2) + (d & 3)] = yb[(size_t)(j0 + row) * D + d];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    __builtin_amdgcn_wave_barrier();
    if (!do_norms) return;  // phase B with the norms of phase A kept in LDS
    // norms of this wave's rows (padding columns hold zeros); rows beyond the chunk get +inf: F = +inf
    float wmax = 0.0f;
    bool wnan = false;
    for (int r0 = 0; r0 < RW; r0 += 64) {
        const int row = row_lo + r0 + lane;
        if (r0 + lane < RW) {
            float t = INFINITY;
            if (row < cn) {
                t = 0.0f;
#pragma unroll
                for (int c = 0; c < PPR; ++c) {
                    const float4 v = *reinterpret_cast<const float4 *>(img + knn_piece_off<DK>(row, c));
                    t = __builtin_fmaf(v.x, v.x, t);
                    t = __builtin_fmaf(v.y, v.y, t);
                    t = __builtin_fmaf(v.z, v.z, t);
                    t = __builtin_fmaf(v.w, v.w, t);
                }
                wnan |= (t != t);
                wmax = fmaxf(wmax, t);
            }
            cnorm[row] = t;
        }
    }
    if (want_cmax) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, m, 64));
        const bool anynan = __ballot(wnan) != 0;
        if (lane == 0) atomicMax(cmax, anynan ? 0x7fc00000u : __builtin_bit_cast(unsigned int, wmax));  // NaN > +inf
    }
}

// Wave-cooperative ranking of ONE query's n survivors on the full (distance bits, index) keys: lane e ranks key e against all n (LDS
// broadcast reads; qd / qj are padded with sentinels up to a multiple of four); keys are unique, so the ranks below kk are a
// permutation and slots[0, kk) is the sorted answer.  The form for the rare tied query of ordinary data: well under a microsecond for
// its wave -- the launch is one round of blocks and waits for it (knn_rank_ties4 below takes ~2.4 us for a single query: measured as
// + 3 us on C4's kernel when it served every case).
__device__ __forceinline__ void knn_rank_ties(const unsigned int *qd, const int *qj, int n, int kk,
                                              unsigned long long *slots, int lane) {
    for (int e = lane; e < n; e += 64) {
        const unsigned int md = qd[e];
        const int mj = qj[e];
        int rank = 0;
        for (int i = 0; i < n; i += 4) {
            const uint4 od = *reinterpret_cast<const uint4 *>(qd + i);
            const int4 oj = *reinterpret_cast<const int4 *>(qj + i);
            rank += (int)(od.x < md) | ((int)(od.x == md) & (int)(oj.x < mj));
            rank += (int)(od.y < md) | ((int)(od.y == md) & (int)(oj.y < mj));
            rank += (int)(od.z < md) | ((int)(od.z == md) & (int)(oj.z < mj));
            rank += (int)(od.w < md) | ((int)(od.w == md) & (int)(oj.w < mj));
        }
        if (rank < kk) slots[rank] = ((unsigned long long)md << 32) | (unsigned int)mj;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}

// Ranking of a wave's tied queries on the full (distance bits, index) keys -- the path of a query whose distance-only ranks collide
// (an exact tie among its first kk): up to SIXTEEN queries at once, four lanes per query (pl = 0..3), every lane called with ITS
// query's key arrays (n = 0: no query).  A lane ranks entries pl, pl + 4, ... against all n (qd / qj are padded with sentinels up to
// a multiple of four); keys are unique, so the ranks below kk are a permutation and slots[0, kk) is the sorted answer.  Lattices and
// duplicated points tie in EVERY query: one query at a time (a wave per query, rounds 2-4) cost the wave 16 x (a chain of LDS round
// trips + the output stores) -- 23 / 28 us of the 53 / 59 us at C4's shape; here the chains overlap (17 us: what is left is the
// n^2 compares of four VALU each, run to the longest list of the wave).
__device__ __forceinline__ void knn_rank_ties4(const unsigned int *qd, const int *qj, int n, int kk, unsigned long long *slots, int pl) {
    for (int e = pl; e < n; e += 8) {  // two entries of this lane per pass over the keys
        const bool two = e + 4 < n;
        const unsigned int md0 = qd[e], md1 = two ? qd[e + 4] : 0xffffffffu;
        const int mj0 = qj[e], mj1 = two ? qj[e + 4] : 0x7fffffff;
        int r0 = 0, r1 = 0;
        for (int i = 0; i < n; i += 4) {
            const uint4 od = *reinterpret_cast<const uint4 *>(qd + i);
            const int4 oj = *reinterpret_cast<const int4 *>(qj + i);
            r0 += (int)(od.x < md0) | ((int)(od.x == md0) & (int)(oj.x < mj0));
            r0 += (int)(od.y < md0) | ((int)(od.y == md0) & (int)(oj.y < mj0));
            r0 += (int)(od.z < md0) | ((int)(od.z == md0) & (int)(oj.z < mj0));
            r0 += (int)(od.w < md0) | ((int)(od.w == md0) & (int)(oj.w < mj0));
            r1 += (int)(od.x < md1) | ((int)(od.x == md1) & (int)(oj.x < mj1));
            r1 += (int)(od.y < md1) | ((int)(od.y == md1) & (int)(oj.y < mj1));
            r1 += (int)(od.z < md1) | ((int)(od.z == md1) & (int)(oj.z < mj1));
            r1 += (int)(od.w < md1) | ((int)(od.w == md1) & (int)(oj.w < mj1));
        }
        if (r0 < kk) slots[r0] = ((unsigned long long)md0 << 32) | (unsigned int)mj0;
        if (two && r1 < kk) slots[r1] = ((unsigned long long)md1 << 32) | (unsigned int)mj1;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}


typedef _Float16 kh8 __attribute__((ext_vector_type(8)));
// Geometry of one instantiation: G query groups (32 queries each) per block, two waves per group (each taking every other pair of
// candidate tiles: two waves per SIMD overlap each other's LDS / shuffle latencies), CAP rows per lane list (CAP - 1 usable + the
// scratch head; a lane sees half the tiles), KCAP keys per query (the four lanes' survivors; three sentinels follow them inside the
// stride KS: 32 queries x b128 reads without bank conflicts), KKMAX = the largest k + drop (SS - 1 rank slots per query).
template <int G_, int CAP_, int KCAP_, int KKMAX_>
struct K3Geom {
    static constexpr int G = G_, W = 2 * G_, T = W * 64, CAP = CAP_, KCAP = KCAP_, KS = KCAP_ + 4, KKMAX = KKMAX_, SS = KKMAX_ + 1;
    static_assert((size_t)W * 32 * 33 * 4 <= (size_t)W * CAP * 64 * 4, "the tau exchange aliases the lists");
    static_assert((size_t)G * 32 * SS * 8 + W * 128 * 4 <= (size_t)W * CAP * 64 * 4, "slots + scratch alias the lists");
    static_assert(CAP <= 64 && KKMAX <= 64 && KKMAX % 16 == 0 && KCAP % 4 == 0, "one list word per lane in the medium path; 16-byte key rows");
};
using K3Base = K3Geom<4, 24, 64, 32>;    // k + drop <= 32: C4 gets 256 blocks of 128 queries, one per CU
using K3Wide = K3Geom<2, 40, 128, 64>;   // 32 < k + drop <= 64: twice the keys and longer lists per query, half the queries per block
// ... and a compact one (round 3) for 32 < k + drop <= 48: an allocation below half a CU's LDS, so that TWO blocks (eight waves) share a
// CU as in the base geometry -- the wide geometry's four waves leave half of every CU's issue slots empty (C4's shape, k = 40:
// 52.4 -> 34.8 us); the LDS image is held to the size of the key arrays (1472 candidates; larger clouds pass through it in
// chunks), the raw coordinates stay in L2.  The same
// form of the base geometry (K3Geom<2, 24, 64, 32>, two blocks per CU) measured equal to it (k = 20: 24.8 vs 25.0 us): not kept.
using K3Mid = K3Geom<2, 28, 88, 48>;
// (48 < k + drop <= 64 as K3Geom<1, 35, 120, 64>, 32 queries per block and three blocks per CU, measured equal to the wide geometry
//  up to k = 56 -- four times the prologues -- and worse beyond, where 120 keys overflow: not kept)
// dynamic LDS limit of a compact block: NB of them (+ ~0.7 KiB static each) fit in a CU's 160 KiB
constexpr size_t k3_compact_lds(int nb) { return (size_t)160 * 1024 / nb - 1024; }
constexpr int kTChunk = 3072;         // candidates per LDS image (32 B each): image + lists + counters <= 152 KiB
constexpr int kTRawMax = 2048;        // clouds up to this size also keep their raw coordinates in LDS
constexpr int kK3FarCap = 16;         // far candidates (robust range, as in nn1_f16_kernel) kept on the exact side list


__device__ __forceinline__ float vmax_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void k3_split2h(float v, _Float16 &h, _Float16 &l) {
    h = (_Float16)v;
    l = (_Float16)(v - (float)h);
}
// fp16 image pieces of one candidate c~ = s (c - mu): same K-slot table as nn1_f16_kernel
constexpr float kK3BetaC = 0x1.2p-18f;  // candidate's share of the filter error, folded into its norm (chamfer.hip kBetaC, + the 17th term)
__device__ __forceinline__ void k3_make_pieces(float cx, float cy, float cz, kh8 &p0, kh8 &p1) {
    _Float16 hx, lx, hy, ly, hz, lz, n1, n2, n3;
    k3_split2h(cx, hx, lx); k3_split2h(cy, hy, ly); k3_split2h(cz, hz, lz);
    const float n0 = ((cx * cx) + (cy * cy)) + (cz * cz);
    const float n = n0 + kK3BetaC * n0;
    n1 = (_Float16)n;
    const float r1 = n - (float)n1;
    n2 = (_Float16)r1;
    n3 = (_Float16)(r1 - (float)n2);
    p0 = kh8{hx, hx, lx, hy, hy, ly, hz, hz};
    p1 = kh8{lz, n1, n2, n3, lx, ly, lz, (_Float16)1.0f};  // slot 15: times the query's -threshold in phase B
}
// ... of a candidate that may lie beyond the robust range (|c~|_inf >= 2^7): zero pieces, norm +inf (its filter value is
// +inf for every query), recorded once (first staging of its chunk) on the block's side list
__device__ __forceinline__ void k3_pieces_far(float sx, float sy, float sz, bool has_far, bool record, int index, int *nfar, int *farlist,
                                              kh8 &p0, kh8 &p1) {
    if (!has_far) { k3_make_pieces(sx, sy, sz, p0, p1); return; }
    const bool far = !(fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz)) < 128.0f);
    k3_make_pieces(far ? 0.f : sx, far ? 0.f : sy, far ? 0.f : sz, p0, p1);
    if (far) {
        p1[1] = (_Float16)INFINITY;
        if (record) {
            const int f = atomicAdd(nfar, 1);
            if (f < kK3FarCap) farlist[f] = index;
        }
    }
}
// ascending sort of NV registers (compile-time indices only): Batcher's odd-even merge sort, 191 compare-exchanges
// for 32 values (the bitonic network needs 240)
template <int NV>
__device__ __forceinline__ void k3_sort_regs(float (&v)[NV]) {
    static_assert((NV & (NV - 1)) == 0, "power of two");
#pragma unroll
    for (int p = 1; p < NV; p <<= 1) {
#pragma unroll
        for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
            for (int j = k % p; j <= NV - 1 - k; j += 2 * k) {
#pragma unroll
                for (int i = 0; i < k; ++i) {
                    if (i <= NV - j - k - 1 && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                        const float lo = vmin_f32(v[i + j], v[i + j + k]), hi = vmax_f32(v[i + j], v[i + j + k]);
                        v[i + j] = lo;
                        v[i + j + k] = hi;
                    }
                }
            }
        }
    }
}

// tau = an upper bound of the kk-th smallest filter value of every query (kk <= 24) from its 128 group minima -- 32 in this lane (mn), 32
// in its partner half-lane, 64 in the other wave of the pair (wave index +- GW) -- shared by knn_f16_d3_kernel and knn_mfma_kernel<PRE>:
// the EIGHT smallest of each HALF of a lane's group minima (16 of its 32) instead of a sort of all 32.  The kk-th smallest of any
// subset of the 128 group minima bounds the kk-th smallest filter value (every group minimum is some candidate's value); the subset
// {8 smallest of each of the query's eight half-lane sets of 16} holds the kk <= 24 smallest of all 128 unless one set holds more
// than 8 of them (kk = 21: Bin(21, 1/8) >= 9, 4e-4 per set, and tau is then the next order statistic).  (The 8 smallest of each
// LANE's 32 -- Bin(21, 1/4) -- was cheaper still but let one query in 10^4 end with 40+ survivors: the rank phase of its block
// doubled, and a one-round launch lasts as long as its slowest block: 21.9 -> 23.5 us.)  Four 19-exchange sorts of 8, two "8
// smallest of two sorted 8" steps (8 v_min + a 12-exchange bitonic merge), a 16-value merge, the partner lane's sixteen by
// v_permlane32_swap, a 32-value merge, the other wave's through LDS (xch: [2 GW][32][27] floats) and the split minimum: ~560 VALU per
// wave where the sort of 32 and two 32-value merges took ~860.  Contains a block barrier: every thread of the block calls it.
template <int GW>
__device__ __forceinline__ float knn_tau_8of16(const float (&mn)[32], float *xch, int wv, int jq, int hh, int kk) {
    float tau;
    {
    float a8[4][8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a8[g][i] = mn[8 * g + i];
            k3_sort_regs<8>(a8[g]);
        }
        auto low8 = [](float (&x)[8], const float (&y)[8]) {  // x <- the 8 smallest of two ascending octets, ascending
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = vmin_f32(x[i], y[7 - i]);  // bitonic
#pragma unroll
            for (int j = 4; j > 0; j >>= 1)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int l = i ^ j;
                    if (l > i) {
                        const float lo = vmin_f32(x[i], x[l]), hi = vmax_f32(x[i], x[l]);
                        x[i] = lo;
                        x[l] = hi;
                    }
                }
        };
        low8(a8[0], a8[1]);
        low8(a8[2], a8[3]);
        float x32[32];  // [0, 16): this lane's sixteen, ascending after the first merge; then the wave's 32
#pragma unroll
        for (int r = 0; r < 8; ++r) { x32[r] = a8[0][r]; x32[8 + r] = a8[2][7 - r]; }  // ascending then descending: bitonic
#pragma unroll
        for (int j = 8; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const float lo = vmin_f32(x32[i], x32[l]), hi = vmax_f32(x32[i], x32[l]);
                    x32[i] = lo;
                    x32[l] = hi;
                }
            }
        {
            float oth[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)  // x32[r] <- lanes 0-31's value, oth[r] <- lanes 32-63's, in every lane
                asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x32[r]), "=&v"(oth[r]));
#pragma unroll
            for (int r = 0; r < 16; ++r) x32[16 + r] = oth[15 - r];
        }
#pragma unroll
        for (int j = 16; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const float lo = vmin_f32(x32[i], x32[l]), hi = vmax_f32(x32[i], x32[l]);
                    x32[i] = lo;
                    x32[l] = hi;
                }
            }
        // exchange rows of 27 words per (wave, query): [0] = +inf, [1] = -inf, [2 + r] = the wave's r-th smallest, r < 24
        float *row = xch + (wv * 32 + jq) * 27;
        if (hh == 0) {
#pragma unroll
            for (int r = 0; r < 24; ++r) row[2 + r] = x32[r];
        } else {
            row[0] = INFINITY;
            row[1] = -INFINITY;
        }
        __syncthreads();
        // the kk-th smallest of the union of X = x32 and the other wave's Y (both ascending) without merging them: min
        // over the splits (i values from X, kk - i from Y) of max(X'[i-1], Y'[kk-i-1]), X'[-1] = Y'[-1] = -inf; a split with
        // i > kk reads +inf.  (The offsets depend on the runtime kk: computed here, behind an opaque copy -- hoisted to the
        // kernel's start they were 25 more long-lived scalars in a kernel that already spills SGPRs.)
        int kko = kk;
        asm volatile("" : "+s"(kko));
        const float *po = xch + (((wv + GW) % (2 * GW)) * 32 + jq) * 27;
        float yv[25];  // (all reads first)
#pragma unroll
        for (int i = 0; i <= 24; ++i) {
            const int o = kko - i + 1;
            yv[i] = po[o > 0 ? o : 0];
        }
        tau = yv[0];   // i = 0: X'[-1] = -inf
#pragma unroll
        for (int i = 1; i <= 24; ++i) tau = vmin_f32(tau, vmax_f32(x32[i - 1], yv[i]));
    }
    return tau;
}

// the shapes the matrix-core kernels take (pure functions of the shape: the dispatch, the scratch planner and the units agree)
// D = 3: k + drop <= 32: the base geometry; 33 ... 64: the wide one (both waves of a group bound tau by their own ceil(kk / 2)-th group minimum)
constexpr int kK3WideMinM = 128;  // every wave of a group needs >= 32 finite group minima: two pairs of tiles
__host__ inline bool knn_f16_d3_shape_ok(int M, int kk) {
    return M < (1 << 21) && (kk <= 32 ? M >= 64 : (kk <= 64 && M >= kK3WideMinM));
}
inline bool knn_mfma_eligible(int M, int D, int kk) { return D >= 4 && D <= 128 && kk <= 32 && M >= 64; }

}  // namespace

namespace fx3d {
// knn_d3.hip
fx3d_status knn_d3_launch(const float *x, int N, const float *y, int M, int B, int k, int drop, int32_t *idx, float *dist, hipStream_t st,
                          float *feat, int layout, int xdiv);
// knn_mfma.hip
fx3d_status knn_mfma_launch(const float *x, int N, const float *y, int M, int B, int D, int k, int drop, int32_t *idx, float *dist,
                            hipStream_t st, void *pre_ws, int xdiv);
bool knn_mfma_pre_shape_ok(int M, int D, int kk);
bool knn_mfma_pre_eligible(const float *x, const float *y, int M, int D, int kk);
size_t knn_mfma_pre_bytes(int M, int B, int D);
}  // namespace fx3d
