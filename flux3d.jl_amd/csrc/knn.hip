// k-NN graph kernels (gfx950).
//
// Replaces CreateSingleKNNGraph (src/models/dgcnn.jl:3-7) and its per-batch-element loop in
// EdgeConv (:36): B KD-tree builds + N*B sorted (K+1)-queries + N*B small gathers become one
// brute-force launch (+ one gather launch).  Ordering is (distance, index) ascending with the
// CPU path's arithmetic (Float32 sum of squared differences in dimension order, unfused), so the
// index lists are bit-identical to oracle/flux3d_oracle.c:fx3d_oracle_knn.
//
// Kernels, in file order (dispatch in launch_knn / fx3d_edgeconv_graph at the end of the file):
//   knn_wave_d3_kernel / knn_wave_generic_kernel   one wave per query, exact distances: every shape the two below do not take
//   knn_exact_bruteforce / knn_rank_ties           wave-cooperative exact selection / tie re-rank shared by all kernels
//   knn_gather[4]_kernel                           X[:, idx] (src/models/dgcnn.jl:6)
//   knn_select_kernel                              any k + drop <= M, any D (M <= 36864): all keys of a query in LDS, radix select;
//                                                  also the fallback of the verified slice merge (flagged queries only)
//   knn_tau_8of16                                  tau of the matrix-core kernels from 128 group minima per query (shared, round 4)
//   knn_f16_d3_kernel<FEAT, K3Geom>                D = 3: fp16-split matrix-core filter + exact re-scan in three geometries --
//                                                  base (k+drop <= 32, M >= 64), compact (<= 48, two blocks per CU), wide (<= 64);
//                                                  FEAT: EdgeConv's cat(X, KNN - X) written by the same kernel
//   knn_pre_*_kernel                               fx3d_knn_ws: per-cloud statistics + fp16 image built once (feature space)
//   knn_mfma_kernel<DK, F16, SPLIT, PRE>           4 <= D <= 128, k+drop <= 32, M >= 64: GEMM filter (fp16 rounded halves, 2-way
//                                                  fp16 split, or Float32) + exact re-scan, medium path for crowded bands
//   edge_features_*_kernel                         cat(X, KNN - X) + permute for any F, and the @nograd adjoint
//   knn_interleave_kernel / knn_merge_slices_kernel  fx3d_knn_ws: candidate slices as virtual clouds (few clouds with many rows;
//                                                  k+drop in 33 ... 128 in feature space: 32 nearest per slice, verified merge)
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

__global__ __launch_bounds__(kWThreads) void knn_wave_d3_kernel(const float *__restrict__ x, int N,
                                                                const float *__restrict__ y, int M, int B,
                                                                int k, int drop, int32_t *__restrict__ idx,
                                                                float *__restrict__ dist) {
    __shared__ unsigned int lst_d[kWThreads / 64][64];
    __shared__ int lst_j[kWThreads / 64][64];
    __shared__ unsigned int best_d[kWThreads / 64][kWQ][64];   // per-query best lists across chunks (distance keys)
    __shared__ int best_j[kWThreads / 64][kWQ][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kk = k + drop;
    const float *xb = x + (size_t)b * N * 3, *yb = y + (size_t)b * M * 3;
    const int q0 = (blockIdx.x * (kWThreads / 64) + wv) * kWQ;
    if (q0 >= N) return;  // wave-uniform; no block-level sync below
#pragma unroll
    for (int qq = 0; qq < kWQ; ++qq) { best_d[wv][qq][lane] = kNoKey; best_j[wv][qq][lane] = 0x7fffffff; }

    for (int j0 = 0; j0 < M; j0 += 1024) {
        float cx[16], cy[16], cz[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = j0 + lane + 64 * i;
            if (j < M) {
                cx[i] = yb[(size_t)j * 3]; cy[i] = yb[(size_t)j * 3 + 1]; cz[i] = yb[(size_t)j * 3 + 2];
            } else {
                cx[i] = 0.0f; cy[i] = 0.0f; cz[i] = 0.0f;  // beyond the cloud: key = kNoKey below, never selected
            }
        }
#pragma unroll 1
        for (int qq = 0; qq < kWQ; ++qq) {
            const int qi = q0 + qq;
            if (qi >= N) break;
            const float qx = xb[(size_t)qi * 3], qy = xb[(size_t)qi * 3 + 1], qz = xb[(size_t)qi * 3 + 2];  // uniform
            unsigned int d[16];
            unsigned int lmin = kNoKey;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float t0 = qx - cx[i], t1 = qy - cy[i], t2 = qz - cz[i];
                d[i] = j0 + lane + 64 * i < M ? dist_key(((t0 * t0) + (t1 * t1)) + (t2 * t2)) : kNoKey;
                lmin = lmin < d[i] ? lmin : d[i];
            }
            unsigned int bd = best_d[wv][qq][lane];
            int bj = best_j[wv][qq][lane];
            unsigned int tau = readlane_u(bd, kk - 1);
            if (j0 == 0) {  // kk-th smallest lane minimum bounds the kk-th smallest distance
                unsigned int v = lmin;
                bitonic64u(v, lane);
                tau = readlane_u(v, kk - 1);
            }
            int cnt = 0;
            const int cap = 64 - kk;  // list + best list must fit one key per lane
            if (cap > 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    bool pred = d[i] <= tau && d[i] != kNoKey;
                    unsigned long long bal = __ballot(pred);
                    while (bal) {  // usually one pass; more only when > cap candidates qualify
                        const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32),
                                              __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0));
                        const bool put = pred && pos < cap;
                        if (put) { lst_d[wv][pos] = d[i]; lst_j[wv][pos] = j0 + lane + 64 * i; }
                        const int np = __builtin_popcountll(bal);
                        const bool overflow = cnt + np > cap;
                        cnt = overflow ? cap : cnt + np;
                        pred = pred && !put;
                        if (overflow) {  // flush: merge the full list into the best list, tighten tau
                            unsigned int sd = lane < kk ? bd : (lane - kk < cnt ? lst_d[wv][lane - kk] : kNoKey);
                            int sj = lane < kk ? bj : (lane - kk < cnt ? lst_j[wv][lane - kk] : 0x7fffffff);
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
            }
            if (cnt > 0 || cap == 0) {
                unsigned int sd;
                int sj;
                if (cap > 0) {
                    sd = lane < kk ? bd : (lane - kk < cnt ? lst_d[wv][lane - kk] : kNoKey);
                    sj = lane < kk ? bj : (lane - kk < cnt ? lst_j[wv][lane - kk] : 0x7fffffff);
                    bitonic64(sd, sj, lane);
                    bd = sd; bj = sj;
                } else {
                    // kk == 64: no room for a list; merge the chunk 64 candidates at a time
#pragma unroll 1
                    for (int i = 0; i < 16; ++i) {
                        unsigned int nd = d[i];
                        int nj = nd != kNoKey ? j0 + lane + 64 * i : 0x7fffffff;
                        bitonic64(nd, nj, lane);                      // ascending new batch
                        const unsigned int rd = (unsigned int)__shfl((int)nd, 63 - lane, 64);   // reversed
                        const int rj = __shfl(nj, 63 - lane, 64);
                        const bool o_less = key_less(rd, rj, bd, bj);
                        bd = o_less ? rd : bd;                         // lower half of the union (bitonic)
                        bj = o_less ? rj : bj;
                        bitonic64(bd, bj, lane);
                    }
                }
            }
            best_d[wv][qq][lane] = bd;
            best_j[wv][qq][lane] = bj;
            if (j0 + 1024 >= M) {  // last chunk: lanes drop..drop+k-1 hold the answer
                const int r = lane - drop;
                if (r >= 0 && r < k) {
                    idx[((size_t)b * N + qi) * k + r] = bj;
                    if (dist) dist[((size_t)b * N + qi) * k + r] = key_dist(bd);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// knn_wave_generic_kernel: any D (the second EdgeConv runs kNN in 64-D feature space,
// src/models/dgcnn.jl:121).  Same wave-per-query selection as knn_wave_d3_kernel; the distance stage
// works on candidate tiles of kGT rows staged in LDS with a padded row stride (D+1 floats: lane j reads
// row j, bank (j+d)%32 -- conflict-free), each wave evaluating its kGQ queries against the tile (query
// values are wave-uniform: scalar loads).  Distances are the oracle's: s = s + t*t in dimension order.
// Per-query state (best list, pending list, count, threshold) lives in LDS across tiles; pending
// candidates are merged lazily (when the list is full, and once at the end).
constexpr int kGT = 128;   // candidates per tile (2 per lane)
constexpr int kGQ = 4;     // queries per wave

__global__ __launch_bounds__(kWThreads) void knn_wave_generic_kernel(const float *__restrict__ x, int N,
                                                                     const float *__restrict__ y, int M, int B,
                                                                     int D, int k, int drop,
                                                                     int32_t *__restrict__ idx,
                                                                     float *__restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) float gl[];
    constexpr int NW = kWThreads / 64;
    const int RS = D + 1;                       // padded row stride
    float *tile = gl;                           // [kGT][RS]
    unsigned int *bestd = reinterpret_cast<unsigned int *>(tile + kGT * RS);   // [NW][kGQ][64] distance keys
    int *bestj = reinterpret_cast<int *>(bestd + NW * kGQ * 64);
    unsigned int *lstd = reinterpret_cast<unsigned int *>(bestj + NW * kGQ * 64);
    int *lstj = reinterpret_cast<int *>(lstd + NW * kGQ * 64);
    unsigned int *taus = reinterpret_cast<unsigned int *>(lstj + NW * kGQ * 64);   // [NW][kGQ]
    int *cnts = reinterpret_cast<int *>(taus + NW * kGQ);            // [NW][kGQ]
    float4 *qs4 = reinterpret_cast<float4 *>(cnts + NW * kGQ);       // [NW][D] : the wave's kGQ=4 queries, interleaved

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kk = k + drop, cap = 64 - kk;
    const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
    const int q0 = (blockIdx.x * NW + wv) * kGQ;
#pragma unroll
    for (int qq = 0; qq < kGQ; ++qq) {
        bestd[(wv * kGQ + qq) * 64 + lane] = kNoKey;
        bestj[(wv * kGQ + qq) * 64 + lane] = 0x7fffffff;
        if (lane == 0) { taus[wv * kGQ + qq] = kNoKey; cnts[wv * kGQ + qq] = 0; }
    }
    for (int d = lane; d < D; d += 64) {  // the wave's queries, component-interleaved: one b128 broadcast per d
        float4 v;
        float *pv = &v.x;
#pragma unroll
        for (int qq = 0; qq < kGQ; ++qq) {
            const int qi = q0 + qq < N ? q0 + qq : N - 1;
            pv[qq] = xb[(size_t)qi * D + d];
        }
        qs4[wv * D + d] = v;
    }

    // staging geometry: D / 4 column groups; kWThreads % (D / 4) == 0 keeps a thread on one group
    const int pr = D >> 2;
    const bool vec_stage = (D & 3) == 0 && pr > 0 && kWThreads % pr == 0 && (reinterpret_cast<uintptr_t>(yb) & 15) == 0;
    const int rows_step = vec_stage ? kWThreads / pr : 1, srow = vec_stage ? tid / pr : 0, scol = vec_stage ? (tid - srow * pr) * 4 : 0;
    for (int j0 = 0; j0 < M; j0 += kGT) {
        const int cntc = (M - j0) < kGT ? (M - j0) : kGT;
        __syncthreads();
        if (vec_stage) {
            // 16-byte pieces, a thread keeps its column group and walks rows (no division in the loop; four loads in flight)
            for (int r0 = srow; r0 < cntc; r0 += 4 * rows_step) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + u * rows_step;
                    v[u] = *reinterpret_cast<const float4 *>(yb + (size_t)(j0 + (r < cntc ? r : cntc - 1)) * D + scol);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + u * rows_step;
                    if (r < cntc) {
                        float *t = tile + r * RS + scol;
                        t[0] = v[u].x; t[1] = v[u].y; t[2] = v[u].z; t[3] = v[u].w;
                    }
                }
            }
        } else {
            for (int e = tid; e < cntc * D; e += kWThreads) {  // coalesced: the tile is contiguous in memory
                const int r = e / D, d = e - r * D;
                tile[r * RS + d] = yb[(size_t)j0 * D + e];
            }
        }
        __syncthreads();
        if (q0 < N) {
            // distances of the wave's 4 queries to its 2 tile rows per lane, all dims
            float acc0[kGQ], acc1[kGQ];
            {
                // the two rows of a lane as one register pair: difference, square and sum are packed (v_pk_add / v_pk_mul_f32,
                // each half the plain IEEE operation: same bits, half the instructions)
                typedef float f32x2v __attribute__((ext_vector_type(2)));
                f32x2v acc[kGQ];
#pragma unroll
                for (int qq = 0; qq < kGQ; ++qq) acc[qq] = f32x2v{0.0f, 0.0f};
                const float *r0 = tile + lane * RS, *r1 = tile + (lane + 64) * RS;
                const float4 *qw = qs4 + wv * D;
#pragma unroll 4
                for (int d = 0; d < D; ++d) {
                    const float4 qd = qw[d];
                    const f32x2v c = f32x2v{r0[d], r1[d]};
                    const float qv4[4] = {qd.x, qd.y, qd.z, qd.w};
#pragma unroll
                    for (int qq = 0; qq < kGQ; ++qq) {
                        const f32x2v t = f32x2v{qv4[qq], qv4[qq]} - c;
                        acc[qq] = acc[qq] + t * t;
                    }
                }
#pragma unroll
                for (int qq = 0; qq < kGQ; ++qq) { acc0[qq] = acc[qq].x; acc1[qq] = acc[qq].y; }
            }
#pragma unroll
            for (int qq = 0; qq < kGQ; ++qq) {
                const int qi = q0 + qq;
                if (qi >= N) break;
                const unsigned int d0 = lane < cntc ? dist_key(acc0[qq]) : kNoKey;
                const unsigned int d1 = lane + 64 < cntc ? dist_key(acc1[qq]) : kNoKey;
                const int sidx = (wv * kGQ + qq) * 64;
                unsigned int tau = taus[wv * kGQ + qq];
                int cnt = cnts[wv * kGQ + qq];
                unsigned int bd = bestd[sidx + lane];
                int bj = bestj[sidx + lane];
                if (j0 == 0) {
                    unsigned int v = d0 < d1 ? d0 : d1;
                    bitonic64u(v, lane);
                    tau = readlane_u(v, kk - 1);
                }
                bool dirty = false;
                if (cap > 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned int di = i ? d1 : d0;
                        bool pred = di <= tau && di != kNoKey;
                        unsigned long long bal = __ballot(pred);
                        while (bal) {
                            const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0));
                            const bool put = pred && pos < cap;
                            if (put) { lstd[sidx + pos] = di; lstj[sidx + pos] = j0 + lane + 64 * i; }
                            const int np = __builtin_popcountll(bal);
                            const bool overflow = cnt + np > cap;
                            cnt = overflow ? cap : cnt + np;
                            pred = pred && !put;
                            if (overflow) {
                                unsigned int sd = lane < kk ? bd : (lane - kk < cnt ? lstd[sidx + lane - kk] : kNoKey);
                                int sj = lane < kk ? bj : (lane - kk < cnt ? lstj[sidx + lane - kk] : 0x7fffffff);
                                bitonic64(sd, sj, lane);
                                bd = sd; bj = sj;
                                tau = readlane_u(sd, kk - 1);
                                cnt = 0;
                                dirty = true;
                                pred = pred && di <= tau;
                            }
                            bal = __ballot(pred);
                        }
                    }
                } else {  // kk == 64: merge 64 candidates at a time
#pragma unroll 1
                    for (int i = 0; i < 2; ++i) {
                        unsigned int nd = i ? d1 : d0;
                        int nj = nd != kNoKey ? j0 + lane + 64 * i : 0x7fffffff;
                        bitonic64(nd, nj, lane);
                        const unsigned int rd = (unsigned int)__shfl((int)nd, 63 - lane, 64);
                        const int rj = __shfl(nj, 63 - lane, 64);
                        const bool o_less = key_less(rd, rj, bd, bj);
                        bd = o_less ? rd : bd;
                        bj = o_less ? rj : bj;
                        bitonic64(bd, bj, lane);
                    }
                    dirty = true;
                }
                const bool last = j0 + kGT >= M;
                if (last && cnt > 0) {
                    unsigned int sd = lane < kk ? bd : (lane - kk < cnt ? lstd[sidx + lane - kk] : kNoKey);
                    int sj = lane < kk ? bj : (lane - kk < cnt ? lstj[sidx + lane - kk] : 0x7fffffff);
                    bitonic64(sd, sj, lane);
                    bd = sd; bj = sj;
                    cnt = 0;
                    dirty = true;
                }
                if (dirty) { bestd[sidx + lane] = bd; bestj[sidx + lane] = bj; }
                if (lane == 0) { taus[wv * kGQ + qq] = tau; cnts[wv * kGQ + qq] = cnt; }
                if (last) {
                    const int r = lane - drop;
                    if (r >= 0 && r < k) {
                        idx[((size_t)b * N + qi) * k + r] = bj;
                        if (dist) dist[((size_t)b * N + qi) * k + r] = key_dist(bd);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// knn_select_kernel: the general path -- ANY k + drop <= M and ANY D (the reference's `knn(kdtree, x, K+1, true)`,
// src/models/dgcnn.jl:3-7, takes any K <= N) -- for the shapes none of the other kernels takes: k + drop > 64, or a D the
// wave kernel's tile does not fit.  One wave per query, all M distance keys of the query in LDS (dist_key: the
// oracle's isless order as unsigned integers):
//   1. keys: lane l evaluates candidates l, l+64, ... with the oracle's unfused dimension-order sum;
//   2. T = the kk-th smallest key VALUE by a most-significant-bit-first search (32 counting sweeps over the LDS keys);
//   3. every candidate with key <= T ranks itself: #(keys below it) + #(equal keys with a lower index) -- the oracle's
//      (distance, index) order without a sort -- and writes slot rank - drop if drop <= rank < kk.
//      (the kk survivors are compacted into a list first and ranked among themselves: kk^2 / 64 comparisons per lane)
// Cost O(M 32 / 64 + kk^2 / 64) LDS reads per lane and query: a correct general path, not a tuned one (C4's shape with
// kk = 101: 0.2 ms at D = 3, 1.3 ms at D = 64 -- the key evaluation); the tuned kernels keep k + drop <= 32 (matrix cores).
constexpr int kSelMaxLds = 144 * 1024;
// keys[j] = canonical distance key of candidate j (kNoKey for j >= M) for j = start, start + stride, ... < Mpad
__device__ __forceinline__ void knn_select_keys(const float *__restrict__ x, int N, const float *__restrict__ y, int M, int D, int b, int qi,
                                                int Mpad, unsigned int *keys, int start, int stride) {
    const float *q = x + ((size_t)b * N + qi) * D, *yb = y + (size_t)b * M * D;
    const bool vec4 = (D & 3) == 0 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(yb)) & 15) == 0;
    for (int j = start; j < Mpad; j += stride) {
        unsigned int key = kNoKey;
        if (j < M) {
            const float *c = yb + (size_t)j * D;
            float sacc = 0.0f;
            if (vec4) {
#pragma unroll 8
                for (int dd = 0; dd < D; dd += 4) {  // (eight 16-byte steps in flight: one at a time the row is a chain of round trips)
                    const float4 qv = *reinterpret_cast<const float4 *>(q + dd), cv = *reinterpret_cast<const float4 *>(c + dd);
                    const float t0 = qv.x - cv.x, t1 = qv.y - cv.y, t2 = qv.z - cv.z, t3 = qv.w - cv.w;
                    sacc = sacc + t0 * t0; sacc = sacc + t1 * t1; sacc = sacc + t2 * t2; sacc = sacc + t3 * t3;
                }
            } else {
                for (int dd = 0; dd < D; ++dd) { const float t = q[dd] - c[dd]; sacc = sacc + t * t; }
            }
            key = dist_key(sacc);
        }
        keys[j] = key;
    }
}
// selection + ranking of ONE query by one wave from its keys in LDS (steps 2 and 3 above)
__device__ __forceinline__ void knn_select_wave(unsigned int *keys, int M, int Mpad, int lcap, int N, int b, int qi, int k, int drop, int lane,
                                                int32_t *__restrict__ idx, float *__restrict__ dist) {
    const uint4 *keys4 = reinterpret_cast<const uint4 *>(keys);
    const int kk = k + drop;
    const int n4 = Mpad / 4;
    // ---- T: largest value with #(keys < T) < kk, i.e. the kk-th smallest key (bit by bit, most significant first) ----
    unsigned int T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned int trial = T | (1u << bit);
        int c = 0;  // (wave-uniform: counted with ballots -- no cross-lane reduction per bit; Mpad % 256 == 0: every lane in every sweep)
        for (int i = lane; i < n4; i += 64) {
            const uint4 v = keys4[i];
            c += __builtin_popcountll(__ballot(v.x < trial)) + __builtin_popcountll(__ballot(v.y < trial)) +
                 __builtin_popcountll(__ballot(v.z < trial)) + __builtin_popcountll(__ballot(v.w < trial));
        }
        if (c < kk) T = trial;
    }
    if (lcap >= kk) {
        // ---- the kk survivors: every key below T (fewer than kk) and, in index order, as many keys equal to T as are still
        //      missing -- compacted into a (key, index) list; a survivor's rank among the survivors IS its rank among all
        //      candidates (whatever precedes it in the (distance, index) order survives too).  Ranking against the list costs
        //      kk^2 / 64 comparisons per lane instead of M kk / 64 (k = 64 at C4's shape: 3.8 ms -> 0.3 ms per call).
        uint2 *lst = reinterpret_cast<uint2 *>(keys + Mpad);
        int nless = 0, neq = 0;
        for (int i = lane; i < n4; i += 64) {
            const uint4 v = keys4[i];
            nless += __builtin_popcountll(__ballot(v.x < T)) + __builtin_popcountll(__ballot(v.y < T)) +
                     __builtin_popcountll(__ballot(v.z < T)) + __builtin_popcountll(__ballot(v.w < T));
        }
        const int quota = kk - nless;  // keys equal to T still wanted (>= 1)
        int S = 0;
        for (int j0 = 0; j0 < M; j0 += 64) {
            const int e = j0 + lane;
            const unsigned int me = e < M ? keys[e] : kNoKey;
            const bool eq = e < M && me == T;
            const unsigned long long beq = __ballot(eq);
            const int eqpos = neq + __builtin_amdgcn_mbcnt_hi((unsigned int)(beq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)beq, 0));
            const bool take = e < M && (me < T || (eq && eqpos < quota));
            const unsigned long long bt = __ballot(take);
            if (take) lst[S + __builtin_amdgcn_mbcnt_hi((unsigned int)(bt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)bt, 0))] = uint2{me, (unsigned int)e};
            S += __builtin_popcountll(bt);
            neq += __builtin_popcountll(beq);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < S; t0 += 64) {  // S == kk
            const int t = t0 + lane;
            const uint2 mine = lst[t < S ? t : 0];
            int rank = 0;
            for (int u = 0; u < S; ++u) {  // every lane reads the same entry: LDS broadcast
                const uint2 o = lst[u];
                rank += (int)(o.x < mine.x) | ((int)(o.x == mine.x) & (int)(o.y < mine.y));
            }
            if (t < S && rank >= drop && rank < kk) {
                idx[((size_t)b * N + qi) * k + rank - drop] = (int)mine.y;
                if (dist) dist[((size_t)b * N + qi) * k + rank - drop] = key_dist(mine.x);
            }
        }
        return;
    }
    // ---- ranks of the candidates at or below T ---------------------------------------------------------------------
    for (int j0 = 0; j0 < M; j0 += 64) {
        const int e = j0 + lane;
        const unsigned int me = e < M ? keys[e] : kNoKey;
        if (me <= T && e < M) {
            int rank = 0;
            for (int i = 0; i < n4; ++i) {  // every lane reads the same address: LDS broadcast
                const uint4 v = keys4[i];
                const int p = 4 * i;
                rank += (int)(v.x < me) | ((int)(v.x == me) & (int)(p < e));
                rank += (int)(v.y < me) | ((int)(v.y == me) & (int)(p + 1 < e));
                rank += (int)(v.z < me) | ((int)(v.z == me) & (int)(p + 2 < e));
                rank += (int)(v.w < me) | ((int)(v.w == me) & (int)(p + 3 < e));
            }
            if (rank >= drop && rank < kk) {
                idx[((size_t)b * N + qi) * k + rank - drop] = e;
                if (dist) dist[((size_t)b * N + qi) * k + rank - drop] = key_dist(me);
            }
        }
    }
}

__global__ __launch_bounds__(256) void knn_select_kernel(const float *__restrict__ x, int N, const float *__restrict__ y,
                                                         int M, int B, int D, int k, int drop,
                                                         int32_t *__restrict__ idx, float *__restrict__ dist, int Mpad, int lcap,
                                                         const unsigned char *__restrict__ only, int only_regions) {
    extern __shared__ __attribute__((aligned(16))) unsigned int selkeys[];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (only) {
        // the verified slice merge's flagged queries: a block looks at 32 consecutive queries of a cloud and answers the flagged ones
        // one after the other, ALL its waves evaluating the keys, wave 0 selecting (a flagged query on one wave was 45 us -- the whole
        // duration of the launch; a block per query made the launch itself 30 us: 32768 blocks that only read a flag)
        const int base = blockIdx.x * 32;
        const bool f = lane < 32 && base + lane < N && only[(size_t)b * N + base + lane] != 0;
        unsigned int m = (unsigned int)__ballot(f);  // (every wave computes the same mask)
        if (__builtin_popcount(m) >= nw && only_regions >= nw) {
            // many flagged queries (a binomial tail at larger k, or data that defeat the interleaving): a wave per query again
            unsigned int *keys = selkeys + (size_t)wv * (Mpad + 2 * lcap);
            int t = 0;
            for (; m; m &= m - 1, ++t) {
                if (t % nw != wv) continue;
                const int qi = base + __builtin_ctz(m);
                knn_select_keys(x, N, y, M, D, b, qi, Mpad, keys, lane, 64);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                knn_select_wave(keys, M, Mpad, lcap, N, b, qi, k, drop, lane, idx, dist);
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
        for (; m; m &= m - 1) {
            const int qi = base + __builtin_ctz(m);
            knn_select_keys(x, N, y, M, D, b, qi, Mpad, selkeys, (int)threadIdx.x, (int)blockDim.x);
            __syncthreads();
            if (wv == 0) knn_select_wave(selkeys, M, Mpad, lcap, N, b, qi, k, drop, lane, idx, dist);
            __syncthreads();  // (the next query's keys overwrite these)
        }
        return;
    }
    const int qi = blockIdx.x * nw + wv;
    if (qi >= N) return;  // wave-uniform; no block-level sync below
    unsigned int *keys = selkeys + (size_t)wv * (Mpad + 2 * lcap);
    knn_select_keys(x, N, y, M, D, b, qi, Mpad, keys, lane, 64);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    knn_select_wave(keys, M, Mpad, lcap, N, b, qi, k, drop, lane, idx, dist);
}

// out[(((b*N+i)*k + r)*F + f] = x[(b*N + idx[(b*N+i)*k + r])*F + f]
__global__ __launch_bounds__(kThreads) void knn_gather_kernel(const float *__restrict__ x, int N, int B,
                                                              int F, int k,
                                                              const int32_t *__restrict__ idx,
                                                              float *__restrict__ out) {
    const long long total = (long long)B * N * k * F;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total;
         e += (long long)gridDim.x * kThreads) {
        const long long row = e / F;  // (b*N+i)*k + r
        const int f = (int)(e - row * F);
        const long long bn = row / k;
        const int b = (int)(bn / N);
        const int j = idx[row];
        out[e] = x[((size_t)b * N + j) * F + f];
    }
}


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
constexpr size_t kStreamingStoreBytes = (size_t)192 << 20;
template <bool NT>
__global__ __launch_bounds__(kThreads) void knn_gather4_kernel(const float *__restrict__ x, int N, int B, int F4, int k,
                                                               const int32_t *__restrict__ idx, float *__restrict__ out) {
    const long long total = (long long)B * N * k * F4;
    const f32x4v *x4 = reinterpret_cast<const f32x4v *>(x);
    f32x4v *o4 = reinterpret_cast<f32x4v *>(out);
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total;
         e += (long long)gridDim.x * kThreads) {
        const long long row = e / F4;  // (b*N+i)*k + r
        const int f = (int)(e - row * F4);
        const int b = (int)(row / k / N);
        const f32x4v v = x4[((size_t)b * N + idx[row]) * F4 + f];
        if (NT) __builtin_nontemporal_store(v, o4 + e);
        else o4[e] = v;
    }
}

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

// Wave-cooperative ranking of ONE query's n survivors on the full (distance bits, index) keys -- the path of the
// rare query whose distance-only ranks collide (an exact tie among its first kk).  Lane e ranks key e against all
// n (LDS broadcast reads; qd / qj are padded with sentinels up to a multiple of four); keys are unique, so the
// ranks below kk are a permutation and slots[0, kk) is the sorted answer.  One tied query costs its wave well
// under a microsecond (a per-lane loop over the query's keys made the whole grid wait ~10 us for one wave).
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

// ------------------------------------------------------------------------------------------------
// knn_f16_d3_kernel: kNN for D = 3 (DGCNN's first EdgeConv, BASELINE config 4) with the chamfer kernel's
// fp16-split filter (chamfer.hip, nn1_f16_kernel: t = |c~|^2 + qm~ . c~ on ONE v_mfma_f32_32x32x16_f16 per
// 32 x 32 tile, |t - s^2 (d_oracle - |q'|^2)| <= 2^-20 (4 + 2S)).
//   A block = 4 groups of 32 queries x 2 waves per group; the two waves of a group take alternate pairs of
//   32-candidate tiles (two waves per SIMD hide each other's LDS latencies).  Lane l = (hh, jq) of a wave holds
//   query jq of its group and the 16 candidate rows (r&3)+8(r>>2)+4hh of its tiles: four lanes per query.
//   Phase A: per lane and register the minimum over its tiles (one v_min3 folds two tiles): 32 group minima per
//            lane, 128 per query, each over M/128 candidates.  The kk-th smallest of them bounds the kk-th
//            smallest filter value: tau.  Selection = 32-element sorting network in registers, the partner
//            lane's values by v_permlane32_swap, the other wave's 32 smallest through LDS, two bitonic merges.
//   Phase B: the filter again with the threshold folded into the MFMA (K slot 15: 1 x -thr16, thr16 the
//            smallest fp16 above the threshold): the sign of the result is the test, one v_alignbit per row
//            shifts it into the tile's 16-bit row mask; (tile, mask) words go to the LANE's private LDS list
//            (unconditional store at the list head, the head advances when the mask is not empty).
//            threshold = tau (1 + 4 beta + ...) + (10.1 beta + ...) |q~|^2 + floor on the upper-bound values of the image
//            (mean centring, 2^7 scale, folded norms: chamfer.hip make_pieces; band_b1 / band_a below).  A superset
//            of the k nearest, ~1.1 kk entries per query at config 4.
//   Exact:   every lane decodes its list and evaluates the oracle's distance of its entries (the query is in its
//            registers); the ranking of a query's keys is shared by its four lanes: rank = number of keys with
//            a smaller distance = output slot, verified by count and rank sum, ties re-ranked on (distance,
//            index).  Bit-identical to fx3d_oracle_knn.
//   Queries outside the fp16 range, with overflowing lists or non-finite thresholds take the brute-force merge.
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

// EdgeConv's features of ONE (point i, neighbour rank r) pair for F = 3 (src/models/dgcnn.jl:36-51): cat(x_i, x_j - x_i),
// layout 0 = (2F,K,N,B), 1 = (K*N,2F,B).  Used by the rare paths of the fused kernel (ties, exact fallback).
__device__ __forceinline__ void knn_d3_feature_entry(float *__restrict__ feat, int layout, int b, int N, int k, int i, int r,
                                                     const float *a, const float *c) {
    if (layout == 0) {
        float *o = feat + (((size_t)b * N + i) * k + r) * 6;
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
        o[3] = c[0] - a[0]; o[4] = c[1] - a[1]; o[5] = c[2] - a[2];
    } else {
        const size_t KN = (size_t)k * N;
        float *o = feat + (size_t)b * 6 * KN + (size_t)i * k + r;
        o[0] = a[0]; o[KN] = a[1]; o[2 * KN] = a[2];
        o[3 * KN] = c[0] - a[0]; o[4 * KN] = c[1] - a[1]; o[5 * KN] = c[2] - a[2];
    }
}

// FEAT: EdgeConv's graph build in one kernel (self-kNN, x == y): the epilogue also writes cat(x_i, x_j - x_i).
template <bool FEAT, class C>
__global__ __launch_bounds__(C::T) void knn_f16_d3_kernel(const float *__restrict__ x, int N,
                                                               const float *__restrict__ y, int M, int B, int k,
                                                               int drop, int32_t *__restrict__ idx,
                                                               float *__restrict__ dist, int CH, int img_bytes,
                                                               int raw_ok, float *__restrict__ feat, int layout, int med_cap, int med_off, int xdiv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char k3sm[];
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * C::W];  // per wave: min, max, sum, sampled second moment (padded to 4 dims)
    __shared__ int nfar;                    // candidates of the cloud beyond the robust range ...
    __shared__ int farlist[kK3FarCap];      // ... their indices: outside the filter, every query evaluates them exactly
    kh8 *imgp = reinterpret_cast<kh8 *>(k3sm);  // piece (blk, half, row) at (blk*2 + half)*32 + row
    constexpr int kListBytes = C::W * C::CAP * 64 * 4;      // lane lists; also the tau exchange and, later, the slots
    constexpr int kCtrInts = C::G * 32 * 8;               // per query: 4 part counts, overflow, n, fast, below
    int *lists_all = reinterpret_cast<int *>(k3sm + img_bytes);                                  // [C::W][C::CAP][64]
    int *ctr = reinterpret_cast<int *>(k3sm + img_bytes + kListBytes);                           // [C::G*32][8]
    const float4 *rawc = reinterpret_cast<const float4 *>(k3sm + img_bytes + kListBytes + kCtrInts * 4);  // [M] when raw_ok

    const int nbx = (N + C::G * 32 - 1) / (C::G * 32);
    const int L = blockIdx.x;
    const bool by_xcd = B >= 8;
    const int b = by_xcd ? ((L >> 3) / nbx) * 8 + (L & 7) : L / nbx;
    const int bxq = by_xcd ? (L >> 3) % nbx : L % nbx;
    if (b >= B) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int jq = lane & 31, hh = lane >> 5;
    const int kk = k + drop;
    const float *xb = x + (size_t)(b / xdiv) * N * 3, *yb = y + (size_t)b * M * 3;  // (xdiv > 1: candidate slices as virtual clouds share their queries)
    const int grp = wv % C::G, half = wv / C::G;  // query group; which pairs of tiles this wave takes
    const int q0 = (bxq * C::G + grp) * 32;
    const bool wave_active = q0 < N;
    for (int e = tid; e < kCtrInts; e += C::T) ctr[e] = 0;
    if (tid == 0) nfar = 0;  // (ordered before its first use by the barrier of the cloud pass)
    const int qi = q0 + jq;  // this lane's query (loaded here: the latency hides behind the pass over the cloud)
    const int qc = qi < N ? qi : N - 1;
    const float qr[3] = {xb[(size_t)qc * 3], xb[(size_t)qc * 3 + 1], xb[(size_t)qc * 3 + 2]};
    KNN_PROBE_MARK(0);

    // ---- one pass over the cloud: bounding box (-> centre mu, power-of-two scale sc with |c~| <= 1) and, for
    //      clouds up to kTRawMax points, the raw coordinates parked in LDS for the staging and the exact phase ----
    float mu[3], cinf = 0.0f, varmax = 0.0f;
    bool allfin = true;
    {
        float4 *raww = reinterpret_cast<float4 *>(k3sm + img_bytes + kListBytes + kCtrInts * 4);
        float mn3[3] = {INFINITY, INFINITY, INFINITY}, mx3[3] = {-INFINITY, -INFINITY, -INFINITY}, sm3[3] = {0.f, 0.f, 0.f};
        float sqt = 0.0f;  // second moment (all three coordinates) of a sample (~M/4 points) about the cloud's first point: the spread
        const float pil[3] = {yb[0], yb[1], yb[2]};
        // thread t takes points t, t + C::T, ...: 12-byte loads, consecutive lanes on consecutive points (coalesced,
        // and the 16-byte LDS slots of a wave's points are consecutive: no bank conflicts)
        const int nsweep = (M + C::T - 1) / C::T;
        for (int i0 = 0; i0 < nsweep; i0 += 4) {
            P3 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = (i0 + e) * C::T + tid;
                v[e] = *reinterpret_cast<const P3 *>(yb + (size_t)(pt < M ? pt : M - 1) * 3);  // (clamped: always valid)
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = (i0 + e) * C::T + tid;
                mn3[0] = fminf(mn3[0], v[e].x); mx3[0] = fmaxf(mx3[0], v[e].x);
                mn3[1] = fminf(mn3[1], v[e].y); mx3[1] = fmaxf(mx3[1], v[e].y);
                mn3[2] = fminf(mn3[2], v[e].z); mx3[2] = fmaxf(mx3[2], v[e].z);
                if (pt < M) {
                    sm3[0] = sm3[0] + v[e].x; sm3[1] = sm3[1] + v[e].y; sm3[2] = sm3[2] + v[e].z;
                    if (raw_ok) raww[pt] = float4{v[e].x, v[e].y, v[e].z, 0.0f};
                }
                if (((i0 + e) & 3) == (wv & 3) && pt < M) {  // (wave-uniform first condition: 64-point runs all over the cloud)
                    sqt = __builtin_fmaf(v[e].x - pil[0], v[e].x - pil[0], sqt); sqt = __builtin_fmaf(v[e].y - pil[1], v[e].y - pil[1], sqt);
                    sqt = __builtin_fmaf(v[e].z - pil[2], v[e].z - pil[2], sqt);
                }
            }
        }
        {   // wave level by DPP (the result is in lane 63), one 16-byte row per (wave, statistic)
            float4 lo4, hi4, st4;
            lo4.x = wave_min_l63(mn3[0]); lo4.y = wave_min_l63(mn3[1]); lo4.z = wave_min_l63(mn3[2]); lo4.w = 0.0f;
            hi4.x = wave_max_l63(mx3[0]); hi4.y = wave_max_l63(mx3[1]); hi4.z = wave_max_l63(mx3[2]); hi4.w = 0.0f;
            st4.x = wave_sum_l63(sm3[0]); st4.y = wave_sum_l63(sm3[1]); st4.z = wave_sum_l63(sm3[2]); st4.w = wave_sum_l63(sqt);
            if (lane == 63) {
                float4 *r4 = reinterpret_cast<float4 *>(red);
                r4[wv * 4] = lo4; r4[wv * 4 + 1] = hi4; r4[wv * 4 + 2] = st4;
            }
        }
        __syncthreads();
        {
            const float4 *r4 = reinterpret_cast<const float4 *>(red);
            float4 lo4 = r4[0], hi4 = r4[1], st4 = r4[2];
#pragma unroll
            for (int w = 1; w < C::W; ++w) {
                const float4 a0 = r4[w * 4], a1 = r4[w * 4 + 1], a2 = r4[w * 4 + 2];
                lo4.x = fminf(lo4.x, a0.x); lo4.y = fminf(lo4.y, a0.y); lo4.z = fminf(lo4.z, a0.z);
                hi4.x = fmaxf(hi4.x, a1.x); hi4.y = fmaxf(hi4.y, a1.y); hi4.z = fmaxf(hi4.z, a1.z);
                st4.x = st4.x + a2.x; st4.y = st4.y + a2.y; st4.z = st4.z + a2.z; st4.w = st4.w + a2.w;
            }
            const float lo3[3] = {lo4.x, lo4.y, lo4.z}, hi3[3] = {hi4.x, hi4.y, hi4.z}, st3[3] = {st4.x, st4.y, st4.z};
            varmax = st4.w / fmaxf(0.25f * (float)M, 1.0f);  // TOTAL variance of the three coordinates, from ~M/4 sampled points (a heuristic's input)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                // centre = the MEAN (a stray far point moves the box centre, hardly the mean); any centre is correct
                mu[d] = fminf(fmaxf(st3[d] / (float)M, lo3[d]), hi3[d]);
                const float off = st3[d] / (float)M - pil[d];
                varmax = varmax - off * off;  // (variance about the mean from the moment about a data point)
                cinf = fmaxf(cinf, fmaxf(hi3[d] - mu[d], mu[d] - lo3[d]));
                allfin = allfin && fabsf(st3[d]) < INFINITY;  // a NaN / +-Inf coordinate makes the sum non-finite (fminf / fmaxf skip NaNs)
            }
        }
        cinf = cinf * 1.000001f;
    }
    // not sane (non-finite or huge coordinates): every query takes the brute-force merge, which orders distances as the
    // oracle does (dist_key); a finite cloud with cinf < 1e16 has no infinite or NaN distance to a usable query
    const bool sane = allfin && cinf < 1.0e16f;  // usable queries lie within 234 cinf of the centre: 3 (235 cinf)^2 stays finite
    // robust range, as in nn1_f16_kernel (chamfer.hip): a few points far from the bulk must not set the scale (one point 10^5 x
    // the extent away sent every query to the exact merge: 143 vs 22 us).  rng = min(cinf, 16 x the mean max-norm deviation
    // about the re-centred mean); gates: kRobustGate / kRobustHarm (fx3d_common.h).  Candidates beyond the range
    // get norm +inf in the image (never below a threshold) and go on a side list that every query appends to its survivors.
    float rng = cinf;
    if (sane && 3.0f * cinf * cinf > kRobustGate * varmax) {
        const float4 r = robust_range3<C::T, true>(yb, M, raw_ok != 0, rawc, red, mu[0], mu[1], mu[2], cinf);
        mu[0] = r.x; mu[1] = r.y; mu[2] = r.z; rng = r.w;
    }
    const bool has_far = sane && rng < cinf;
    float sc = 1.0f;
    if (sane && rng > 1.0e-30f) {
        int e;
        (void)frexpf(rng, &e);
        sc = ldexpf(1.0f, 7 - e);  // |c~| < 2^7: a bulk far smaller than the farthest point stays out of fp16's subnormals
    }
    KNN_PROBE_MARK(1);

    // ---- this lane's query: B operand, band --------------------------------------------------------------------
    const float m0 = -2.0f * ((qr[0] - mu[0]) * sc), m1 = -2.0f * ((qr[1] - mu[1]) * sc), m2 = -2.0f * ((qr[2] - mu[2]) * sc);
    const float S = (fabsf(m0) + fabsf(m1)) + fabsf(m2);
    const bool qok = S < 3.0e4f;  // inside the fp16 range (also false for NaN)
    // band (chamfer.hip, make_pieces: the image holds upper bounds U_c = t^ + beta n_c): a candidate among the kk nearest
    // has U_c <= tau_U (1 + 18 beta) + 22.3 beta |q~|^2 + floor, + the oracle's rounding, + the threshold as a 17th
    // MFMA term in phase B (2^-21 of its magnitude)
    const float qn = 0.25f * ((m0 * m0 + m1 * m1) + m2 * m2);
    const float band_b1 = 1.0f + 18.0f * kK3BetaC + 0x1p-20f + 0x1p-21f;
    const float band_a = (22.3f * kK3BetaC + 0x1p-19f + 0x1p-21f) * qn + 0x1p-24f * (S + 4.0f);
    kh8 bq;
    {
        _Float16 hx, lx, hy, ly, hz, lz;
        k3_split2h(qok ? m0 : 0.f, hx, lx); k3_split2h(qok ? m1 : 0.f, hy, ly); k3_split2h(qok ? m2 : 0.f, hz, lz);
        const _Float16 one = (_Float16)1.0f, z = (_Float16)0.0f;
        bq = hh == 0 ? kh8{hx, lx, hx, hy, ly, hy, hz, lz} : kh8{hz, one, one, one, lx, ly, lz, z};
    }

    float mn[32];  // group minima: [r] first / [16 + r] second tile of this wave's pairs -> 128 groups per query
#pragma unroll
    for (int r = 0; r < 32; ++r) mn[r] = INFINITY;
    float thr = 0.0f;
    int cnt = 0, tot = 0;
    int *mylist = lists_all + wv * C::CAP * 64 + lane;  // entry e at mylist[e * 64]
    f32x16v zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
    const int nchunk = (M + CH - 1) / CH;

    for (int phase = 0; phase < 2; ++phase) {
        for (int ci = 0; ci < nchunk; ++ci) {
            const int j0 = (phase == 0 ? ci : nchunk - 1 - ci) * CH;  // phase B backwards: its first chunk is staged
            const int cn = (M - j0) < CH ? (M - j0) : CH;
            const int cn_pad = (cn + 63) & ~63;
            if (!(phase == 1 && ci == 0)) {
                __syncthreads();
                // (two separate loops: a select between an LDS and a global pointer trips the compiler)
                if (raw_ok) {
                    for (int pt = tid; pt < cn_pad; pt += C::T) {
                        const int i0 = ((pt >> 5) * 2) * 32 + (pt & 31);
                        kh8 p0, p1;
                        if (pt < cn) {
                            const float4 rc = rawc[j0 + pt];
                            k3_pieces_far((rc.x - mu[0]) * sc, (rc.y - mu[1]) * sc, (rc.z - mu[2]) * sc, has_far, phase == 0, j0 + pt, &nfar, farlist, p0, p1);
                        } else {  // padding: n1 = +inf => t = +inf, never below a finite threshold
                            k3_make_pieces(0.f, 0.f, 0.f, p0, p1);
                            p1[1] = (_Float16)INFINITY;
                        }
                        imgp[i0] = p0;
                        imgp[i0 + 32] = p1;
                    }
                } else {
                    for (int pt = tid; pt < cn_pad; pt += C::T) {
                        const int i0 = ((pt >> 5) * 2) * 32 + (pt & 31);
                        kh8 p0, p1;
                        if (pt < cn) {
                            const float *src = yb + (size_t)(j0 + pt) * 3;
                            k3_pieces_far((src[0] - mu[0]) * sc, (src[1] - mu[1]) * sc, (src[2] - mu[2]) * sc, has_far, phase == 0, j0 + pt, &nfar, farlist, p0, p1);
                        } else {
                            k3_make_pieces(0.f, 0.f, 0.f, p0, p1);
                            p1[1] = (_Float16)INFINITY;
                        }
                        imgp[i0] = p0;
                        imgp[i0 + 32] = p1;
                    }
                }
                __syncthreads();
                KNN_PROBE_MARK(2);
            }
            if (wave_active) {
                const kh8 *pa = imgp + hh * 32 + jq;
                const int npair = cn_pad / 64;
                const int tile0 = j0 / 32;
                int pr = half;  // this wave's pairs of 32-candidate tiles: half, half + 2, ...
                if (phase == 0) {
                    for (; pr + 2 < npair; pr += 4) {  // two pairs per step: one v_min3 folds two tiles' rows
                        f32x16v a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2) * 64], bq, zero, 0, 0, 0);
                        f32x16v b0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2 + 4) * 64], bq, zero, 0, 0, 0);
                        f32x16v a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2 + 1) * 64], bq, zero, 0, 0, 0);
                        f32x16v b1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2 + 5) * 64], bq, zero, 0, 0, 0);
                        KNN_MFMA_SETTLE4(a0, b0, a1, b1);
#pragma unroll
                        for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[r]) : "v"(a0[r]), "v"(b0[r]));
#pragma unroll
                        for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[16 + r]) : "v"(a1[r]), "v"(b1[r]));
                    }
                }
                for (; pr < npair; pr += 2) {
                    f32x16v acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2) * 64], bq, zero, 0, 0, 0);
                    f32x16v acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2 + 1) * 64], bq, zero, 0, 0, 0);
                    if (phase == 0) {
                        KNN_MFMA_SETTLE2(acc0, acc1);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mn[r] = vmin_acc(mn[r], acc0[r]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mn[16 + r] = vmin_acc(mn[16 + r], acc1[r]);
                    } else {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            // one word per tile: (tile index << 16) | mask of the rows with t - thr16 < 0 (row r at
                            // bit 15 - r: one v_alignbit shifts the sign in); stored at the list head
                            // unconditionally, the head advances when the mask is not empty
                            unsigned int m = 0;
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                m = __builtin_amdgcn_alignbit(m, __builtin_bit_cast(unsigned int, tt ? acc1[r] : acc0[r]), 31);
                            const int pp = cnt < C::CAP - 1 ? cnt : C::CAP - 1;
                            mylist[pp * 64] = (int)((unsigned int)(tile0 + pr * 2 + tt) << 16 | m);
                            cnt += m != 0 ? 1 : 0;
                            tot += __builtin_popcount(m);
                        }
                    }
                }
            }
        }
        KNN_PROBE_MARK(phase ? 5 : 3);
        if (phase == 0) {
            // ---- tau: kk-th smallest of the 128 group minima of every query: 32 in this lane, 32 in its partner
            //      lane, 64 in the other wave of the group.  Sorting network in registers, one exchange with the
            //      partner lane, one exchange with the other wave through LDS, bitonic merges in between.
            float tau = INFINITY;
            float *xch = reinterpret_cast<float *>(lists_all);  // the lists are not in use yet
            if (C::KKMAX <= 32 && kk <= 24 && M >= 128) {
                // (round 4) the eight smallest of each half-lane set of 16 instead of a sort of all 32: knn_tau_8of16
                tau = knn_tau_8of16<C::G>(mn, xch, wv, jq, hh, kk);
            } else {
                k3_sort_regs<32>(mn);
                {
                    float oth[32];  // the partner lane's values (v_permlane32_swap: no LDS round trip)
    #pragma unroll
                    for (int r = 0; r < 32; ++r)  // mn[r] <- lanes 0-31's value, oth[r] <- lanes 32-63's, in every lane
                        // (inline asm: this compiler's __builtin_amdgcn_permlane32_swap returns its first result twice)
                        asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(mn[r]), "=&v"(oth[r]));
    #pragma unroll
                    for (int r = 0; r < 32; ++r)  // the 32 smallest of the wave's 64 (a bitonic sequence), in both half-lanes
                        mn[r] = vmin_f32(mn[r], oth[31 - r]);
                }
    #pragma unroll
                for (int j = 16; j > 0; j >>= 1) {  // one bitonic merge sorts it ascending
    #pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        const int l = i ^ j;
                        if (l > i) {
                            const float lo = vmin_f32(mn[i], mn[l]), hi = vmax_f32(mn[i], mn[l]);
                            mn[i] = lo;
                            mn[l] = hi;
                        }
                    }
                }
                if (hh == 0) {
    #pragma unroll
                    for (int r = 0; r < 32; ++r) xch[(wv * 32 + jq) * 33 + r] = mn[r];
                }
                __syncthreads();
                // the kk-th smallest of the union of this wave's 32 smallest X and the other wave's Y (both ascending) without merging
                // them: min over the splits (i values from X, kk - i from Y) of max(X[i-1], Y[kk-i-1]).  For kk <= 32 that is the kk-th
                // smallest of all 128 group minima; for 32 < kk <= 64 the kk-th smallest of these 64 -- an upper bound of it (equal unless
                // one wave holds more than 32 of the kk smallest).  ~3 VALU + one LDS read per split instead of 32 reads + a 32-value
                // bitonic merge (192 VALU).
                if (C::KKMAX <= 32) {
                    const float *po = xch + (((wv + C::G) % C::W) * 32 + jq) * 33;  // the group's other wave
    #pragma unroll
                    for (int r = 0; r < 32; ++r) mn[r] = vmin_f32(mn[r], po[31 - r]);     // the 32 smallest of the 128
    #pragma unroll
                    for (int j = 16; j > 0; j >>= 1) {
    #pragma unroll
                        for (int i = 0; i < 32; ++i) {
                            const int l = i ^ j;
                            if (l > i) {
                                const float lo = vmin_f32(mn[i], mn[l]), hi = vmax_f32(mn[i], mn[l]);
                                mn[i] = lo;
                                mn[l] = hi;
                            }
                        }
                    }
                    tau = mn[0];  // kk <= 32: among the 32 smallest
    #pragma unroll
                    for (int r = 1; r < 32; ++r) tau = (kk - 1) == r ? mn[r] : tau;
                } else
                {
                    const float *po = xch + (((wv + C::G) % C::W) * 32 + jq) * 33;  // the group's other wave
                    float yv[33];  // (all reads first, clamped addresses: a branch per split made them a chain of LDS round trips)
    #pragma unroll
                    for (int i = 0; i <= 32; ++i) {
                        const int yi = kk - i - 1;
                        yv[i] = po[yi < 0 ? 0 : (yi > 31 ? 31 : yi)];
                    }
    #pragma unroll
                    for (int i = 0; i <= 32; ++i) {
                        const int ny = kk - i;  // (uniform: the selects below take scalar conditions)
                        const float a = i >= 1 ? mn[i >= 1 ? i - 1 : 0] : -INFINITY;
                        const float t = vmax_f32(a, ny >= 1 ? yv[i] : -INFINITY);
                        tau = vmin_f32(tau, (ny >= 0 && ny <= 32) ? t : INFINITY);
                    }
                }
            }
            thr = __builtin_fmaf(tau, band_b1, band_a);
            {
                // phase B subtracts the threshold inside the MFMA (K slot 15: candidate side 1, query side -thr16)
                // and keeps the sign: thr16 = the smallest fp16 value strictly above thr, so that t <= thr gives a
                // negative difference (no -0) -- at most a few more survivors than the Float32 threshold
                _Float16 h = (_Float16)thr;
                unsigned short hb = __builtin_bit_cast(unsigned short, h);
                if ((float)h <= thr) hb = (hb & 0x7fffu) == 0 ? 0x0001u : ((hb & 0x8000u) ? hb - 1 : hb + 1);
                h = __builtin_bit_cast(_Float16, hb);
                if (!((float)h < INFINITY)) thr = INFINITY;  // (also NaN) -> the query is not usable
                if (hh == 1) bq[7] = -h;
            }
            __syncthreads();  // the exchange space becomes the lane lists
            KNN_PROBE_MARK(4);
        }
    }

    // ---- exact phase: the four lanes of a query (two half-lanes x two waves) share its survivors ------------------
    const int part = half * 2 + hh;
    const int qslot = grp * 32 + jq;
    int *qctr = ctr + qslot * 8;  // [0..3] entries decoded by part, [4] overflow, [5] below
    const int need = kk < M ? kk : M;
    const int nv = cnt < C::CAP - 1 ? cnt : C::CAP - 1;
    const int nf = has_far ? nfar : 0;              // (complete: every chunk was staged before the last barrier)
    const bool far_ok = nf <= kK3FarCap;             // more far candidates than the side list holds: no query is usable
    qctr[part] = tot + (part == 3 && far_ok ? nf : 0);  // the query's last lane appends the far candidates to its own entries
    if (cnt > C::CAP - 1) qctr[4] = 1;
    if (med_cap > 0) atomicOr(&qctr[6], nv << (8 * part));  // list lengths of the four parts (< 24 each): the medium path's decode
    __syncthreads();  // every wave is done with the image: its space now holds the keys
    KNN_PROBE_MARK(6);
    unsigned int *qd = reinterpret_cast<unsigned int *>(k3sm) + (size_t)qslot * C::KS;                          // distance bits
    int *qj = reinterpret_cast<int *>(k3sm) + (size_t)C::G * 32 * C::KS + (size_t)qslot * C::KS;     // indices
    const int c0 = qctr[0], c1 = qctr[1], c2 = qctr[2], c3 = qctr[3];
    const int n = c0 + c1 + c2 + c3;
    const int off = part == 0 ? 0 : (part == 1 ? c0 : (part == 2 ? c0 + c1 : c0 + c1 + c2));
    const bool usable = sane && far_ok && qok && thr < INFINITY;
    const bool fast = wave_active && qi < N && usable && qctr[4] == 0 && n <= C::KCAP && n >= need;  // (+ 3 sentinels: inside the stride)
    // ---- medium path (tight clusters, duplicated points, lattices: more candidates inside the band than the key arrays hold):
    //      a wave decodes the query's four lane lists into an id list (+ the far candidates) and selects exactly among those,
    //      instead of scanning all M candidates in the fallback.  The lists are intact until the barrier after the decode.
    const bool medium = med_cap > 0 && wave_active && qi < N && usable && qctr[4] == 0 && n > C::KCAP && n <= med_cap && n >= need;
    if (med_cap > 0 && wave_active) {
        const unsigned long long mm = __ballot(medium);
        int *ids = reinterpret_cast<int *>(k3sm + med_off) + wv * (med_cap + 128);
        for (unsigned int bm = (unsigned int)mm | (unsigned int)(mm >> 32); bm; bm &= bm - 1) {
            const int j = __builtin_ctz(bm);
            if ((j & 1) != half) continue;  // the group's two waves share the queries
            const int *cj = ctr + (grp * 32 + j) * 8;
            int total = 0;
            for (int p2 = 0; p2 < 4; ++p2) {  // p2 = 2 * (wave of the group) + half-wave
                const int src = (p2 & 1) * 32 + j;
                const int nv2 = (cj[6] >> (8 * p2)) & 0xff;
                const unsigned int w = lane < nv2 ? (unsigned int)lists_all[((grp + C::G * (p2 >> 1)) * C::CAP + lane) * 64 + src] : 0u;
                const int pc = __builtin_popcount(w & 0xffffu);
                int incl = pc;
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) {
                    const int t = __shfl_up(incl, m, 64);
                    if (lane >= m) incl += t;
                }
                int pos = total + incl - pc;
                unsigned int m16 = w & 0xffffu;
                const int rowbase = (int)(w >> 16) * 32 + 4 * (p2 & 1);
                while (m16) {
                    const int r = 15 - __builtin_ctz(m16);  // (row r at bit 15 - r)
                    m16 &= m16 - 1;
                    ids[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                }
                total += __shfl(incl, 63, 64);
            }
            for (int f = lane; f < nf; f += 64) ids[total + f] = farlist[f];  // (nf = 0 unless far_ok and has_far: usable)
            total += nf;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float bd;
            int bj;
            knn_exact_bruteforce<(C::KKMAX > 32)>(xb + (size_t)(q0 + j) * 3, yb, total, 3, kk, lane, reinterpret_cast<float *>(ids + med_cap), ids + med_cap + 64,
                                 bd, bj, ids);
            const int r = lane - drop;
            if (r >= 0 && r < k) {
                idx[((size_t)b * N + q0 + j) * k + r] = bj;
                if (dist) dist[((size_t)b * N + q0 + j) * k + r] = bd;
                if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, q0 + j, r, xb + (size_t)(q0 + j) * 3, yb + (size_t)bj * 3);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (fast) {
        // (1) decode the (tile, mask) words into candidate ids: integer work only, no memory latency in the chain
        int pos = off;
        for (int e0 = 0; e0 < nv; e0 += 4) {  // four list words in flight
            unsigned int w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = (unsigned int)mylist[(e0 + u < C::CAP ? e0 + u : C::CAP - 1) * 64];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned int m = e0 + u < nv ? (w[u] & 0xffffu) : 0u;
                const int rowbase = (int)(w[u] >> 16) * 32 + 4 * hh;
                while (m) {
                    const int r = 15 - __builtin_ctz(m);
                    m &= m - 1;
                    qj[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                }
            }
        }
        if (part == 3)
            for (int f = 0; f < nf; ++f) qj[pos++] = farlist[f];  // (never inside a mask: their filter value is +inf)
        // (2) the oracle's distance of the ids this lane just wrote (its own LDS writes: no barrier), eight in flight
        for (int p0 = off; p0 < pos; p0 += 8) {
            int id[8];
            float c0f[8], c1f[8], c2f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) id[u] = qj[p0 + u < pos ? p0 + u : off];
            if (raw_ok) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 rc = rawc[id[u]];
                    c0f[u] = rc.x; c1f[u] = rc.y; c2f[u] = rc.z;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float *c = yb + (size_t)id[u] * 3;
                    c0f[u] = c[0]; c1f[u] = c[1]; c2f[u] = c[2];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float t0 = qr[0] - c0f[u], t1 = qr[1] - c1f[u], t2 = qr[2] - c2f[u];
                float sd = t0 * t0;
                sd = sd + t1 * t1;
                sd = sd + t2 * t2;
                if (p0 + u < pos) qd[p0 + u] = __builtin_bit_cast(unsigned int, sd);
            }
        }
        if (part == 3) {  // sentinels: the rank loops read four keys at a time
            qd[n] = 0xffffffffu; qd[n + 1] = 0xffffffffu; qd[n + 2] = 0xffffffffu;
            qj[n] = 0x7fffffff; qj[n + 1] = 0x7fffffff; qj[n + 2] = 0x7fffffff;
        }
    }
    __syncthreads();  // keys visible to the query's four lanes; the lane lists are dead: their space holds the slots
    KNN_PROBE_MARK(8);
    const int per = (n + 3) >> 2;  // the ranking is shared evenly
    const int mystart = part * per < n ? part * per : n;
    const int mycount = mystart + per <= n ? per : n - mystart;
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(lists_all) + (size_t)qslot * C::SS;  // [..][KKMAX + 1 pad]
    // ---- order: rank of a survivor = number of survivors of its query with a smaller distance (squared distances
    //      are >= +0: unsigned order of the bits).  Ties in the distance are resolved by the index in the
    //      reference; they are rare, so the ranks are computed on the distances alone and VERIFIED: the ranks below
    //      kk are a permutation of 0..kk-1 iff exactly kk entries have rank < kk and their ranks sum to
    //      kk (kk - 1) / 2: an entry's rank is at most its position in the sorted order, strictly less for every
    //      entry tied with an earlier one; a tie straddling the kk boundary makes the count kk + 1.  A query that
    //      fails is ranked again on the full keys.
    if (fast) {
        int below = 0;  // own entries with rank < kk: count | sum of ranks << 8
        for (int e0 = 0; e0 < mycount; e0 += 8) {
            unsigned int md[8];
            int rank[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                md[u] = e0 + u < mycount ? qd[mystart + e0 + u] : 0xffffffffu;
                rank[u] = 0;
            }
            for (int i = 0; i < n; i += 4) {
                const uint4 o = *reinterpret_cast<const uint4 *>(qd + i);
#pragma unroll
                for (int u = 0; u < 8; ++u) {  // compare + add-with-carry: two VALU ops per pair
                    unsigned long long cc;
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.x), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.y), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.z), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.w), "v"(md[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u < mycount && rank[u] < kk) {
                    slots[rank[u]] = ((unsigned long long)md[u] << 32) | (unsigned int)qj[mystart + e0 + u];
                    below += 1 + (rank[u] << 8);
                }
        }
        if (below) atomicAdd(&qctr[5], below);
    }
    __syncthreads();
    KNN_PROBE_MARK(9);
    const bool bad = fast && qctr[5] != kk + ((kk * (kk - 1) / 2) << 8);  // (n >= kk here)
    if (part == 0 && wave_active && qi < N) {
        KNN_PROBE_STAT(0, 1);
        KNN_PROBE_STAT(1, !fast);
        KNN_PROBE_STAT(2, !usable);
        KNN_PROBE_STAT(3, qctr[4] != 0);
        KNN_PROBE_STAT(4, n > C::KCAP);
        KNN_PROBE_STAT(5, n < need);
        KNN_PROBE_STAT(6, n);
        KNN_PROBE_STAT(8, bad);
    }
    if (fast && !bad) {
        // slots [drop, kk) are the answer, in order; the query's four lanes share the writes, 16 bytes at a time
        const size_t obase = ((size_t)b * N + qi) * k;
        if ((k & 3) == 0 && ((reinterpret_cast<uintptr_t>(idx) | (dist ? reinterpret_cast<uintptr_t>(dist) : 0)) & 15) == 0) {
#pragma unroll
            for (int u = 0; u < C::KKMAX / 16; ++u) {
                const int v = part + 4 * u;
                if (4 * v < k) {
                    unsigned long long key[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) key[e] = slots[drop + 4 * v + e];
                    *reinterpret_cast<int4 *>(idx + obase + 4 * v) =
                        int4{(int)(unsigned int)key[0], (int)(unsigned int)key[1], (int)(unsigned int)key[2], (int)(unsigned int)key[3]};
                    if (dist)
                        *reinterpret_cast<float4 *>(dist + obase + 4 * v) =
                            float4{__builtin_bit_cast(float, (unsigned int)(key[0] >> 32)), __builtin_bit_cast(float, (unsigned int)(key[1] >> 32)),
                                   __builtin_bit_cast(float, (unsigned int)(key[2] >> 32)), __builtin_bit_cast(float, (unsigned int)(key[3] >> 32))};
                    if (FEAT) {
                        // four neighbours of point qi: 16-byte stores along the rank dimension (mlp layout: one per
                        // channel row; cat layout: 24 contiguous floats)
                        float cx[4], cy[4], cz[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = (int)(unsigned int)key[e];
                            if (raw_ok) {
                                const float4 rc = rawc[j];
                                cx[e] = rc.x - qr[0]; cy[e] = rc.y - qr[1]; cz[e] = rc.z - qr[2];
                            } else {
                                const float *c = yb + (size_t)j * 3;
                                cx[e] = c[0] - qr[0]; cy[e] = c[1] - qr[1]; cz[e] = c[2] - qr[2];
                            }
                        }
                        if (layout == 1 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0) {
                            const size_t KN = (size_t)k * N;
                            float *o = feat + (size_t)b * 6 * KN + (size_t)qi * k + 4 * v;
                            *reinterpret_cast<float4 *>(o) = float4{qr[0], qr[0], qr[0], qr[0]};
                            *reinterpret_cast<float4 *>(o + KN) = float4{qr[1], qr[1], qr[1], qr[1]};
                            *reinterpret_cast<float4 *>(o + 2 * KN) = float4{qr[2], qr[2], qr[2], qr[2]};
                            *reinterpret_cast<float4 *>(o + 3 * KN) = float4{cx[0], cx[1], cx[2], cx[3]};
                            *reinterpret_cast<float4 *>(o + 4 * KN) = float4{cy[0], cy[1], cy[2], cy[3]};
                            *reinterpret_cast<float4 *>(o + 5 * KN) = float4{cz[0], cz[1], cz[2], cz[3]};
                        } else if (layout == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0) {
                            float4 *o = reinterpret_cast<float4 *>(feat + (obase + 4 * v) * 6);
                            o[0] = float4{qr[0], qr[1], qr[2], cx[0]};
                            o[1] = float4{cy[0], cz[0], qr[0], qr[1]};
                            o[2] = float4{qr[2], cx[1], cy[1], cz[1]};
                            o[3] = float4{qr[0], qr[1], qr[2], cx[2]};
                            o[4] = float4{cy[2], cz[2], qr[0], qr[1]};
                            o[5] = float4{qr[2], cx[3], cy[3], cz[3]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float cc[3] = {cx[e], cy[e], cz[e]};
                                float *o0 = layout == 0 ? feat + (obase + 4 * v + e) * 6 : feat + (size_t)b * 6 * k * N + (size_t)qi * k + 4 * v + e;
                                const size_t st = layout == 0 ? 1 : (size_t)k * N;
                                o0[0] = qr[0]; o0[st] = qr[1]; o0[2 * st] = qr[2];
                                o0[3 * st] = cc[0]; o0[4 * st] = cc[1]; o0[5 * st] = cc[2];
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < C::KKMAX / 4; ++u) {
                const int r = drop + part + 4 * u;
                if (r < kk) {
                    const unsigned long long key = slots[r];
                    idx[obase + r - drop] = (int)(unsigned int)key;
                    if (dist) dist[obase + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
                    if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, qi, r - drop, qr, yb + (size_t)(unsigned int)key * 3);
                }
            }
        }
    }
    KNN_PROBE_MARK(10);
    if (!wave_active) return;
    // ties are ranked again, and the leftovers (exact merge) answered, by the group's two waves on alternate queries
    const bool slowq = qi < N && !fast && !medium;
    const unsigned long long badmask = __ballot(bad), slowmask = __ballot(slowq);
    if ((badmask | slowmask) == 0) return;
    for (unsigned int bm = (unsigned int)badmask | (unsigned int)(badmask >> 32); bm; bm &= bm - 1) {
        // a tie in the distance among the first kk of query j: the whole wave ranks its keys again, on (distance, index);
        // the group's two waves take alternate queries (lattices and duplicated points tie in every query)
        const int j = __builtin_ctz(bm);
        if ((j & 1) != half) continue;
        const int qs = grp * 32 + j;
        const int *cj = ctr + qs * 8;
        unsigned long long *sj = reinterpret_cast<unsigned long long *>(lists_all) + (size_t)qs * C::SS;
        knn_rank_ties(reinterpret_cast<const unsigned int *>(k3sm) + (size_t)qs * C::KS,
                      reinterpret_cast<const int *>(k3sm) + (size_t)C::G * 32 * C::KS + (size_t)qs * C::KS,
                      cj[0] + cj[1] + cj[2] + cj[3], kk, sj, lane);
        for (int r = drop + lane; r < kk; r += 64) {
            const unsigned long long key = sj[r];
            idx[((size_t)b * N + q0 + j) * k + r - drop] = (int)(unsigned int)key;
            if (dist) dist[((size_t)b * N + q0 + j) * k + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
            if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, q0 + j, r - drop, xb + (size_t)(q0 + j) * 3, yb + (size_t)(unsigned int)key * 3);
        }
    }
    // leftovers, wave-cooperative (scratch: behind the slots)
    int *wscratch = lists_all + C::G * 32 * C::SS * 2 + wv * 128;
    const unsigned int slow32 = (unsigned int)slowmask | (unsigned int)(slowmask >> 32);
    for (int j = half; j < 32; j += 2) {
        if (!((slow32 >> j) & 1u) || q0 + j >= N) continue;
        float bd;
        int bj;
        __builtin_amdgcn_wave_barrier();
        knn_exact_bruteforce<(C::KKMAX > 32)>(xb + (size_t)(q0 + j) * 3, yb, M, 3, kk, lane, reinterpret_cast<float *>(wscratch), wscratch + 64, bd, bj);
        const int r = lane - drop;
        if (r >= 0 && r < k) {
            idx[((size_t)b * N + q0 + j) * k + r] = bj;
            if (dist) dist[((size_t)b * N + q0 + j) * k + r] = bd;
            if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, q0 + j, r, xb + (size_t)(q0 + j) * 3, yb + (size_t)bj * 3);
        }
    }
    KNN_PROBE_MARK(11);
}

// dynamic LDS of geometry C for a cloud of M candidates without the optional parts (raw coordinates, medium-path scratch)
template <class C>
size_t knn_f16_d3_core_lds(int M) {
    int CH = (M + 63) / 64 * 64;
    if (CH > kTChunk) CH = kTChunk;
    size_t img = (size_t)CH * 32;
    const size_t keys = (size_t)C::G * 32 * C::KS * 8;
    if (img < keys) img = keys;
    return img + (size_t)C::W * C::CAP * 64 * 4 + (size_t)C::G * 32 * 8 * 4;
}
template <class C, int COMPACT = 0>  // COMPACT = blocks per CU the allocation is held to (0: one block, the whole CU)
fx3d_status launch_knn_f16_d3_geom(const float *x, int N, const float *y, int M, int B, int k, int drop, int32_t *idx,
                                   float *dist, hipStream_t st, float *feat, int layout, int xdiv) {
    int CH = (M + 63) / 64 * 64;
    if (CH > kTChunk) CH = kTChunk;
    const size_t keys = (size_t)C::G * 32 * C::KS * 8;  // distance bits + indices
    // (COMPACT: the LDS image no larger than the key arrays -- larger clouds go through it in several chunks)
    if (COMPACT && (size_t)CH * 32 > keys) CH = (int)(keys / 32 / 64 * 64);
    size_t img = (size_t)CH * 32;
    if (img < keys) img = keys;
    const size_t fixed = (size_t)C::W * C::CAP * 64 * 4 + (size_t)C::G * 32 * 8 * 4;  // lists (exchange, slots) + counters
    // COMPACT: the whole allocation stays below half a CU's LDS (two blocks per CU); the optional parts only if they fit under that
    const size_t raw_limit = COMPACT ? k3_compact_lds(COMPACT ? COMPACT : 1) : 152 * 1024, all_limit = COMPACT ? k3_compact_lds(COMPACT ? COMPACT : 1) : 156 * 1024;
    const int raw_ok = M <= kTRawMax && img + fixed + (size_t)M * 16 <= raw_limit;
    size_t lds = img + fixed + (raw_ok ? (size_t)M * 16 : 0);
    // medium path scratch (id list + merge lists per wave), when it fits next to everything else
    int med_cap = 0, med_off = 0;
    for (int cap = 512; cap >= 128; cap >>= 1)
        if (lds + (size_t)C::W * (cap + 128) * 4 <= all_limit) {
            med_cap = cap; med_off = (int)lds; lds += (size_t)C::W * (cap + 128) * 4;
            break;
        }
    const fx3d_status arc = feat ? ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_f16_d3_kernel<true, C>), 156 * 1024, "knn_f16_d3_kernel<feat>")
                                 : ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_f16_d3_kernel<false, C>), 156 * 1024, "knn_f16_d3_kernel");
    if (arc != FX3D_OK) return arc;
    const int nbx = (N + C::G * 32 - 1) / (C::G * 32);
    const int bpad = B >= 8 ? (B + 7) / 8 * 8 : B;
    if (feat)
        hipLaunchKernelGGL((knn_f16_d3_kernel<true, C>), dim3(nbx * bpad), dim3(C::T), lds, st, x, N, y, M, B, k, drop, idx, dist,
                           CH, (int)img, raw_ok, feat, layout, med_cap, med_off, xdiv);
    else
        hipLaunchKernelGGL((knn_f16_d3_kernel<false, C>), dim3(nbx * bpad), dim3(C::T), lds, st, x, N, y, M, B, k, drop, idx, dist,
                           CH, (int)img, raw_ok, feat, layout, med_cap, med_off, xdiv);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}
// k + drop <= 32: the base geometry; 33 ... 64: the wide one (both waves of a group bound tau by their own ceil(kk / 2)-th group minimum)
constexpr int kK3WideMinM = 128;  // every wave of a group needs >= 32 finite group minima: two pairs of tiles
__host__ inline bool knn_f16_d3_shape_ok(int M, int kk) {
    return M < (1 << 21) && (kk <= 32 ? M >= 64 : (kk <= 64 && M >= kK3WideMinM));
}
fx3d_status launch_knn_f16_d3(const float *x, int N, const float *y, int M, int B, int k, int drop, int32_t *idx,
                              float *dist, hipStream_t st, float *feat = nullptr, int layout = 0, int xdiv = 1) {
    const bool compact = true;
    if (k + drop <= 32) return launch_knn_f16_d3_geom<K3Base>(x, N, y, M, B, k, drop, idx, dist, st, feat, layout, xdiv);
    // (measured, tools/knn_compact_ab.py: 1.3-1.6 x over the wide geometry at M = 1600 ... 8192 too -- several image chunks --, except
    //  where many queries overflow the 88 keys and fall to the exact merge over a large cloud: k + drop > 44 with M > 4096)
    if (k + drop <= K3Mid::KKMAX && compact && (k + drop <= 44 || M <= 4096))
        return launch_knn_f16_d3_geom<K3Mid, 2>(x, N, y, M, B, k, drop, idx, dist, st, feat, layout, xdiv);
    return launch_knn_f16_d3_geom<K3Wide>(x, N, y, M, B, k, drop, idx, dist, st, feat, layout, xdiv);
}

// ------------------------------------------------------------------------------------------------
// knn_mfma_kernel<DK>: feature-space kNN (4 <= D <= 128, k+drop <= 32; the second EdgeConv runs at D = 64).
// Same selection scheme as knn_f16_d3_kernel (lane = query; 64 group minima -> tau in registers; per-lane mask
// lists; verified distance-only ranking), with the filter as a dense Float32 GEMM:
//   F[c][q] = fl(|c|^2) + sum_d c_d (-2 q_d)   on v_mfma_f32_32x32x2_f32 (rows = 32 candidates of a tile, columns
//   = the wave's 32 queries, the accumulator starts at |c|^2; two tiles on two accumulators).  With u = 2^-24 and
//   the usual gamma_n bounds, |F + |q|^2 - d_oracle| <= eps_q = 8 (D+4) u (|q|^2 + Cmax^2) for every candidate
//   (Cmax = largest candidate norm of the cloud; 2x head-room for the matrix core's internal rounding), so the
//   candidates with F <= tau + 2 eps_q are a superset of the k nearest.
// Block = 4 consumer waves (32 queries each) + 4 producer waves that stage the next candidate chunk into the other
// LDS buffer with global_load_lds_dwordx4 while the consumers work (one consumer wave per SIMD: nothing else
// would hide global latency; VALU work of any wave delays that SIMD's MFMAs, hence the direct loads).  The
// reduction dimension is permuted so that half-wave h owns d in [h*DP/2, (h+1)*DP/2): every operand fetch is a
// b128 (4 k-steps); the LDS image is lane-linear, the conflict-free rotation sits on the source addresses.
// In the exact phase all eight waves work: the four lanes of a query (two halves x consumer/producer) split its
// survivors; candidate and query rows are gathered from L2.
// Template modes: F16 = false: the Float32 GEMM above.  F16 = true (default for D % 4 == 0, M <= 4096): the cloud is
// centred per dimension and scaled by a power of two, operands are fp16; SPLIT = false (default) uses the rounded halves
// alone (one v_mfma_f32_32x32x16_f16 per K block, band 2^-10 (|q~|^2 + C~max^2)), SPLIT = true the 2-way split
// hi*hi + lo*hi + hi*lo (band 2^-18).  Producers convert while staging.
// Queries whose band holds more candidates than the key arrays (60) but whose lane lists are intact take the medium
// path (exact selection among their own survivors, up to kMMedCap); the rest of the leftovers the full exact merge.
// PRE (fx3d_knn_ws, the pre-pass has built the cloud's fp16 image): both waves of a pair run the filter (DUAL, 128 group minima per
// query).  Round 4, in this instantiation: every lane reads its pieces of its query row and of the centre straight from memory (no
// LDS staging, no barrier before the first chunk's); the image chunks come through registers (option knn_direct_lds: direct-to-LDS
// loads); one instantiation of the four-tile loop per phase (phase A folds two tiles per v_min3 on the MFMA registers, phase B starts
// the accumulators at n_c - thr and shifts the signs in with v_alignbit); tau by knn_tau_8of16; survivors counted in phase B; the
// exact phase on 16-dimension COLUMN SLICES of the whole cloud (M <= 1024, D % 16 == 0, D <= 64; option knn_row_stages: the row
// stages of rounds 2-3), pairs in registers across the slices.  C4': kernel 60.8 -> 50.8 us (profiles/r04_v5_*).
constexpr int kMLCap = 40;        // rows of a lane's mask list (39 usable + the scratch head)
constexpr int kMKeyCap = 64;      // survivors per query handled by the fast path (three sentinels follow them inside the stride of 68)
constexpr int kMMedCap = 512;     // ... by the medium path: exact selection among the query's own survivors
constexpr int kMKeyStride = 68;   // row stride of the key arrays in words: 32 queries x b128 reads without bank conflicts

// fp16-split staging of a candidate chunk (producer side, F16 filter): unit = (row, group of 8 dimensions).
// A thread converts two float4 of a row (scaled by sc) into one hi piece and one lo piece of 8 halves each.
// LDS row: pieces [0, PPR/2) = hi of dimension groups, [PPR/2, PPR) = lo; piece c of row r sits at (c + r) mod PPR.
// (row, group) units per producer thread and chunk: CH * (DP/8) <= units * 256.  The single-piece image is half
// the size, so its chunks can be twice as long (fewer steps: a step cannot be shorter than the latency of the
// global loads issued one step ahead)
constexpr int kMUnitsSplit = 5, kMUnitsSingle = 6;
template <int DK, int kMUnits>
__device__ __forceinline__ void knn_f16_load_chunk(const float *__restrict__ yb, int D, int j0, int cn, int CH, int ptid,
                                                   float4 (&reg)[kMUnits][2]) {
    constexpr int DP = DK * 32, G = DP / 8;
    // unconditional loads from clamped (always valid) addresses, zeroed afterwards: predicated loads would be
    // issued one branch at a time, each waiting for its data
    const float4 zero4 = float4{0.f, 0.f, 0.f, 0.f};
    const int cnm1 = cn - 1;
#pragma unroll
    for (int u = 0; u < kMUnits; ++u) {
        const int un = ptid + u * kMProd;
        const int row = un / G, g = un - row * G;
        const int rowc = row < cnm1 ? row : cnm1;
        const int d0 = 8 * g < D ? 8 * g : 0, d1 = 8 * g + 4 < D ? 8 * g + 4 : 0;
        const float *src = yb + (size_t)(j0 + rowc) * D;
        reg[u][0] = *reinterpret_cast<const float4 *>(src + d0);
        reg[u][1] = *reinterpret_cast<const float4 *>(src + d1);
    }
#pragma unroll
    for (int u = 0; u < kMUnits; ++u) {
        const int un = ptid + u * kMProd;
        const int row = un / G, g = un - row * G;
        const bool ok = un < CH * G && row < cn;
        if (!(ok && 8 * g < D)) reg[u][0] = zero4;
        if (!(ok && 8 * g + 4 < D)) reg[u][1] = zero4;
    }
}
// fp16 single-piece image (SPLIT = false): rows of PPI = DP/8 pieces (16 bytes = 8 halves); piece c of row r sits
// at (c + r / RPB) mod PPI, RPB = rows per 256 bytes, so that 16 consecutive rows cover all LDS banks.
template <int PPI>
__device__ __forceinline__ int knn_hpiece_off(int row, int c) {
    constexpr int RPB = PPI >= 16 ? 1 : 16 / PPI;
    return (row * PPI + ((c + row / RPB) & (PPI - 1))) * 4;
}
// Converts and stores the units; the G = DP/8 consecutive lanes that hold one row also sum its scaled norm
// (3..4 butterfly steps).  norms != nullptr: phase A, norms[row] and the running maximum are recorded.
template <int CTRL>
__device__ __forceinline__ float knn_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int DK, bool SPLIT, int kMUnits>
__device__ __forceinline__ void knn_f16_store_chunk(float *img, int CH, int cn, float sc, const float *mu_lds, int ptid,
                                                    const float4 (&reg)[kMUnits][2], float *norms, float *norms_m,
                                                    const float *acoef_lds, float &tmax, bool &tnan) {
    constexpr int DP = DK * 32, G = DP / 8, PPR = DK * 8;
    static_assert(kMProd % G == 0, "a producer thread always converts the same group of eight dimensions");
    static_assert(G == 4 || G == 8 || G == 16, "the row sum below");
    // (centre and error coefficient are re-read from LDS per call: holding them in registers across the chunk loop spills)
    float mu8[8];
    {
        const float4 m0 = *reinterpret_cast<const float4 *>(mu_lds + 8 * (ptid % G)), m1 = *reinterpret_cast<const float4 *>(mu_lds + 8 * (ptid % G) + 4);
        mu8[0] = m0.x; mu8[1] = m0.y; mu8[2] = m0.z; mu8[3] = m0.w; mu8[4] = m1.x; mu8[5] = m1.y; mu8[6] = m1.z; mu8[7] = m1.w;
    }
    const float acoef = norms_m ? *acoef_lds : 0.0f;
#pragma unroll
    for (int u = 0; u < kMUnits; ++u) {
        const int un = ptid + u * kMProd;
        const int row = un / G, g = un - row * G;
        const float v[8] = {(reg[u][0].x - mu8[0]) * sc, (reg[u][0].y - mu8[1]) * sc, (reg[u][0].z - mu8[2]) * sc,
                            (reg[u][0].w - mu8[3]) * sc, (reg[u][1].x - mu8[4]) * sc, (reg[u][1].y - mu8[5]) * sc,
                            (reg[u][1].z - mu8[6]) * sc, (reg[u][1].w - mu8[7]) * sc};
        kh8 hi, lo;
        float part = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 hh_ = (_Float16)v[e];
            hi[e] = hh_;
            if (SPLIT) lo[e] = (_Float16)(v[e] - (float)hh_);
            part = __builtin_fmaf(v[e], v[e], part);
        }
        if (un < CH * G) {
            if (SPLIT) {
                *reinterpret_cast<kh8 *>(img + knn_piece_off<DK>(row, g)) = hi;
                *reinterpret_cast<kh8 *>(img + knn_piece_off<DK>(row, PPR / 2 + g)) = lo;
            } else {
                *reinterpret_cast<kh8 *>(img + knn_hpiece_off<G>(row, g)) = hi;
            }
        }
        if (norms) {  // wave-uniform
            // sum over the row's G lanes by DPP (VALU speed; the same tree as an xor butterfly in the row's first lane,
            // the only one that uses it)
            part = part + knn_dpp<0xB1>(part);                // quad_perm [1,0,3,2]
            part = part + knn_dpp<0x4E>(part);                // quad_perm [2,3,0,1]
            if (G >= 8) part = part + knn_dpp<0x141>(part);   // row_half_mirror
            if (G >= 16) part = part + knn_dpp<0x140>(part);  // row_mirror
            if (un < CH * G && g == 0) {
                const float t = row < cn ? part : INFINITY;  // rows beyond the cloud: F = +inf
                // norms_m: the candidate's own share of the filter error is folded into its norm, upwards for the
                // threshold search (phase A), downwards for the test (phase B)
                norms[row] = norms_m ? t + acoef * t : t;
                if (norms_m) norms_m[row] = row < cn ? t - acoef * t : INFINITY;
                if (row < cn) { tnan |= (t != t); tmax = fmaxf(tmax, t); }
            }
        }
    }
}

// staged exact phase: the oracle's squared distance of one query row (registers) to two staged candidate rows (LDS),
// dimension by dimension in order.  Differences and squares two dimensions per instruction (v_pk_add/mul_f32 on the
// natural register pairs), the sums one by one; the next four pieces of both rows are in flight while four are summed.
// FULL: D == DP (no guards).
template <int DP, bool FULL>
__device__ __forceinline__ void knn_pair_dist(const f32x4v (&q)[DP / 4], const float *cp0, const float *cp1, int D, float &s0, float &s1) {
    constexpr int NB = DP / 16;
    f32x4v c0[2][4], c1[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (FULL || 4 * t < D) {
            c0[0][t] = *reinterpret_cast<const f32x4v *>(cp0 + 4 * t);
            c1[0][t] = *reinterpret_cast<const f32x4v *>(cp1 + 4 * t);
        }
#pragma unroll
    for (int bk = 0; bk < NB; ++bk) {
        if (bk + 1 < NB) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (FULL || 16 * (bk + 1) + 4 * t < D) {
                    c0[(bk + 1) & 1][t] = *reinterpret_cast<const f32x4v *>(cp0 + 16 * (bk + 1) + 4 * t);
                    c1[(bk + 1) & 1][t] = *reinterpret_cast<const f32x4v *>(cp1 + 16 * (bk + 1) + 4 * t);
                }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (FULL || 16 * bk + 4 * t < D) {
                const f32x4v d0 = q[4 * bk + t] - c0[bk & 1][t], d1 = q[4 * bk + t] - c1[bk & 1][t];
                const f32x4v m0 = d0 * d0, m1 = d1 * d1;
                s0 = s0 + m0.x; s0 = s0 + m0.y; s0 = s0 + m0.z; s0 = s0 + m0.w;
                s1 = s1 + m1.x; s1 = s1 + m1.y; s1 = s1 + m1.z; s1 = s1 + m1.w;
            }
    }
}

// staged exact phase: a thread's share of one stage of candidate rows (8 pieces of 16 bytes, rows srow + i * RPI of
// the stage that starts at row g0; clamped addresses: always valid, unused rows are never stored)
__device__ __forceinline__ void knn_stage_fetch(const float *__restrict__ yb, int D, int M, int g0, int srow, int RPI, int scol,
                                                f32x4v (&reg)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int g = g0 + srow + i * RPI;
        g = g < M ? g : M - 1;
        reg[i] = *reinterpret_cast<const f32x4v *>(yb + (size_t)g * D + scol);
    }
}

// ------------------------------------------------------------------------------------------------
// Pre-pass of the feature-space kNN (fx3d_knn_ws): the per-cloud statistics and the fp16 image are built ONCE per cloud
// instead of by every block of the cloud (8 blocks per cloud at C4': the scale pass alone was 8 us of the 78, bound by the
// L2 -- every block read the whole cloud -- and the producers' conversion VALU delayed the consumers' MFMAs in both phases).
//   knn_pre_stats_kernel   grid (kPreParts, B): minima / maxima / sums per dimension of one eighth of the cloud's rows
//   knn_pre_image_kernel   grid (kPreParts, B): combines the eight parts (every block the same arithmetic: identical centre
//                          and scale), robust centre as in knn_mfma_kernel, converts its rows: fp16 image [Mpad][DP], scaled
//                          row norms (the candidate's error share folded in, upwards / downwards), largest norm of the part
// knn_mfma_kernel then reads the header, and its producer waves bring image chunks and norms in with direct-to-LDS loads (no
// VALU).  Workspace per cloud: KnnPre::cloud_bytes(M, DP).
constexpr int kPreParts = 8;
constexpr int kPreThreads = 256;
struct KnnPre {
    unsigned char *base;  // workspace
    size_t stride;        // bytes per cloud
    int Mpad;             // rows of the image (multiple of 256: chunks never need a clamp), DP halves per row
    // offsets inside a cloud's slab (bytes)
    size_t off_parts, off_hdr, off_cmax, off_nup, off_ndn, off_img;
    __host__ __device__ static KnnPre make(void *ws, int M, int DP) {
        KnnPre k{};
        k.base = static_cast<unsigned char *>(ws);
        k.Mpad = (M + 255) / 256 * 256;
        size_t o = 0;
        k.off_parts = o; o += (size_t)kPreParts * (3 * DP + 4) * 4;
        k.off_hdr = o; o += (size_t)(8 + DP) * 4;
        k.off_cmax = o; o += (size_t)kPreParts * 4;
        o = (o + 63) & ~(size_t)63;
        k.off_nup = o; o += (size_t)k.Mpad * 4;
        k.off_ndn = o; o += (size_t)k.Mpad * 4;
        k.off_img = o; o += (size_t)k.Mpad * DP * 2;
        k.stride = (o + 255) & ~(size_t)255;
        return k;
    }
    __host__ __device__ float *parts(int b) const { return reinterpret_cast<float *>(base + (size_t)b * stride + off_parts); }
    __host__ __device__ float *hdr(int b) const { return reinterpret_cast<float *>(base + (size_t)b * stride + off_hdr); }
    __host__ __device__ unsigned int *cmaxp(int b) const { return reinterpret_cast<unsigned int *>(base + (size_t)b * stride + off_cmax); }
    __host__ __device__ float *nup(int b) const { return reinterpret_cast<float *>(base + (size_t)b * stride + off_nup); }
    __host__ __device__ float *ndn(int b) const { return reinterpret_cast<float *>(base + (size_t)b * stride + off_ndn); }
    __host__ __device__ _Float16 *img(int b) const { return reinterpret_cast<_Float16 *>(base + (size_t)b * stride + off_img); }
};
// header floats: [0] sc  [1] funit  [2] acoef (candidate side)  [3] bits: 1 = non-finite / overflow-prone cloud  [8 ...] mu[DP]

// The parts' statistics travel between the blocks of a cloud INSIDE the fused pre-pass kernel: device-coherent accesses (relaxed
// atomics at agent scope: write-through stores, loads that do not hit a stale line -- the blocks may sit on different XCDs, each
// with its own L2) instead of agent-scope fences, which write back / invalidate a whole L2 per block (measured: + 0.1 us per block
// of the grid, serialised per XCD: C4' 76 -> 104 us).
__device__ __forceinline__ void knn_pre_put(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float knn_pre_get(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void knn_pre_stats_body(const float *__restrict__ y, int M, int D, int DP, const KnnPre &pre, int part, int b,
                                                   float *red) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *yb = y + (size_t)b * M * D;
    const int rq = D / 4;  // (kPreThreads % rq == 0: thread t always sees dimensions 4 (t % rq) ...)
    const int per = (M + kPreParts - 1) / kPreParts;
    const int r_lo = part * per < M ? part * per : M, r_hi = r_lo + per < M ? r_lo + per : M;
    const float4 *c4 = reinterpret_cast<const float4 *>(yb) + (size_t)r_lo * rq;
    const int total4 = (r_hi - r_lo) * rq;
    float4 lo4 = float4{INFINITY, INFINITY, INFINITY, INFINITY}, hi4 = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float4 sum4 = float4{0.f, 0.f, 0.f, 0.f};
    float poison = 0.0f;
    for (int e0 = tid; e0 < total4; e0 += 8 * kPreThreads) {
        float4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c4[e0 + e * kPreThreads < total4 ? e0 + e * kPreThreads : e0];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            poison = __builtin_fmaf(v[e].x, 0.0f, poison); poison = __builtin_fmaf(v[e].y, 0.0f, poison);
            poison = __builtin_fmaf(v[e].z, 0.0f, poison); poison = __builtin_fmaf(v[e].w, 0.0f, poison);
            lo4.x = fminf(lo4.x, v[e].x); lo4.y = fminf(lo4.y, v[e].y); lo4.z = fminf(lo4.z, v[e].z); lo4.w = fminf(lo4.w, v[e].w);
            hi4.x = fmaxf(hi4.x, v[e].x); hi4.y = fmaxf(hi4.y, v[e].y); hi4.z = fmaxf(hi4.z, v[e].z); hi4.w = fmaxf(hi4.w, v[e].w);
            if (e0 + e * kPreThreads < total4) {
                sum4.x = sum4.x + v[e].x; sum4.y = sum4.y + v[e].y; sum4.z = sum4.z + v[e].z; sum4.w = sum4.w + v[e].w;
            }
        }
    }
    const bool anynan = __syncthreads_or(poison != poison) != 0;
    for (int m = rq; m < 64; m <<= 1) {  // lanes with equal lane % rq hold the same dimensions
        lo4.x = fminf(lo4.x, __shfl_xor(lo4.x, m, 64)); lo4.y = fminf(lo4.y, __shfl_xor(lo4.y, m, 64));
        lo4.z = fminf(lo4.z, __shfl_xor(lo4.z, m, 64)); lo4.w = fminf(lo4.w, __shfl_xor(lo4.w, m, 64));
        hi4.x = fmaxf(hi4.x, __shfl_xor(hi4.x, m, 64)); hi4.y = fmaxf(hi4.y, __shfl_xor(hi4.y, m, 64));
        hi4.z = fmaxf(hi4.z, __shfl_xor(hi4.z, m, 64)); hi4.w = fmaxf(hi4.w, __shfl_xor(hi4.w, m, 64));
        sum4.x = sum4.x + __shfl_xor(sum4.x, m, 64); sum4.y = sum4.y + __shfl_xor(sum4.y, m, 64);
        sum4.z = sum4.z + __shfl_xor(sum4.z, m, 64); sum4.w = sum4.w + __shfl_xor(sum4.w, m, 64);
    }
    if (lane < rq) {
        float *r8 = red + (size_t)(wv * 32 + lane) * 12;
        r8[0] = lo4.x; r8[1] = lo4.y; r8[2] = lo4.z; r8[3] = lo4.w;
        r8[4] = hi4.x; r8[5] = hi4.y; r8[6] = hi4.z; r8[7] = hi4.w;
        r8[8] = sum4.x; r8[9] = sum4.y; r8[10] = sum4.z; r8[11] = sum4.w;
    }
    __syncthreads();
    float *out = pre.parts(b) + (size_t)part * (3 * DP + 4);
    if (tid < rq) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float lo = INFINITY, hi = -INFINITY, sm = 0.0f;
            for (int w = 0; w < kPreThreads / 64; ++w) {
                const float *r8 = red + (size_t)(w * 32 + tid) * 12;
                lo = fminf(lo, r8[c]); hi = fmaxf(hi, r8[4 + c]); sm = sm + r8[8 + c];
            }
            knn_pre_put(out + 4 * tid + c, lo); knn_pre_put(out + DP + 4 * tid + c, hi); knn_pre_put(out + 2 * DP + 4 * tid + c, sm);
        }
    }
    if (tid == 0) knn_pre_put(out + 3 * DP, __builtin_bit_cast(float, anynan ? 1 : 0));
}
__global__ __launch_bounds__(kPreThreads) void knn_pre_stats_kernel(const float *__restrict__ y, int M, int D, int DP, KnnPre pre) {
    __shared__ float red[(kPreThreads / 64) * 32 * 12];
    knn_pre_stats_body(y, M, D, DP, pre, blockIdx.x, blockIdx.y, red);
}

// mu: [DP] floats, sh: 4 words of LDS ([0] bits of the extent  [1] skew flag  [2] bits of the bulk radius  [3] largest norm of the part)
template <int DK>
__device__ __forceinline__ void knn_pre_image_body(const float *__restrict__ y, int M, int D, int two_norms, const KnnPre &pre, int part, int b,
                                                   float *mu, unsigned int *sh) {
    constexpr int DP = DK * 32, G = DP / 8;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *yb = y + (size_t)b * M * D;
    const int rq = D / 4;
    const float *parts = pre.parts(b);
    // this part's rows: the first four sweeps (a whole part at C4': 128 rows) are requested BEFORE the parts' statistics are read --
    // the rows do not depend on them, and the kernel is two dependent global round trips otherwise (round 4)
    const int per = (M + kPreParts - 1) / kPreParts;
    const int r_lo = part * per < M ? part * per : M, r_hi = r_lo + per < M ? r_lo + per : M;
    const int g = tid % G;  // (kPreThreads % G == 0: a thread always converts the same eight dimensions)
    constexpr int RPS = kPreThreads / G;  // rows per sweep of the block
    const int d0 = 8 * g < D ? 8 * g : 0, d1 = 8 * g + 4 < D ? 8 * g + 4 : 0;
    float4 a0[4], a1[4];
    auto load_rows = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // every load of the four sweeps before the first use (a part is a few sweeps: latency, not bandwidth)
            const int row = r0 + u * RPS + tid / G;
            const float *src = yb + (size_t)(row < r_hi ? row : (r_lo < M ? r_lo : 0)) * D;
            a0[u] = *reinterpret_cast<const float4 *>(src + d0);
            a1[u] = *reinterpret_cast<const float4 *>(src + d1);
        }
    };
    load_rows(r_lo);
    if (tid < 4) sh[tid] = 0u;
    bool anynan = false;
    for (int p = 0; p < kPreParts; ++p) anynan |= __builtin_bit_cast(int, knn_pre_get(parts + (size_t)p * (3 * DP + 4) + 3 * DP)) != 0;
    __syncthreads();
    if (tid < rq) {
        float amax = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float lo = INFINITY, hi = -INFINITY, sm = 0.0f;
            for (int p = 0; p < kPreParts; ++p) {  // (fixed order: every block of the cloud gets the same centre)
                const float *q = parts + (size_t)p * (3 * DP + 4);
                lo = fminf(lo, knn_pre_get(q + 4 * tid + c)); hi = fmaxf(hi, knn_pre_get(q + DP + 4 * tid + c)); sm = sm + knn_pre_get(q + 2 * DP + 4 * tid + c);
            }
            float m0 = sm / (float)M;
            m0 = fminf(fmaxf(m0, lo), hi);
            mu[4 * tid + c] = m0;
            amax = fmaxf(amax, fmaxf(hi - m0, m0 - lo));
            if (fabsf(m0 - 0.5f * (lo + hi)) > 0.25f * (hi - lo)) sh[1] = 1u;
        }
        atomicMax(&sh[0], __builtin_bit_cast(unsigned int, amax));
    } else if (tid < DP / 4) {
        mu[4 * tid] = 0.0f; mu[4 * tid + 1] = 0.0f; mu[4 * tid + 2] = 0.0f; mu[4 * tid + 3] = 0.0f;
    }
    __syncthreads();
    if (sh[1] != 0u && !anynan) {  // robust centre: see knn_mfma_kernel (the same rule on the same 16 sampled rows)
        if (wv == 0) {
            float shiftmax = 0.0f, iqr2 = 0.0f;
            bool sw = false;
            float medv[DP / 64 > 0 ? DP / 64 : 1];
#pragma unroll
            for (int t = 0; t < (DP + 63) / 64; ++t) {
                const int d = lane + 64 * t;
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = yb[(size_t)((long long)i * M / 16) * D + (d < D ? d : 0)];
                float med = v[0], q1 = v[0], q3 = v[0];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    int rk = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) rk += (v[j] < v[i] || (v[j] == v[i] && j < i)) ? 1 : 0;
                    med = rk == 8 ? v[i] : med; q1 = rk == 4 ? v[i] : q1; q3 = rk == 12 ? v[i] : q3;
                }
                const float shift = d < D ? fabsf(mu[d < D ? d : 0] - med) : 0.0f;
                sw = sw || (shift > 8.0f * (q3 - q1));
                shiftmax = fmaxf(shiftmax, shift);
                if (d < D) iqr2 = __builtin_fmaf(q3 - q1, q3 - q1, iqr2);
                medv[t] = med;
            }
            if (__ballot(sw) != 0ull) {
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) {
                    shiftmax = fmaxf(shiftmax, __shfl_xor(shiftmax, m, 64));
                    iqr2 = iqr2 + __shfl_xor(iqr2, m, 64);
                }
#pragma unroll
                for (int t = 0; t < (DP + 63) / 64; ++t)
                    if (lane + 64 * t < D) mu[lane + 64 * t] = medv[t];
                if (lane == 0) {
                    sh[0] = __builtin_bit_cast(unsigned int, __builtin_bit_cast(float, sh[0]) + shiftmax);
                    sh[2] = __builtin_bit_cast(unsigned int, sqrtf(iqr2));
                }
            }
        }
        __syncthreads();
    }
    const float cinf = __builtin_bit_cast(float, sh[0]);
    float sc = 1.0f;
    if (cinf > 1.0e-30f && cinf < 1.0e30f) {
        int e;
        (void)frexpf(cinf * 1.000001f, &e);
        sc = ldexpf(1.0f, 10 - e);
    }
    const float rad = __builtin_bit_cast(float, sh[2]);
    const float funit = rad > 0.0f ? fminf(1.0f, fmaxf(sc * rad, 0x1p-12f)) : 1.0f;
    const float aq = 8.0f * (float)(4 * D + 8) * 0x1p-24f + 0x1.01p-10f;
    const float acoef = aq * 1.01f + 0x1p-26f * sqrtf((float)D) / funit + 0x1p-23f;
    const bool bad = anynan || !(cinf < 1.0e15f);
    if (part == 0) {
        float *h = pre.hdr(b);
        if (tid == 0) { h[0] = sc; h[1] = funit; h[2] = acoef; h[3] = bad ? 1.0f : 0.0f; }
        if (tid < DP) h[8 + tid] = mu[tid];
    }
    // ---- this part's rows -> image, norms
    _Float16 *img = pre.img(b);
    float *nup = pre.nup(b), *ndn = pre.ndn(b);
    float mu8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) mu8[e] = mu[8 * g + e];
    float tmax = 0.0f;
    bool tnan = false;
    for (int r0 = r_lo; r0 < r_hi; r0 += 4 * RPS) {
        if (r0 != r_lo) load_rows(r0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = r0 + u * RPS + tid / G;
            const bool ok = row < r_hi;
            if (!(8 * g < D)) a0[u] = float4{mu8[0], mu8[1], mu8[2], mu8[3]};       // padding dimensions: zero pieces
            if (!(8 * g + 4 < D)) a1[u] = float4{mu8[4], mu8[5], mu8[6], mu8[7]};
            const float v[8] = {(a0[u].x - mu8[0]) * sc, (a0[u].y - mu8[1]) * sc, (a0[u].z - mu8[2]) * sc, (a0[u].w - mu8[3]) * sc,
                                (a1[u].x - mu8[4]) * sc, (a1[u].y - mu8[5]) * sc, (a1[u].z - mu8[6]) * sc, (a1[u].w - mu8[7]) * sc};
            kh8 hi;
            float pt = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hi[e] = (_Float16)v[e];
                pt = __builtin_fmaf(v[e], v[e], pt);
            }
            pt = pt + knn_dpp<0xB1>(pt);
            pt = pt + knn_dpp<0x4E>(pt);
            if (G >= 8) pt = pt + knn_dpp<0x141>(pt);
            if (G >= 16) pt = pt + knn_dpp<0x140>(pt);
            if (ok) {
                *reinterpret_cast<kh8 *>(img + ((size_t)row * G + g) * 8) = hi;
                if (g == 0) {
                    nup[row] = two_norms ? pt + acoef * pt : pt;
                    ndn[row] = two_norms ? pt - acoef * pt : pt;
                    tnan |= (pt != pt);
                    tmax = fmaxf(tmax, pt);
                }
            }
        }
    }
    if (part == kPreParts - 1) {  // rows [M, Mpad): zero pieces, norm +inf (F = +inf: never selected)
        kh8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
        for (int un = tid; un < (pre.Mpad - M) * G; un += kPreThreads) *reinterpret_cast<kh8 *>(img + ((size_t)M * G + un) * 8) = z;
        for (int r = M + tid; r < pre.Mpad; r += kPreThreads) { nup[r] = INFINITY; ndn[r] = INFINITY; }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, m, 64));
    const bool anyn = __ballot(tnan) != 0;
    if (lane == 0) atomicMax(&sh[3], anyn ? 0x7fc00000u : __builtin_bit_cast(unsigned int, tmax));
    __syncthreads();
    if (tid == 0) pre.cmaxp(b)[part] = bad ? 0x7fc00000u : sh[3];
}
template <int DK>
__global__ __launch_bounds__(kPreThreads) void knn_pre_image_kernel(const float *__restrict__ y, int M, int D, int two_norms, KnnPre pre) {
    __shared__ float mu[DK * 32];
    __shared__ unsigned int sh[4];
    knn_pre_image_body<DK>(y, M, D, two_norms, pre, blockIdx.x, blockIdx.y, mu, sh);
}
// (Round 4 tried the whole pre-pass as ONE launch of one 1024-thread block per cloud -- the cloud in registers between the statistics
//  and the conversion, no meeting: 12.7 us under rocprofv3 against 4.8 + 6.4 for the two launches, calls 1-2 us slower at four of five
//  shapes (profiles/r04_v4_knn_prepass_ab.txt): 32 CUs stream 256 KB each at ~50 GB/s.  Removed; the two launches stay.)

// producer wave pw brings the norms of chunk [j0, j0 + CH) into the block's norm arrays (direct-to-LDS; not waited for here)
__device__ __forceinline__ void knn_pre_stage_norms(const float *__restrict__ gnup, const float *__restrict__ gndn, int j0, int CH, float *nup,
                                                    float *ndn, int pw, int lane) {
    const int nin = CH / 4 / 64;  // wave-instructions per norm array (CH / 4 pieces of four floats); CH = 64 -> a quarter wave
    for (int i = pw; i < (nin > 0 ? nin : 1); i += kMWaves)
        if (i * 64 + lane < CH / 4) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gnup + j0 + (size_t)(i * 64 + lane) * 4),
                                             (__attribute__((address_space(3))) void *)(nup + (size_t)i * 256), 16, 0, 0);
            if (ndn)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gndn + j0 + (size_t)(i * 64 + lane) * 4),
                                                 (__attribute__((address_space(3))) void *)(ndn + (size_t)i * 256), 16, 0, 0);
        }
}
// producer wave pw brings chunk [j0, j0 + CH) of the pre-pass image into `img` (single-piece layout of knn_hpiece_off: the
// rotation sits on the source address) and, in phase A, its norms into the block's norm arrays -- direct-to-LDS loads only
template <int DK, bool WAIT = true>
__device__ __forceinline__ void knn_pre_stage_chunk(const _Float16 *__restrict__ gimg, const float *__restrict__ gnup,
                                                    const float *__restrict__ gndn, int j0, int CH, float *img, float *nup,
                                                    float *ndn, bool norms, int pw, int lane) {
    constexpr int PPI = DK * 4;                       // 16-byte pieces per image row (DP halves)
    constexpr int RPB = PPI >= 16 ? 1 : 16 / PPI;
    const int ninstr = CH * PPI / 64 / kMWaves;       // wave-instructions of this producer wave (CH is a multiple of 64)
    for (int i = 0; i < ninstr; ++i) {
        const int S0 = (pw * ninstr + i) * 64;        // first 16-byte slot of this wave-instruction
        const int S = S0 + lane;
        const int row = S / PPI, pos = S & (PPI - 1);
        const int c = (pos - row / RPB) & (PPI - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gimg + ((size_t)(j0 + row) * PPI + c) * 8),
                                         (__attribute__((address_space(3))) void *)(img + (size_t)S0 * 4), 16, 0, 0);
    }
    if (norms) {
        const int nin = CH / 4 / 64;  // wave-instructions per norm array (CH / 4 pieces of four floats); CH = 64 -> a quarter wave
        for (int i = pw; i < (nin > 0 ? nin : 1); i += kMWaves)
            if (i * 64 + lane < CH / 4) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gnup + j0 + (size_t)(i * 64 + lane) * 4),
                                                 (__attribute__((address_space(3))) void *)(nup + (size_t)i * 256), 16, 0, 0);
                if (ndn)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gndn + j0 + (size_t)(i * 64 + lane) * 4),
                                                     (__attribute__((address_space(3))) void *)(ndn + (size_t)i * 256), 16, 0, 0);
            }
    }
    if (WAIT) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces have landed
        __builtin_amdgcn_wave_barrier();
    }
}

template <int DK, bool F16, bool SPLIT, bool PRE = false>
__global__ __launch_bounds__(kMThreads) void knn_mfma_kernel(const float *__restrict__ x, int N,
                                                             const float *__restrict__ y, int M, int B, int D,
                                                             int k, int drop, int32_t *__restrict__ idx,
                                                             float *__restrict__ dist, int CH, int img_floats,
                                                             int keep_norms, int two_norms, int srl, void *pre_ws, int xdiv, int csl, int regstage) {
    constexpr int DP = DK * 32;      // padded feature dimension
    constexpr int RS = DP + 4;       // row stride of the query rows staged in the prologue (floats)
    constexpr int PPR = DK * 8;      // 16-byte pieces per candidate row
    constexpr int NT = DP / 8;       // b128 operand fetches per tile and half
    constexpr int NB16 = DP / 16;    // K blocks of the fp16 filter
    constexpr int RSI = (F16 && !SPLIT) ? DP / 2 : DP;  // image row stride in floats (single-piece fp16: hi halves only)
    constexpr int PPI = RSI / 4;     // 16-byte pieces per image row
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int buf_floats = CH * RSI + CH;                                  // image [CH][RSI] + norms [CH]
    // fixed-size bookkeeping first, then lists | med | nall: from the lists on everything is dead once the survivors are
    // decoded, so the staged exact phase uses that whole tail of the allocation for candidate rows
    int *lcnt = reinterpret_cast<int *>(sm + img_floats);                  // [kMWaves][64]  list lengths
    int *qn_n = lcnt + kMWaves * 64;                                       // [kMWaves][32]  survivors per query
    int *qflag = qn_n + kMWaves * 32;                                      // [kMWaves][32]  1 = fast path
    int *qbelow = qflag + kMWaves * 32;                                    // [kMWaves][32]  entries with rank < kk
    unsigned int *cmax = reinterpret_cast<unsigned int *>(qbelow + kMWaves * 32);  // bits of max |c|^2 (>= 0)
    float *mu = reinterpret_cast<float *>(cmax + 4);                       // [DP] F16: per-dimension centre of the cloud
    unsigned long long *qstpk = reinterpret_cast<unsigned long long *>(mu + DP);  // [kMWaves][32] survivors per row stage (packed prefix)
    unsigned short *lcnt2 = reinterpret_cast<unsigned short *>(lcnt);      // DUAL (below): [2 kMWaves][64] list lengths, in lcnt's space
    int *lists = reinterpret_cast<int *>(qstpk + kMWaves * 32);            // [kMWaves][kMLCap][64] mask words (DUAL: [2 kMWaves][kMLCap / 2][64])
    int *med = lists + kMWaves * kMLCap * 64;                              // [2 kMWaves][kMMedCap + 128] medium path: ids + merge lists
    float *nall = reinterpret_cast<float *>(med + 2 * kMWaves * (kMMedCap + 128));  // [nchunk*CH] all candidate norms (keep_norms)
    // DUAL: the four parts' packed stage counts per query [kMWaves][32][4] (4 KiB) live in the norm arrays, which are dead after
    // the chunk loop and at least that large (one array of >= 2304 floats, or two of >= 512)
    unsigned long long *qpk = reinterpret_cast<unsigned long long *>(nall);
    // block L runs on XCD L % 8: give every cloud's blocks ids with equal L % 8 so that its candidates stay in
    // one L2 (8 or more clouds; fewer: plain order, a cloud's blocks spread over all XCDs)
    const int nbx = (N + kMWaves * 32 - 1) / (kMWaves * 32);
    const int L = blockIdx.x;
    const bool by_xcd = B >= 8;
    const int b = by_xcd ? ((L >> 3) / nbx) * 8 + (L & 7) : L / nbx;
    const int bxq = by_xcd ? (L >> 3) % nbx : L % nbx;
    if (b >= B) return;
    // pre-pass workspace (fx3d_knn_ws): this cloud's image and norms; the layout is a function of (M, DP)
    constexpr bool use_pre = PRE;  // (a separate instantiation: the kernel without a pre-pass keeps its registers)
    const _Float16 *pre_img = nullptr;
    const float *pre_nup = nullptr, *pre_ndn = nullptr;
    if (use_pre) {
        const KnnPre pre = KnnPre::make(pre_ws, M, DP);
        pre_img = pre.img(b); pre_nup = pre.nup(b); pre_ndn = pre.ndn(b);
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool consumer = wv < kMWaves;
    const int cw = consumer ? wv : wv - kMWaves;   // the consumer wave this wave is paired with
    const int ptid = tid - kMProd;                 // producer thread id (negative for consumers)
    // PRE: the image arrives by direct-to-LDS loads, the "producer" waves are free -- BOTH waves of a pair (they share a SIMD)
    // run the filter, on alternate double pairs of tiles: two waves per SIMD hide each other's LDS latencies and MFMA -> VALU
    // dependencies (a lone consumer wave stalled for more than half of its cycles).  Per query 128 group minima instead of 64
    // (a tighter tau), four lane lists instead of two (the decode is shared by four lanes).
    constexpr bool DUAL = PRE;
    constexpr int LCAP = DUAL ? kMLCap / 2 : kMLCap;  // rows of a lane's mask list
    const int half = consumer ? 0 : 1;
    const int h = lane >> 5, jl = lane & 31;
    const int kk = k + drop;
    const float *xb = x + (size_t)(b / xdiv) * N * D, *yb = y + (size_t)b * M * D;  // (xdiv > 1: candidate slices as virtual clouds share their queries)
    const int q0 = (bxq * kMWaves + cw) * 32;
    const bool wave_active = q0 < N;
    const int qi = q0 + jl;
    const bool vec4y = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(yb) & 15) == 0);
    const bool vec4x = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(xb) & 15) == 0);
    const int nchunk = (M + CH - 1) / CH;
    // filter error per unit of (candidate norm + query norm), scaled units: fp32 accumulation + centring + the oracle's own
    // rounding 8 (4D + 8) u (4x head-room), operand representation 2^-10 (rounded halves) or 2^-18 (2-way split)
    float *nallm = two_norms ? nall + (size_t)nchunk * CH : nullptr;  // [nchunk*CH] norms for the phase-B test
    KNN_PROBE_MARK(0);

    if (tid == 0) {
        *cmax = 0u;
        cmax[2] = 0u;  // F16: a mean far from the middle of its range was seen (scale pass)
        cmax[3] = 0u;  // F16: bulk radius of a cloud centred on its medians (0: not in use)
        const float aq = 8.0f * (float)(4 * D + 8) * 0x1p-24f + (SPLIT ? 0x1p-18f : 0x1.01p-10f);
        // candidate side: + its share of the subnormal floor, + the rounding of n (1 +- A); parked in LDS (cmax[1])
        reinterpret_cast<float *>(cmax)[1] = aq * 1.01f + 0x1p-26f * sqrtf((float)D) + 0x1p-23f;
    }
    float sc = 1.0f;  // F16: power-of-two scale with |sc * c| < 1 for every candidate
    float funit = 1.0f;  // F16: unit of the absolute error terms (see the scale pass)
    // (round 4) behind the pre-pass the kernel's start is ONE global round trip: the first chunk of the image (and the norms) is requested
    // right here, and every lane reads its pieces of its query row and of the centre straight from memory (below) -- no staging of the
    // query rows through LDS, no barrier before the first chunk's.  (It was three dependent round trips -- header, query rows in a loop
    // of load -> LDS store, first chunk -- and two block barriers: 10.4 k cycles.)
    const bool early = F16 && use_pre && vec4x;
    if (F16 && use_pre) {
        // ---- the pre-pass (knn_pre_*_kernel) has the centre, the scale and the largest scaled norm of this cloud
        if (!early) __syncthreads();
        const KnnPre pre = KnnPre::make(pre_ws, M, DP);
        const float *h = pre.hdr(b);
        sc = h[0];
        funit = h[1];
        if (!early && tid < DP) mu[tid] = h[8 + tid];
        if (tid == 0) {  // (the same thread zeroed these words above; they are read after the chunk loop's barriers)
            reinterpret_cast<float *>(cmax)[1] = h[2];
            unsigned int m = h[3] != 0.0f ? 0x7fc00000u : 0u;
            const unsigned int *cp = pre.cmaxp(b);
            for (int p = 0; p < kPreParts; ++p) m = cp[p] > m ? cp[p] : m;  // (NaN pattern > every finite norm)
            *cmax = m;
        }
        if (!early) __syncthreads();
    } else if (F16) {
        // ---- centre and scale: per-dimension MEAN mu (robust against a few far points, unlike the mid-range) and the
        //      largest |c - mu| of the cloud, one coalesced pass (F16 => 16-byte loads are legal).  Distances do not
        //      depend on the origin, the fp16 band does: it grows with |q~|^2 + |c~|^2, so a common offset of a few
        //      standard deviations would flood the lists.  Thread t always sees the same four dimensions when the
        //      block size is a multiple of D/4.  (Any mu is correct; it only has to be the same for all points.)
        __syncthreads();
        const int rq = D / 4;
        const bool centre = (kMThreads % rq) == 0 && rq <= 32;
        float4 lo4 = float4{INFINITY, INFINITY, INFINITY, INFINITY}, hi4 = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        float4 sum4 = float4{0.f, 0.f, 0.f, 0.f};
        bool tnan = false;
        float poison = 0.0f;
        const float4 *c4 = reinterpret_cast<const float4 *>(yb);
        const int total4 = M * (D / 4);
        constexpr int kInFlight = 8;  // 16-byte loads in flight per thread (16 did not help: the pass is bound by the L2, every block reads its whole cloud)
        for (int e0 = tid; e0 < total4; e0 += kInFlight * kMThreads) {
            float4 v[kInFlight];
#pragma unroll
            for (int e = 0; e < kInFlight; ++e) v[e] = c4[e0 + e * kMThreads < total4 ? e0 + e * kMThreads : e0];  // (clamped: same dimensions)
#pragma unroll
            for (int e = 0; e < kInFlight; ++e) {
                // NaN or +-inf coordinates (x * 0 is NaN for them): no scale exists, every query takes the exact path
                poison = __builtin_fmaf(v[e].x, 0.0f, poison); poison = __builtin_fmaf(v[e].y, 0.0f, poison);
                poison = __builtin_fmaf(v[e].z, 0.0f, poison); poison = __builtin_fmaf(v[e].w, 0.0f, poison);
                lo4.x = vmin_f32(lo4.x, v[e].x); lo4.y = vmin_f32(lo4.y, v[e].y); lo4.z = vmin_f32(lo4.z, v[e].z); lo4.w = vmin_f32(lo4.w, v[e].w);
                hi4.x = vmax_f32(hi4.x, v[e].x); hi4.y = vmax_f32(hi4.y, v[e].y); hi4.z = vmax_f32(hi4.z, v[e].z); hi4.w = vmax_f32(hi4.w, v[e].w);
                if (e0 + e * kMThreads < total4) {  // (the clamped duplicates must not enter the mean)
                    sum4.x = sum4.x + v[e].x; sum4.y = sum4.y + v[e].y; sum4.z = sum4.z + v[e].z; sum4.w = sum4.w + v[e].w;
                }
            }
        }
        tnan = poison != poison;
        const bool anynan = __syncthreads_or(tnan) != 0;
        float *red = sm;  // [kMThreads / 64][32][12] scratch in the (still unused) chunk buffers
        if (centre) {
            for (int m = rq; m < 64; m <<= 1) {  // lanes with equal lane % rq hold the same dimensions
                lo4.x = fminf(lo4.x, __shfl_xor(lo4.x, m, 64)); lo4.y = fminf(lo4.y, __shfl_xor(lo4.y, m, 64));
                lo4.z = fminf(lo4.z, __shfl_xor(lo4.z, m, 64)); lo4.w = fminf(lo4.w, __shfl_xor(lo4.w, m, 64));
                hi4.x = fmaxf(hi4.x, __shfl_xor(hi4.x, m, 64)); hi4.y = fmaxf(hi4.y, __shfl_xor(hi4.y, m, 64));
                hi4.z = fmaxf(hi4.z, __shfl_xor(hi4.z, m, 64)); hi4.w = fmaxf(hi4.w, __shfl_xor(hi4.w, m, 64));
                sum4.x = sum4.x + __shfl_xor(sum4.x, m, 64); sum4.y = sum4.y + __shfl_xor(sum4.y, m, 64);
                sum4.z = sum4.z + __shfl_xor(sum4.z, m, 64); sum4.w = sum4.w + __shfl_xor(sum4.w, m, 64);
            }
            if (lane < rq) {
                float *r8 = red + (size_t)(wv * 32 + lane) * 12;
                r8[0] = lo4.x; r8[1] = lo4.y; r8[2] = lo4.z; r8[3] = lo4.w;
                r8[4] = hi4.x; r8[5] = hi4.y; r8[6] = hi4.z; r8[7] = hi4.w;
                r8[8] = sum4.x; r8[9] = sum4.y; r8[10] = sum4.z; r8[11] = sum4.w;
            }
        }
        __syncthreads();
        float amax = 0.0f;
        if (centre) {
            if (tid < rq) {
                float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                float sm4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int w = 0; w < kMThreads / 64; ++w) {
                    const float *r8 = red + (size_t)(w * 32 + tid) * 12;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { lo[c] = fminf(lo[c], r8[c]); hi[c] = fmaxf(hi[c], r8[4 + c]); sm4[c] = sm4[c] + r8[8 + c]; }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float m0 = sm4[c] / (float)M;
                    m0 = fminf(fmaxf(m0, lo[c]), hi[c]);  // (rounding of the sum cannot leave the range)
                    mu[4 * tid + c] = m0;
                    amax = fmaxf(amax, fmaxf(hi[c] - m0, m0 - lo[c]));
                    // a mean far from the middle of its range: skewed data or a few far points (checked below on a sample)
                    if (fabsf(m0 - 0.5f * (lo[c] + hi[c])) > 0.25f * (hi[c] - lo[c])) cmax[2] = 1u;
                }
            } else if (tid < DP / 4) {
                mu[4 * tid] = 0.0f; mu[4 * tid + 1] = 0.0f; mu[4 * tid + 2] = 0.0f; mu[4 * tid + 3] = 0.0f;
            }
        } else {
            if (tid < DP / 4) { mu[4 * tid] = 0.0f; mu[4 * tid + 1] = 0.0f; mu[4 * tid + 2] = 0.0f; mu[4 * tid + 3] = 0.0f; }
            if (tid < total4)  // (threads without an element hold +-inf)
                amax = fmaxf(fmaxf(fmaxf(fabsf(lo4.x), fabsf(hi4.x)), fmaxf(fabsf(lo4.y), fabsf(hi4.y))),
                             fmaxf(fmaxf(fabsf(lo4.z), fabsf(hi4.z)), fmaxf(fabsf(lo4.w), fabsf(hi4.w))));
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
        if (lane == 0) atomicMax(cmax, anynan ? 0x7fc00000u : __builtin_bit_cast(unsigned int, amax));
        __syncthreads();
        if (centre && cmax[2] != 0u && !anynan) {  // (block-uniform)
            // ---- robust centre.  A few points far from the bulk pull the mean towards them (one point 10^6 x the extent away
            //      among 1024: by 10^3 extents), every query then sits |q~| >> extent from the centre and its band ~ 2^-10 |q~|^2
            //      swallows the whole cloud (1.7 ms instead of 80 us).  Per-dimension MEDIAN and quartiles of 16 rows spread
            //      over the cloud; when a mean lies more than 8 interquartile ranges from the median, every dimension is
            //      centred on its median instead (any centre is correct) and the extent grows by the largest shift (an upper
            //      bound, no second pass).  Skewed but clean data (one-sided features) keep their means.
            if (wv == 0) {
                float shiftmax = 0.0f, iqr2 = 0.0f;
                bool sw = false;
                float medv[DP / 64 > 0 ? DP / 64 : 1];
#pragma unroll
                for (int t = 0; t < (DP + 63) / 64; ++t) {
                    const int d = lane + 64 * t;
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = yb[(size_t)((long long)i * M / 16) * D + (d < D ? d : 0)];
                    float med = v[0], q1 = v[0], q3 = v[0];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        int rk = 0;
#pragma unroll
                        for (int j = 0; j < 16; ++j) rk += (v[j] < v[i] || (v[j] == v[i] && j < i)) ? 1 : 0;
                        med = rk == 8 ? v[i] : med; q1 = rk == 4 ? v[i] : q1; q3 = rk == 12 ? v[i] : q3;
                    }
                    const float shift = d < D ? fabsf(mu[d < D ? d : 0] - med) : 0.0f;
                    sw = sw || (shift > 8.0f * (q3 - q1));
                    shiftmax = fmaxf(shiftmax, shift);
                    if (d < D) iqr2 = __builtin_fmaf(q3 - q1, q3 - q1, iqr2);
                    medv[t] = med;
                }
                if (__ballot(sw) != 0ull) {
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) {
                        shiftmax = fmaxf(shiftmax, __shfl_xor(shiftmax, m, 64));
                        iqr2 = iqr2 + __shfl_xor(iqr2, m, 64);
                    }
#pragma unroll
                    for (int t = 0; t < (DP + 63) / 64; ++t)
                        if (lane + 64 * t < D) mu[lane + 64 * t] = medv[t];
                    if (lane == 0) {
                        *cmax = __builtin_bit_cast(unsigned int, __builtin_bit_cast(float, *cmax) + shiftmax);
                        reinterpret_cast<float *>(cmax)[3] = sqrtf(iqr2);  // the bulk's radius (unscaled): the unit of the absolute error terms below
                    }
                }
            }
            __syncthreads();
        }
        const float cinf = __builtin_bit_cast(float, *cmax);
        if (cinf > 1.0e-30f && cinf < 1.0e30f) {
            int e;
            (void)frexpf(cinf * 1.000001f, &e);  // = m 2^e, m in [0.5, 1)
            // |sc (c - mu)| < 2^10: ten binades above 1 so that a bulk far smaller than the largest |c - mu| (a few far
            // points) still sits in fp16's normal range; queries up to 30 x the cloud's extent stay below 6e4
            sc = ldexpf(1.0f, 10 - e);
        }
        {
            // unit s of the absolute (fp16 subnormal) error terms: |x| <= (x^2 / s + s) / 2 for any s > 0 turns the linear bound
            // 2^-24 sqrt(D) (|q~| + |c~| / 2) into shares of the squared norms.  s = 1 unless the cloud was centred on its medians
            // because of far points: then the bulk may sit far below 1 in scaled units, and with s = 1 the constant term
            // 2^-23 sqrt(D) would dwarf its squared distances (the whole cloud inside every band).
            const float rad = reinterpret_cast<const float *>(cmax)[3];
            funit = rad > 0.0f ? fminf(1.0f, fmaxf(sc * rad, 0x1p-12f)) : 1.0f;
        }
        __syncthreads();
        if (tid == 0 && funit < 1.0f) {
            const float aq = 8.0f * (float)(4 * D + 8) * 0x1p-24f + (SPLIT ? 0x1p-18f : 0x1.01p-10f);
            reinterpret_cast<float *>(cmax)[1] = aq * 1.01f + 0x1p-26f * sqrtf((float)D) / funit + 0x1p-23f;
        }
        // from here on: bits of the largest SCALED squared norm.  A cloud whose extent lets exact Float32 distances overflow
        // (D (61 cinf)^2 >= 3.4e38 for usable queries) is handled like a non-finite one: its +Inf ties are ordered by index in
        // the oracle, which only the brute-force merge reproduces.
        if (tid == 0) *cmax = (anynan || !(cinf < 1.0e15f)) ? 0x7fc00000u : 0u;
    }

    // ---- B operand: the wave's 32 query rows, staged through LDS (coalesced), then -2 q in registers ------------
    float4 a[NT];            // f32 filter: -2 q, this lane's half of the permuted reduction dimension
    kh8 ah[NB16], al[NB16];  // fp16 filter: hi / lo halves of -2 sc q, 8 dimensions per K block and half-wave
    float qn = 0.0f;
    bool qok = true;
    if (early) {
        // K block bb covers dimensions 16 bb + 8 h + [0, 8) in half-wave h: hi halves of -2 sc (q - mu), every piece read from memory
        // (16 bytes of the lane's own row -- rows beyond N read row N - 1 and are never used -- and of the pre-pass header's centre,
        // zero beyond D); all loads in flight together
        const float *qg = xb + (size_t)(qi < N ? qi : N - 1) * D;
        const float *mg = KnnPre::make(pre_ws, M, DP).hdr(b) + 8;
        float4 qv[NB16][2], mv[NB16][2];
#pragma unroll
        for (int bb = 0; bb < NB16; ++bb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int d0 = 16 * bb + 8 * h + 4 * u;
                qv[bb][u] = *reinterpret_cast<const float4 *>(qg + (d0 < D ? d0 : 0));
                mv[bb][u] = *reinterpret_cast<const float4 *>(mg + d0);
            }
        // the first chunk's direct-to-LDS loads (and the norms) go out BEHIND this wave's own loads: the memory counter retires in
        // order, so requested first they made the producer waves wait for the whole chunk before they could touch their header
        // values (their operands were ready 3 k cycles after the consumers', and the block's first barrier with them)
        if (!consumer) {
            if (DK <= 2 && regstage)
                for (int c = 1; c < nchunk; ++c)
                    knn_pre_stage_norms(pre_nup, pre_ndn, c * CH, CH, nall + (size_t)c * CH, nallm ? nallm + (size_t)c * CH : nullptr, wv - kMWaves, lane);
            knn_pre_stage_chunk<DK, false>(pre_img, pre_nup, pre_ndn, 0, CH, sm, nall, nallm, true, wv - kMWaves, lane);
        }
        float amax = 0.0f;
#pragma unroll
        for (int bb = 0; bb < NB16; ++bb) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool in = 16 * bb + 8 * h + 4 * u < D;
                const float v[4] = {qv[bb][u].x, qv[bb][u].y, qv[bb][u].z, qv[bb][u].w}, m4[4] = {mv[bb][u].x, mv[bb][u].y, mv[bb][u].z, mv[bb][u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qs_ = in ? (v[e] - m4[e]) * sc : 0.0f;  // centred like the candidates
                    qn = qn + qs_ * qs_;
                    const float av = -2.0f * qs_;
                    amax = fmaxf(amax, fabsf(av));
                    ah[bb][4 * u + e] = (_Float16)av;
                }
            }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        qok = amax < 6.0e4f;  // inside the fp16 range (false for NaN too)
        qn = qn + __shfl_xor(qn, 32, 64);
    } else if (consumer || DUAL) {  // (DUAL: both waves of a pair stage the same rows -- identical values -- and derive the same operands)
        float *qs = sm + (size_t)cw * 32 * RS;
        const int nrow = wave_active ? ((N - q0) < 32 ? (N - q0) : 32) : 0;
        const float *src = xb + (size_t)q0 * D;
        if (vec4x) {
            const int rq = D / 4;
            for (int e = lane; e < nrow * rq; e += 64) {
                const int row = e / rq, c4 = e - row * rq;
                *reinterpret_cast<float4 *>(qs + (size_t)row * RS + 4 * c4) = reinterpret_cast<const float4 *>(src)[e];
            }
        } else {
            for (int e = lane; e < nrow * D; e += 64) {
                const int row = e / D, d = e - row * D;
                qs[(size_t)row * RS + d] = src[e];
            }
        }
        for (int e = lane; e < 32 * DP; e += 64) {  // zero padding: columns >= D, rows >= nrow
            const int row = e / DP, d = e - row * DP;
            if (row >= nrow || d >= D) qs[(size_t)row * RS + d] = 0.0f;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (F16) {
            // K block bb covers dimensions 16 bb + 8 h + [0, 8) in half-wave h: hi and lo halves of -2 sc q
            float amax = 0.0f;
#pragma unroll
            for (int bb = 0; bb < NB16; ++bb) {
                const float *qr = qs + (size_t)jl * RS + 16 * bb + 8 * h;
                const float4 v0 = *reinterpret_cast<const float4 *>(qr), v1 = *reinterpret_cast<const float4 *>(qr + 4);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float qs_ = (v[e] - mu[16 * bb + 8 * h + e]) * sc;  // centred like the candidates
                    qn = qn + qs_ * qs_;
                    const float av = -2.0f * qs_;
                    amax = fmaxf(amax, fabsf(av));
                    const _Float16 hh_ = (_Float16)av;
                    ah[bb][e] = hh_;
                    if (SPLIT) al[bb][e] = (_Float16)(av - (float)hh_);
                }
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            qok = amax < 6.0e4f;  // inside the fp16 range (false for NaN too)
        } else {
            const float *qr = qs + (size_t)jl * RS + h * (DP / 2);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 v = *reinterpret_cast<const float4 *>(qr + 4 * t);
                qn = qn + v.x * v.x;
                qn = qn + v.y * v.y;
                qn = qn + v.z * v.z;
                qn = qn + v.w * v.w;
                a[t] = float4{-2.0f * v.x, -2.0f * v.y, -2.0f * v.z, -2.0f * v.w};
            }
        }
        qn = qn + __shfl_xor(qn, 32, 64);
    }
    if (!early) __syncthreads();
    KNN_PROBE_MARK(1);

    // ---- chunk schedule: phase A walks the chunks forwards, phase B backwards (its first chunk is resident) ----
    const int nstep = 2 * nchunk;
    constexpr int kMUnits = SPLIT ? kMUnitsSplit : kMUnitsSingle;
    float4 preg[kMUnits][2];  // F16 producers: the chunk after next, loaded one step ahead
    float pmax = 0.0f;        // F16 producers: largest scaled norm seen
    bool pnan = false;
    int stage_ev = 0;                    // F16 producers: staging events done (chunks 0..n-1, n-2..0)
    const int nevents = 2 * nchunk - 1;
    if (F16 && use_pre) {
        if (!consumer && early) {
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of the first chunk (requested at the kernel's start) have landed
            __builtin_amdgcn_wave_barrier();
        } else if (!consumer) {
            // (regstage: the image chunks of the loop below come through registers; the norms of ALL chunks arrive here, once)
            if (DK <= 2 && regstage)
                for (int c = 1; c < nchunk; ++c)
                    knn_pre_stage_norms(pre_nup, pre_ndn, c * CH, CH, nall + (size_t)c * CH, nallm ? nallm + (size_t)c * CH : nullptr, wv - kMWaves, lane);
            knn_pre_stage_chunk<DK>(pre_img, pre_nup, pre_ndn, 0, CH, sm, nall, nallm, true, wv - kMWaves, lane);
        }
    } else if (F16) {
        if (!consumer) {
            knn_f16_load_chunk<DK, kMUnits>(yb, D, 0, M < CH ? M : CH, CH, ptid, preg);
            knn_f16_store_chunk<DK, SPLIT, kMUnits>(sm, CH, M < CH ? M : CH, sc, mu, ptid, preg, nall, nallm, reinterpret_cast<const float *>(cmax + 1), pmax, pnan);
            if (nchunk == 1) {
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, m, 64));
                const bool anyn = __ballot(pnan) != 0;
                if (lane == 0) atomicMax(cmax, anyn ? 0x7fc00000u : __builtin_bit_cast(unsigned int, pmax));
            }
            if (nevents > 1) {
                const int c1 = 1 < nchunk ? 1 : 2 * nchunk - 3;
                knn_f16_load_chunk<DK, kMUnits>(yb, D, c1 * CH, (M - c1 * CH) < CH ? (M - c1 * CH) : CH, CH, ptid, preg);
            }
            stage_ev = 1;
        }
    } else {
        if (D < DP || !vec4y) {  // padding columns must read as zeros; the direct loads never touch them
            for (int e = tid; e < 2 * buf_floats / 4; e += kMThreads)
                reinterpret_cast<float4 *>(sm)[e] = float4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
        }
        if (!consumer) {
            const int cn = M < CH ? M : CH;
            knn_stage_chunk<DK>(yb, D, 0, cn, CH, sm, keep_norms ? nall : sm + (size_t)CH * DP, cmax, true, true, vec4y,
                                wv - kMWaves, lane);
        }
    }
    __syncthreads();
    KNN_PROBE_MARK(2);

    float mn[32];  // group minima: [r] even tiles, [16 + r] odd tiles -> 64 groups per query
#pragma unroll
    for (int r = 0; r < 32; ++r) mn[r] = INFINITY;
    float thr = 0.0f;
    int cnt = 0, totb = 0;  // list words with a survivor; survivors seen by phase B (all of them: the overflow flag tells when words were lost)
    int *mylist = lists + (DUAL ? wv * LCAP : cw * kMLCap) * 64 + lane;  // entry e at mylist[e * 64]

    int cur = 0;  // buffer holding the chunk of this step
    for (int step = 0; step < nstep; ++step) {
        const int phase = step >= nchunk ? 1 : 0;
        const int ci = phase ? nstep - 1 - step : step;
        const int j0 = ci * CH;
        const int cn = (M - j0) < CH ? (M - j0) : CH;
        const int cn_pad = (cn + 63) & ~63;
        const int nstep1 = step + 1;
        const int ci_next = nstep1 >= nchunk ? nstep - 1 - nstep1 : nstep1;
        const bool stage_next = nstep1 < nstep && ci_next != ci;
        // regstage (round 4): the next chunk of the image through REGISTERS -- every thread requests its 16-byte pieces now and writes
        // them to the other buffer after its share of the filter (no VALU either way).  The direct-to-LDS loads moved ~16 bytes per
        // cycle and CU and did not overlap the compute (a 256-row step = its compute, 3.0 k cycles, + its staging, 2.4 k); loads to
        // registers run at the L1's 64 bytes per cycle.
        constexpr bool REGST = PRE && DK <= 2;  // (D > 64: eight pieces per thread -- 32 registers the kernel does not have)
        constexpr int NCR = REGST ? (PPI / 2 > 0 ? PPI / 2 : 1) : 1;  // 16-byte pieces per thread and 256-row chunk
        f32x4v creg[NCR];
        if (REGST && stage_next && regstage) {
            const int j0n = ci_next * CH;
            constexpr int RPBc = PPI >= 16 ? 1 : 16 / PPI;
#pragma unroll
            for (int i = 0; i < NCR; ++i) {
                const int S = tid + i * kMThreads;
                const int row = S / PPI, pos = S & (PPI - 1);
                const int c = (pos - row / RPBc) & (PPI - 1);
                if (S < CH * PPI) creg[i] = *reinterpret_cast<const f32x4v *>(pre_img + ((size_t)(j0n + row) * PPI + c) * 8);
            }
        } else if (DUAL && !consumer && stage_next) {  // the next chunk's direct loads first: they land while this wave computes
            const int j0n = ci_next * CH;
            knn_pre_stage_chunk<DK, false>(pre_img, pre_nup, pre_ndn, j0n, CH, sm + (size_t)(1 - cur) * buf_floats, nall + (size_t)ci_next * CH,
                                           nallm ? nallm + (size_t)ci_next * CH : nullptr, nstep1 < nchunk, wv - kMWaves, lane);
        }
        if (consumer || DUAL) {
            if (wave_active) {
                const float *cand = sm + (size_t)cur * buf_floats;
                const float *cnorm = keep_norms ? (phase && nallm ? nallm : nall) + (size_t)ci * CH : cand + (size_t)CH * DP;
                const int npair = cn_pad / 64;
                const int tile0 = j0 / 32;
                int pr_first = 0;
                // phase B of the fp16 filters accumulates on n_c - thr: the sign of the result is the test.  (The Float32 GEMM keeps the
                // compare: its staging leaves the rows beyond the cloud's end unwritten -- norm +inf, stale pieces -- and inf + NaN has
                // no usable sign; the fp16 images are zero there.)
                const float tsub = (F16 && phase) ? thr : 0.0f;
                if (F16 && !SPLIT && PRE) {  // (without the pre-pass the producers' staging registers leave no room: 68 spills)
                    // single-piece fp16 filter: TWO pairs of tiles per iteration -- the second pair's operand fetches and MFMAs are
                    // issued before the first pair's results are folded, so the fold (VALU) of one overlaps the matrix work of the
                    // other and one round of LDS latency serves four tiles (a lone consumer wave per SIMD hides nothing otherwise).
                    // One instantiation per phase (round 4): phase A's accumulators ARE the norm loads' destinations (no VALU) and two
                    // tiles fold per v_min3 straight from the MFMA registers; phase B starts them at n_c - thr (one v_sub each).
                    constexpr int RPB2 = PPI >= 16 ? 1 : 16 / PPI;
                    auto run4 = [&](auto phc) {
                        constexpr bool PHB = decltype(phc)::value;
                        for (; pr_first + 1 < npair; pr_first += 2) {
                            if (DUAL && ((pr_first >> 1) & 1) != half) continue;  // the pair's waves take alternate double pairs
                            f32x16v accs[4];
                            kh8 ops[4][NB16];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {  // q = 2 * (pair) + (tile of the pair)
                                const int rbase = (pr_first + (q >> 1)) * 64 + 32 * (q & 1);
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const float4 n0 = *reinterpret_cast<const float4 *>(cnorm + rbase + 8 * g + 4 * h);
                                    if (PHB) {
                                        accs[q][4 * g] = n0.x - thr; accs[q][4 * g + 1] = n0.y - thr; accs[q][4 * g + 2] = n0.z - thr; accs[q][4 * g + 3] = n0.w - thr;
                                    } else {
                                        accs[q][4 * g] = n0.x; accs[q][4 * g + 1] = n0.y; accs[q][4 * g + 2] = n0.z; accs[q][4 * g + 3] = n0.w;
                                    }
                                }
                                const float *cq = cand + (size_t)(rbase + jl) * RSI;
#pragma unroll
                                for (int bb = 0; bb < NB16; ++bb)
                                    ops[q][bb] = *reinterpret_cast<const kh8 *>(cq + ((2 * bb + h + jl / RPB2) & (PPI - 1)) * 4);
                            }
#pragma unroll
                            for (int bb = 0; bb < NB16; ++bb) {  // four independent accumulators in turn: no MFMA waits for its predecessor
#pragma unroll
                                for (int q = 0; q < 4; ++q) accs[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ops[q][bb], ah[bb], accs[q], 0, 0, 0);
                            }
                            // (round 4) the fold / mask work runs at raised wave priority: the pair's other wave is then the one whose MFMAs are
                            // in the pipe while this one issues VALU -- same-box A/B 51.95 -> 50.7 us (the reverse, priority on the MFMA block, costs
                            // 12 us: the issuing wave hogs the slots its partner's fold needs; static priorities by wave role: no effect)
                            __builtin_amdgcn_s_setprio(1);
                            if (!PHB) {
                                // (any partition of the tiles into the 32 groups of a lane will do; the two accumulators issued last are
                                //  read 20+ issue slots after their MFMAs: knn_f16_d3_kernel's order)
                                KNN_MFMA_SETTLE4(accs[0], accs[1], accs[2], accs[3]);
#pragma unroll
                                for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[r]) : "v"(accs[0][r]), "v"(accs[1][r]));
#pragma unroll
                                for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[16 + r]) : "v"(accs[2][r]), "v"(accs[3][r]));
                            } else {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    unsigned int m = 0;
#pragma unroll
                                    for (int i = 0; i < 16; ++i) {  // the sign of F - thr is the test: one v_alignbit per row shifts it in (row r at bit r)
                                        const float av = accs[q][15 - i];
                                        m = __builtin_amdgcn_alignbit(m, __builtin_bit_cast(unsigned int, av), 31);
                                    }
                                    const int pp = cnt < LCAP - 1 ? cnt : LCAP - 1;
                                    mylist[pp * 64] = (int)((unsigned int)(tile0 + pr_first * 2 + q) << 16 | m);
                                    cnt += m != 0 ? 1 : 0;
                                    totb += __builtin_popcount(m);
                                }
                            }
                            __builtin_amdgcn_s_setprio(0);
                        }
                    };
                    if (phase == 0) run4(std::false_type{});
                    else run4(std::true_type{});
                }
                for (int pr = (DUAL && half) ? npair : pr_first; pr < npair; ++pr) {  // (DUAL: a last lone pair goes to the first wave)
                    // rows pr*64 + jl and + 32 share (row mod PPR) = jl mod PPR: one rotated offset per fetch
                    const float *c0 = cand + (size_t)(pr * 64 + jl) * RSI, *c1 = c0 + (size_t)32 * RSI;
                    // accumulators start at the candidate norms: register r of half h is row (r&3) + 8(r>>2) + 4h
                    f32x16v acc0, acc1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 n0 = *reinterpret_cast<const float4 *>(cnorm + pr * 64 + 8 * g + 4 * h);
                        const float4 n1 = *reinterpret_cast<const float4 *>(cnorm + pr * 64 + 32 + 8 * g + 4 * h);
                        acc0[4 * g] = n0.x - tsub; acc0[4 * g + 1] = n0.y - tsub; acc0[4 * g + 2] = n0.z - tsub; acc0[4 * g + 3] = n0.w - tsub;
                        acc1[4 * g] = n1.x - tsub; acc1[4 * g + 1] = n1.y - tsub; acc1[4 * g + 2] = n1.z - tsub; acc1[4 * g + 3] = n1.w - tsub;
                    }
                    if (F16 && SPLIT) {
                        // A = candidate pieces (rows), B = query pieces (columns); hi*hi + lo*hi + hi*lo
                        kh8 h0[NB16], l0[NB16], h1[NB16], l1[NB16];
#pragma unroll
                        for (int bb = 0; bb < NB16; ++bb) {
                            const int ph = ((2 * bb + h + jl) & (PPR - 1)) * 4, pl = ((PPR / 2 + 2 * bb + h + jl) & (PPR - 1)) * 4;
                            h0[bb] = *reinterpret_cast<const kh8 *>(c0 + ph);
                            l0[bb] = *reinterpret_cast<const kh8 *>(c0 + pl);
                            h1[bb] = *reinterpret_cast<const kh8 *>(c1 + ph);
                            l1[bb] = *reinterpret_cast<const kh8 *>(c1 + pl);
                        }
#pragma unroll
                        for (int bb = 0; bb < NB16; ++bb) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0[bb], ah[bb], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1[bb], ah[bb], acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l0[bb], ah[bb], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l1[bb], ah[bb], acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0[bb], al[bb], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1[bb], al[bb], acc1, 0, 0, 0);
                        }
                    } else if (F16) {
                        // single-piece filter: one MFMA per K block and tile on the rounded (hi) halves
                        constexpr int RPB = PPI >= 16 ? 1 : 16 / PPI;
                        kh8 h0[NB16], h1[NB16];
#pragma unroll
                        for (int bb = 0; bb < NB16; ++bb) {
                            const int ph = ((2 * bb + h + jl / RPB) & (PPI - 1)) * 4;
                            h0[bb] = *reinterpret_cast<const kh8 *>(c0 + ph);
                            h1[bb] = *reinterpret_cast<const kh8 *>(c1 + ph);
                        }
#pragma unroll
                        for (int bb = 0; bb < NB16; ++bb) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0[bb], ah[bb], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1[bb], ah[bb], acc1, 0, 0, 0);
                        }
                    } else {
                        float4 b0[NT], b1[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int po = ((h * NT + t + jl) & (PPR - 1)) * 4;
                            b0[t] = *reinterpret_cast<const float4 *>(c0 + po);
                            b1[t] = *reinterpret_cast<const float4 *>(c1 + po);
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) {  // A = candidates (rows), B = queries (columns)
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t].x, a[t].x, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t].x, a[t].x, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t].y, a[t].y, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t].y, a[t].y, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t].z, a[t].z, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t].z, a[t].z, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t].w, a[t].w, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t].w, a[t].w, acc1, 0, 0, 0);
                        }
                    }
                    if (phase == 0) {
                        KNN_MFMA_SETTLE2(acc0, acc1);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mn[r] = vmin_acc(mn[r], acc0[r]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mn[16 + r] = vmin_acc(mn[16 + r], acc1[r]);
                    } else {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            // one word per tile: (tile index << 16) | mask of the rows with F <= thr; stored at the
                            // list head unconditionally, the head advances when the mask is not empty
                            unsigned int m = 0;
#pragma unroll
                            for (int i = 0; i < 16; ++i) {  // (ascending i: a descending unrolled loop over the vector's elements read element 0 every time)
                                const float av = tt ? acc1[15 - i] : acc0[15 - i];
                                if (F16) m = __builtin_amdgcn_alignbit(m, __builtin_bit_cast(unsigned int, av), 31);
                                else m |= (av <= thr) ? (1u << (15 - i)) : 0u;
                            }
                            const int pp = cnt < LCAP - 1 ? cnt : LCAP - 1;
                            mylist[pp * 64] = (int)((unsigned int)(tile0 + pr * 2 + tt) << 16 | m);
                            cnt += m != 0 ? 1 : 0;
                            totb += __builtin_popcount(m);
                        }
                    }
                }
            }
        } else if (stage_next) {
            const int j0n = ci_next * CH;
            const int cnn = (M - j0n) < CH ? (M - j0n) : CH;
            float *img = sm + (size_t)(1 - cur) * buf_floats;
            if (F16 && use_pre) {  // (not reached when DUAL: kept for a PRE build without it)
                const bool phase_a = nstep1 < nchunk;  // (the norms of all chunks stay in LDS: phase B brings the image only)
                knn_pre_stage_chunk<DK>(pre_img, pre_nup, pre_ndn, j0n, CH, img, nall + (size_t)ci_next * CH,
                                        nallm ? nallm + (size_t)ci_next * CH : nullptr, phase_a, wv - kMWaves, lane);
            } else if (F16) {
                // the registers hold chunk ci_next (loaded one step ago); then fetch the chunk after it
                knn_f16_store_chunk<DK, SPLIT, kMUnits>(img, CH, cnn, sc, mu, ptid, preg, stage_ev < nchunk ? nall + (size_t)stage_ev * CH : nullptr,
                                        stage_ev < nchunk && nallm ? nallm + (size_t)stage_ev * CH : nullptr,
                                        reinterpret_cast<const float *>(cmax + 1), pmax, pnan);
                if (stage_ev == nchunk - 1) {  // last phase-A chunk: publish this wave's maximum norm
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, m, 64));
                    const bool anyn = __ballot(pnan) != 0;
                    if (lane == 0) atomicMax(cmax, anyn ? 0x7fc00000u : __builtin_bit_cast(unsigned int, pmax));
                }
                ++stage_ev;
                if (stage_ev < nevents) {
                    const int cnx = stage_ev < nchunk ? stage_ev : 2 * nchunk - 2 - stage_ev;
                    knn_f16_load_chunk<DK, kMUnits>(yb, D, cnx * CH, (M - cnx * CH) < CH ? (M - cnx * CH) : CH, CH, ptid, preg);
                }
            } else {
                const bool phase_a = nstep1 < nchunk;
                knn_stage_chunk<DK>(yb, D, j0n, cnn, CH, img, keep_norms ? nall + (size_t)ci_next * CH : img + (size_t)CH * DP,
                                    cmax, phase_a, phase_a || !keep_norms, vec4y, wv - kMWaves, lane);
            }
        }
        if (REGST && stage_next && regstage) {
            float *nimg = sm + (size_t)(1 - cur) * buf_floats;
#pragma unroll
            for (int i = 0; i < NCR; ++i) {
                const int S = tid + i * kMThreads;
                if (S < CH * PPI) *reinterpret_cast<f32x4v *>(nimg + (size_t)S * 4) = creg[i];
            }
        } else if (DUAL && !consumer && stage_next) {
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of the next chunk have landed
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        KNN_PROBE_MARK(3 + step);
        if (stage_next) cur = 1 - cur;
        if (step == nchunk - 1 && (consumer || DUAL)) {
            // ---- tau: kk-th smallest of the 64 group minima of every query (32 in this lane, 32 in its partner) ----
            const float c2 = __builtin_bit_cast(float, *cmax);
            float tau;
            if (DUAL && kk <= 24 && M >= 128) {
                // 128 group minima per query in the layout of the D = 3 kernel (32 per lane x two half-lanes x the pair's two waves):
                // its reduced selection (round 4; the sort of all 32 + two 32-value merges below were 5 us of this kernel)
                tau = knn_tau_8of16<kMWaves>(mn, reinterpret_cast<float *>(lists), wv, jl, h, kk);
                __syncthreads();  // the exchange space becomes the lane lists
            } else {
            k3_sort_regs<32>(mn);
            float oth[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) oth[r] = __shfl_xor(mn[31 - r], 32, 64);
#pragma unroll
            for (int r = 0; r < 32; ++r)  // half 0 keeps the 32 smallest of the 64 (a bitonic sequence)
                mn[r] = h ? vmax_f32(mn[r], oth[r]) : vmin_f32(mn[r], oth[r]);
#pragma unroll
            for (int j = 16; j > 0; j >>= 1) {  // one bitonic merge sorts it ascending
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int l = i ^ j;
                    if (l > i) {
                        const float lo = vmin_f32(mn[i], mn[l]), hi = vmax_f32(mn[i], mn[l]);
                        mn[i] = lo;
                        mn[l] = hi;
                    }
                }
            }
            if (DUAL) {
                // the other wave of the pair holds the minima of the other tiles: the 32 smallest of the 128 through LDS (the
                // lane lists are not in use yet), one more bitonic merge
                float *xch = reinterpret_cast<float *>(lists);  // [2 kMWaves][32][33]
                if (h == 0) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) xch[(wv * 32 + jl) * 33 + r] = mn[r];
                }
                __syncthreads();
                {
                    const float *po = xch + (((wv + kMWaves) % (2 * kMWaves)) * 32 + jl) * 33;
#pragma unroll
                    for (int r = 0; r < 32; ++r) mn[r] = vmin_f32(mn[r], po[31 - r]);
#pragma unroll
                    for (int j = 16; j > 0; j >>= 1) {
#pragma unroll
                        for (int i = 0; i < 32; ++i) {
                            const int l = i ^ j;
                            if (l > i) {
                                const float lo = vmin_f32(mn[i], mn[l]), hi = vmax_f32(mn[i], mn[l]);
                                mn[i] = lo;
                                mn[l] = hi;
                            }
                        }
                    }
                }
                __syncthreads();  // the exchange space becomes the lane lists
            }
            float val = mn[0];
#pragma unroll
            for (int r = 1; r < 32; ++r) val = (kk - 1) == r ? mn[r] : val;
            tau = __shfl(val, jl, 64);  // kk <= 32: always among the 32 smallest (half 0)
            }
            float eps;
            if (F16) {
                // scaled units (c~ = sc c, |c~| < 1; qn = |sc q|^2): split representation 3 2^-22 |a~||c~|, fp32
                // accumulation of the 3D exact products (3D+1) u, the oracle's own (D+2) u, fp16 underflow floor
                // single piece: the rounded operands differ by 2^-11 relative each, sum |c~_d a_d| <= 2 |c~||q~| <= qn + c2
                // |F^ - (sc^2 d_oracle - qn)| <= A (n_c + qn) + floor for candidate c with scaled norm n_c (A = acoef_q).
                // two_norms: the candidate's share A n_c is already inside the norms (upwards in phase A, downwards
                // in phase B), the query keeps B_q = A qn + floor_q: a far candidate no longer widens everybody's
                // band.  Otherwise n_c <= c2 for all of them.
                const float acoef_q = 8.0f * (float)(4 * D + 8) * 0x1p-24f + (SPLIT ? 0x1p-18f : 0x1.01p-10f);
                const float floor_q = 0x1p-24f * sqrtf((float)D) * (qn / funit + 2.0f * funit);
                eps = two_norms ? acoef_q * qn + floor_q : acoef_q * (qn + c2) + floor_q + 0x1p-26f * sqrtf((float)D) * c2 / funit;
                eps = (qok && c2 == c2) ? eps : INFINITY;  // c2 is NaN for a non-finite / overflow-prone cloud (scale pass)
            } else {
                eps = (8.0f * (float)(D + 4) * 0x1p-24f) * (qn + c2);
                eps = qn + c2 < 1.0e38f ? eps : INFINITY;  // (|q| + |c|)^2 <= 2 (qn + c2): no exact distance overflows
            }
            thr = tau + 2.0f * eps;  // NaN / inf => slow path below
            // (round 4) phase B starts the accumulators at n_c - thr instead of n_c and keeps the SIGN of the result (one v_alignbit per
            // row where the compare cost v_cmp + v_cndmask + v_or and two wait states).  The accumulation now carries thr through its
            // D + 1 roundings: against the compare form the result moves by at most (D + 1) u (2 n_c + |thr| + 2 sum |products|)
            // (65 u (4 n_c + 2 qn + |thr|) at D = 64), u = 2^-24.  The candidate's and the query's own shares sit inside the budget the
            // filter already grants them (8 (4 D + 8) u = 2112 u each at D = 64, of which the compare form uses ~130 u); the threshold's share,
            // (K + 1) u |thr| with K = D products (3 D for the split filter), is added here four times over -- at D = 64 2^-16 |thr|,
            // 1/64 of the band of a candidate at the boundary (its norm is of the threshold's size) -- and it makes the test strict
            // (a candidate at the threshold gives a negative result, never +-0).
            thr = thr + (4.0f * (float)((SPLIT ? 3 : 1) * D + 1) * 0x1p-24f) * (fabsf(thr) + qn);
        }
    }
    KNN_PROBE_MARK(20);

    // ---- exact phase -----------------------------------------------------------------------------------------------
    // (1) consumers: list lengths, survivors per query, fast-path flag
    const int need = kk < M ? kk : M;
    const int part = (consumer ? 0 : 2) + h;  // the query's four lanes: two half-waves x the pair's two waves
    if (DUAL) {
        // every lane publishes the per-stage counts of its own list (bytes of a 64-bit word; srl == 0: the total in byte 0)
        // and whether the list overflowed (top bit: a stage holds < 128 survivors of the <= 64 that matter)
        const int nv = cnt < LCAP - 1 ? cnt : LCAP - 1;
        const int tsh = srl > 0 ? srl - 5 : 31;
        unsigned long long pk = 0;
        int totx = 0;
        if (srl > 0) {
            for (int e = 0; e < nv; ++e) {
                const unsigned int w = (unsigned int)mylist[e * 64];
                const int pc = __builtin_popcount(w & 0xffffu);
                totx += pc;
                pk += (unsigned long long)pc << (((w >> 16) >> tsh) * 8);
            }
        } else {  // no row stages (column slices, gather): the total phase B counted -- no walk over the list (a chain of LDS round trips)
            totx = totb;
            pk = (unsigned long long)(unsigned int)totb;
        }
        // (the bytes are only meaningful while none can carry: a part with more than 63 survivors -- the query is not a fast one
        //  then -- publishes its plain total behind a marker bit instead)
        qpk[(cw * 32 + jl) * 4 + part] = (totx <= 63 ? pk : (1ull << 62) | (unsigned long long)totx) | (cnt > LCAP - 1 ? 1ull << 63 : 0ull);
        lcnt2[wv * 64 + lane] = (unsigned short)nv;
    } else if (consumer) {
        const int nv = cnt < kMLCap - 1 ? cnt : kMLCap - 1;
        int tot = 0;
        for (int e = 0; e < nv; ++e) tot += __builtin_popcount((unsigned int)mylist[e * 64] & 0xffffu);
        const int totp = __shfl_xor(tot, 32, 64);
        const int cntp = __shfl_xor(cnt, 32, 64);
        const int n = tot + totp;
        const bool lists_ok = wave_active && qi < N && thr < INFINITY && cnt <= kMLCap - 1 && cntp <= kMLCap - 1 && n >= need;
        const bool fast = lists_ok && n <= kMKeyCap;
        const bool medium = lists_ok && n > kMKeyCap && n <= kMMedCap;  // too many for the key arrays, lists intact
        lcnt[cw * 64 + lane] = (h ? totp : 0) | (nv << 16);  // start offset of this lane's ids | entries
        if (h == 0) { qn_n[cw * 32 + jl] = n; qflag[cw * 32 + jl] = fast ? 1 : (medium ? 2 : 0); qbelow[cw * 32 + jl] = 0; }
    }
    __syncthreads();  // the chunk buffers are free from here on: they hold the keys
    KNN_PROBE_MARK(21);
    unsigned int *qd = reinterpret_cast<unsigned int *>(sm) + (size_t)(cw * 32 + jl) * kMKeyStride;                       // distance bits
    int *qj = reinterpret_cast<int *>(sm) + (size_t)kMWaves * 32 * kMKeyStride + (size_t)(cw * 32 + jl) * kMKeyStride;     // indices
    int n = qn_n[cw * 32 + jl];
    bool fast = qflag[cw * 32 + jl] == 1;
    bool handled = qflag[cw * 32 + jl] == 2;  // answered by the medium path
    unsigned long long dpk[4] = {0ull, 0ull, 0ull, 0ull};  // DUAL: the four parts' packed counts
    if (DUAL) {
        bool ovf = false, big = false;
        int tot4 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long v = qpk[(cw * 32 + jl) * 4 + q];
            ovf |= (v >> 63) != 0;
            const bool bigp = ((v >> 62) & 1ull) != 0;  // more than 63 survivors in this part alone: its plain total
            big |= bigp;
            dpk[q] = bigp ? 0ull : v & ~(3ull << 62);
            tot4 += bigp ? (int)(unsigned int)v : (int)((dpk[q] * 0x0101010101010101ull) >> 56);  // sum of the bytes
        }
        n = tot4;
        const bool lists_ok = wave_active && qi < N && thr < INFINITY && !ovf && n >= need;
        fast = lists_ok && !big && n <= kMKeyCap;
        handled = lists_ok && !fast && n <= kMMedCap;
        if (part == 0) { qn_n[cw * 32 + jl] = n; qflag[cw * 32 + jl] = fast ? 1 : (handled ? 2 : 0); qbelow[cw * 32 + jl] = 0; }
    }
    // staged exact phase (srl > 0): the thread's share of the first stage of candidate rows is requested here, so that
    // it arrives while the lists are decoded
    f32x4v sreg[8];
    // (D > 64: the rows are staged and evaluated in two column halves of <= 64 dimensions -- see the exact phase below)
    const int DS = DP > 64 && D > 64 ? 64 : D;                     // staged width of a row (half), floats
    const int PR = DS >> 2, RPI = srl > 0 ? kMThreads / PR : 0;  // rows per sweep of the block (PR divides the block size)
    const int srow = srl > 0 ? tid / PR : 0, scol = (tid - srow * PR) * 4;
    if (srl > 0) knn_stage_fetch(yb, D, M, 0, srow, RPI, scol, sreg);
    if (csl > 0) knn_stage_fetch(yb, D, M, 0, tid >> 2, kMThreads / 4, 4 * (tid & 3), sreg);  // column slices: rows (tid >> 2) + 128 i, piece tid & 3
    if (wave_active) {
        // ---- medium path (tight clusters, many duplicates: more candidates inside the band than the key arrays hold):
        //      the wave decodes the query's two lane lists into an id list and selects exactly among those ids,
        //      instead of scanning all M candidates in the fallback.  The lists are intact until the barrier after (2).
        const unsigned long long mmask = __ballot(handled);
        int *ids = med + wv * (kMMedCap + 128);
        for (unsigned int bm = (unsigned int)mmask | (unsigned int)(mmask >> 32); bm; bm &= bm - 1) {
            const int j = __builtin_ctz(bm);
            if ((j & 1) != (consumer ? 0 : 1)) continue;  // the pair's two waves share the queries
            int total = 0;
            for (int h2 = 0; h2 < (DUAL ? 4 : 2); ++h2) {  // (DUAL: h2 = 2 * (wave of the pair) + half-wave)
                const int src = (h2 & 1) * 32 + j;
                const int lw = DUAL ? cw + kMWaves * (h2 >> 1) : cw;  // the wave that holds the list
                const int nv2 = DUAL ? lcnt2[lw * 64 + src] : lcnt[cw * 64 + src] >> 16;
                const unsigned int w = lane < nv2 ? (unsigned int)lists[((DUAL ? lw * LCAP : cw * kMLCap) + lane) * 64 + src] : 0u;  // nv2 < 64
                const int pc = __builtin_popcount(w & 0xffffu);
                int incl = pc;
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) {
                    const int t = __shfl_up(incl, m, 64);
                    if (lane >= m) incl += t;
                }
                int pos = total + incl - pc;
                unsigned int m16 = w & 0xffffu;
                const int rowbase = (int)(w >> 16) * 32 + 4 * (h2 & 1);
                while (m16) {
                    const int r = __builtin_ctz(m16);
                    m16 &= m16 - 1;
                    ids[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                }
                total += __shfl(incl, 63, 64);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float bd;
            int bj;
            knn_exact_bruteforce(xb + (size_t)(q0 + j) * D, yb, total, D, kk, lane, reinterpret_cast<float *>(ids + kMMedCap),
                                 ids + kMMedCap + 64, bd, bj, ids);
            const int r = lane - drop;
            if (r >= 0 && r < k) {
                idx[((size_t)b * N + q0 + j) * k + r] = bj;
                if (dist) dist[((size_t)b * N + q0 + j) * k + r] = bd;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // (2) consumers decode their mask words into candidate ids (integer work only).  srl > 0 (staged exact phase): the
    //     ids of a query are grouped by row stage (2^srl candidate rows, at most 8 stages): per-stage counts of the two
    //     half-wave lists in the bytes of a 64-bit word (n <= 60 < 256), prefix sums by one multiplication
    if (DUAL) {
        if (fast) {  // every one of the query's four lanes decodes its own list behind the lists of the parts before it
            const int nv = lcnt2[wv * 64 + lane];
            const unsigned long long incl = (dpk[0] + dpk[1] + dpk[2] + dpk[3]) * 0x0101010101010101ull;  // byte s: survivors in stages 0..s
            unsigned long long before = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) before += q < part ? dpk[q] : 0ull;
            unsigned long long startpk = (srl > 0 ? incl << 8 : 0ull) + before;  // byte s: where this lane's ids of stage s go
            const int tsh = srl > 0 ? srl - 5 : 31;
            for (int e0 = 0; e0 < nv; e0 += 4) {  // four list words in flight
                unsigned int w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = (unsigned int)mylist[(e0 + u < LCAP ? e0 + u : LCAP - 1) * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned int m = e0 + u < nv ? (w[u] & 0xffffu) : 0u;
                    const int rowbase = (int)(w[u] >> 16) * 32 + 4 * h;
                    const int sh = srl > 0 ? (int)((w[u] >> 16) >> tsh) * 8 : 0;
                    int pos = (int)(startpk >> sh) & 0xff;
                    startpk += (unsigned long long)__builtin_popcount(m) << sh;
                    while (m) {
                        const int r = __builtin_ctz(m);
                        m &= m - 1;
                        qj[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                    }
                }
            }
            if (part == 3) { qd[n] = 0xffffffffu; qd[n + 1] = 0xffffffffu; qd[n + 2] = 0xffffffffu;
                             qj[n] = 0x7fffffff; qj[n + 1] = 0x7fffffff; qj[n + 2] = 0x7fffffff; }  // sentinels for the b128 sweeps
        }
    } else if (consumer && fast) {
        const int meta = lcnt[cw * 64 + lane];
        const int nv = meta >> 16;
        if (srl > 0) {
            const int tsh = srl - 5;  // tile -> stage
            unsigned long long pk = 0;
            for (int e = 0; e < nv; ++e) {
                const unsigned int w = (unsigned int)mylist[e * 64];
                pk += (unsigned long long)__builtin_popcount(w & 0xffffu) << (((w >> 16) >> tsh) * 8);
            }
            const unsigned long long pko = ((unsigned long long)(unsigned int)__shfl_xor((int)(pk >> 32), 32, 64) << 32) |
                                           (unsigned int)__shfl_xor((int)pk, 32, 64);
            const unsigned long long incl = (pk + pko) * 0x0101010101010101ull;  // byte s: survivors in stages 0..s
            unsigned long long startpk = (incl << 8) + (h ? pko : 0ull);         // byte s: where this lane's ids of stage s go
            if (h == 0) qstpk[cw * 32 + jl] = incl;
            for (int e0 = 0; e0 < nv; e0 += 4) {  // four list words in flight
                unsigned int w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = (unsigned int)mylist[(e0 + u < kMLCap ? e0 + u : kMLCap - 1) * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned int m = e0 + u < nv ? (w[u] & 0xffffu) : 0u;
                    const int rowbase = (int)(w[u] >> 16) * 32 + 4 * h;
                    const int sh = (int)((w[u] >> 16) >> tsh) * 8;
                    int pos = (int)(startpk >> sh) & 0xff;
                    startpk += (unsigned long long)__builtin_popcount(m) << sh;
                    while (m) {
                        const int r = __builtin_ctz(m);
                        m &= m - 1;
                        qj[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                    }
                }
            }
        } else {
            int pos = meta & 0xffff;
            for (int e0 = 0; e0 < nv; e0 += 4) {  // four list words in flight
                unsigned int w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = (unsigned int)mylist[(e0 + u < kMLCap ? e0 + u : kMLCap - 1) * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned int m = e0 + u < nv ? (w[u] & 0xffffu) : 0u;
                    const int rowbase = (int)(w[u] >> 16) * 32 + 4 * h;
                    while (m) {
                        const int r = __builtin_ctz(m);
                        m &= m - 1;
                        qj[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                    }
                }
            }
        }
        if (h) { qd[n] = 0xffffffffu; qd[n + 1] = 0xffffffffu; qd[n + 2] = 0xffffffffu;
                 qj[n] = 0x7fffffff; qj[n + 1] = 0x7fffffff; qj[n + 2] = 0x7fffffff; }  // sentinels for the b128 sweeps
    }
    __syncthreads();  // ids visible to the producer partners; the lane lists are dead: their space holds the slots
    KNN_PROBE_MARK(22);
    // (3) the query's survivors are split over its four lanes (two halves x consumer / producer wave)
    const int per = (n + 3) >> 2;
    const int mystart = part * per < n ? part * per : n;
    const int mycount = (mystart + per <= n ? per : n - mystart);
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(lists) + (size_t)(cw * 32 + jl) * 33;  // [..][32 + 1 pad]
    if (csl > 0) {
        // ---- column slices (round 4): ALL candidate rows pass through LDS, 16 dimensions at a time (csl = D / 16 slices), as four
        //      planes of 16-byte pieces ([piece c][row], plane stride 16 Mp + 32 bytes: the coalesced staging writes and the reads of
        //      consecutive rows are conflict-free).  In every slice the four lanes of a query share ALL its survivors evenly -- the
        //      row stages below share the survivors of one stage at a time: ~1.7 per lane against a fullest lane of 3-4 (42 % of the
        //      lane slots held a pair); whole queries differ far less (27 +- 5 survivors) -- and a pair's running sum waits in its
        //      distance slot between slices: the oracle's order of additions.  The query's slice is 16 floats in registers (the row
        //      stages held the whole row: 64), the next slice's pieces of rows and query are in flight while this one is summed.
        const int MPc = (M + kMThreads / 4 - 1) / (kMThreads / 4) * (kMThreads / 4);
        const int PS = MPc * 4 + 8;  // plane stride, floats
        float *stg = reinterpret_cast<float *>(lists);
        const bool act = wave_active && fast;
        const int crow = tid >> 2, cc = tid & 3;
        const float *qrow = xb + (size_t)(act ? qi : 0) * D;
        f32x4v qs[4], qnx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) qs[t] = *reinterpret_cast<const f32x4v *>(qrow + 4 * t);
        // this lane's pairs stay in registers across the slices: the candidate's plane offset and the running sum (n <= 64 survivors per
        // query: at most 16 per lane); four pairs are in flight -- their 16 pieces are requested together, their four chains of additions
        // interleave (a slice of ONE pair is 16 dependent additions: two pairs in flight left the phase latency bound, 2.5 us per slice)
        const int cnt_l = act ? mycount : 0;
        int jo[16];
        float acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            jo[u] = u < cnt_l ? qj[mystart + u] * 4 : 0;
            acc[u] = 0.0f;
        }
        for (int s = 0; s < csl; ++s) {
            if (s) __syncthreads();  // every lane is done with the previous slice
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = crow + i * (kMThreads / 4);
                if (row < MPc) *reinterpret_cast<f32x4v *>(stg + (size_t)cc * PS + (size_t)row * 4) = sreg[i];
            }
            __syncthreads();
            if (s < 3) KNN_PROBE_MARK(26 + 2 * s);
            if (s + 1 < csl) {
                knn_stage_fetch(yb, D, M, 0, crow, kMThreads / 4, 16 * (s + 1) + 4 * cc, sreg);
#pragma unroll
                for (int t = 0; t < 4; ++t) qnx[t] = *reinterpret_cast<const f32x4v *>(qrow + 16 * (s + 1) + 4 * t);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (__ballot(4 * g < cnt_l) != 0ull) {  // (wave-uniform: some lane still has a pair in this group)
                    f32x4v c[4][4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < 4; ++t) c[e][t] = *reinterpret_cast<const f32x4v *>(stg + jo[4 * g + e] + t * PS);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x4v m[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x4v d = qs[t] - c[e][t];
                            m[e] = d * d;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = acc[4 * g + e] + m[e].x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = acc[4 * g + e] + m[e].y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = acc[4 * g + e] + m[e].z;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = acc[4 * g + e] + m[e].w;
                    }
                }
            }
            if (s < 3) KNN_PROBE_MARK(27 + 2 * s);
#pragma unroll
            for (int t = 0; t < 4; ++t) qs[t] = qnx[t];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < cnt_l) qd[mystart + u] = __builtin_bit_cast(unsigned int, acc[u]);
    } else if (srl > 0) {
        // staged: the candidate rows come through LDS one stage (2^srl rows) at a time, loaded coalesced once per block
        // (every row exactly once: M * 4D bytes from L2 instead of 4D per survivor), rows 16 bytes apart in the banks.
        // Per stage the four lanes of a query split its survivors of that stage; the query row sits in registers.
        // The oracle's distance of every id: same operations in the same order as the gather below.
        const int SRW = 1 << srl, RSX = DS + 4;
        float *stg = reinterpret_cast<float *>(lists);
        const int nstage = (M + SRW - 1) >> srl;
        const bool act = wave_active && fast;
        const unsigned long long incl = !act ? 0ull : (DUAL ? (dpk[0] + dpk[1] + dpk[2] + dpk[3]) * 0x0101010101010101ull : qstpk[cw * 32 + jl]);
        if constexpr (DP <= 64) {
            f32x4v qreg[DP / 4];
            {
                const float *qrow = xb + (size_t)(act ? qi : 0) * D;
#pragma unroll
                for (int t = 0; t < DP / 4; ++t)
                    qreg[t] = 4 * t < D ? *reinterpret_cast<const f32x4v *>(qrow + 4 * t) : f32x4v{0.f, 0.f, 0.f, 0.f};
            }
            for (int s = 0; s < nstage; ++s) {
                if (s) __syncthreads();  // every lane is done with the previous stage
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = srow + i * RPI;
                    if (row < SRW) *reinterpret_cast<f32x4v *>(stg + (size_t)row * RSX + scol) = sreg[i];
                }
                __syncthreads();
                if (s < 3) KNN_PROBE_MARK(26 + 2 * s);
                if (s + 1 < nstage) knn_stage_fetch(yb, D, M, (s + 1) << srl, srow, RPI, scol, sreg);  // in flight while this stage is evaluated
                if (act) {
                    const int start = s ? (int)(incl >> (8 * (s - 1))) & 0xff : 0, end = (int)(incl >> (8 * s)) & 0xff;
                    const int per = (end - start + 3) >> 2;
                    const int a0 = start + part * per < end ? start + part * per : end;
                    const int a1 = a0 + per < end ? a0 + per : end;
                    for (int p0 = a0; p0 < a1; p0 += 2) {
                        const bool two = p0 + 1 < a1;
                        const float *cp0 = stg + (size_t)(qj[p0] - (s << srl)) * RSX;
                        const float *cp1 = stg + (size_t)(qj[two ? p0 + 1 : p0] - (s << srl)) * RSX;
                        float s0 = 0.0f, s1 = 0.0f;
                        if (D == DP) knn_pair_dist<DP, true>(qreg, cp0, cp1, D, s0, s1);
                        else knn_pair_dist<DP, false>(qreg, cp0, cp1, D, s0, s1);
                        qd[p0] = __builtin_bit_cast(unsigned int, s0);
                        if (two) qd[p0 + 1] = __builtin_bit_cast(unsigned int, s1);
                    }
                }
                if (s < 3) KNN_PROBE_MARK(27 + 2 * s);
            }
        } else {
            // D > 64 (the fourth EdgeConv's 128 features): a 128-dimension query row is 128 registers -- with the stage registers
            // and the pair buffers the kernel spilled 330-380 of them.  The exact phase runs once per column half instead:
            // dimensions 0..63 of every row are staged and summed first (the partial sum waits in the pair's distance slot),
            // then 64..D-1 continue it -- the oracle's order; each half is the D = 64 phase (same stage size, same traffic).
            const int nhalf = D > 64 ? 2 : 1;
            for (int hf = 0; hf < nhalf; ++hf) {
                const int hoff = 64 * hf, wdt = hf ? D - 64 : DS;  // this half's first column and width
                f32x4v qreg[16];
                {
                    const float *qrow = xb + (size_t)(act ? qi : 0) * D + hoff;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        qreg[t] = 4 * t < wdt ? *reinterpret_cast<const f32x4v *>(qrow + 4 * t) : f32x4v{0.f, 0.f, 0.f, 0.f};
                }
                for (int s = 0; s < nstage; ++s) {
                    if (s || hf) __syncthreads();  // every lane is done with the previous stage
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int row = srow + i * RPI;
                        if (row < SRW) *reinterpret_cast<f32x4v *>(stg + (size_t)row * RSX + scol) = sreg[i];
                    }
                    __syncthreads();
                    if (s < 3 && hf == 0) KNN_PROBE_MARK(26 + 2 * s);
                    // the next stage of this half, or the first stage of the second half, in flight while this one is evaluated
                    // (a half narrower than 64 columns: the pieces beyond it re-read its last piece and are never used)
                    if (s + 1 < nstage) {
                        knn_stage_fetch(yb + hoff, D, M, (s + 1) << srl, srow, RPI, scol < wdt ? scol : wdt - 4, sreg);
                    } else if (hf + 1 < nhalf) {
                        knn_stage_fetch(yb + 64, D, M, 0, srow, RPI, scol < D - 64 ? scol : D - 68, sreg);
                    }
                    if (act) {
                        const int start = s ? (int)(incl >> (8 * (s - 1))) & 0xff : 0, end = (int)(incl >> (8 * s)) & 0xff;
                        const int per = (end - start + 3) >> 2;
                        const int a0 = start + part * per < end ? start + part * per : end;
                        const int a1 = a0 + per < end ? a0 + per : end;
                        for (int p0 = a0; p0 < a1; p0 += 2) {
                            const bool two = p0 + 1 < a1;
                            const float *cp0 = stg + (size_t)(qj[p0] - (s << srl)) * RSX;
                            const float *cp1 = stg + (size_t)(qj[two ? p0 + 1 : p0] - (s << srl)) * RSX;
                            float s0 = hf ? __builtin_bit_cast(float, qd[p0]) : 0.0f;
                            float s1 = hf && two ? __builtin_bit_cast(float, qd[p0 + 1]) : 0.0f;
                            if (wdt == 64) knn_pair_dist<64, true>(qreg, cp0, cp1, wdt, s0, s1);
                            else knn_pair_dist<64, false>(qreg, cp0, cp1, wdt, s0, s1);
                            qd[p0] = __builtin_bit_cast(unsigned int, s0);
                            if (two) qd[p0 + 1] = __builtin_bit_cast(unsigned int, s1);
                        }
                    }
                    if (s < 3 && hf == 0) KNN_PROBE_MARK(27 + 2 * s);
                }
            }
        }
    } else if (wave_active && fast) {
        // the oracle's distance of every id.  The query row sits in registers; candidate rows are gathered from L2
        // one full 128-byte line per request (32 dimensions), two candidates in flight
        const float *qrow = xb + (size_t)qi * D;
        if (vec4y && vec4x) {
            constexpr int QR = DP > 64 ? 1 : DP / 4;  // D > 64: the query pieces are re-read (L1) with every 32-dimension block
            float4 qreg[QR];
            if (DP <= 64) {
#pragma unroll
                for (int t = 0; t < QR; ++t)
                    qreg[t] = 4 * t < D ? *reinterpret_cast<const float4 *>(qrow + 4 * t) : float4{0.f, 0.f, 0.f, 0.f};
            }
            for (int p0 = mystart; p0 < mystart + mycount; p0 += 2) {
                const bool two = p0 + 1 < mystart + mycount;
                const float *cp0 = yb + (size_t)qj[p0] * D;
                const float *cp1 = yb + (size_t)qj[two ? p0 + 1 : p0] * D;
                float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                for (int d0 = 0; d0 < DP; d0 += 32) {
                    if (d0 < D) {
                        float4 c0[8], c1[8], q8[DP > 64 ? 8 : 1];
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (d0 + 4 * t < D) {
                                c0[t] = *reinterpret_cast<const float4 *>(cp0 + d0 + 4 * t);
                                c1[t] = *reinterpret_cast<const float4 *>(cp1 + d0 + 4 * t);
                                if (DP > 64) q8[t] = *reinterpret_cast<const float4 *>(qrow + d0 + 4 * t);
                            }
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (d0 + 4 * t < D) {
                                const float4 qv = DP > 64 ? q8[DP > 64 ? t : 0] : qreg[DP > 64 ? 0 : d0 / 4 + t];
                                float t0 = qv.x - c0[t].x, t1 = qv.y - c0[t].y, t2 = qv.z - c0[t].z, t3 = qv.w - c0[t].w;
                                s0 = s0 + t0 * t0;
                                s0 = s0 + t1 * t1;
                                s0 = s0 + t2 * t2;
                                s0 = s0 + t3 * t3;
                                t0 = qv.x - c1[t].x; t1 = qv.y - c1[t].y; t2 = qv.z - c1[t].z; t3 = qv.w - c1[t].w;
                                s1 = s1 + t0 * t0;
                                s1 = s1 + t1 * t1;
                                s1 = s1 + t2 * t2;
                                s1 = s1 + t3 * t3;
                            }
                    }
                }
                qd[p0] = __builtin_bit_cast(unsigned int, s0);
                if (two) qd[p0 + 1] = __builtin_bit_cast(unsigned int, s1);
            }
        } else {
            for (int p0 = mystart; p0 < mystart + mycount; ++p0) {
                const float *cp = yb + (size_t)qj[p0] * D;
                float sd = 0.0f;
                for (int d = 0; d < D; ++d) {
                    const float t = qrow[d] - cp[d];
                    sd = sd + t * t;
                }
                qd[p0] = __builtin_bit_cast(unsigned int, sd);
            }
        }
    }
    __syncthreads();
    KNN_PROBE_MARK(23);
    // (4) rank on the distance bits (squared distances are >= +0: unsigned order), verified as in knn_f16_d3_kernel.  Passes of eight of
    //     the lane's entries against all n keys; a remainder of at most four / two entries in every lane of the wave takes a narrower
    //     pass (round 4: n = 33 ... 36 survivors -- nine entries per lane -- cost a second full pass, 1152 instead of 720 operations).
    {
        const int myc = (wave_active && fast) ? mycount : 0;
        const int nloop = (wave_active && fast) ? n : 0;
        int below = 0;
        auto rank_pass = [&](auto wc, int e0) {
            constexpr int W = decltype(wc)::value;
            unsigned int md[W];
            int rank[W];
#pragma unroll
            for (int u = 0; u < W; ++u) {
                md[u] = e0 + u < myc ? qd[mystart + e0 + u] : 0xffffffffu;
                rank[u] = 0;
            }
            for (int i = 0; i < nloop; i += 4) {  // (nloop: n for the lanes of a fast query, 0 for the others)
                const uint4 o = *reinterpret_cast<const uint4 *>(qd + i);
#pragma unroll
                for (int u = 0; u < W; ++u) {  // compare + add-with-carry: two VALU ops per pair
                    unsigned long long cc;
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.x), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.y), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.z), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.w), "v"(md[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < W; ++u)
                if (e0 + u < myc && rank[u] < kk) {
                    slots[rank[u]] = ((unsigned long long)md[u] << 32) | (unsigned int)qj[mystart + e0 + u];
                    below += 1 + (rank[u] << 8);
                }
        };
        for (int e0 = 0;;) {
            const int rem = myc - e0;
            if (__ballot(rem > 0) == 0ull) break;
            if (__ballot(rem > 2) == 0ull) { rank_pass(std::integral_constant<int, 2>{}, e0); e0 += 2; }
            else if (__ballot(rem > 4) == 0ull) { rank_pass(std::integral_constant<int, 4>{}, e0); e0 += 4; }
            else { rank_pass(std::integral_constant<int, 8>{}, e0); e0 += 8; }
        }
        if (below) atomicAdd(&qbelow[cw * 32 + jl], below);
    }
    __syncthreads();
    KNN_PROBE_MARK(24);
    // (5) verify (count and rank sum, see knn_f16_d3_kernel); slots [drop, kk) are the answer, in order: the query's
    //     four lanes share the writes, 16 bytes at a time
    const bool bad = wave_active && qi < N && fast && qbelow[cw * 32 + jl] != kk + ((kk * (kk - 1) / 2) << 8);  // (n >= kk here)
    if (wave_active && qi < N && fast && !bad) {
        const size_t obase = ((size_t)b * N + qi) * k;
        if ((k & 3) == 0 && ((reinterpret_cast<uintptr_t>(idx) | (dist ? reinterpret_cast<uintptr_t>(dist) : 0)) & 15) == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int v = part + 4 * u;
                if (4 * v < k) {
                    unsigned long long key[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) key[e] = slots[drop + 4 * v + e];
                    *reinterpret_cast<int4 *>(idx + obase + 4 * v) =
                        int4{(int)(unsigned int)key[0], (int)(unsigned int)key[1], (int)(unsigned int)key[2], (int)(unsigned int)key[3]};
                    if (dist)
                        *reinterpret_cast<float4 *>(dist + obase + 4 * v) =
                            float4{__builtin_bit_cast(float, (unsigned int)(key[0] >> 32)), __builtin_bit_cast(float, (unsigned int)(key[1] >> 32)),
                                   __builtin_bit_cast(float, (unsigned int)(key[2] >> 32)), __builtin_bit_cast(float, (unsigned int)(key[3] >> 32))};
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = drop + part + 4 * u;
                if (r < kk) {
                    const unsigned long long key = slots[r];
                    idx[obase + r - drop] = (int)(unsigned int)key;
                    if (dist) dist[obase + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
                }
            }
        }
    }
    if (!wave_active) return;
    // tied queries are ranked again on the full keys, and the leftovers (exact merge) answered, by the pair's two waves on
    // alternate queries
    const bool slowq = qi < N && !fast && !handled;
    const unsigned long long badmask = __ballot(bad);
    for (unsigned int bm = (unsigned int)badmask | (unsigned int)(badmask >> 32); bm; bm &= bm - 1) {
        const int j = __builtin_ctz(bm);  // a tie in the distance among the first kk of query j
        if ((j & 1) != (consumer ? 0 : 1)) continue;  // (the pair's two waves take alternate queries)
        const int qs = cw * 32 + j;
        unsigned long long *sj = reinterpret_cast<unsigned long long *>(lists) + (size_t)qs * 33;
        knn_rank_ties(reinterpret_cast<const unsigned int *>(sm) + (size_t)qs * kMKeyStride,
                      reinterpret_cast<const int *>(sm) + (size_t)kMWaves * 32 * kMKeyStride + (size_t)qs * kMKeyStride, qn_n[qs], kk,
                      sj, lane);
        for (int r = drop + lane; r < kk; r += 64) {
            const unsigned long long key = sj[r];
            idx[((size_t)b * N + q0 + j) * k + r - drop] = (int)(unsigned int)key;
            if (dist) dist[((size_t)b * N + q0 + j) * k + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
        }
    }
    // leftovers (overflowing lists, non-finite bands), wave-cooperative (scratch: behind all the slots)
    int *wscratch = lists + kMWaves * 32 * 33 * 2 + wv * 128;
    const unsigned long long slowmask = __ballot(slowq);
    const unsigned int slow32 = (unsigned int)slowmask | (unsigned int)(slowmask >> 32);
    for (int j = consumer ? 0 : 1; j < 32; j += 2) {
        if (!((slow32 >> j) & 1u) || q0 + j >= N) continue;
        float bd;
        int bj;
        __builtin_amdgcn_wave_barrier();
        knn_exact_bruteforce(xb + (size_t)(q0 + j) * D, yb, M, D, kk, lane, reinterpret_cast<float *>(wscratch), wscratch + 64, bd, bj);
        const int r = lane - drop;
        if (r >= 0 && r < k) {
            idx[((size_t)b * N + q0 + j) * k + r] = bj;
            if (dist) dist[((size_t)b * N + q0 + j) * k + r] = bd;
        }
    }
    KNN_PROBE_MARK(25);
}

// ---- EdgeConv graph features (src/models/dgcnn.jl:36-51): cat(X, KNNGraph - X, dims=1) in one pass --------
// layout 0: out (2F,K,N,B) as the reference holds it after `cat(..., dims = 1)` (:45)
__global__ __launch_bounds__(kThreads) void edge_features_cat_kernel(const float *__restrict__ x, int N, int B,
                                                                     int F, int k,
                                                                     const int32_t *__restrict__ idx,
                                                                     float *__restrict__ out) {
    const long long total = (long long)B * N * k * 2 * F;
    const int F2 = 2 * F;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total;
         e += (long long)gridDim.x * kThreads) {
        const long long row = e / F2;  // (b*N+i)*k + r
        const int f = (int)(e - row * F2);
        const long long bn = row / k;  // b*N + i
        if (f < F) {
            out[e] = x[(size_t)bn * F + f];
        } else {
            const int b = (int)(bn / N);
            const int j = idx[row];
            out[e] = x[((size_t)b * N + j) * F + (f - F)] - x[(size_t)bn * F + (f - F)];
        }
    }
}

// layout 1: out (K*N, 2F, B), what reaches the 1x1 conv after PermutedDimsArray + reshape (:48-51).
// One thread per (r,i) position (the contiguous dimension of the output), looping over features, so every
// feature row is written with unit stride; the two source rows are read as float4 when F % 4 == 0.
template <bool VEC4>
__global__ __launch_bounds__(kThreads) void edge_features_mlp_kernel(const float *__restrict__ x, int N, int B,
                                                                     int F, int k,
                                                                     const int32_t *__restrict__ idx,
                                                                     float *__restrict__ out) {
    const int b = blockIdx.y;
    const long long KN = (long long)k * N;
    const long long e = (long long)blockIdx.x * kThreads + threadIdx.x;  // i*k + r
    if (e >= KN) return;
    const int i = (int)(e / k);
    const int j = idx[(size_t)b * KN + e];
    const float *xi = x + ((size_t)b * N + i) * F;
    const float *xj = x + ((size_t)b * N + j) * F;
    float *o = out + (size_t)b * 2 * F * KN + e;
    if (VEC4) {
        for (int f = 0; f < F; f += 4) {
            const float4 a = *reinterpret_cast<const float4 *>(xi + f);
            const float4 c = *reinterpret_cast<const float4 *>(xj + f);
            o[(size_t)(f + 0) * KN] = a.x;
            o[(size_t)(f + 1) * KN] = a.y;
            o[(size_t)(f + 2) * KN] = a.z;
            o[(size_t)(f + 3) * KN] = a.w;
            o[(size_t)(F + f + 0) * KN] = c.x - a.x;
            o[(size_t)(F + f + 1) * KN] = c.y - a.y;
            o[(size_t)(F + f + 2) * KN] = c.z - a.z;
            o[(size_t)(F + f + 3) * KN] = c.w - a.w;
        }
    } else {
        for (int f = 0; f < F; ++f) {
            const float a = xi[f];
            o[(size_t)f * KN] = a;
            o[(size_t)(F + f) * KN] = xj[f] - a;
        }
    }
}

// Same, four consecutive (r,i) positions per thread: the 4 x 4 block (4 positions x 4 features) is read as
// float4 along the features and written as float4 along the positions -- every store is 16 bytes, a wave writes
// 1 KiB runs.  Needs F % 4 == 0, (k*N) % 4 == 0 and 16-byte aligned x / out.
// Round 4: blockIdx.z splits the feature loop (fper features per block) -- F = 64 at C4' is 335 MB written by what used to be 640
// blocks (2.5 per CU, ten waves per CU, each a serial loop of load -> 8 stores); the write stream wants many more waves in
// flight (tools/ubench_hbm.hip: 4.7 TB/s from 2048 blocks, 6.1 from 32768) -- and NT selects streaming (non-temporal) stores:
// the tensor is larger than the Infinity Cache and nobody reads it back inside the launch.
template <bool NT>
__global__ __launch_bounds__(kThreads) void edge_features_mlp4_kernel(const float *__restrict__ x, int N, int B,
                                                                      int F, int k,
                                                                      const int32_t *__restrict__ idx,
                                                                      float *__restrict__ out, int fper) {
    const int b = blockIdx.y;
    const long long KN = (long long)k * N;
    const long long e0 = ((long long)blockIdx.x * kThreads + threadIdx.x) * 4;  // i*k + r of the first position
    if (e0 >= KN) return;
    const int f_lo = blockIdx.z * fper, f_hi = f_lo + fper < F ? f_lo + fper : F;
    const int4 jj = *reinterpret_cast<const int4 *>(idx + (size_t)b * KN + e0);
    const float *xb = x + (size_t)b * N * F;
    const float *xi0 = xb + (size_t)(e0 / k) * F, *xi1 = xb + (size_t)((e0 + 1) / k) * F;
    const float *xi2 = xb + (size_t)((e0 + 2) / k) * F, *xi3 = xb + (size_t)((e0 + 3) / k) * F;
    const float *xj0 = xb + (size_t)jj.x * F, *xj1 = xb + (size_t)jj.y * F, *xj2 = xb + (size_t)jj.z * F,
                *xj3 = xb + (size_t)jj.w * F;
    float *o = out + (size_t)b * 2 * F * KN + e0;
    auto put = [](float *p, const f32x4v &v) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4v *>(p));
        else *reinterpret_cast<f32x4v *>(p) = v;
    };
    for (int f = f_lo; f < f_hi; f += 4) {
        const float4 a0 = *reinterpret_cast<const float4 *>(xi0 + f), a1 = *reinterpret_cast<const float4 *>(xi1 + f);
        const float4 a2 = *reinterpret_cast<const float4 *>(xi2 + f), a3 = *reinterpret_cast<const float4 *>(xi3 + f);
        const float4 c0 = *reinterpret_cast<const float4 *>(xj0 + f), c1 = *reinterpret_cast<const float4 *>(xj1 + f);
        const float4 c2 = *reinterpret_cast<const float4 *>(xj2 + f), c3 = *reinterpret_cast<const float4 *>(xj3 + f);
        put(o + (size_t)(f + 0) * KN, f32x4v{a0.x, a1.x, a2.x, a3.x});
        put(o + (size_t)(f + 1) * KN, f32x4v{a0.y, a1.y, a2.y, a3.y});
        put(o + (size_t)(f + 2) * KN, f32x4v{a0.z, a1.z, a2.z, a3.z});
        put(o + (size_t)(f + 3) * KN, f32x4v{a0.w, a1.w, a2.w, a3.w});
        put(o + (size_t)(F + f + 0) * KN, f32x4v{c0.x - a0.x, c1.x - a1.x, c2.x - a2.x, c3.x - a3.x});
        put(o + (size_t)(F + f + 1) * KN, f32x4v{c0.y - a0.y, c1.y - a1.y, c2.y - a2.y, c3.y - a3.y});
        put(o + (size_t)(F + f + 2) * KN, f32x4v{c0.z - a0.z, c1.z - a1.z, c2.z - a2.z, c3.z - a3.z});
        put(o + (size_t)(F + f + 3) * KN, f32x4v{c0.w - a0.w, c1.w - a1.w, c2.w - a2.w, c3.w - a3.w});
    }
}

// Adjoint w.r.t. X.  CreateSingleKNNGraph is @nograd (src/models/dgcnn.jl:9), so the gathered neighbours are
// constants and dX[f,i,b] = sum_r (g[f,r,i,b] - g[F+f,r,i,b]), accumulated in rank order.
__global__ __launch_bounds__(kThreads) void edge_features_bwd_kernel(const float *__restrict__ g, int N, int B, int F,
                                                                     int k, int layout, float *__restrict__ gx) {
    const long long total = (long long)B * N * F;
    const long long KN = (long long)k * N;
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total;
         e += (long long)gridDim.x * kThreads) {
        long long bn;
        int f;
        if (layout == 0) {  // consecutive threads -> consecutive f (reads stride 1 in f)
            bn = e / F;
            f = (int)(e - bn * F);
        } else {            // consecutive threads -> consecutive i (reads k-float runs)
            const long long bf = e / N;
            const int i = (int)(e - bf * N);
            const int b = (int)(bf / F);
            f = (int)(bf - (long long)b * F);
            bn = (long long)b * N + i;
        }
        const int b = (int)(bn / N);
        const int i = (int)(bn - (long long)b * N);
        float acc = 0.0f;
        for (int r = 0; r < k; ++r) {
            float a, c;
            if (layout == 0) {
                const size_t base = ((size_t)bn * k + r) * 2 * F;
                a = g[base + f];
                c = g[base + F + f];
            } else {
                const size_t base = (size_t)b * 2 * F * KN + (size_t)i * k + r;
                a = g[base + (size_t)f * KN];
                c = g[base + (size_t)(F + f) * KN];
            }
            acc = acc + (a - c);
        }
        gx[(size_t)bn * F + f] = acc;
    }
}


bool knn_pre_shape_ok(int M, int D, int kk);
// the shapes fx3d_knn_ws serves with the pre-pass: the single-piece fp16 filter (the default of knn_mfma_kernel)
bool knn_pre_eligible(const float *x, const float *y, int M, int D, int kk) {
    if (!knn_pre_shape_ok(M, D, kk)) return false;
    if (((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) != 0) return false;
    return !(opt(OPT_KNN_NO_MFMA) || opt(OPT_KNN_NO_PREPASS));
}

template <int DK, bool F16, bool SPLIT>
fx3d_status launch_knn_mfma_dk(const float *x, int N, const float *y, int M, int B, int D, int k, int drop,
                               int32_t *idx, float *dist, hipStream_t st, void *pre_ws = nullptr, int xdiv = 1) {
    constexpr int DP = DK * 32, RS = DP + 4, RSI = (F16 && !SPLIT) ? DP / 2 : DP;
    // list lengths + per-query counters + cmax + per-dimension centre + per-stage survivor counts ...
    const bool use_pre = pre_ws != nullptr && F16 && !SPLIT;
    const size_t small = (size_t)kMWaves * 64 * 4 + (size_t)3 * kMWaves * 32 * 4 + 64 + (size_t)DP * 4 + (size_t)kMWaves * 32 * 8;
    static_assert(kMWaves * 32 * 33 * 8 + 2 * kMWaves * 128 * 4 <= kMWaves * kMLCap * 64 * 4, "rank slots + fallback scratch alias the mask lists");
    const int keep_norms = M <= 4096;  // all candidate norms stay in LDS: phase B does not recompute them
    // ... then the tail that is dead after the decode: lists (later the slots), medium path (id lists + merge scratch, one
    // per wave), candidate norms
    size_t tail = (size_t)kMWaves * kMLCap * 64 * 4 + (size_t)2 * kMWaves * (kMMedCap + 128) * 4;
    if (keep_norms) tail += (size_t)((M + 255) / 256 * 256 + 256) * 4;
    // fp16 filter, room permitting: a second norms array (the candidate's error share folded in, upwards / downwards)
    const int two_norms = F16 && keep_norms && M <= 2048;
    if (two_norms) tail += (size_t)((M + 255) / 256 * 256 + 256) * 4;
    const size_t fixed = small + tail;
    const size_t budget = 150 * 1024 - fixed;                                  // floats*4 for the two chunk buffers
    int CH = (int)(budget / 2 / ((size_t)RSI * 4 + 4)) / 64 * 64;
    if (CH > 256) CH = 256;
    constexpr int kMUnits = SPLIT ? kMUnitsSplit : kMUnitsSingle;
    if (F16 && !use_pre && CH > kMUnits * kMProd * 8 / DP / 64 * 64) CH = kMUnits * kMProd * 8 / DP / 64 * 64;  // producer register budget
    const int mpad = (M + 63) / 64 * 64;
    if (CH > mpad) CH = mpad;
    if (use_pre) CH = CH >= 256 ? 256 : (CH >= 128 ? 128 : 64);  // chunks tile the image's 256-row padding exactly
    size_t img = 2 * ((size_t)CH * RSI + CH);                                  // floats
    const size_t qstage = (size_t)kMWaves * 32 * RS;                           // prologue: query rows
    const size_t exact = (size_t)2 * kMWaves * 32 * kMKeyStride;               // exact phase: distance bits + indices
    if (img < qstage) img = qstage;
    if (img < exact) img = exact;
    img = (img + 3) & ~(size_t)3;
    size_t lds = img * 4 + fixed;
    // staged exact phase: candidate rows pass through the tail in stages of 2^srl rows of 4D + 16 bytes (at most 8 stages,
    // at most 8 sweeps of the block per stage; the allocation may grow up to the limit for it).  0 = gather from L2.
    int srl = 0;
    const int DS = DP > 64 && D > 64 ? 64 : D;  // staged width of a row: D > 64 goes through in two column halves
    const int PR = DS / 4;
    const bool stageable = D % 4 == 0 && (kMThreads % PR) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (stageable) {
        const size_t room = 152 * 1024 - (img * 4 + small);
        for (int l = 8; l >= 5; --l) {
            const size_t need = ((size_t)1 << l) * ((size_t)DS * 4 + 16);
            if (need <= room && ((size_t)1 << l) * PR <= 8 * (size_t)kMThreads && ((M + (1 << l) - 1) >> l) <= 8) {
                srl = l;
                if (img * 4 + small + need > lds) lds = img * 4 + small + need;
                break;
            }
        }
    }
    // column slices (round 4) instead of row stages: every row of the cloud, 16 dimensions at a time, as four planes of 16-byte
    // pieces -- when the whole cloud's slice fits the same tail (M <= 1024 at the kernel's 512 threads x 8 pieces)
    int csl = 0;
    if (stageable && D % 16 == 0 && D <= 64 && M <= 8 * (kMThreads / 4)) {
        const size_t room = 152 * 1024 - (img * 4 + small);
        const size_t mpc = (size_t)(M + kMThreads / 4 - 1) / (kMThreads / 4) * (kMThreads / 4);
        const size_t need = 4 * (mpc * 16 + 32);
        if (need <= room) {
            csl = D / 16;
            srl = 0;  // (the decode does not group the ids by row stage)
            if (img * 4 + small + need > lds) lds = img * 4 + small + need;
        }
    }
    const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_mfma_kernel<DK, F16, SPLIT>), 152 * 1024, "knn_mfma_kernel");
    if (arc != FX3D_OK) return arc;
    FX3D_REQUIRE(lds <= 152 * 1024, "fx3d_knn: internal LDS plan exceeds the CU (D=%d)", D);
    const int qpb = kMWaves * 32;
    const int nbx = (N + qpb - 1) / qpb;
    const int bpad = B >= 8 ? (B + 7) / 8 * 8 : B;
    KnnPre pre{};
    if (use_pre) {
        pre = KnnPre::make(pre_ws, M, DP);
        hipLaunchKernelGGL(knn_pre_stats_kernel, dim3(kPreParts, B), dim3(kPreThreads), 0, st, y, M, D, DP, pre);
        hipLaunchKernelGGL((knn_pre_image_kernel<DK>), dim3(kPreParts, B), dim3(kPreThreads), 0, st, y, M, D, two_norms, pre);
    }
    if (use_pre) {
        const fx3d_status arc2 = ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_mfma_kernel<DK, F16, SPLIT, F16 && !SPLIT>), 152 * 1024,
                                                    "knn_mfma_kernel<pre>");
        if (arc2 != FX3D_OK) return arc2;
        hipLaunchKernelGGL((knn_mfma_kernel<DK, F16, SPLIT, F16 && !SPLIT>), dim3(nbx * bpad), dim3(kMThreads), lds, st, x, N, y, M, B, D,
                           k, drop, idx, dist, CH, (int)img, keep_norms, two_norms, srl, pre_ws, xdiv, csl, 1);
    } else
        hipLaunchKernelGGL((knn_mfma_kernel<DK, F16, SPLIT>), dim3(nbx * bpad), dim3(kMThreads), lds, st, x, N, y, M, B, D,
                           k, drop, idx, dist, CH, (int)img, keep_norms, two_norms, srl, nullptr, xdiv, csl, 0);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status launch_knn_mfma(const float *x, int N, const float *y, int M, int B, int D, int k, int drop, int32_t *idx,
                            float *dist, hipStream_t st, void *pre_ws = nullptr, int xdiv = 1) {
    const int dk = (D + 31) / 32;
    // fp16-split filter: needs 16-byte loads (D % 4 == 0, aligned clouds) and all norms in LDS up front
    const bool f16 = D % 4 == 0 && M <= 4096 && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                     ((size_t)M * D * 4) % 16 == 0;
    if (f16) {
        switch (dk) {
            case 1: return launch_knn_mfma_dk<1, true, false>(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
            case 2: return launch_knn_mfma_dk<2, true, false>(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
            default: return launch_knn_mfma_dk<4, true, false>(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
        }
    }
    switch (dk) {
        case 1: return launch_knn_mfma_dk<1, false, true>(x, N, y, M, B, D, k, drop, idx, dist, st, nullptr, xdiv);
        case 2: return launch_knn_mfma_dk<2, false, true>(x, N, y, M, B, D, k, drop, idx, dist, st, nullptr, xdiv);
        default: return launch_knn_mfma_dk<4, false, true>(x, N, y, M, B, D, k, drop, idx, dist, st, nullptr, xdiv);
    }
}

size_t knn_wave_generic_lds(int D) {
    return (size_t)kGT * (D + 1) * 4 + (kWThreads / 64) * (kGQ * (64 * 16 + 8) + D * 16) + 16;
}
bool knn_mfma_eligible(int M, int D, int kk) { return D >= 4 && D <= 128 && kk <= 32 && M >= 64; }

// the general path's geometry: as many waves per block as their key arrays fit in LDS (0 = M too large)
int knn_select_waves(int M) {
    const size_t per_wave = (size_t)((M + 255) / 256 * 256) * 4;
    int w = (int)(kSelMaxLds / per_wave);
    return w > 4 ? 4 : w;
}
// ... and the survivor list of a wave (entries; 0: the keys of M candidates leave no room, or kk is so large that ranking
// against all keys costs less than kk^2 / 64)
int knn_select_list(int M, int kk, int *nw) {
    const int Mpad = (M + 255) / 256 * 256, lcap = (kk + 63) / 64 * 64;
    if (4ll * kk * kk > 3ll * M * M) return 0;  // (kk^2 / 64 x 4 operations against M^2 / 256 x 12 for the ranks over all keys)
    int w = (int)(kSelMaxLds / ((size_t)(Mpad + 2 * lcap) * 4));
    if (w < 1) return 0;
    *nw = w > 4 ? 4 : w;
    return lcap;
}
bool knn_needs_select(int M, int D, int kk) {
    // (D = 3, 44 < kk <= 64: the wave kernel's candidate list holds 64 - kk entries between merges -- 546 us at kk = 64 and C4's
    //  shape against 188 us here, 160 against ~185 at kk = 41)
    if (D == 3 && knn_f16_d3_shape_ok(M, kk)) return false;  // (round 3: the matrix-core kernel up to kk = 64)
    if (D == 3 && kk > 44 && kk <= 64 && knn_select_waves(M) >= 1) return true;
    return kk > 64 || (D != 3 && !knn_mfma_eligible(M, D, kk) && knn_wave_generic_lds(D) > 64 * 1024);
}

// ---- candidate slices (few clouds with many rows; fx3d_knn_ws) -----------------------------------------------------------------
// A block of the matrix-core kernels takes 128 (64) queries against ALL candidates of their cloud: B = 1, N = M = 8192 is 64
// blocks on 256 CUs, each sweeping 8192 candidates (D = 64: 758 us, and beyond the sizes the fp16 filter / the staged exact phase
// take).  With scratch the search runs on S contiguous slices of every cloud as B x S virtual clouds of M / S rows -- the same
// kernels, the queries' batch index is b / S --, each slice's kk = k + drop nearest land, in order, in the scratch, and one wave
// per query merges the S lists on the full (distance, index) keys: the exact answer (a slice's kk nearest contain every member of
// the cloud's kk nearest that lies in the slice; the global index = slice offset + local index keeps the oracle's tie order).
// y (D, M, B) -> (D, M / S, B x S) with slice s of cloud b = its rows s, s + S, s + 2S, ...: the verified merge's slices must be
// samples of the whole cloud -- contiguous slices of a cloud whose index neighbours are spatial neighbours (a scan line, a sorted
// mesh) hold ALL of a query's neighbours in one slice, and every query would be flagged.
__global__ __launch_bounds__(256) void knn_interleave_kernel(const float *__restrict__ y, int M, int D, int B, int S, float *__restrict__ out,
                                                             int vec4) {
    const int Ms = M / S;
    if (vec4) {
        const int D4 = D / 4;
        const long long total = (long long)B * M * D4;
        const float4 *y4 = reinterpret_cast<const float4 *>(y);
        float4 *o4 = reinterpret_cast<float4 *>(out);
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
            const long long row = e / D4;  // output row: (b * S + sl) * Ms + l
            const int f = (int)(e - row * D4);
            const int l = (int)(row % Ms);
            const long long bs = row / Ms;
            const int sl = (int)(bs % S);
            const long long b = bs / S;
            o4[e] = y4[(b * M + (long long)l * S + sl) * D4 + f];
        }
    } else {
        const long long total = (long long)B * M * D;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
            const long long row = e / D;
            const int f = (int)(e - row * D);
            const int l = (int)(row % Ms);
            const long long bs = row / Ms;
            const int sl = (int)(bs % S);
            const long long b = bs / S;
            out[e] = y[(b * M + (long long)l * S + sl) * D + f];
        }
    }
}

// Block of 256 threads = 4 / wpq queries x wpq waves per query (wpq = 1, 2, 4: n <= 64, 128, more -- every entry its own thread up to
// n = 256, two per thread beyond): the chain load -> LDS -> searches -> store runs once per thread (a wave per query with n / 64
// entries per lane ran it n / 64 times back to back: 17 / 44 us at n = 64 / 128 and C4's shape).  Dynamic LDS: (4 / wpq) x n keys.
__global__ __launch_bounds__(256) void knn_merge_slices_kernel(const int32_t *__restrict__ widx, const float *__restrict__ wdist, int N, int B,
                                                               int S, int Ms, int kl, int kk, int k, int drop, int32_t *__restrict__ idx,
                                                               float *__restrict__ dist, unsigned char *__restrict__ flags, int interleaved,
                                                               int wpq) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long mkeys[];  // [4 / wpq][n] (distance key, global index): unique,
                                                                                // their unsigned order is the oracle's
    __shared__ unsigned long long tkey[4];
    const int n = S * kl;  // kl entries per slice: kk, or 32 < kk with `flags` (verified below)
    const int qpb = 4 / wpq;
    const int qs = (threadIdx.x >> 6) / wpq;                       // query slot of this thread's wave
    const int t0 = threadIdx.x - qs * wpq * 64;                    // thread index within the query's wpq waves
    const int i = blockIdx.x * qpb + qs, b = blockIdx.y;
    const bool live = i < N;                                        // (uniform per wave; every thread reaches the barriers)
    unsigned long long *keys = mkeys + (size_t)qs * n;
    const size_t q = (size_t)b * N + (live ? i : 0);
    const int nt = wpq * 64;
    unsigned long long me[2] = {0ull, 0ull};
    int sl2[2] = {0, 0}, r2[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = t0 + u * nt;
        if (live && e < n) {
            const int sl = (int)((unsigned int)e / (unsigned int)kl), r = e - sl * kl;
            const size_t src = (((size_t)b * S + sl) * N + i) * kl + r;
            // global index: contiguous slices -- offset + local; interleaved slices (slice sl = rows sl, sl + S, ...) -- local * S + sl;
            // either way increasing with the local index inside a slice, so the slices' own tie order is the global one
            me[u] = ((unsigned long long)dist_key(wdist[src]) << 32) | (unsigned int)(interleaved ? widx[src] * S + sl : widx[src] + sl * Ms);
            keys[e] = me[u];
            sl2[u] = sl; r2[u] = r;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = t0 + u * nt;
        if (live && e < n) {
            // every slice's list is ascending in these keys (the search kernels' order): the rank is the position in the own list
            // plus, per other slice, the number of its keys below `me` -- a binary search each
            int rank = r2[u];
            for (int t = 0; t < S; ++t) {
                if (t == sl2[u]) continue;
                const unsigned long long *L = keys + t * kl;
                int lo = 0, hi = kl;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (L[mid] < me[u]) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            if (rank >= drop && rank < kk) {
                idx[q * k + rank - drop] = (int)(unsigned int)me[u];
                if (dist) dist[q * k + rank - drop] = wdist[(((size_t)b * S + sl2[u]) * N + i) * kl + r2[u]];  // (the slice's own bits)
            }
            if (flags && rank == kk - 1) tkey[qs] = me[u];  // (exactly one entry: the keys are unique and n >= kk)
        }
    }
    if (flags) {
        // lists shorter than kk: the answer stands iff no slice can hide a candidate below the kk-th merged key T -- a slice's
        // unlisted candidates lie above its last listed key, so a slice whose last key is >= T hides nothing
        __syncthreads();
        if (live && t0 < 64) {  // the query's first wave
            const unsigned long long T = tkey[qs];
            const bool hides = t0 < S && keys[t0 * kl + kl - 1] < T;
            const unsigned long long any = __ballot(hides);
            if (t0 == 0) flags[q] = any ? 1 : 0;
        }
    }
}

bool knn_mfma_eligible(int M, int D, int kk);
// number of slices for a shape (1 = none): a function of the shape alone, so that fx3d_knn_workspace_bytes and the call agree
int knn_slices(int N, int M, int B, int D, int kk) {
    const int force = opt(OPT_KNN_SLICES);  // 0 = automatic, 1 = never, 2 / 4 / 8 = forced (when the shape allows it)
    if (force == 1) return 1;
    const bool d3 = D == 3;
    if (d3 ? !knn_f16_d3_shape_ok(M, kk) : (opt(OPT_KNN_NO_MFMA) || !knn_mfma_eligible(M, D, kk))) return 1;
    const int qpb = d3 && kk > 32 ? 64 : 128;
    const long long blocks = (long long)B * ((N + qpb - 1) / qpb);
    const int ncu = device_cus();
    auto fits = [&](int S) {
        if (M % S) return false;
        const int Ms = M / S;
        if (Ms < (d3 ? 1024 : 512) || Ms < 4 * kk || S * kk > 512) return false;
        if (!d3 && ((size_t)Ms * D * 4) % 16 != 0) return false;  // the slices keep the clouds' 16-byte alignment
        return (long long)B * S <= 65535;
    };
    if (force > 1) return (force == 2 || force == 4 || force == 8) && fits(force) ? force : 1;
    // automatic (tools/knn_slices_time.py): slice until the grid fills the chip; in feature space also down to the size the fp16
    // filter takes (4096 rows)
    int best = 1;
    for (int S = 2; S <= 8; S *= 2) {
        if (!fits(S)) continue;
        if (d3) {  // (measured, tools/knn_slices_time.py: only long sweeps on an under-filled chip pay for the second exact phase)
            if (S == 2 && blocks < ncu && M >= 8192) best = S;
            continue;
        }
        const bool underfilled = blocks * (S / 2) < ncu;
        const bool too_long = M / (S / 2) > 4096;  // beyond the fp16 filter: the Float32 GEMM + the L2 gather cost 2.7x even on a full grid
        if (underfilled || too_long) best = S;
    }
    return best;
}
size_t knn_pre_bytes(int M, int B, int D) {
    const int DP = (D + 31) / 32 * 32 == 96 ? 128 : (D + 31) / 32 * 32;
    return KnnPre::make(nullptr, M, DP).stride * (size_t)B;
}
bool knn_pre_shape_ok(int M, int D, int kk) {
    return D >= 4 && D <= 128 && kk <= 32 && M >= 64 && M <= 4096 && D % 4 == 0 && kPreThreads % (D / 4) == 0 && D / 4 <= 32;
}
// 32 < k + drop <= 64 in feature space (the matrix-core kernel selects up to 32): S slices, each slice's 32 nearest, a VERIFIED
// merge -- the kk nearest of the cloud spread over the slices (~kk / S each), so 32 per slice almost always hold them all; the merge
// checks it per query (knn_merge_slices_kernel) and the flagged queries are answered again by the general selection kernel.
int knn_wide_slices(int M, int D, int kk) {
    if (D < 4 || D > 128 || kk <= 32 || kk > 128 || opt(OPT_KNN_NO_MFMA) || opt(OPT_KNN_SLICES) == 1) return 0;
    const int force = opt(OPT_KNN_SLICES);
    const int S = kk > 64 ? 8 : (force == 2 || force == 4 || force == 8 ? force : (kk <= 45 ? 2 : 4));  // (65 ... 128: eight slices, ~kk / 8 each)  // (measured at C4's shape, D = 64, us: kk = 41 / 45 / 49 / 53
                                                                                         //  S = 2: 180 / 188 / 219 / 342 -- the flagged queries --, S = 4: 228 flat)
    if (M % S || M / S < 64 || ((size_t)(M / S) * D * 4) % 16 != 0 || knn_select_waves(M) < 1) return 0;
    return S;
}
// scratch of a call: [pre-pass slabs of the (virtual) clouds][slice results: indices, distances][flags][interleaved clouds], 256-byte aligned parts
struct KnnScratch {
    int S;        // candidate slices per cloud (1 = none)
    int kl;       // entries per slice list: k + drop, or 32 with `verify`
    bool verify;  // the slices' lists are shorter than k + drop: verified merge + fallback for the flagged queries
    size_t pre_bytes, list_bytes, flag_bytes, copy_bytes, total;
    static KnnScratch plan(int N, int M, int B, int D, int kk) {
        KnnScratch p{};
        p.S = knn_slices(N, M, B, D, kk);
        p.kl = kk;
        if (p.S == 1) {
            const int W = knn_wide_slices(M, D, kk);
            if (W > 1 && (long long)B * W <= 65535) { p.S = W; p.kl = 32; p.verify = true; }
        }
        const int Ms = M / p.S;
        p.pre_bytes = knn_pre_shape_ok(Ms, D, p.kl) ? (knn_pre_bytes(Ms, B * p.S, D) + 255) & ~(size_t)255 : 0;
        p.list_bytes = p.S > 1 ? (((size_t)p.kl * N * B * p.S * 4 + 255) & ~(size_t)255) : 0;
        p.flag_bytes = p.verify ? (((size_t)N * B + 255) & ~(size_t)255) : 0;
        p.copy_bytes = p.verify ? (((size_t)M * D * B * 4 + 255) & ~(size_t)255) : 0;  // the interleaved copy of the candidate clouds
        p.total = p.pre_bytes + 2 * p.list_bytes + p.flag_bytes + p.copy_bytes;
        return p;
    }
};

fx3d_status launch_knn(const float *x, int N, const float *y, int M, int B, int D, int k, int drop,
                       int32_t *idx, float *dist, hipStream_t st, void *pre_ws = nullptr, int xdiv = 1) {
    ProfileScope prof("knn", st);
    const int kk = k + drop;
    const bool grid_y = knn_needs_select(M, D, kk) || (D == 3 ? !knn_f16_d3_shape_ok(M, kk) : !knn_mfma_eligible(M, D, kk));
    FX3D_REQUIRE(!grid_y || B <= 65535, "fx3d_knn: B=%d exceeds the grid's y range for this shape", B);
    if (knn_needs_select(M, D, kk)) {
        int nw = knn_select_waves(M);
        const int Mpad = (M + 255) / 256 * 256;
        const int lcap = knn_select_list(M, kk, &nw);
        const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_select_kernel), kSelMaxLds, "knn_select_kernel");
        if (arc != FX3D_OK) return arc;
        hipLaunchKernelGGL(knn_select_kernel, dim3((N + nw - 1) / nw, B), dim3(64 * nw), (size_t)nw * (Mpad + 2 * lcap) * 4, st, x, N, y, M,
                           B, D, k, drop, idx, dist, Mpad, lcap, nullptr, 0);
        FX3D_LAUNCH_CHECK();
        return FX3D_OK;
    }
    if (D == 3 && knn_f16_d3_shape_ok(M, kk))
        return launch_knn_f16_d3(x, N, y, M, B, k, drop, idx, dist, st, nullptr, 0, xdiv);
    FX3D_REQUIRE(xdiv == 1 || (D != 3 && !opt(OPT_KNN_NO_MFMA) && knn_mfma_eligible(M, D, kk)),
                 "fx3d_knn: internal: candidate slices on a kernel without them");
    if (D == 3) {
        const int qpb = (kWThreads / 64) * kWQ;
        hipLaunchKernelGGL(knn_wave_d3_kernel, dim3((N + qpb - 1) / qpb, B), dim3(kWThreads), 0, st, x, N, y, M, B, k,
                           drop, idx, dist);
    } else if (!opt(OPT_KNN_NO_MFMA) && knn_mfma_eligible(M, D, kk)) {
        return launch_knn_mfma(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
    } else {
        const int qpb = (kWThreads / 64) * kGQ;
        hipLaunchKernelGGL(knn_wave_generic_kernel, dim3((N + qpb - 1) / qpb, B), dim3(kWThreads), knn_wave_generic_lds(D),
                           st, x, N, y, M, B, D, k, drop, idx, dist);
    }
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

}  // namespace

extern "C" {

fx3d_status fx3d_knn(const float *x, int32_t N, const float *y, int32_t M, int32_t B, int32_t D,
                     int32_t k, int32_t drop_first, int32_t *idx, float *dist, fx3d_stream_t s) {
    FX3D_REQUIRE(x && y && idx, "fx3d_knn: null pointer");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0 && k > 0, "fx3d_knn: bad sizes (N=%d M=%d B=%d D=%d k=%d)",
                 N, M, B, D, k);
    const int drop = drop_first ? 1 : 0;
    const int kk = k + drop;
    FX3D_REQUIRE(kk <= M, "fx3d_knn: k+drop_first=%d exceeds the number of candidates M=%d", kk, M);
    if (knn_needs_select(M, D, kk) && knn_select_waves(M) < 1) {
        set_error("fx3d_knn: k+drop_first=%d > 64 (or D=%d beyond the wave kernel) is supported for M <= %d candidates, got M=%d", kk, D,
                  kSelMaxLds / 4, M);
        return FX3D_ERR_UNSUPPORTED;
    }
    return launch_knn(x, N, y, M, B, D, k, drop, idx, dist, as_stream(s));
}

fx3d_status fx3d_knn_workspace_bytes(int32_t N, int32_t M, int32_t B, int32_t D, int32_t k, int32_t drop_first, size_t *bytes) {
    FX3D_REQUIRE(bytes, "fx3d_knn_workspace_bytes: null output");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0 && k > 0, "fx3d_knn_workspace_bytes: bad sizes");
    const int kk = k + (drop_first ? 1 : 0);
    // (alignment of x / y is checked at the call: an ineligible call simply does not use the workspace)
    *bytes = kk <= M ? KnnScratch::plan(N, M, B, D, kk).total : 0;
    return FX3D_OK;
}

fx3d_status fx3d_knn_ws(const float *x, int32_t N, const float *y, int32_t M, int32_t B, int32_t D, int32_t k,
                        int32_t drop_first, int32_t *idx, float *dist, void *ws, size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(x && y && idx, "fx3d_knn_ws: null pointer");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0 && k > 0, "fx3d_knn_ws: bad sizes (N=%d M=%d B=%d D=%d k=%d)", N, M, B, D, k);
    const int drop = drop_first ? 1 : 0;
    const int kk = k + drop;
    FX3D_REQUIRE(kk <= M, "fx3d_knn_ws: k+drop_first=%d exceeds the number of candidates M=%d", kk, M);
    const bool ws_ok = ws && (reinterpret_cast<uintptr_t>(ws) & 255) == 0;
    const KnnScratch p = KnnScratch::plan(N, M, B, D, kk);
    if (p.S > 1 && ws_ok && ws_bytes >= p.total) {
        // candidate slices: the search on B x S virtual clouds of M / S rows (no drop: the merge drops), then the merge
        const int Ms = M / p.S;
        unsigned char *w8 = static_cast<unsigned char *>(ws);
        int32_t *widx = reinterpret_cast<int32_t *>(w8 + p.pre_bytes);
        float *wdist = reinterpret_cast<float *>(w8 + p.pre_bytes + p.list_bytes);
        unsigned char *flags = p.verify ? w8 + p.pre_bytes + 2 * p.list_bytes : nullptr;
        const float *ys = y;  // the clouds the slices are cut from
        if (p.verify) {       // 32 per slice must hold the cloud's kk nearest: interleaved slices (samples of the whole cloud)
            float *yc = reinterpret_cast<float *>(w8 + p.pre_bytes + 2 * p.list_bytes + p.flag_bytes);
            const int vec4 = D % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
            const long long items = (long long)B * M * (vec4 ? D / 4 : D);
            const long long nb = (items + 255) / 256;
            hipLaunchKernelGGL(knn_interleave_kernel, dim3((unsigned int)(nb < 8192 ? nb : 8192)), dim3(256), 0, as_stream(s), y, M, D, B, p.S, yc, vec4);
            FX3D_LAUNCH_CHECK();
            ys = yc;
        }
        void *pre_ws = p.pre_bytes && knn_pre_eligible(x, ys, Ms, D, p.kl) ? ws : nullptr;
        const fx3d_status rc = launch_knn(x, N, ys, Ms, B * p.S, D, p.kl, 0, widx, wdist, as_stream(s), pre_ws, p.S);
        if (rc != FX3D_OK) return rc;
        const int nent = p.S * p.kl;                                     // entries per query (<= 512)
        const int wpq = nent <= 64 ? 1 : (nent <= 128 ? 2 : 4), qpb = 4 / wpq;
        FX3D_REQUIRE(B <= 65535, "fx3d_knn_ws: B=%d exceeds the grid's y range for this shape", B);
        hipLaunchKernelGGL(knn_merge_slices_kernel, dim3((unsigned int)((N + qpb - 1) / qpb), (unsigned int)B), dim3(256), (size_t)qpb * nent * 8,
                           as_stream(s), widx, wdist, N, B, p.S, Ms, p.kl, kk, k, drop, idx, dist, flags, p.verify ? 1 : 0, wpq);
        FX3D_LAUNCH_CHECK();
        if (p.verify) {  // the flagged queries (a slice held more than 32 of their kk nearest) again, on all M candidates
            int nw = knn_select_waves(M);
            const int Mpad = (M + 255) / 256 * 256;
            const int lcap = knn_select_list(M, kk, &nw);
            const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_select_kernel), kSelMaxLds, "knn_select_kernel");
            if (arc != FX3D_OK) return arc;
            FX3D_REQUIRE(B <= 65535, "fx3d_knn_ws: B=%d exceeds the grid's y range for this shape", B);
            const int regions = nw >= 4 ? 4 : 1;  // key arrays in LDS: one per wave when they fit (many flagged queries), else one
            hipLaunchKernelGGL(knn_select_kernel, dim3((N + 31) / 32, B), dim3(256), (size_t)regions * (Mpad + 2 * lcap) * 4, as_stream(s), x, N,
                               y, M, B, D, k, drop, idx, dist, Mpad, lcap, flags, regions);
            FX3D_LAUNCH_CHECK();
        }
        return FX3D_OK;
    }
    const bool pre = ws_ok && knn_pre_eligible(x, y, M, D, kk) && ws_bytes >= knn_pre_bytes(M, B, D);
    if (!pre) return fx3d_knn(x, N, y, M, B, D, k, drop_first, idx, dist, s);
    return launch_knn(x, N, y, M, B, D, k, drop, idx, dist, as_stream(s), ws);
}

fx3d_status fx3d_knn_gather(const float *x, int32_t N, int32_t B, int32_t F, int32_t k,
                            const int32_t *idx, float *out, fx3d_stream_t s) {
    FX3D_REQUIRE(x && idx && out, "fx3d_knn_gather: null pointer");
    FX3D_REQUIRE(N > 0 && B > 0 && F > 0 && k > 0, "fx3d_knn_gather: bad sizes");
    if (F % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {  // 16-byte copies
        const long long total4 = (long long)B * N * k * (F / 4);
        long long g4 = (total4 + kThreads - 1) / kThreads;
        if (g4 > 16384) g4 = 16384;
        ProfileScope prof4("knn_gather", as_stream(s));
        if ((size_t)F * k * N * B * 4 > kStreamingStoreBytes)
            hipLaunchKernelGGL(knn_gather4_kernel<true>, dim3((unsigned)g4), dim3(kThreads), 0, as_stream(s), x, N, B, F / 4, k, idx, out);
        else
            hipLaunchKernelGGL(knn_gather4_kernel<false>, dim3((unsigned)g4), dim3(kThreads), 0, as_stream(s), x, N, B, F / 4, k, idx, out);
        FX3D_LAUNCH_CHECK();
        return FX3D_OK;
    }
    const long long total = (long long)B * N * k * F;
    long long g = (total + kThreads - 1) / kThreads;
    if (g > 8192) g = 8192;
    ProfileScope prof("knn_gather", as_stream(s));
    hipLaunchKernelGGL(knn_gather_kernel, dim3((unsigned)g), dim3(kThreads), 0, as_stream(s), x, N, B, F, k, idx, out);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}


fx3d_status fx3d_edge_features(const float *x, int32_t N, int32_t B, int32_t F, int32_t k, const int32_t *idx,
                               int32_t layout, float *out, fx3d_stream_t s) {
    FX3D_REQUIRE(x && idx && out, "fx3d_edge_features: null pointer");
    FX3D_REQUIRE(N > 0 && B > 0 && F > 0 && k > 0, "fx3d_edge_features: bad sizes");
    FX3D_REQUIRE(layout == 0 || layout == 1, "fx3d_edge_features: layout must be 0 (2F,K,N,B) or 1 (K*N,2F,B)");
    ProfileScope prof("edge_features", as_stream(s));
    if (layout == 0) {
        const long long total = (long long)B * N * k * 2 * F;
        long long g = (total + kThreads - 1) / kThreads;
        if (g > 16384) g = 16384;
        hipLaunchKernelGGL(edge_features_cat_kernel, dim3((unsigned)g), dim3(kThreads), 0, as_stream(s), x, N, B, F, k,
                           idx, out);
    } else {
        const long long KN = (long long)k * N;
        dim3 grid((unsigned)((KN + kThreads - 1) / kThreads), B);
        const bool al16 = (((uintptr_t)x | (uintptr_t)out | (uintptr_t)idx) & 15) == 0;
        if (F % 4 == 0 && KN % 4 == 0 && al16) {
            // the feature loop split over blockIdx.z until the grid holds ~16 blocks per CU
            const long long gx = (KN / 4 + kThreads - 1) / kThreads;
            int fper = F;
            while (fper > 4 && gx * B * ((F + fper - 1) / fper) < 16ll * device_cus()) fper = (fper / 2 + 3) / 4 * 4;
            const unsigned gz = (unsigned)((F + fper - 1) / fper);
            const bool nt = (size_t)2 * F * KN * B * 4 > kStreamingStoreBytes;  // (smaller tensors may be read back from the caches)
            if (nt)
                hipLaunchKernelGGL(edge_features_mlp4_kernel<true>, dim3((unsigned)gx, B, gz), dim3(kThreads), 0, as_stream(s), x, N, B, F, k, idx, out, fper);
            else
                hipLaunchKernelGGL(edge_features_mlp4_kernel<false>, dim3((unsigned)gx, B, gz), dim3(kThreads), 0, as_stream(s), x, N, B, F, k, idx, out, fper);
        }
        else if (F % 4 == 0 && ((uintptr_t)x & 15) == 0)
            hipLaunchKernelGGL(edge_features_mlp_kernel<true>, grid, dim3(kThreads), 0, as_stream(s), x, N, B, F, k, idx, out);
        else
            hipLaunchKernelGGL(edge_features_mlp_kernel<false>, grid, dim3(kThreads), 0, as_stream(s), x, N, B, F, k, idx, out);
    }
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_edge_features_bwd(const float *gout, int32_t N, int32_t B, int32_t F, int32_t k, int32_t layout,
                                   float *gx, fx3d_stream_t s) {
    FX3D_REQUIRE(gout && gx, "fx3d_edge_features_bwd: null pointer");
    FX3D_REQUIRE(N > 0 && B > 0 && F > 0 && k > 0, "fx3d_edge_features_bwd: bad sizes");
    FX3D_REQUIRE(layout == 0 || layout == 1, "fx3d_edge_features_bwd: bad layout");
    const long long total = (long long)B * N * F;
    long long g = (total + kThreads - 1) / kThreads;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(edge_features_bwd_kernel, dim3((unsigned)g), dim3(kThreads), 0, as_stream(s), gout, N, B, F, k,
                       layout, gx);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_edgeconv_graph(const float *x, int32_t N, int32_t B, int32_t F, int32_t k, int32_t layout,
                                int32_t *idx, float *out, fx3d_stream_t s) {
    FX3D_REQUIRE(idx, "fx3d_edgeconv_graph: idx (k,N,B) is required (it is also the adjoint's side input)");
    FX3D_REQUIRE(x && out && N > 0 && B > 0 && F > 0 && k > 0, "fx3d_edgeconv_graph: bad argument");
    FX3D_REQUIRE(layout == 0 || layout == 1, "fx3d_edgeconv_graph: layout must be 0 (2F,K,N,B) or 1 (K*N,2F,B)");
    if (F == 3 && k + 1 <= N && knn_f16_d3_shape_ok(N, k + 1) &&
        !opt(OPT_EDGECONV_UNFUSED)) {
        // first EdgeConv (coordinates): neighbour search and features in ONE kernel
        ProfileScope prof("edgeconv_graph", as_stream(s));
        return launch_knn_f16_d3(x, N, x, N, B, k, 1, idx, nullptr, as_stream(s), out, layout);
    }
    fx3d_status rc = fx3d_knn(x, N, x, N, B, F, k, 1, idx, nullptr, s);
    if (rc != FX3D_OK) return rc;
    return fx3d_edge_features(x, N, B, F, k, idx, layout, out, s);
}

}  // extern "C"
