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
//   knn_exact_bruteforce / knn_rank_ties4          wave-cooperative exact selection / tie re-rank shared by all kernels
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
// (round 5: one translation unit per kernel family -- knn_d3.hip, knn_mfma.hip; the shared helpers in knn_common.h)
#include "knn_common.h"

namespace {

#ifndef FX3D_KNN_ONE_TU  // (tools/knn_probe.hip includes the three units into one: the names below are then the units' own)
// the other units' entry points under the names the dispatch below was written with
inline fx3d_status launch_knn_f16_d3(const float *x, int N, const float *y, int M, int B, int k, int drop, int32_t *idx, float *dist,
                                     hipStream_t st, float *feat = nullptr, int layout = 0, int xdiv = 1) {
    return knn_d3_launch(x, N, y, M, B, k, drop, idx, dist, st, feat, layout, xdiv);
}
inline fx3d_status launch_knn_mfma(const float *x, int N, const float *y, int M, int B, int D, int k, int drop, int32_t *idx, float *dist,
                                   hipStream_t st, void *pre_ws = nullptr, int xdiv = 1) {
    return knn_mfma_launch(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
}
inline bool knn_pre_shape_ok(int M, int D, int kk) { return knn_mfma_pre_shape_ok(M, D, kk); }
inline bool knn_pre_eligible(const float *x, const float *y, int M, int D, int kk) { return knn_mfma_pre_eligible(x, y, M, D, kk); }
inline size_t knn_pre_bytes(int M, int B, int D) { return knn_mfma_pre_bytes(M, B, D); }
#endif

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

size_t knn_wave_generic_lds(int D) {
    return (size_t)kGT * (D + 1) * 4 + (kWThreads / 64) * (kGQ * (64 * 16 + 8) + D * 16) + 16;
}

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
