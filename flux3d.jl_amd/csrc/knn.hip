// k-NN graph kernels (gfx950).
//
// Replaces CreateSingleKNNGraph (src/models/dgcnn.jl:3-7) and its per-batch-element loop in
// EdgeConv (:36): B KD-tree builds + N*B sorted (K+1)-queries + N*B small gathers become one
// brute-force launch (+ one gather launch).  Ordering is (distance, index) ascending with the
// CPU path's arithmetic (Float32 sum of squared differences in dimension order, unfused), so the
// index lists are bit-identical to oracle/flux3d_oracle.c:fx3d_oracle_knn.
//
// knn_d3_kernel<KMAX>: one thread per query, candidates broadcast from LDS (SoA, like chamfer.hip),
// the running top-KMAX list lives in registers (fully unrolled insertion network, guarded by one
// compare against the current worst).  KMAX in {8,16,32,64} >= k+drop_first.
// knn_generic_kernel<KMAX>: any D (the second EdgeConv runs in 64-D feature space,
// src/models/dgcnn.jl:121): the query lives in LDS transposed ([d][thread], conflict-free), the
// candidate row is read through the scalar/L1 path (same address for every lane).
#include <cmath>
#include <cstdlib>

#include "fx3d_common.h"

using namespace fx3d;

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 2048;  // candidates staged per LDS pass (D=3: 24 KiB)

template <int KMAX>
struct TopK {
    float d[KMAX];
    int j[KMAX];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int p = 0; p < KMAX; ++p) { d[p] = INFINITY; j[p] = 0x7fffffff; }
    }
    // insert (dist, idx) keeping (d, j) sorted ascending; equal distances keep arrival order
    // (candidates arrive in ascending index => ties resolve to the lower index first).
    __device__ __forceinline__ void insert(float dist, int idx) {
        if (dist < d[KMAX - 1]) {
            bool lt[KMAX];
#pragma unroll
            for (int p = 0; p < KMAX; ++p) lt[p] = dist < d[p];
#pragma unroll
            for (int p = KMAX - 1; p > 0; --p) {
                // lt[p-1] => shift slot p-1 up; else if lt[p] => land here
                d[p] = lt[p - 1] ? d[p - 1] : (lt[p] ? dist : d[p]);
                j[p] = lt[p - 1] ? j[p - 1] : (lt[p] ? idx : j[p]);
            }
            d[0] = lt[0] ? dist : d[0];
            j[0] = lt[0] ? idx : j[0];
        }
    }
};

template <int KMAX>
__global__ __launch_bounds__(kThreads) void knn_d3_kernel(const float *__restrict__ x, int N,
                                                          const float *__restrict__ y, int M, int B,
                                                          int k, int drop, int32_t *__restrict__ idx,
                                                          float *__restrict__ dist) {
    __shared__ __attribute__((aligned(16))) float lds[3 * kChunk];
    const int b = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const float *xb = x + (size_t)b * N * 3, *yb = y + (size_t)b * M * 3;
    const int ic = i < N ? i : N - 1;
    const float q0 = xb[3ll * ic], q1 = xb[3ll * ic + 1], q2 = xb[3ll * ic + 2];
    TopK<KMAX> top;
    top.init();
    for (int j0 = 0; j0 < M; j0 += kChunk) {
        const int cnt = (M - j0) < kChunk ? (M - j0) : kChunk;
        if (j0 > 0) __syncthreads();
        for (int e = threadIdx.x; e < cnt * 3; e += kThreads) {
            const float v = yb[(size_t)j0 * 3 + e];
            const int pt = e / 3, cc = e - pt * 3;
            lds[cc * kChunk + pt] = v;
        }
        __syncthreads();
        for (int jj = 0; jj < cnt; ++jj) {
            const float t0 = q0 - lds[jj], t1 = q1 - lds[kChunk + jj], t2 = q2 - lds[2 * kChunk + jj];
            const float dd = ((t0 * t0) + (t1 * t1)) + (t2 * t2);
            top.insert(dd, j0 + jj);
        }
    }
    if (i < N) {
        int32_t *o = idx + ((size_t)b * N + i) * k;
        float *od = dist ? dist + ((size_t)b * N + i) * k : nullptr;
#pragma unroll
        for (int p = 0; p < KMAX; ++p) {
            const int r = p - drop;
            if (r >= 0 && r < k) {
                o[r] = top.j[p];
                if (od) od[r] = top.d[p];
            }
        }
    }
}

template <int KMAX>
__global__ __launch_bounds__(kThreads) void knn_generic_kernel(const float *__restrict__ x, int N,
                                                               const float *__restrict__ y, int M,
                                                               int B, int D, int k, int drop,
                                                               int32_t *__restrict__ idx,
                                                               float *__restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [D][kThreads]
    const int b = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
    const int ic = i < N ? i : N - 1;
    for (int d = 0; d < D; ++d) qs[d * kThreads + threadIdx.x] = xb[(size_t)ic * D + d];
    TopK<KMAX> top;
    top.init();
    for (int j = 0; j < M; ++j) {
        const float *c = yb + (size_t)j * D;  // wave-uniform address
        float s = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float t = qs[d * kThreads + threadIdx.x] - c[d];
            s = s + t * t;
        }
        top.insert(s, j);
    }
    if (i < N) {
        int32_t *o = idx + ((size_t)b * N + i) * k;
        float *od = dist ? dist + ((size_t)b * N + i) * k : nullptr;
#pragma unroll
        for (int p = 0; p < KMAX; ++p) {
            const int r = p - drop;
            if (r >= 0 && r < k) {
                o[r] = top.j[p];
                if (od) od[r] = top.d[p];
            }
        }
    }
}

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

struct Key { float d; int j; };
__device__ __forceinline__ bool key_less(float d, int j, float od, int oj) { return d < od || (d == od && j < oj); }

// ascending bitonic sort of one (d, j) key per lane
__device__ __forceinline__ void bitonic64(float &d, int &j, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int s = k >> 1; s > 0; s >>= 1) {
            const float od = __shfl_xor(d, s, 64);
            const int oj = __shfl_xor(j, s, 64);
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
__device__ __forceinline__ void bitonic64f(float &v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int s = k >> 1; s > 0; s >>= 1) {
            const float o = __shfl_xor(v, s, 64);
            const bool up = (lane & k) == 0 || k == 64;
            const bool lower = (lane & s) == 0;
            v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
        }
    }
}

__global__ __launch_bounds__(kWThreads) void knn_wave_d3_kernel(const float *__restrict__ x, int N,
                                                                const float *__restrict__ y, int M, int B,
                                                                int k, int drop, int32_t *__restrict__ idx,
                                                                float *__restrict__ dist) {
    __shared__ float lst_d[kWThreads / 64][64];
    __shared__ int lst_j[kWThreads / 64][64];
    __shared__ float best_d[kWThreads / 64][kWQ][64];   // per-query best lists across chunks
    __shared__ int best_j[kWThreads / 64][kWQ][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kk = k + drop;
    const float *xb = x + (size_t)b * N * 3, *yb = y + (size_t)b * M * 3;
    const int q0 = (blockIdx.x * (kWThreads / 64) + wv) * kWQ;
    if (q0 >= N) return;  // wave-uniform; no block-level sync below
#pragma unroll
    for (int qq = 0; qq < kWQ; ++qq) { best_d[wv][qq][lane] = INFINITY; best_j[wv][qq][lane] = 0x7fffffff; }

    for (int j0 = 0; j0 < M; j0 += 1024) {
        float cx[16], cy[16], cz[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = j0 + lane + 64 * i;
            if (j < M) {
                cx[i] = yb[(size_t)j * 3]; cy[i] = yb[(size_t)j * 3 + 1]; cz[i] = yb[(size_t)j * 3 + 2];
            } else {
                cx[i] = INFINITY; cy[i] = INFINITY; cz[i] = INFINITY;  // d = +inf: never selected
            }
        }
#pragma unroll 1
        for (int qq = 0; qq < kWQ; ++qq) {
            const int qi = q0 + qq;
            if (qi >= N) break;
            const float qx = xb[(size_t)qi * 3], qy = xb[(size_t)qi * 3 + 1], qz = xb[(size_t)qi * 3 + 2];  // uniform
            float d[16];
            float lmin = INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float t0 = qx - cx[i], t1 = qy - cy[i], t2 = qz - cz[i];
                d[i] = ((t0 * t0) + (t1 * t1)) + (t2 * t2);
                lmin = fminf(lmin, d[i]);
            }
            float bd = best_d[wv][qq][lane];
            int bj = best_j[wv][qq][lane];
            float tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bd), kk - 1));
            if (j0 == 0) {  // kk-th smallest lane minimum bounds the kk-th smallest distance
                float v = lmin;
                bitonic64f(v, lane);
                tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), kk - 1));
            }
            int cnt = 0;
            const int cap = 64 - kk;  // list + best list must fit one key per lane
            if (cap > 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    bool pred = d[i] <= tau && d[i] < INFINITY;
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
                            float sd = lane < kk ? bd : (lane - kk < cnt ? lst_d[wv][lane - kk] : INFINITY);
                            int sj = lane < kk ? bj : (lane - kk < cnt ? lst_j[wv][lane - kk] : 0x7fffffff);
                            bitonic64(sd, sj, lane);
                            bd = lane < kk ? sd : INFINITY;
                            bj = lane < kk ? sj : 0x7fffffff;
                            tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sd), kk - 1));
                            cnt = 0;
                            pred = pred && d[i] <= tau;
                        }
                        bal = __ballot(pred);
                    }
                }
            }
            if (cnt > 0 || cap == 0) {
                float sd, sj_f;
                int sj;
                if (cap > 0) {
                    sd = lane < kk ? bd : (lane - kk < cnt ? lst_d[wv][lane - kk] : INFINITY);
                    sj = lane < kk ? bj : (lane - kk < cnt ? lst_j[wv][lane - kk] : 0x7fffffff);
                    bitonic64(sd, sj, lane);
                    bd = sd; bj = sj;
                } else {
                    // kk == 64: no room for a list; merge the chunk 64 candidates at a time
#pragma unroll 1
                    for (int i = 0; i < 16; ++i) {
                        float nd = d[i];
                        int nj = nd < INFINITY ? j0 + lane + 64 * i : 0x7fffffff;
                        bitonic64(nd, nj, lane);                      // ascending new batch
                        const float rd = __shfl(nd, 63 - lane, 64);   // reversed
                        const int rj = __shfl(nj, 63 - lane, 64);
                        const bool o_less = key_less(rd, rj, bd, bj);
                        bd = o_less ? rd : bd;                         // lower half of the union (bitonic)
                        bj = o_less ? rj : bj;
                        bitonic64(bd, bj, lane);
                    }
                }
                (void)sj_f;
            }
            best_d[wv][qq][lane] = bd;
            best_j[wv][qq][lane] = bj;
            if (j0 + 1024 >= M) {  // last chunk: lanes drop..drop+k-1 hold the answer
                const int r = lane - drop;
                if (r >= 0 && r < k) {
                    idx[((size_t)b * N + qi) * k + r] = bj;
                    if (dist) dist[((size_t)b * N + qi) * k + r] = bd;
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
    float *bestd = tile + kGT * RS;             // [NW][kGQ][64]
    int *bestj = reinterpret_cast<int *>(bestd + NW * kGQ * 64);
    float *lstd = reinterpret_cast<float *>(bestj + NW * kGQ * 64);
    int *lstj = reinterpret_cast<int *>(lstd + NW * kGQ * 64);
    float *taus = reinterpret_cast<float *>(lstj + NW * kGQ * 64);   // [NW][kGQ]
    int *cnts = reinterpret_cast<int *>(taus + NW * kGQ);            // [NW][kGQ]
    float4 *qs4 = reinterpret_cast<float4 *>(cnts + NW * kGQ);       // [NW][D] : the wave's kGQ=4 queries, interleaved

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kk = k + drop, cap = 64 - kk;
    const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
    const int q0 = (blockIdx.x * NW + wv) * kGQ;
#pragma unroll
    for (int qq = 0; qq < kGQ; ++qq) {
        bestd[(wv * kGQ + qq) * 64 + lane] = INFINITY;
        bestj[(wv * kGQ + qq) * 64 + lane] = 0x7fffffff;
        if (lane == 0) { taus[wv * kGQ + qq] = INFINITY; cnts[wv * kGQ + qq] = 0; }
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

    for (int j0 = 0; j0 < M; j0 += kGT) {
        const int cntc = (M - j0) < kGT ? (M - j0) : kGT;
        __syncthreads();
        for (int e = tid; e < cntc * D; e += kWThreads) {  // coalesced: the tile is contiguous in memory
            const int r = e / D, d = e - r * D;
            tile[r * RS + d] = yb[(size_t)j0 * D + e];
        }
        __syncthreads();
        if (q0 < N) {
            // distances of the wave's 4 queries to its 2 tile rows per lane, all dims
            float acc0[kGQ], acc1[kGQ];
#pragma unroll
            for (int qq = 0; qq < kGQ; ++qq) { acc0[qq] = 0.0f; acc1[qq] = 0.0f; }
            {
                const float *r0 = tile + lane * RS, *r1 = tile + (lane + 64) * RS;
                const float4 *qw = qs4 + wv * D;
#pragma unroll 4
                for (int d = 0; d < D; ++d) {
                    const float4 qd = qw[d];
                    const float c0 = r0[d], c1 = r1[d];
                    const float qv4[4] = {qd.x, qd.y, qd.z, qd.w};
#pragma unroll
                    for (int qq = 0; qq < kGQ; ++qq) {
                        const float t0 = qv4[qq] - c0, t1 = qv4[qq] - c1;
                        acc0[qq] = acc0[qq] + t0 * t0;
                        acc1[qq] = acc1[qq] + t1 * t1;
                    }
                }
            }
#pragma unroll
            for (int qq = 0; qq < kGQ; ++qq) {
                const int qi = q0 + qq;
                if (qi >= N) break;
                float d0 = acc0[qq], d1 = acc1[qq];
                if (lane >= cntc) d0 = INFINITY;
                if (lane + 64 >= cntc) d1 = INFINITY;
                const int sidx = (wv * kGQ + qq) * 64;
                float tau = taus[wv * kGQ + qq];
                int cnt = cnts[wv * kGQ + qq];
                float bd = bestd[sidx + lane];
                int bj = bestj[sidx + lane];
                if (j0 == 0) {
                    float v = fminf(d0, d1);
                    bitonic64f(v, lane);
                    tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), kk - 1));
                }
                bool dirty = false;
                if (cap > 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float di = i ? d1 : d0;
                        bool pred = di <= tau && di < INFINITY;
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
                                float sd = lane < kk ? bd : (lane - kk < cnt ? lstd[sidx + lane - kk] : INFINITY);
                                int sj = lane < kk ? bj : (lane - kk < cnt ? lstj[sidx + lane - kk] : 0x7fffffff);
                                bitonic64(sd, sj, lane);
                                bd = sd; bj = sj;
                                tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sd), kk - 1));
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
                        float nd = i ? d1 : d0;
                        int nj = nd < INFINITY ? j0 + lane + 64 * i : 0x7fffffff;
                        bitonic64(nd, nj, lane);
                        const float rd = __shfl(nd, 63 - lane, 64);
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
                    float sd = lane < kk ? bd : (lane - kk < cnt ? lstd[sidx + lane - kk] : INFINITY);
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
                        if (dist) dist[((size_t)b * N + qi) * k + r] = bd;
                    }
                }
            }
        }
    }
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

template <int KMAX>
fx3d_status launch_knn(const float *x, int N, const float *y, int M, int B, int D, int k, int drop,
                       int32_t *idx, float *dist, hipStream_t st) {
    dim3 grid((N + kThreads - 1) / kThreads, B);
    ProfileScope prof("knn", st);
    static const bool legacy = [] { const char *e = getenv("FX3D_KNN_LEGACY"); return e && atoi(e); }();
    if (D == 3 && !legacy) {
        const int qpb = (kWThreads / 64) * kWQ;
        hipLaunchKernelGGL(knn_wave_d3_kernel, dim3((N + qpb - 1) / qpb, B), dim3(kWThreads), 0, st, x, N, y, M, B, k,
                           drop, idx, dist);
    } else if (D == 3) {
        hipLaunchKernelGGL(knn_d3_kernel<KMAX>, grid, dim3(kThreads), 0, st, x, N, y, M, B, k, drop, idx, dist);
    } else if (!legacy && (size_t)kGT * (D + 1) * 4 + (kWThreads / 64) * (kGQ * (64 * 16 + 8) + D * 16) + 16 <= 64 * 1024) {
        const size_t lds = (size_t)kGT * (D + 1) * 4 + (kWThreads / 64) * (kGQ * (64 * 16 + 8) + D * 16) + 16;
        const int qpb = (kWThreads / 64) * kGQ;
        hipLaunchKernelGGL(knn_wave_generic_kernel, dim3((N + qpb - 1) / qpb, B), dim3(kWThreads), lds, st, x, N, y, M,
                           B, D, k, drop, idx, dist);
    } else {
        const size_t lds = sizeof(float) * (size_t)D * kThreads;
        hipLaunchKernelGGL(knn_generic_kernel<KMAX>, grid, dim3(kThreads), lds, st, x, N, y, M, B, D, k,
                           drop, idx, dist);
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
    if (kk > 64) {
        set_error("fx3d_knn: k+drop_first=%d > 64 is not supported", kk);
        return FX3D_ERR_UNSUPPORTED;
    }
    if (D != 3 && (size_t)D * kThreads * sizeof(float) > 64 * 1024) {
        set_error("fx3d_knn: D=%d > 64 is not supported", D);
        return FX3D_ERR_UNSUPPORTED;
    }
    hipStream_t st = as_stream(s);
    if (kk <= 8) return launch_knn<8>(x, N, y, M, B, D, k, drop, idx, dist, st);
    if (kk <= 16) return launch_knn<16>(x, N, y, M, B, D, k, drop, idx, dist, st);
    if (kk <= 32) return launch_knn<32>(x, N, y, M, B, D, k, drop, idx, dist, st);
    return launch_knn<64>(x, N, y, M, B, D, k, drop, idx, dist, st);
}

fx3d_status fx3d_knn_gather(const float *x, int32_t N, int32_t B, int32_t F, int32_t k,
                            const int32_t *idx, float *out, fx3d_stream_t s) {
    FX3D_REQUIRE(x && idx && out, "fx3d_knn_gather: null pointer");
    FX3D_REQUIRE(N > 0 && B > 0 && F > 0 && k > 0, "fx3d_knn_gather: bad sizes");
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
        if (F % 4 == 0 && ((uintptr_t)x & 15) == 0)
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
    fx3d_status rc = fx3d_knn(x, N, x, N, B, F, k, 1, idx, nullptr, s);
    if (rc != FX3D_OK) return rc;
    return fx3d_edge_features(x, N, B, F, k, idx, layout, out, s);
}

}  // extern "C"
