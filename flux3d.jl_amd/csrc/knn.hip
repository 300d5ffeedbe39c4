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

template <int KMAX>
fx3d_status launch_knn(const float *x, int N, const float *y, int M, int B, int D, int k, int drop,
                       int32_t *idx, float *dist, hipStream_t st) {
    dim3 grid((N + kThreads - 1) / kThreads, B);
    ProfileScope prof("knn", st);
    if (D == 3) {
        hipLaunchKernelGGL(knn_d3_kernel<KMAX>, grid, dim3(kThreads), 0, st, x, N, y, M, B, k, drop, idx, dist);
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

}  // extern "C"
