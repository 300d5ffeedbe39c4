/*
 * flux3d_oracle.c -- CPU ORACLE for the Flux3D.jl geometric-metric hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (flux3d.jl_amd/ +
 * libflux3d_hip.so) never calls into this file and has no CPU fallback.
 *
 * It is a plain-C restatement of the reference's *algorithms* (the reference is Julia and
 * cannot be executed in this image, see DESIGN.md "Oracle").  Every function cites the
 * reference file:line whose semantics it follows.  Build with -ffp-contract=off so that no
 * multiply-add is fused: the reference's CPU arithmetic is Julia Float32 without muladd.
 *
 * Pinning (tests/test_oracle_golden.py): reference known answers -- face areas
 * (test/rep.jl:259-260), teapot laplacian_loss 0.05888283f0 (README.md:111-112), the 3-mesh
 * batch dense-Laplacian identity (test/metrics.jl:8-73), edge list identity
 * (test/rep.jl:135-156), chamfer vs dense naive formula (test/metrics.jl:94-111), sphere
 * radius property of sample_points (test/transforms/mesh_func.jl:4-14).
 * Third-party arithmetic absent from /root/reference: NearestNeighbors.jl (compat 0.4,
 * unpinned) `knn(KDTree(y), x, 1)`, Euclidean metric: sum over dims of (a-b)^2 in Float32 in
 * dimension order; tie order unspecified there, first (lowest) index here.
 *
 * Layouts are the reference's (Julia column-major): points (D,N,B) => x[(b*N+i)*D+d].
 * All indices crossing this API are 0-based int32/int64 unless stated.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FX_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Squared Euclidean distance exactly as Distances.jl's Euclidean pre-metric evaluates it for
 * Float32 data: s = 0; for d in 1:D  s += (a[d]-b[d])^2   (no FMA, dimension order).
 * Reference call sites: src/metrics/pcloud.jl:57,64 ; src/models/dgcnn.jl:5-6.
 * ---------------------------------------------------------------------------------------- */
static inline float sqdist(const float *a, const float *b, int D) {
    float s = 0.0f;
    for (int d = 0; d < D; ++d) {
        float t = a[d] - b[d];
        s = s + t * t;
    }
    return s;
}

/* Order of distances: Julia's `isless` total order on Float32 -- ascending, every NaN after +Inf, all NaNs
 * equal -- then the lower index.  For finite data this is the plain `<` scan with the first minimum winning.
 * Non-finite data is outside anything the reference pins (NearestNeighbors' `dist <= best` skips NaN candidates and
 * leaves its -1 sentinel when none qualifies; the CuArray method's argmin, src/metrics/pcloud.jl:80-81, puts NaN first):
 * this order keeps every index valid, skips NaN candidates while a comparable one exists and lets a NaN query
 * propagate NaN into the loss, as the reference's mean would. */
static inline int fless(float a, float b) { return (a < b) || (b != b && a == a); }

/* 1-NN of every x point in y (same batch element), brute force, first minimum wins.
 * Semantics of `knn(KDTree(y[:,:,b]), x[:,:,b], 1)[1]`, src/metrics/pcloud.jl:54-61. */
static void nn1_dir(const float *x, int N, const float *y, int M, int D, int32_t *idx,
                    float *dmin) {
    for (int i = 0; i < N; ++i) {
        const float *a = x + (size_t)i * D;
        float best = sqdist(a, y, D);
        int32_t bi = 0;
        for (int j = 1; j < M; ++j) {
            float d = sqdist(a, y + (size_t)j * D, D);
            if (fless(d, best)) { best = d; bi = j; }
        }
        idx[i] = bi;
        if (dmin) dmin[i] = best;
    }
}

/* _nearest_neighbors(x::Array{Float32,3}, y::Array{Float32,3}), src/metrics/pcloud.jl:54-70.
 * idx_x[b*N+i] = 0-based index into y of the NN of x_i ; idx_y[b*M+j] likewise into x. */
FX_API int fx3d_oracle_nn1(const float *x, int N, const float *y, int M, int B, int D,
                           int32_t *idx_x, int32_t *idx_y, float *dmin_x, float *dmin_y) {
    if (N <= 0 || M <= 0 || B <= 0 || D <= 0) return -1;
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
        nn1_dir(xb, N, yb, M, D, idx_x + (size_t)b * N, dmin_x ? dmin_x + (size_t)b * N : NULL);
        nn1_dir(yb, M, xb, N, D, idx_y + (size_t)b * M, dmin_y ? dmin_y + (size_t)b * M : NULL);
    }
    return 0;
}

/* _chamfer_distance(A,B,w1,w2), src/metrics/pcloud.jl:39-52:
 *   dist_A_to_B = mean((A .- B[:, nn_for_A]).^2) * 3.0f0   (mean over D*N*B elements)
 *   dist_B_to_A = mean((B .- A[:, nn_for_B]).^2) * 3.0f0
 *   w1*dist_A_to_B + w2*dist_B_to_A
 * Element squares are Float32; the mean is accumulated here in double, where the reference's `mean` is a
 * Float32 pairwise sum (Base.mean -> sum / n): the two differ by O(log2(n) * 2^-24) relative, inside the 1e-5 of
 * north_star; fx3d_chamfer_finalize (csrc/chamfer.hip) follows THIS definition (double sums, one rounding). */
FX_API int fx3d_oracle_chamfer_fwd(const float *x, int N, const float *y, int M, int B, int D,
                                   float w1, float w2, float *loss, int32_t *idx_x,
                                   int32_t *idx_y, double *sums /* [2] optional */) {
    if (N <= 0 || M <= 0 || B <= 0 || D <= 0) return -1;
    int32_t *ix = idx_x ? idx_x : (int32_t *)malloc(sizeof(int32_t) * (size_t)N * B);
    int32_t *iy = idx_y ? idx_y : (int32_t *)malloc(sizeof(int32_t) * (size_t)M * B);
    fx3d_oracle_nn1(x, N, y, M, B, D, ix, iy, NULL, NULL);
    double sa = 0.0, sb = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
        for (int i = 0; i < N; ++i) {
            const float *p = xb + (size_t)i * D, *q = yb + (size_t)ix[(size_t)b * N + i] * D;
            for (int d = 0; d < D; ++d) { float t = p[d] - q[d]; sa += (double)(t * t); }
        }
        for (int j = 0; j < M; ++j) {
            const float *p = yb + (size_t)j * D, *q = xb + (size_t)iy[(size_t)b * M + j] * D;
            for (int d = 0; d < D; ++d) { float t = p[d] - q[d]; sb += (double)(t * t); }
        }
    }
    float dA = (float)(sa / ((double)D * N * B)) * 3.0f;
    float dB = (float)(sb / ((double)D * M * B)) * 3.0f;
    *loss = (w1 * dA) + (w2 * dB);
    if (sums) { sums[0] = sa; sums[1] = sb; }
    if (!idx_x) free(ix);
    if (!idx_y) free(iy);
    return 0;
}

/* The same loss in the REFERENCE's own arithmetic: `mean(T)` of the materialised Float32 array
 * T = (A .- B[:, nn_for_A]) .^ 2, size (D, N, B), column-major, is `sum(T) / length(T)` with Base's pairwise sum
 * (base/reduce.jl mapreduce_impl, pairwise_blocksize = 1024): a range [ifirst, ilast] with ilast - ifirst < 1024 is summed
 * left to right starting from T[ifirst] + T[ifirst + 1], a longer one is split at imid = ifirst + (ilast - ifirst) >> 1 and
 * its halves are added.  Everything in Float32: v / Float32(length) * 3.0f0, then w1 * dA + w2 * dB.
 * Caveat (stated, not hidden): the leaf loop of Base carries @simd, so a Julia build may re-associate a leaf's sum into as
 * many partial sums as its target's vector width x unroll factor; this restatement is the un-vectorised definition, which is
 * what `julia -O0` / a scalar target computes.  The PRODUCT's default (fx3d_chamfer_fwd) stays the Float64 sum above -- one
 * rounding, order independent to 2^-53 --; fx3d_chamfer_loss_pairwise_f32 (csrc/chamfer.hip) reproduces THIS function bit
 * for bit from the indices. */
static float pairwise_sum_f32(const float *a, size_t ifirst, size_t ilast) { /* inclusive, 0-based */
    if (ifirst == ilast) return a[ifirst];
    if (ilast - ifirst < 1024) {
        float v = a[ifirst] + a[ifirst + 1];
        for (size_t i = ifirst + 2; i <= ilast; ++i) v = v + a[i];
        return v;
    }
    const size_t imid = ifirst + ((ilast - ifirst) >> 1);
    const float v1 = pairwise_sum_f32(a, ifirst, imid);
    const float v2 = pairwise_sum_f32(a, imid + 1, ilast);
    return v1 + v2;
}

FX_API int fx3d_oracle_chamfer_loss_pairwise(const float *x, int N, const float *y, int M, int B, int D,
                                             const int32_t *idx_x, const int32_t *idx_y, float w1, float w2, float *loss) {
    if (N <= 0 || M <= 0 || B <= 0 || D <= 0 || !idx_x || !idx_y) return -1;
    float mean2[2];
    for (int dir = 0; dir < 2; ++dir) {
        const float *a = dir ? y : x, *c = dir ? x : y;
        const int32_t *idx = dir ? idx_y : idx_x;
        const int R = dir ? M : N, S = dir ? N : M;
        const size_t len = (size_t)D * R * B;
        float *T = (float *)malloc(sizeof(float) * len);
        if (!T) return -2;
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < R; ++i)
                for (int d = 0; d < D; ++d) {
                    const float t = a[((size_t)b * R + i) * D + d] - c[((size_t)b * S + idx[(size_t)b * R + i]) * D + d];
                    T[((size_t)b * R + i) * D + d] = t * t;
                }
        mean2[dir] = pairwise_sum_f32(T, 0, len - 1) / (float)len;
        free(T);
    }
    *loss = (w1 * (mean2[0] * 3.0f)) + (w2 * (mean2[1] * 3.0f));
    return 0;
}

/* Zygote adjoint of src/metrics/pcloud.jl:47-48 with the indices held constant (@ignore,:45):
 *   gA = g*w1*(6/(D*N*B))*(A - B[nn_A])  -  scatter_add_{nn_B}( g*w2*(6/(D*M*B))*(B - A[nn_B]) )
 * and symmetrically for gB.   */
FX_API int fx3d_oracle_chamfer_bwd(const float *x, int N, const float *y, int M, int B, int D,
                                   const int32_t *idx_x, const int32_t *idx_y, float w1,
                                   float w2, float gout, float *gx, float *gy) {
    float ca = gout * w1 * (float)(6.0 / ((double)D * N * B));
    float cb = gout * w2 * (float)(6.0 / ((double)D * M * B));
    memset(gx, 0, sizeof(float) * (size_t)N * B * D);
    memset(gy, 0, sizeof(float) * (size_t)M * B * D);
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
        float *gxb = gx + (size_t)b * N * D, *gyb = gy + (size_t)b * M * D;
        for (int i = 0; i < N; ++i) {
            int j = idx_x[(size_t)b * N + i];
            for (int d = 0; d < D; ++d) {
                float t = ca * (xb[(size_t)i * D + d] - yb[(size_t)j * D + d]);
                gxb[(size_t)i * D + d] += t;
                gyb[(size_t)j * D + d] -= t;
            }
        }
        for (int j = 0; j < M; ++j) {
            int i = idx_y[(size_t)b * M + j];
            for (int d = 0; d < D; ++d) {
                float t = cb * (yb[(size_t)j * D + d] - xb[(size_t)i * D + d]);
                gyb[(size_t)j * D + d] += t;
                gxb[(size_t)i * D + d] -= t;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * KD-tree 1-NN: algorithmic twin of the reference CPU path (per-batch-element KDTree build +
 * N queries, serial), src/metrics/pcloud.jl:54-70.  NearestNeighbors.jl defaults: leafsize 10,
 * split on the widest dimension at the median.  Used (a) as the like-for-like cpu_baseline in
 * bench.py, (b) as an independent check that the brute-force result is what a tree returns.
 * Ties resolve to the lowest index so the answer is identical to nn1_dir (finite data only: the tree's
 * pruning test assumes comparable distances; non-finite inputs are defined by nn1_dir's order alone).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float *pts;
    int D;
    int32_t *perm;   /* point indices, permuted in place */
    int32_t *split_dim;
    float *split_val;
    int n, leafsize, nnodes_cap;
} kdt_t;

static int cmp_dim_D;
static const float *cmp_pts;
static int cmp_dim;
static int cmp_idx(const void *a, const void *b) {
    int32_t ia = *(const int32_t *)a, ib = *(const int32_t *)b;
    float fa = cmp_pts[(size_t)ia * cmp_dim_D + cmp_dim], fb = cmp_pts[(size_t)ib * cmp_dim_D + cmp_dim];
    if (fa < fb) return -1;
    if (fa > fb) return 1;
    return (ia > ib) - (ia < ib);
}

static void kdt_build_rec(kdt_t *t, int node, int lo, int hi) {
    if (hi - lo <= t->leafsize) {
        t->split_dim[node] = -1;
        return;
    }
    int D = t->D, best = 0;
    float bestw = -1.0f;
    for (int d = 0; d < D; ++d) {
        float mn = INFINITY, mx = -INFINITY;
        for (int k = lo; k < hi; ++k) {
            float v = t->pts[(size_t)t->perm[k] * D + d];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        if (mx - mn > bestw) { bestw = mx - mn; best = d; }
    }
    cmp_pts = t->pts; cmp_dim = best; cmp_dim_D = D;
    qsort(t->perm + lo, (size_t)(hi - lo), sizeof(int32_t), cmp_idx);
    int mid = (lo + hi) / 2;
    t->split_dim[node] = best;
    t->split_val[node] = t->pts[(size_t)t->perm[mid] * D + best];
    kdt_build_rec(t, 2 * node + 1, lo, mid);
    kdt_build_rec(t, 2 * node + 2, mid, hi);
}

static void kdt_query_rec(const kdt_t *t, int node, int lo, int hi, const float *q,
                          float *best, int32_t *bi) {
    int sd = t->split_dim[node];
    if (sd < 0) {
        for (int k = lo; k < hi; ++k) {
            int32_t j = t->perm[k];
            float d = sqdist(q, t->pts + (size_t)j * t->D, t->D);
            if (d < *best || (d == *best && j < *bi)) { *best = d; *bi = j; }
        }
        return;
    }
    int mid = (lo + hi) / 2;
    float diff = q[sd] - t->split_val[node];
    int near = diff < 0.0f ? 0 : 1;
    if (near == 0) kdt_query_rec(t, 2 * node + 1, lo, mid, q, best, bi);
    else kdt_query_rec(t, 2 * node + 2, mid, hi, q, best, bi);
    /* visit the far side unless the splitting plane is strictly farther than the best hit
     * (<= keeps equal-distance candidates so ties resolve exactly like brute force) */
    if (diff * diff <= *best) {
        if (near == 0) kdt_query_rec(t, 2 * node + 2, mid, hi, q, best, bi);
        else kdt_query_rec(t, 2 * node + 1, lo, mid, q, best, bi);
    }
}

static void kdt_nn1_dir(const float *x, int N, const float *y, int M, int D, int32_t *idx) {
    kdt_t t;
    t.pts = y; t.D = D; t.n = M; t.leafsize = 10;
    int cap = 1;
    while (cap < 4 * (M / t.leafsize + 2)) cap *= 2;
    t.nnodes_cap = cap;
    t.perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)M);
    t.split_dim = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    t.split_val = (float *)malloc(sizeof(float) * (size_t)cap);
    for (int j = 0; j < M; ++j) t.perm[j] = j;
    kdt_build_rec(&t, 0, 0, M);
    for (int i = 0; i < N; ++i) {
        float best = INFINITY;
        int32_t bi = 0x7fffffff;
        kdt_query_rec(&t, 0, 0, M, x + (size_t)i * D, &best, &bi);
        idx[i] = bi;
    }
    free(t.perm); free(t.split_dim); free(t.split_val);
}

FX_API int fx3d_oracle_nn1_kdtree(const float *x, int N, const float *y, int M, int B, int D,
                                  int32_t *idx_x, int32_t *idx_y) {
    if (N <= 0 || M <= 0 || B <= 0 || D <= 0) return -1;
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
        kdt_nn1_dir(xb, N, yb, M, D, idx_x + (size_t)b * N);
        kdt_nn1_dir(yb, M, xb, N, D, idx_y + (size_t)b * M);
    }
    return 0;
}

/* All-core variants for the cpu_baseline leg only ("brute-force fp32 on 1 core and on all
 * cores"): the 2*B independent (batch element, direction) searches are spread over OpenMP threads; every
 * search is the serial code above, so the results are identical.  Returns the number of threads used. */
#include <omp.h>
FX_API int fx3d_oracle_nn1_allcores(const float *x, int N, const float *y, int M, int B, int D,
                                    int32_t *idx_x, int32_t *idx_y, int use_kdtree, int want_threads) {
    if (N <= 0 || M <= 0 || B <= 0 || D <= 0) return -1;
    int nthreads = 1;
    if (want_threads > 0) omp_set_num_threads(want_threads);
    if (use_kdtree) {  /* one task per (batch element, direction): tree build + its queries */
#pragma omp parallel
        {
#pragma omp single
            nthreads = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
            for (int t = 0; t < 2 * B; ++t) {
                const int b = t >> 1;
                const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
                if ((t & 1) == 0) kdt_nn1_dir(xb, N, yb, M, D, idx_x + (size_t)b * N);
                else kdt_nn1_dir(yb, M, xb, N, D, idx_y + (size_t)b * M);
            }
        }
        return nthreads;
    }
    /* brute force: tasks of 128 queries */
    const int QB = 128;
    const int nbx = (N + QB - 1) / QB, nby = (M + QB - 1) / QB;
    const long long ntask = (long long)B * (nbx + nby);
#pragma omp parallel
    {
#pragma omp single
        nthreads = omp_get_num_threads();
#pragma omp for schedule(dynamic, 4)
        for (long long t = 0; t < ntask; ++t) {
            const int b = (int)(t / (nbx + nby));
            const int r = (int)(t % (nbx + nby));
            const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
            if (r < nbx) {
                const int lo = r * QB, n = (N - lo) < QB ? (N - lo) : QB;
                nn1_dir(xb + (size_t)lo * D, n, yb, M, D, idx_x + (size_t)b * N + lo, NULL);
            } else {
                const int lo = (r - nbx) * QB, n = (M - lo) < QB ? (M - lo) : QB;
                nn1_dir(yb + (size_t)lo * D, n, xb, N, D, idx_y + (size_t)b * M + lo, NULL);
            }
        }
    }
    return nthreads;
}

/* Whole reference CPU forward (KD-tree NN + gather + mean), for the cpu_baseline timing leg. */
FX_API int fx3d_oracle_chamfer_fwd_kdtree(const float *x, int N, const float *y, int M, int B,
                                          int D, float w1, float w2, float *loss) {
    int32_t *ix = (int32_t *)malloc(sizeof(int32_t) * (size_t)N * B);
    int32_t *iy = (int32_t *)malloc(sizeof(int32_t) * (size_t)M * B);
    fx3d_oracle_nn1_kdtree(x, N, y, M, B, D, ix, iy);
    double sa = 0.0, sb = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
        for (int i = 0; i < N; ++i)
            sa += (double)sqdist(xb + (size_t)i * D, yb + (size_t)ix[(size_t)b * N + i] * D, D);
        for (int j = 0; j < M; ++j)
            sb += (double)sqdist(yb + (size_t)j * D, xb + (size_t)iy[(size_t)b * M + j] * D, D);
    }
    float dA = (float)(sa / ((double)D * N * B)) * 3.0f;
    float dB = (float)(sb / ((double)D * M * B)) * 3.0f;
    *loss = (w1 * dA) + (w2 * dB);
    free(ix); free(iy);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * k-NN graph: CreateSingleKNNGraph(X,K), src/models/dgcnn.jl:3-7:
 *   knn(kdtree, X[:,i], K+1, true)[1][2:K+1]   -- K+1 nearest sorted ascending, first dropped.
 * Here: sorted by (distance in the isless order above, index); drop_first removes rank 0 (the reference assumes it is
 * the query itself).  idx[(b*N+i)*k + r], dist likewise (squared distance, Float32).
 * ---------------------------------------------------------------------------------------- */
FX_API int fx3d_oracle_knn(const float *x, int N, const float *y, int M, int B, int D, int k,
                           int drop_first, int32_t *idx, float *dist) {
    int kk = k + (drop_first ? 1 : 0);
    if (N <= 0 || M <= 0 || B <= 0 || D <= 0 || k <= 0 || kk > M) return -1;
    float *bd = (float *)malloc(sizeof(float) * (size_t)kk);
    int32_t *bj = (int32_t *)malloc(sizeof(int32_t) * (size_t)kk);
    for (int b = 0; b < B; ++b) {
        const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
        for (int i = 0; i < N; ++i) {
            int cnt = 0;
            for (int j = 0; j < M; ++j) {
                float d = sqdist(xb + (size_t)i * D, yb + (size_t)j * D, D);
                if (cnt == kk && !fless(d, bd[kk - 1])) continue; /* ties keep the earlier index */
                int p = cnt < kk ? cnt : kk - 1;
                while (p > 0 && fless(d, bd[p - 1])) { bd[p] = bd[p - 1]; bj[p] = bj[p - 1]; --p; }
                bd[p] = d; bj[p] = j;
                if (cnt < kk) ++cnt;
            }
            int off = drop_first ? 1 : 0;
            for (int r = 0; r < k; ++r) {
                idx[((size_t)b * N + i) * k + r] = bj[r + off];
                if (dist) dist[((size_t)b * N + i) * k + r] = bd[r + off];
            }
        }
    }
    free(bd); free(bj);
    return 0;
}

/* Gather of neighbour features: X[:, idxs] -> (F,K,N) per batch element, cat -> (F,K,N,B),
 * src/models/dgcnn.jl:6,36.  out[(((b*N+i)*k + r)*F + f] */
FX_API int fx3d_oracle_knn_gather(const float *x, int N, int B, int F, int k,
                                  const int32_t *idx, float *out) {
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i)
            for (int r = 0; r < k; ++r) {
                int j = idx[((size_t)b * N + i) * k + r];
                memcpy(out + (((size_t)b * N + i) * k + r) * F, x + ((size_t)b * N + j) * F,
                       sizeof(float) * (size_t)F);
            }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * compute_faces_areas_packed, src/rep/mesh.jl:765-780, with _lg_cross (src/rep/utils.jl:4-21)
 * and _norm (:29):  area = ||(v2-v1) x (v3-v1)|| / 2.   verts (3,V) packed, faces (3,F) packed
 * global indices (0-based here).
 * ---------------------------------------------------------------------------------------- */
static inline float tri_area(const float *v1, const float *v2, const float *v3) {
    float a1 = v2[0] - v1[0], a2 = v2[1] - v1[1], a3 = v2[2] - v1[2];
    float b1 = v3[0] - v1[0], b2 = v3[1] - v1[1], b3 = v3[2] - v1[2];
    float c1 = (a2 * b3) - (a3 * b2);
    float c2 = (a3 * b1) - (a1 * b3);
    float c3 = (a1 * b2) - (a2 * b1);
    float s = ((c1 * c1) + (c2 * c2)) + (c3 * c3);
    return sqrtf(s) / 2.0f;
}

FX_API int fx3d_oracle_faces_areas_packed(const float *verts, int64_t V, const int64_t *faces,
                                          int64_t F, float *areas) {
    for (int64_t f = 0; f < F; ++f) {
        int64_t i1 = faces[3 * f], i2 = faces[3 * f + 1], i3 = faces[3 * f + 2];
        if (i1 < 0 || i2 < 0 || i3 < 0 || i1 >= V || i2 >= V || i3 >= V) return -2;
        areas[f] = tri_area(verts + 3 * i1, verts + 3 * i2, verts + 3 * i3);
    }
    return 0;
}

/* compute_faces_areas_padded, src/rep/mesh.jl:799-808: areas (1,Fmax,B), zero padded.
 * verts_padded (3,Vmax,B), faces_padded (3,Fmax,B) mesh-local 0-based (pad entries ignored). */
FX_API int fx3d_oracle_faces_areas_padded(const float *verts_padded, int Vmax,
                                          const int64_t *faces_padded, int Fmax,
                                          const int64_t *faces_len, int B, float *areas) {
    for (int b = 0; b < B; ++b) {
        const float *vb = verts_padded + (size_t)b * Vmax * 3;
        for (int f = 0; f < Fmax; ++f) {
            float a = 0.0f;
            if (f < faces_len[b]) {
                const int64_t *fc = faces_padded + ((size_t)b * Fmax + f) * 3;
                a = tri_area(vb + 3 * fc[0], vb + 3 * fc[1], vb + 3 * fc[2]);
            }
            areas[(size_t)b * Fmax + f] = a;
        }
    }
    return 0;
}

/* Float64 summation order shared with the device sampler: a radix-32 tree, every node summed left to right.
 *   level 0:  t0[c]   = (((0.0 + v[32c]) + v[32c+1]) + ... + v[32c+31])        (elements past n are absent)
 *   level l:  t_l[g]  = ((0.0 + t_{l-1}[32g]) + t_{l-1}[32g+1]) + ...           until one value is left: the total
 *   scan[f] = off0[c] + l[f],   l = running inclusive sum inside chunk c = f / 32,
 *   off_l[i] = off_{l+1}[i / 32] + e_l[i],  e_l[i] = the left-to-right sum of the entries of i's group before i
 *   (0.0 for the first), and off = e at the top level (one group).  Up to 1024 elements this is the plain "chunks of 32,
 *   then the chunk totals in order"; beyond, no chain is longer than 32 additions, so a cloud of any size is a few
 *   parallel passes on the device (round 2: the one-chain version made a 2 M-face mesh an 8 ms kernel).
 * The reference sums with Julia's pairwise `sum` (src/transforms/mesh_func.jl:35-37); any order
 * differs from it by O(1e-16) relative, far below what the Categorical draw can resolve, but the
 * oracle and the kernel must agree bit-for-bit, so the order is part of the specification. */
#define FX_SCAN_CHUNK 32
/* totals of one level: t[g] = left-to-right sum of v[32g .. 32g+31]; returns the number of totals */
static int level_totals(const double *v, int n, double *t) {
    int m = 0;
    for (int c0 = 0; c0 < n; c0 += FX_SCAN_CHUNK) {
        double s = 0.0;
        for (int k = c0; k < c0 + FX_SCAN_CHUNK && k < n; ++k) s += v[k];
        t[m++] = s;
    }
    return m;
}
static double blocked_total(const double *v, int n) {
    if (n <= 0) return 0.0;
    double *a = (double *)malloc(sizeof(double) * (size_t)((n + FX_SCAN_CHUNK - 1) / FX_SCAN_CHUNK));
    double *b2 = (double *)malloc(sizeof(double) * (size_t)((n + FX_SCAN_CHUNK - 1) / FX_SCAN_CHUNK));
    int m = level_totals(v, n, a);
    while (m > 1) {
        m = level_totals(a, m, b2);
        double *t = a; a = b2; b2 = t;
    }
    const double tot = a[0];
    free(a); free(b2);
    return tot;
}
/* exclusive offsets of the n entries of one level (recursive over the levels above it) */
static void level_offsets(const double *t, int n, double *off) {
    int ng = (n + FX_SCAN_CHUNK - 1) / FX_SCAN_CHUNK;
    double *goff = NULL;
    if (ng > 1) {
        double *u = (double *)malloc(sizeof(double) * (size_t)ng);
        goff = (double *)malloc(sizeof(double) * (size_t)ng);
        level_totals(t, n, u);
        level_offsets(u, ng, goff);
        free(u);
    }
    for (int g = 0; g < ng; ++g) {
        double e = 0.0;
        for (int i = g * FX_SCAN_CHUNK; i < (g + 1) * FX_SCAN_CHUNK && i < n; ++i) {
            off[i] = goff ? goff[g] + e : e;
            e += t[i];
        }
    }
    free(goff);
}
static void blocked_scan(const double *v, int n, double *out) {
    if (n <= 0) return;
    const int nch = (n + FX_SCAN_CHUNK - 1) / FX_SCAN_CHUNK;
    double *t0 = (double *)malloc(sizeof(double) * (size_t)nch), *off0 = (double *)malloc(sizeof(double) * (size_t)nch);
    level_totals(v, n, t0);
    level_offsets(t0, nch, off0);
    for (int c = 0; c < nch; ++c) {
        double l = 0.0;
        for (int k = c * FX_SCAN_CHUNK; k < (c + 1) * FX_SCAN_CHUNK && k < n; ++k) { l += v[k]; out[k] = off0[c] + l; }
    }
    free(t0); free(off0);
}

/* Face probabilities, src/transforms/mesh_func.jl:32-39 (Float64):
 *   p = area ./ max.(sum(area; dims=2), eps);  p[:, end, :] += max.(1 .- sum(p; dims=2), 0)
 * NB the fix-up lands on the last *padded* column Fmax-1; for a mesh with faces_len < Fmax it
 * is outside 1:_len and is dropped by the slice at :45 -- reproduced here. */
FX_API int fx3d_oracle_face_probs(const float *areas_padded, int Fmax, int B, double eps,
                                  double *probs) {
    double *a64 = (double *)malloc(sizeof(double) * (size_t)Fmax);
    for (int b = 0; b < B; ++b) {
        const float *a = areas_padded + (size_t)b * Fmax;
        double *p = probs + (size_t)b * Fmax;
        for (int f = 0; f < Fmax; ++f) a64[f] = (double)a[f];
        double s = blocked_total(a64, Fmax);
        double den = s > eps ? s : eps;
        for (int f = 0; f < Fmax; ++f) p[f] = a64[f] / den;
        double fix = 1.0 - blocked_total(p, Fmax);
        p[Fmax - 1] += fix > 0.0 ? fix : 0.0;
    }
    free(a64);
    return 0;
}

/* _sample_points + _rand_barycentric_coords, src/transforms/mesh_func.jl:60-82, with the random
 * draws made explicit: face (0-based mesh-local), r1, r2 in [0,1).
 *   u = sqrt(r1); w1 = 1-u; w2 = u*(1-r2); w3 = u*r2;  s = (w1*v1 + w2*v2) + w3*v3
 * out (3,n,B). */
FX_API int fx3d_oracle_sample_points_explicit(const float *verts_padded, int Vmax,
                                              const int64_t *faces_padded, int Fmax, int B,
                                              int n, const int32_t *face_idx, const float *r1,
                                              const float *r2, float *out) {
    for (int b = 0; b < B; ++b) {
        const float *vb = verts_padded + (size_t)b * Vmax * 3;
        for (int s = 0; s < n; ++s) {
            size_t k = (size_t)b * n + s;
            const int64_t *fc = faces_padded + ((size_t)b * Fmax + face_idx[k]) * 3;
            const float *v1 = vb + 3 * fc[0], *v2 = vb + 3 * fc[1], *v3 = vb + 3 * fc[2];
            float u = sqrtf(r1[k]), v = r2[k];
            float w1 = 1.0f - u, w2 = u * (1.0f - v), w3 = u * v;
            for (int d = 0; d < 3; ++d)
                out[k * 3 + d] = ((w1 * v1[d]) + (w2 * v2[d])) + (w3 * v3[d]);
        }
    }
    return 0;
}

/* Adjoint of the weighted gather above w.r.t. verts_padded for fixed draws (Zygote through
 * src/transforms/mesh_func.jl:64-73: samples = w1 .* v[:, f1] + w2 .* v[:, f2] + w3 .* v[:, f3]; the pullback of each
 * getindex accumulates w_t * gout over the draws that hit the vertex).  Zygote fixes no order for that accumulation; this
 * restatement fixes the one the device uses, so that the gradient is reproducible bit for bit:
 *   inner(f, t) = 0 + sum over the draws k of face f, k ascending, of  w_t(k) * gout[k]      (per face and corner)
 *   g[v] = base[v] + sum over the (face, corner) pairs holding v, ascending (f, t), of inner(f, t)
 * (Float32, unfused; base = the incoming gverts when accumulate, else 0; an undrawn face adds its +0).
 * faces_padded (3,Fmax,B) 0-based mesh-local, faces_len (B); gout (3,n,B); gverts (3,Vmax,B). */
FX_API int fx3d_oracle_sample_points_bwd(const int64_t *faces_padded, const int64_t *faces_len, int Vmax, int Fmax, int B,
                                         int n, const int32_t *face_idx, const float *r1, const float *r2,
                                         const float *gout, float *gverts, int accumulate) {
    float *inner = (float *)malloc(sizeof(float) * 9 * (size_t)Fmax);
    if (!inner) return -1;
    for (int b = 0; b < B; ++b) {
        for (size_t i = 0; i < 9 * (size_t)Fmax; ++i) inner[i] = 0.0f;
        for (int s = 0; s < n; ++s) {  /* draws in ascending order: every face's sums see its draws ascending */
            size_t k = (size_t)b * n + s;
            int f = face_idx[k];
            if (f < 0 || f >= Fmax) continue;
            float u = sqrtf(r1[k]), v = r2[k];
            float w[3] = {1.0f - u, u * (1.0f - v), u * v};
            for (int t = 0; t < 3; ++t)
                for (int d = 0; d < 3; ++d) inner[(size_t)f * 9 + t * 3 + d] = inner[(size_t)f * 9 + t * 3 + d] + w[t] * gout[k * 3 + d];
        }
        float *gb = gverts + (size_t)b * Vmax * 3;
        if (!accumulate) for (size_t i = 0; i < 3 * (size_t)Vmax; ++i) gb[i] = 0.0f;
        for (int64_t f = 0; f < faces_len[b]; ++f)  /* ascending (face, corner): the order every vertex sees its pairs in */
            for (int t = 0; t < 3; ++t) {
                int64_t v = faces_padded[((size_t)b * Fmax + f) * 3 + t];
                if (v < 0 || v >= Vmax) continue;
                for (int d = 0; d < 3; ++d) gb[3 * v + d] = gb[3 * v + d] + inner[(size_t)f * 9 + t * 3 + d];
            }
    }
    free(inner);
    return 0;
}

/* Philox4x32-10 (Salmon et al. 2011), the counter-based generator of the on-device sampler.
 * counter = (sample, mesh, stream, 0), key = (seed_lo, seed_hi). */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

FX_API void fx3d_oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                               uint32_t k1, uint32_t *out4) {
    uint32_t c[4] = {c0, c1, c2, c3};
    philox4x32_10(c, k0, k1);
    memcpy(out4, c, sizeof(c));
}

/* sample_points with the on-device draw scheme (what fx3d_sample_points does in seed mode):
 *   cdf[f] = blocked_scan (above) of the face probabilities, used over 1:faces_len (:45)
 *   (r0,r1,r2,r3) = philox(sample, mesh, 0, 0; seed)
 *   uf = ((r0<<32 | r1) >> 11) * 2^-53 ; face = first f with cdf[f] > uf*cdf[last] (clamped)
 *   r1 = (r2>>8)*2^-24 ; r2 = (r3>>8)*2^-24
 * Categorical(probvec) in the reference normalises nothing but requires sum==1 within tol; the
 * inverse-CDF draw against cdf[last] is distribution-identical to its alias sampler.
 * Also returns the drawn faces / uniforms when the pointers are non-NULL. */
FX_API int fx3d_oracle_sample_points_seeded(const float *verts_padded, int Vmax,
                                            const int64_t *faces_padded, int Fmax,
                                            const int64_t *faces_len, int B, int n, double eps,
                                            uint64_t seed, float *out, int32_t *face_out,
                                            float *r1_out, float *r2_out) {
    float *areas = (float *)malloc(sizeof(float) * (size_t)B * Fmax);
    double *probs = (double *)malloc(sizeof(double) * (size_t)B * Fmax);
    double *cdf = (double *)malloc(sizeof(double) * (size_t)Fmax);
    int32_t *fi = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * n);
    float *r1 = (float *)malloc(sizeof(float) * (size_t)B * n);
    float *r2 = (float *)malloc(sizeof(float) * (size_t)B * n);
    fx3d_oracle_faces_areas_padded(verts_padded, Vmax, faces_padded, Fmax, faces_len, B, areas);
    fx3d_oracle_face_probs(areas, Fmax, B, eps, probs);
    for (int b = 0; b < B; ++b) {
        int L = (int)faces_len[b];
        blocked_scan(probs + (size_t)b * Fmax, Fmax, cdf);
        for (int s = 0; s < n; ++s) {
            uint32_t c[4] = {(uint32_t)s, (uint32_t)b, 0u, 0u};
            philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            uint64_t bits = (((uint64_t)c[0] << 32) | c[1]) >> 11;
            double uf = (double)bits * (1.0 / 9007199254740992.0) * cdf[L - 1];
            int lo = 0, hi = L - 1; /* first f with cdf[f] > uf */
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (cdf[mid] > uf) hi = mid; else lo = mid + 1;
            }
            size_t k = (size_t)b * n + s;
            fi[k] = lo;
            r1[k] = (float)(c[2] >> 8) * (1.0f / 16777216.0f);
            r2[k] = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
        }
    }
    fx3d_oracle_sample_points_explicit(verts_padded, Vmax, faces_padded, Fmax, B, n, fi, r1, r2,
                                       out);
    if (face_out) memcpy(face_out, fi, sizeof(int32_t) * (size_t)B * n);
    if (r1_out) memcpy(r1_out, r1, sizeof(float) * (size_t)B * n);
    if (r2_out) memcpy(r2_out, r2, sizeof(float) * (size_t)B * n);
    free(areas); free(probs); free(cdf); free(fi); free(r1); free(r2);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * _compute_edges_packed, src/rep/mesh.jl:907-955.  faces (3,F) packed, 0-based global vertex
 * ids here (the reference is 1-based; the hash (V+1)*v0+v1 is order-isomorphic under the
 * shift).  edges (E,2) sorted lexicographically, unique; faces_to_edges (F,3) in the
 * reference's column order (e23, e31, e12), :946.  Returns E (or <0).
 * ---------------------------------------------------------------------------------------- */
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

FX_API int64_t fx3d_oracle_edges_packed(const int64_t *faces, int64_t F, int64_t V,
                                        int64_t *edges /* [3F*2] */,
                                        int64_t *faces_to_edges /* [F*3] or NULL */) {
    uint64_t Vh = (uint64_t)V + 1;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(3 * F));
    for (int64_t f = 0; f < F; ++f) {
        int64_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
        if (a < 0 || b < 0 || c < 0 || a >= V || b >= V || c >= V) { free(h); return -2; }
        int64_t p[3][2] = {{a, b}, {b, c}, {c, a}}; /* e12, e23, e31 (:914-916) */
        for (int e = 0; e < 3; ++e) {
            int64_t v0 = p[e][0] < p[e][1] ? p[e][0] : p[e][1];
            int64_t v1 = p[e][0] < p[e][1] ? p[e][1] : p[e][0];
            h[(size_t)e * F + f] = Vh * (uint64_t)(v0 + 1) + (uint64_t)(v1 + 1);
        }
    }
    uint64_t *s = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(3 * F));
    memcpy(s, h, sizeof(uint64_t) * (size_t)(3 * F));
    qsort(s, (size_t)(3 * F), sizeof(uint64_t), cmp_u64);
    int64_t E = 0;
    for (int64_t i = 0; i < 3 * F; ++i)
        if (i == 0 || s[i] != s[i - 1]) s[E++] = s[i];
    for (int64_t e = 0; e < E; ++e) {
        edges[2 * e] = (int64_t)(s[e] / Vh) - 1;
        edges[2 * e + 1] = (int64_t)(s[e] % Vh) - 1;
    }
    if (faces_to_edges) {
        const int col_of[3] = {2, 0, 1}; /* e12 -> col 3, e23 -> col 1, e31 -> col 2 */
        for (int e = 0; e < 3; ++e)
            for (int64_t f = 0; f < F; ++f) {
                uint64_t key = h[(size_t)e * F + f];
                int64_t lo = 0, hi = E - 1;
                while (lo < hi) {
                    int64_t mid = (lo + hi) >> 1;
                    if (s[mid] < key) lo = mid + 1; else hi = mid;
                }
                faces_to_edges[3 * f + col_of[e]] = lo;
            }
    }
    free(h); free(s);
    return E;
}

/* _compute_laplacian_packed, src/rep/mesh.jl:957-1002, as CSR (rows sorted by column, duplicate
 * (i,j) entries summed like SparseArrays.sparse does):
 *   A = sparse([e1;e2],[e2;e1],1) ; deg = sum(A,dims=2) ; L[i,j] = T(1/deg[i]) for edges,
 *   L[i,i] += -1.   rowptr[V+1], colind/vals capacity 2E+V.  Returns nnz. */
typedef struct { int64_t r, c; float v; int64_t ord; } trip_t;
static int cmp_trip(const void *a, const void *b) {
    const trip_t *x = (const trip_t *)a, *y = (const trip_t *)b;
    if (x->r != y->r) return (x->r > y->r) - (x->r < y->r);
    if (x->c != y->c) return (x->c > y->c) - (x->c < y->c);
    return (x->ord > y->ord) - (x->ord < y->ord);
}

FX_API int64_t fx3d_oracle_laplacian_csr(const int64_t *edges, int64_t E, int64_t V,
                                         int64_t *rowptr, int64_t *colind, float *vals) {
    int64_t *deg = (int64_t *)calloc((size_t)V, sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) { deg[edges[2 * e]]++; deg[edges[2 * e + 1]]++; }
    trip_t *t = (trip_t *)malloc(sizeof(trip_t) * (size_t)(2 * E + V));
    int64_t n = 0;
    for (int64_t e = 0; e < E; ++e) { /* Is=[e1;e2;1:V] Js=[e2;e1;1:V] Vs=[deg1;deg2;diag] */
        int64_t i = edges[2 * e], j = edges[2 * e + 1];
        t[n] = (trip_t){i, j, deg[i] > 0 ? (float)(1.0 / (double)deg[i]) : (float)deg[i], n}; ++n;
    }
    for (int64_t e = 0; e < E; ++e) {
        int64_t i = edges[2 * e], j = edges[2 * e + 1];
        t[n] = (trip_t){j, i, deg[j] > 0 ? (float)(1.0 / (double)deg[j]) : (float)deg[j], n}; ++n;
    }
    for (int64_t i = 0; i < V; ++i) { t[n] = (trip_t){i, i, -1.0f, n}; ++n; }
    qsort(t, (size_t)n, sizeof(trip_t), cmp_trip);
    int64_t nnz = 0;
    memset(rowptr, 0, sizeof(int64_t) * (size_t)(V + 1));
    for (int64_t k = 0; k < n; ++k) {
        if (nnz > 0 && k > 0 && t[k].r == t[k - 1].r && t[k].c == t[k - 1].c) {
            vals[nnz - 1] = vals[nnz - 1] + t[k].v;
        } else {
            colind[nnz] = t[k].c; vals[nnz] = t[k].v; rowptr[t[k].r + 1]++; ++nnz;
        }
    }
    for (int64_t i = 0; i < V; ++i) rowptr[i + 1] += rowptr[i];
    free(deg); free(t);
    return nnz;
}

/* laplacian_loss, src/metrics/mesh.jl:9-15:  mean_i || (L * verts')[i,:] ||.
 * SparseArrays' CSC x dense product adds, for output row i, terms in ascending column order in
 * Float32 (C[i,k] += nzv * B[j,k], no muladd) -- the CSR row order here.  Mean in double. */
FX_API int fx3d_oracle_laplacian_loss(const float *verts, int64_t V, const int64_t *rowptr,
                                      const int64_t *colind, const float *vals, float *loss) {
    double acc = 0.0;
    for (int64_t i = 0; i < V; ++i) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const float *v = verts + 3 * colind[k];
            s0 = s0 + vals[k] * v[0];
            s1 = s1 + vals[k] * v[1];
            s2 = s2 + vals[k] * v[2];
        }
        float nrm = sqrtf(((s0 * s0) + (s1 * s1)) + (s2 * s2));
        acc += (double)nrm;
    }
    *loss = (float)(acc / (double)V);
    return 0;
}

/* edge_loss, src/metrics/mesh.jl:24-32:  mean_e (||v[e1]-v[e2]|| - target)^2. */
FX_API int fx3d_oracle_edge_loss(const float *verts, int64_t V, const int64_t *edges, int64_t E,
                                 float target, float *loss) {
    double acc = 0.0;
    (void)V;
    for (int64_t e = 0; e < E; ++e) {
        const float *a = verts + 3 * edges[2 * e], *b = verts + 3 * edges[2 * e + 1];
        float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
        float nrm = sqrtf(((d0 * d0) + (d1 * d1)) + (d2 * d2));
        float t = nrm - target;
        acc += (double)(t * t);
    }
    *loss = (float)(acc / (double)E);
    return 0;
}

/* Gradients of the two mesh losses w.r.t. packed verts (Zygote adjoints of the expressions
 * above), used to check the backward kernels.  d||r||/dr = r/||r|| (0 where ||r||==0). */
FX_API int fx3d_oracle_laplacian_loss_bwd(const float *verts, int64_t V, const int64_t *rowptr,
                                          const int64_t *colind, const float *vals, float gout,
                                          float *gverts) {
    memset(gverts, 0, sizeof(float) * (size_t)V * 3);
    float c = gout / (float)V;
    for (int64_t i = 0; i < V; ++i) {
        float s[3] = {0.0f, 0.0f, 0.0f};
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
            for (int d = 0; d < 3; ++d) s[d] = s[d] + vals[k] * verts[3 * colind[k] + d];
        float nrm = sqrtf(((s[0] * s[0]) + (s[1] * s[1])) + (s[2] * s[2]));
        if (!(nrm > 0.0f)) continue;
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
            for (int d = 0; d < 3; ++d) gverts[3 * colind[k] + d] += c * vals[k] * (s[d] / nrm);
    }
    return 0;
}

FX_API int fx3d_oracle_edge_loss_bwd(const float *verts, int64_t V, const int64_t *edges,
                                     int64_t E, float target, float gout, float *gverts) {
    memset(gverts, 0, sizeof(float) * (size_t)V * 3);
    float c = gout / (float)E;
    for (int64_t e = 0; e < E; ++e) {
        int64_t i = edges[2 * e], j = edges[2 * e + 1];
        float d[3] = {verts[3 * i] - verts[3 * j], verts[3 * i + 1] - verts[3 * j + 1],
                      verts[3 * i + 2] - verts[3 * j + 2]};
        float nrm = sqrtf(((d[0] * d[0]) + (d[1] * d[1])) + (d[2] * d[2]));
        if (!(nrm > 0.0f)) continue;
        float g = c * 2.0f * (nrm - target) / nrm;
        for (int k = 0; k < 3; ++k) { gverts[3 * i + k] += g * d[k]; gverts[3 * j + k] -= g * d[k]; }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * pointcloud_to_voxel, src/conversions.jl:91-131, restated literally:
 *   verts_max/min: one scalar per cloud (maximum over coordinates and over points, :94-95);
 *   cloud = (p .- min) ./ (max - min) in Float32 (:96);
 *   grid_points = (ind .+ 0.5) ./ res with ind = 1..res, loop order x outer, z inner (:102-113), Float64;
 *   nearest cloud point of every grid point (knn(KDTree(cloud), grid, 1), :115-123; Euclidean in the
 *   promoted type = Float64, dimension order; first minimum here, tie order is unspecified upstream and
 *   cannot change the distance);
 *   dists = sum((grid - cloud[nn]).^2, dims=1) (Float64), voxel = dists <= 0.6/(res*res) (:125-129);
 *   reshape(dists, res,res,res,B): column-major, so the first dimension is z.
 * vox: (res,res,res,B) Float32 0/1.
 * ---------------------------------------------------------------------------------------- */
FX_API int fx3d_oracle_pointcloud_to_voxel(const float *points, int N, int B, int res, float *vox) {
    const size_t R3 = (size_t)res * res * res;
    float *cloud = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    const double thr = 0.6 / (double)((long long)res * res);
    for (int b = 0; b < B; ++b) {
        const float *p = points + (size_t)b * N * 3;
        float lo = p[0], hi = p[0];
        int has_nan = 0;
        for (int e = 0; e < 3 * N; ++e) {
            if (p[e] != p[e]) has_nan = 1;
            if (p[e] < lo) lo = p[e];
            if (p[e] > hi) hi = p[e];
        }
        if (has_nan) lo = hi = NAN;
        for (int e = 0; e < 3 * N; ++e) cloud[e] = (p[e] - lo) / (hi - lo);
        size_t g = 0;
        for (int x = 1; x <= res; ++x)
            for (int y = 1; y <= res; ++y)
                for (int z = 1; z <= res; ++z, ++g) {
                    const double gx = ((double)x + 0.5) / (double)res;
                    const double gy = ((double)y + 0.5) / (double)res;
                    const double gz = ((double)z + 0.5) / (double)res;
                    double best = INFINITY;
                    int bj = 0;
                    for (int j = 0; j < N; ++j) {
                        const double dx = gx - (double)cloud[3 * j], dy = gy - (double)cloud[3 * j + 1],
                                     dz = gz - (double)cloud[3 * j + 2];
                        const double d = ((dx * dx) + (dy * dy)) + (dz * dz);
                        if (d < best) { best = d; bj = j; }
                    }
                    const double dx = gx - (double)cloud[3 * bj], dy = gy - (double)cloud[3 * bj + 1],
                                 dz = gz - (double)cloud[3 * bj + 2];
                    const double d = ((dx * dx) + (dy * dy)) + (dz * dz);
                    vox[(size_t)b * R3 + g] = (d <= thr) ? 1.0f : 0.0f;
                }
    }
    free(cloud);
    return 0;
}
