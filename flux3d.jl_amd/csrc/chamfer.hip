// Nearest-neighbour + chamfer kernels (gfx950).
//
// Replaces _nearest_neighbors(::CuArray,::CuArray) + the gather/mean of _chamfer_distance
// (src/metrics/pcloud.jl:39-52, 72-86): instead of materialising the (N,M,B) matrix P and running
// two argmin passes over it, one launch streams every reference cloud through LDS once per query
// tile and keeps the running minimum in registers.  HBM traffic is O((N+M)*B); the kernel is
// bound by fp32 VALU issue (DESIGN.md "Roofline").
//
// Arithmetic is the CPU method's (src/metrics/pcloud.jl:54-70 via NearestNeighbors' Euclidean):
//   d = ((dx*dx) + dy*dy) + dz*dz   in Float32, no fused multiply-add (-ffp-contract=off),
//   lowest index wins ties  =>  indices are bit-identical to oracle/flux3d_oracle.c:nn1_dir.
//
// Structure of nn1_small_d_kernel<DIM,R>:
//   * block = 256 threads (4 wave64); each thread owns R query points in registers.
//   * the reference cloud is staged chunk-wise into LDS as structure-of-arrays (x[],y[],z[]):
//     every lane reads the SAME address (ds_read_b128 broadcast of 4 consecutive x's), so LDS
//     reads are conflict-free and cost 3 instructions per 4 candidates per wave.
//   * candidates are consumed in tiles of T=32: the tile minimum is folded with v_min3_f32
//     (0.5 instruction per pair instead of compare+2 selects), and only once per tile the
//     running (best, best_tile) is updated.  The argmin is recovered by re-scanning the single
//     winning tile from LDS (1/128 of the work at M=4096) with the reference's strict `<`.
//   * grid is 1-D and XCD-aware: block L runs on XCD L%8 (MI355X_MICROARCH.md), so all query
//     tiles of one (direction,batch) cloud are given ids with equal L%8 and share that XCD's L2.
#include <cmath>

#include "fx3d_common.h"

using namespace fx3d;

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 32;       // candidates per min3 tile
constexpr int kChunkMax = 4096; // candidates staged in LDS at once (DIM*16 KiB)

struct Nn1Params {
    const float *x;  // (D,N,B)
    const float *y;  // (D,M,B)
    int N, M, B;
    int32_t *idx_x, *idx_y;  // optional
    float *dmin_x, *dmin_y;  // optional
    double *partials;        // [2*Bpad8... ] one per (cloud, tile); optional
    int tiles;               // max(tiles_x, tiles_y)
    int tiles_x, tiles_y;
    int chunk;               // LDS chunk capacity (multiple of kTile)
};

__device__ __forceinline__ float min3f(float a, float b, float c) {
    return __builtin_fminf(__builtin_fminf(a, b), c);
}

template <int DIM>
__device__ __forceinline__ float sqd(const float (&q)[DIM], const float (&c)[DIM]) {
    float t0 = q[0] - c[0];
    float s = t0 * t0;
#pragma unroll
    for (int d = 1; d < DIM; ++d) {
        float t = q[d] - c[d];
        s = s + t * t;
    }
    return s;
}

template <int DIM, int R, bool WANT_IDX>
__global__ __launch_bounds__(kThreads) void nn1_small_d_kernel(Nn1Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // ---- XCD-aware decode of the linear block id -> (cloud c, query tile) ------------------
    const int L = blockIdx.x;
    const int xcd = L & 7, slot = L >> 3;
    const int c = (slot / p.tiles) * 8 + xcd;  // cloud id in [0, 2B): dir = c / B
    const int tile = slot % p.tiles;
    if (c >= 2 * p.B) return;
    const int dir = c >= p.B ? 1 : 0;
    const int b = dir ? c - p.B : c;
    const int NQ = dir ? p.M : p.N;  // queries
    const int NC = dir ? p.N : p.M;  // candidates
    if (tile >= (dir ? p.tiles_y : p.tiles_x)) return;
    const float *__restrict__ qb = (dir ? p.y : p.x) + (size_t)b * NQ * DIM;
    const float *__restrict__ cb = (dir ? p.x : p.y) + (size_t)b * NC * DIM;

    const int tid = threadIdx.x;
    const int CH = p.chunk;
    const int CH4 = CH >> 2;
    const float4 *lds4 = reinterpret_cast<const float4 *>(lds);

    // ---- queries into registers (out-of-range lanes clamp to the last point) ---------------
    float q[R][DIM];
    int qi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        qi[r] = tile * (kThreads * R) + r * kThreads + tid;
        const int qc = qi[r] < NQ ? qi[r] : NQ - 1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) q[r][d] = qb[(size_t)qc * DIM + d];
    }

    float best[R];
    int btile[R], bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = INFINITY; btile[r] = -1; bidx[r] = 0; }

    for (int j0 = 0; j0 < NC; j0 += CH) {
        const int cnt = (NC - j0) < CH ? (NC - j0) : CH;
        const int cnt_pad = (cnt + kTile - 1) / kTile * kTile;
        if (j0 > 0) __syncthreads();
        // ---- stage chunk: AoS global stream -> SoA LDS (coalesced dword reads) -------------
        for (int e = tid; e < cnt * DIM; e += kThreads) {
            const float v = cb[(size_t)j0 * DIM + e];
            const int pt = e / DIM, cc = e - pt * DIM;
            lds[cc * CH + pt] = v;
        }
        for (int e = cnt + tid; e < cnt_pad; e += kThreads) {
#pragma unroll
            for (int d = 0; d < DIM; ++d) lds[d * CH + e] = INFINITY;
        }
        __syncthreads();

        const int ntile = cnt_pad / kTile;
        const int tile_base = j0 / kTile;  // CH is a multiple of kTile
        for (int t = 0; t < ntile; ++t) {
            float tm[R];
#pragma unroll
            for (int r = 0; r < R; ++r) tm[r] = INFINITY;
#pragma unroll
            for (int jj = 0; jj < kTile; jj += 4) {
                float4 cv[DIM];
#pragma unroll
                for (int d = 0; d < DIM; ++d)  // float4 units: CH % 32 == 0 => always ds_read_b128
                    cv[d] = lds4[d * CH4 + t * (kTile / 4) + jj / 4];
                float c0[DIM], c1[DIM], c2[DIM], c3[DIM];
#pragma unroll
                for (int d = 0; d < DIM; ++d) { c0[d] = cv[d].x; c1[d] = cv[d].y; c2[d] = cv[d].z; c3[d] = cv[d].w; }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float d0 = sqd<DIM>(q[r], c0), d1 = sqd<DIM>(q[r], c1);
                    const float d2 = sqd<DIM>(q[r], c2), d3 = sqd<DIM>(q[r], c3);
                    tm[r] = min3f(tm[r], d0, d1);
                    tm[r] = min3f(tm[r], d2, d3);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool better = tm[r] < best[r];  // strict: first tile holding the minimum
                best[r] = better ? tm[r] : best[r];
                btile[r] = better ? tile_base + t : btile[r];
            }
        }

        if (WANT_IDX) {
            // ---- exact argmin: re-scan the winning tile while its chunk is still in LDS ----
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (btile[r] >= tile_base) {  // improved within this chunk
                    const int off = (btile[r] - tile_base) * kTile;
                    float cur = INFINITY;
                    int ci = 0;
                    for (int jj = 0; jj < kTile; ++jj) {
                        float cc[DIM];
#pragma unroll
                        for (int d = 0; d < DIM; ++d) cc[d] = lds[d * CH + off + jj];
                        const float dd = sqd<DIM>(q[r], cc);
                        if (dd < cur) { cur = dd; ci = jj; }
                    }
                    bidx[r] = j0 + off + ci;
                }
            }
        }
    }

    // ---- outputs ------------------------------------------------------------------------------
    int32_t *idx_out = dir ? p.idx_y : p.idx_x;
    float *dmin_out = dir ? p.dmin_y : p.dmin_x;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (qi[r] < NQ) {
            if (WANT_IDX && idx_out) idx_out[(size_t)b * NQ + qi[r]] = bidx[r];
            if (dmin_out) dmin_out[(size_t)b * NQ + qi[r]] = best[r];
            acc += (double)best[r];
        }
    }
    if (p.partials) {
        __shared__ double sm[kThreads / 64];
        const double tot = block_sum<kThreads>(acc, sm);
        if (tid == 0) p.partials[(size_t)c * p.tiles + tile] = tot;
    }
}

// Generic dimension (D == 1 or D > 3): one thread per query, candidates read through L1/L2.
// Correct for any D; not the tuned path (the chamfer configs are all D = 3).
__global__ __launch_bounds__(kThreads) void nn1_generic_kernel(Nn1Params p, int D) {
    const int c = blockIdx.y;
    const int dir = c >= p.B ? 1 : 0;
    const int b = dir ? c - p.B : c;
    const int NQ = dir ? p.M : p.N, NC = dir ? p.N : p.M;
    const int tile = blockIdx.x;
    const float *__restrict__ qb = (dir ? p.y : p.x) + (size_t)b * NQ * D;
    const float *__restrict__ cb = (dir ? p.x : p.y) + (size_t)b * NC * D;
    const int i = tile * kThreads + threadIdx.x;
    double acc = 0.0;
    if (i < NQ) {
        float best = INFINITY;
        int bi = 0;
        const float *a = qb + (size_t)i * D;
        for (int j = 0; j < NC; ++j) {
            const float *cc = cb + (size_t)j * D;
            float s = 0.0f;
            for (int d = 0; d < D; ++d) { float t = a[d] - cc[d]; s = s + t * t; }
            if (s < best) { best = s; bi = j; }
        }
        int32_t *idx_out = dir ? p.idx_y : p.idx_x;
        float *dmin_out = dir ? p.dmin_y : p.dmin_x;
        if (idx_out) idx_out[(size_t)b * NQ + i] = bi;
        if (dmin_out) dmin_out[(size_t)b * NQ + i] = best;
        acc = (double)best;
    }
    if (p.partials) {
        __shared__ double sm[kThreads / 64];
        const double tot = block_sum<kThreads>(acc, sm);
        // blocks past this direction's tile count still write (zero) so the reduce is uniform
        if (threadIdx.x == 0) p.partials[(size_t)c * p.tiles + tile] = tot;
    }
}

// Fixed-order reduction of the per-block partials into sums[0..1] (+ optional loss).
struct FinalizeParams {
    const double *partials;
    int B, tiles, tiles_x, tiles_y;
    double *sums;  // [2]
    // optional loss (loss != nullptr)
    float *loss;
    int N, M, D;
    long long Bg;
    float w1, w2;
};

__device__ __forceinline__ float chamfer_loss_from_sums(double sa, double sb, int N, int M, int D,
                                                        long long Bg, float w1, float w2) {
    // mean(...) * 3.0f0, src/metrics/pcloud.jl:47-48 ; w1*dA + w2*dB, :50
    const float dA = (float)(sa / ((double)D * (double)N * (double)Bg)) * 3.0f;
    const float dB = (float)(sb / ((double)D * (double)M * (double)Bg)) * 3.0f;
    return (w1 * dA) + (w2 * dB);
}

__global__ __launch_bounds__(kThreads) void chamfer_finalize_partials_kernel(FinalizeParams f) {
    __shared__ double sm[kThreads / 64];
    double tot[2];
    for (int dir = 0; dir < 2; ++dir) {
        const int nt = dir ? f.tiles_y : f.tiles_x;
        const long long n = (long long)f.B * nt;
        double acc = 0.0;
        for (long long k = threadIdx.x; k < n; k += kThreads) {
            const int b = (int)(k / nt), t = (int)(k % nt);
            acc += f.partials[((size_t)(dir * f.B + b)) * f.tiles + t];
        }
        __syncthreads();
        tot[dir] = block_sum<kThreads>(acc, sm);
    }
    if (threadIdx.x == 0) {
        if (f.sums) { f.sums[0] = tot[0]; f.sums[1] = tot[1]; }
        if (f.loss) *f.loss = chamfer_loss_from_sums(tot[0], tot[1], f.N, f.M, f.D, f.Bg, f.w1, f.w2);
    }
}

__global__ void chamfer_loss_kernel(const double *sums, int N, int M, int D, long long Bg,
                                    float w1, float w2, float *loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *loss = chamfer_loss_from_sums(sums[0], sums[1], N, M, D, Bg, w1, w2);
}

// Backward: gather-difference + atomic scatter-add (adjoint of the two gathers at :47-48).
__global__ __launch_bounds__(kThreads) void chamfer_bwd_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ y, int M, int B, int D,
    const int32_t *__restrict__ idx_x, const int32_t *__restrict__ idx_y, float ca, float cb,
    float *gx, float *gy) {
    const long long total = (long long)B * (N + M);
    for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < total;
         k += (long long)gridDim.x * kThreads) {
        const int b = (int)(k / (N + M));
        const int r = (int)(k % (N + M));
        const float *xb = x + (size_t)b * N * D, *yb = y + (size_t)b * M * D;
        float *gxb = gx + (size_t)b * N * D, *gyb = gy + (size_t)b * M * D;
        if (r < N) {
            const int i = r, j = idx_x[(size_t)b * N + i];
            for (int d = 0; d < D; ++d) {
                const float t = ca * (xb[(size_t)i * D + d] - yb[(size_t)j * D + d]);
                atomicAdd(&gxb[(size_t)i * D + d], t);
                atomicAdd(&gyb[(size_t)j * D + d], -t);
            }
        } else {
            const int j = r - N, i = idx_y[(size_t)b * M + j];
            for (int d = 0; d < D; ++d) {
                const float t = cb * (yb[(size_t)j * D + d] - xb[(size_t)i * D + d]);
                atomicAdd(&gyb[(size_t)j * D + d], t);
                atomicAdd(&gxb[(size_t)i * D + d], -t);
            }
        }
    }
}

struct Plan {
    int R, tiles_x, tiles_y, tiles, chunk, grid;
    size_t lds_bytes;
};

Plan make_plan(int N, int M, int B, int D) {
    Plan pl{};
    // R queries per thread: enough blocks to fill 256 CUs x ~2 blocks, but as much register
    // blocking (LDS-read amortisation, ILP) as the problem size allows.
    const long long work = (long long)B * ((long long)N + M);  // total queries, both directions
    int R = 4;
    while (R > 1 && work / (kThreads * R) < 512) R >>= 1;
    pl.R = R;
    const int per_block = kThreads * R;
    pl.tiles_x = (N + per_block - 1) / per_block;
    pl.tiles_y = (M + per_block - 1) / per_block;
    pl.tiles = pl.tiles_x > pl.tiles_y ? pl.tiles_x : pl.tiles_y;
    const int maxc = N > M ? N : M;
    int chunk = (maxc + kTile - 1) / kTile * kTile;
    if (chunk > kChunkMax) chunk = kChunkMax;
    pl.chunk = chunk;
    pl.lds_bytes = (size_t)chunk * (D <= 3 ? D : 0) * sizeof(float);
    const int clouds8 = (2 * B + 7) / 8;
    pl.grid = clouds8 * 8 * pl.tiles;
    return pl;
}

size_t partials_count(const Plan &pl, int B) { return (size_t)2 * B * pl.tiles; }

template <int DIM, bool WANT_IDX>
fx3d_status launch_small(const Nn1Params &p, const Plan &pl, hipStream_t st) {
    switch (pl.R) {
        case 4:
            hipLaunchKernelGGL((nn1_small_d_kernel<DIM, 4, WANT_IDX>), dim3(pl.grid), dim3(kThreads), pl.lds_bytes, st, p);
            break;
        case 2:
            hipLaunchKernelGGL((nn1_small_d_kernel<DIM, 2, WANT_IDX>), dim3(pl.grid), dim3(kThreads), pl.lds_bytes, st, p);
            break;
        default:
            hipLaunchKernelGGL((nn1_small_d_kernel<DIM, 1, WANT_IDX>), dim3(pl.grid), dim3(kThreads), pl.lds_bytes, st, p);
            break;
    }
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status check_shapes(const char *fn, const void *x, int N, const void *y, int M, int B, int D) {
    FX3D_REQUIRE(x && y, "%s: null input pointer", fn);
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0, "%s: empty input (N=%d M=%d B=%d D=%d)", fn, N, M, B, D);
    FX3D_REQUIRE((long long)B * 2 * (((long long)(N > M ? N : M) + 255) / 256) < (1ll << 30),
                 "%s: problem too large for one launch", fn);
    return FX3D_OK;
}

fx3d_status run_nn1(const float *x, int N, const float *y, int M, int B, int D, int32_t *idx_x,
                    int32_t *idx_y, float *dmin_x, float *dmin_y, double *partials,
                    const Plan &pl, hipStream_t st) {
    Nn1Params p{};
    p.x = x; p.y = y; p.N = N; p.M = M; p.B = B;
    p.idx_x = idx_x; p.idx_y = idx_y; p.dmin_x = dmin_x; p.dmin_y = dmin_y;
    p.partials = partials;
    p.tiles = pl.tiles; p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.chunk = pl.chunk;
    const bool want_idx = idx_x || idx_y;
    ProfileScope prof("nn1", st);
    if (D == 3) return want_idx ? launch_small<3, true>(p, pl, st) : launch_small<3, false>(p, pl, st);
    if (D == 2) return want_idx ? launch_small<2, true>(p, pl, st) : launch_small<2, false>(p, pl, st);
    // generic D: tiles are 256 queries per block
    Nn1Params g = p;
    g.tiles_x = (N + kThreads - 1) / kThreads;
    g.tiles_y = (M + kThreads - 1) / kThreads;
    g.tiles = g.tiles_x > g.tiles_y ? g.tiles_x : g.tiles_y;
    hipLaunchKernelGGL(nn1_generic_kernel, dim3(g.tiles, 2 * B), dim3(kThreads), 0, st, g, D);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

// tiles used for the partial-sum layout (generic path uses 256-query tiles)
void partial_layout(const Plan &pl, int N, int M, int D, int *tiles, int *tx, int *ty) {
    if (D == 2 || D == 3) { *tiles = pl.tiles; *tx = pl.tiles_x; *ty = pl.tiles_y; return; }
    *tx = (N + kThreads - 1) / kThreads;
    *ty = (M + kThreads - 1) / kThreads;
    *tiles = *tx > *ty ? *tx : *ty;
}

}  // namespace

extern "C" {

fx3d_status fx3d_nn1(const float *x, int32_t N, const float *y, int32_t M, int32_t B, int32_t D,
                     int32_t *idx_x, int32_t *idx_y, float *dmin_x, float *dmin_y,
                     fx3d_stream_t s) {
    fx3d_status rc = check_shapes("fx3d_nn1", x, N, y, M, B, D);
    if (rc) return rc;
    const Plan pl = make_plan(N, M, B, D);
    return run_nn1(x, N, y, M, B, D, idx_x, idx_y, dmin_x, dmin_y, nullptr, pl, as_stream(s));
}

fx3d_status fx3d_chamfer_workspace_bytes(int32_t N, int32_t M, int32_t B, int32_t D, size_t *bytes) {
    FX3D_REQUIRE(bytes, "fx3d_chamfer_workspace_bytes: null output");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0, "fx3d_chamfer_workspace_bytes: empty input");
    const Plan pl = make_plan(N, M, B, D);
    int tiles, tx, ty;
    partial_layout(pl, N, M, D, &tiles, &tx, &ty);
    *bytes = ((size_t)2 * B * tiles + 2) * sizeof(double);
    return FX3D_OK;
}

static fx3d_status chamfer_common(const float *x, int N, const float *y, int M, int B, int D,
                                  double *sums_dev, float *loss_dev, long long Bg, float w1,
                                  float w2, int32_t *idx_x, int32_t *idx_y, void *ws,
                                  size_t ws_bytes, hipStream_t st, const char *fn) {
    fx3d_status rc = check_shapes(fn, x, N, y, M, B, D);
    if (rc) return rc;
    const Plan pl = make_plan(N, M, B, D);
    int tiles, tx, ty;
    partial_layout(pl, N, M, D, &tiles, &tx, &ty);
    const size_t need = ((size_t)2 * B * tiles + 2) * sizeof(double);
    if (!ws || ws_bytes < need) {
        set_error("%s: workspace too small (%zu < %zu bytes)", fn, ws ? ws_bytes : (size_t)0, need);
        return FX3D_ERR_WORKSPACE;
    }
    double *partials = reinterpret_cast<double *>(ws);
    rc = run_nn1(x, N, y, M, B, D, idx_x, idx_y, nullptr, nullptr, partials, pl, st);
    if (rc) return rc;
    FinalizeParams f{};
    f.partials = partials; f.B = B; f.tiles = tiles; f.tiles_x = tx; f.tiles_y = ty;
    f.sums = sums_dev ? sums_dev : partials + (size_t)2 * B * tiles;
    f.loss = loss_dev; f.N = N; f.M = M; f.D = D; f.Bg = Bg; f.w1 = w1; f.w2 = w2;
    hipLaunchKernelGGL(chamfer_finalize_partials_kernel, dim3(1), dim3(kThreads), 0, st, f);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_chamfer_sums(const float *x, int32_t N, const float *y, int32_t M, int32_t B,
                              int32_t D, double *sums_dev, int32_t *idx_x, int32_t *idx_y,
                              void *ws, size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(sums_dev, "fx3d_chamfer_sums: null sums_dev");
    return chamfer_common(x, N, y, M, B, D, sums_dev, nullptr, B, 1.f, 1.f, idx_x, idx_y, ws,
                          ws_bytes, as_stream(s), "fx3d_chamfer_sums");
}

fx3d_status fx3d_chamfer_finalize(const double *sums_dev, int32_t N, int32_t M, int64_t B_global,
                                  int32_t D, float w1, float w2, float *loss_dev,
                                  fx3d_stream_t s) {
    FX3D_REQUIRE(sums_dev && loss_dev, "fx3d_chamfer_finalize: null pointer");
    FX3D_REQUIRE(N > 0 && M > 0 && B_global > 0 && D > 0, "fx3d_chamfer_finalize: bad sizes");
    hipLaunchKernelGGL(chamfer_loss_kernel, dim3(1), dim3(64), 0, as_stream(s), sums_dev, N, M, D,
                       (long long)B_global, w1, w2, loss_dev);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_chamfer_fwd(const float *x, int32_t N, const float *y, int32_t M, int32_t B,
                             int32_t D, float w1, float w2, float *loss_dev, float *loss_host,
                             int32_t *idx_x, int32_t *idx_y, void *ws, size_t ws_bytes,
                             fx3d_stream_t s) {
    FX3D_REQUIRE(loss_dev, "fx3d_chamfer_fwd: null loss_dev");
    fx3d_status rc = chamfer_common(x, N, y, M, B, D, nullptr, loss_dev, B, w1, w2, idx_x, idx_y,
                                    ws, ws_bytes, as_stream(s), "fx3d_chamfer_fwd");
    if (rc) return rc;
    if (loss_host) {
        FX3D_HIP(hipMemcpyAsync(loss_host, loss_dev, sizeof(float), hipMemcpyDeviceToHost, as_stream(s)));
        FX3D_HIP(hipStreamSynchronize(as_stream(s)));
    }
    return FX3D_OK;
}

fx3d_status fx3d_chamfer_bwd(const float *x, int32_t N, const float *y, int32_t M, int32_t B,
                             int32_t D, const int32_t *idx_x, const int32_t *idx_y, float w1,
                             float w2, float gout, int64_t B_global, float *gx, float *gy,
                             fx3d_stream_t s) {
    fx3d_status rc = check_shapes("fx3d_chamfer_bwd", x, N, y, M, B, D);
    if (rc) return rc;
    FX3D_REQUIRE(idx_x && idx_y && gx && gy, "fx3d_chamfer_bwd: null pointer");
    FX3D_REQUIRE(B_global >= B, "fx3d_chamfer_bwd: B_global < B");
    hipStream_t st = as_stream(s);
    FX3D_HIP(hipMemsetAsync(gx, 0, sizeof(float) * (size_t)N * B * D, st));
    FX3D_HIP(hipMemsetAsync(gy, 0, sizeof(float) * (size_t)M * B * D, st));
    const float ca = gout * w1 * (float)(6.0 / ((double)D * N * (double)B_global));
    const float cb = gout * w2 * (float)(6.0 / ((double)D * M * (double)B_global));
    const long long total = (long long)B * (N + M);
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, st, x, N, y, M,
                       B, D, idx_x, idx_y, ca, cb, gx, gy);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

}  // extern "C"
