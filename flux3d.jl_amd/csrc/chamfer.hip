// Nearest-neighbour + chamfer forward kernels (gfx950).  (The adjoints live in chamfer_bwd.hip.)
//
// Replaces _nearest_neighbors(::CuArray,::CuArray) + the gather / mean of _chamfer_distance (src/metrics/pcloud.jl:39-52,
// 72-86): the reference materialises the (N,M,B) matrix and runs two argmin passes over it; here every cloud is read from HBM
// once per launch (O((N + M) B) bytes, measured 1.09 x that) and the N x M x B matrix never exists.  Results are the CPU
// method's (src/metrics/pcloud.jl:54-70 via NearestNeighbors' Euclidean), bit for bit:
//   d = ((dx dx) + dy dy) + dz dz  in Float32, no fused multiply-add (-ffp-contract=off), ordered like Julia's `isless`,
//   lowest index on ties  ==  oracle/flux3d_oracle.c: nn1_dir.
//
// Kernels, by the launch plan's choice (make_plan; fx3d_nn1_plan_describe prints it):
//   * nn1_f16_kernel (D = 3, the default) -- FILTER on the matrix cores, exact re-scan of what survives.  The
//     argmin of the distance is the argmin of t = |c'|^2 - 2 q'.c' (cloud centred on its mean, scaled by a power of two), a
//     K = 16 inner product of 2-way fp16 splits: ONE v_mfma_f32_32x32x16_f16 per 32 x 32 pairs, 8 v_min3 fold 16 rows, a few
//     VALU ops track each lane's three smallest lane tiles; every candidate inside the error band of the running minimum goes
//     to a wave-cooperative exact phase (the oracle's arithmetic, 64-bit LDS atomicMin on (distance bits, index)).  1024-thread
//     blocks share one fp16 image of up to 4096 candidates (128 KiB of LDS), 512 queries per pass, XCD-aware block ids
//     (block L runs on XCD L % 8: a cloud's blocks share that XCD's L2); larger clouds run in chunks, or as candidate SUBSETS
//     in parallel blocks whose per-query rows the last subset of a query tile merges itself (round 5: one launch); a robust
//     range + exact side list for far outliers; a per-query power-of-two scale for queries far outside the cloud.  The loss is
//     finalised in the same launch (Float64 partials, agent-scope ticket, fixed summation order).  Bound by the matrix pipe +
//     the VALU fold behind it: roofline.frac 0.27 of the dense f16 peak.
//   * nn1_tiny_kernel (D = 3, problems below ~24 M pair evaluations) -- the exact loop with candidates broadcast along DPP
//     rows: no statistics, no image, no barrier before the arithmetic.
//   * nn1_small_d_kernel<DIM, R> (D = 2, and D = 3 under option nn1_variant = 0: the A/B reference) -- the exact VALU loop of
//     round 1: candidates staged through LDS as structure-of-arrays, tiles of 32 folded by v_min3, the winning tile re-scanned
//     with the reference's strict `<`.
//   * nn1_generic_kernel (any other D).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "fx3d_common.h"

using namespace fx3d;

// Phase timestamps for tools/nn1_probe.hip (compiled out of the product build).
#ifdef FX3D_PROBE
__device__ unsigned long long g_probe[4096 * 16];
#define FX3D_PROBE_MARK(k)                                                                   \
    do {                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                         \
            g_probe[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter();                   \
            if ((k) == 0) g_probe[blockIdx.x * 16 + 13] = wall_clock64();                    \
            if ((k) == 12) g_probe[blockIdx.x * 16 + 14] = wall_clock64();                   \
        }                                                                                    \
    } while (0)
__device__ unsigned long long g_probe2[256 * 16];
#define FX3D_PROBE_MARK2(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_probe2[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
__device__ unsigned long long g_probe3[16 * 16 * 8];   // per WAVE stamps of the first 16 blocks
#define FX3D_PROBE_MARKW(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 16) g_probe3[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define FX3D_PROBE_MARK(k) do { } while (0)
#define FX3D_PROBE_MARK2(k) do { } while (0)
#define FX3D_PROBE_MARKW(k) do { } while (0)
#endif

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
    int tpb;                 // fp16 variant: query-tile passes per block (>1 only for one-chunk clouds), x -> y direction
    int tpb_y;               // ... y -> x direction (clouds of different sizes: the plan balances the two directions' blocks)
    // fused finalisation (fp16 variant): the last block to arrive reduces the partials in fixed order
    unsigned int *ticket;    // library-owned arrival counter, zero between launches; nullptr = no fusion
    unsigned int nvalid;     // number of blocks that deliver a partial
    double *sums_out;        // [2] optional
    float *loss_out;         // optional
    float w1, w2;
    long long Bg;
    // candidate split (few, large clouds): blocks of one query tile take different chunk subsets and
    // merge per query through 64-bit atomics in global scratch; nn1_split_finalize_kernel unpacks
    int nsplit;                  // 1 = off
    unsigned long long *gres;    // [nsplit][2B][qstride] packed (d_bits << 32 | index): every split block stores its own row (no init, no atomics)
    int qstride;
    int fuse_split;              // 1: the last block of a query tile's chunk subsets merges their rows itself (arrival counters in the ticket slot's spare words)
    int tail;                    // > 0: a cloud of chunk + (1 .. tail) points is ONE chunk + a tail every query evaluates exactly
    // spatial pruning (round 6, nn1_f16_kernel<.., PRUNE = true>): per-block scratch in the caller's workspace -- the block's candidate
    // cloud in image order and its window of the query cloud in processing order, as (x, y, z, original index) rows
    float4 *pscr;                // nullptr: the launch runs the PRUNE = false instantiation
    int pscr_stride;             // rows per block: kHChunkMax candidates + the query window (passes x 512 + kHTail)
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

// Order of distances = the oracle's (oracle/flux3d_oracle.c: fless): Julia's isless on Float32 -- ascending, every NaN
// after +Inf, all NaNs equal -- then the lower index.  Squared distances are >= +0 or NaN, so the order is the unsigned
// order of the bit patterns once NaNs are canonical; finite data never produces a NaN distance (at worst +Inf), and
// the hot paths below only pay for this on data that is not finite.
__device__ __forceinline__ bool fless(float a, float b) { return (a < b) || (b != b && a == a); }
__device__ __forceinline__ unsigned int dist_key(float d) { return d != d ? 0x7fc00000u : __builtin_bit_cast(unsigned int, d); }

// exact scan of a whole cloud in that order (the rare exit of the exact-loop kernels: no distance below +Inf)
template <int DIM>
__device__ __forceinline__ void nn1_scan_isless(const float (&q)[DIM], const float *__restrict__ cb, int NC, float &best, int &bi) {
    float c0[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) c0[d] = cb[d];
    best = sqd<DIM>(q, c0);
    bi = 0;
    for (int j = 1; j < NC; ++j) {
        float cc[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) cc[d] = cb[(size_t)j * DIM + d];
        const float dd = sqd<DIM>(q, cc);
        if (fless(dd, best)) { best = dd; bi = j; }
    }
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
            // nothing below +Inf (overflowing or non-finite coordinates): min3 / `<` skipped every candidate
            if (!(best[r] < INFINITY)) nn1_scan_isless<DIM>(q[r], cb, NC, best[r], bidx[r]);
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

// four consecutive points (12 floats) as three 16-byte loads; needs 16-B aligned base and p0 % 4 == 0
__device__ __forceinline__ void load4pts(const float *__restrict__ base, int p0, float (&px)[4],
                                         float (&py)[4], float (&pz)[4]) {
    const float4 *s4 = reinterpret_cast<const float4 *>(base + (size_t)p0 * 3);
    const float4 f0 = s4[0], f1 = s4[1], f2 = s4[2];
    px[0] = f0.x; py[0] = f0.y; pz[0] = f0.z;
    px[1] = f0.w; py[1] = f1.x; pz[1] = f1.y;
    px[2] = f1.z; py[2] = f1.w; pz[2] = f2.x;
    px[3] = f2.y; py[3] = f2.z; pz[3] = f2.w;
}

// ------------------------------------------------------------------------------------------------
// nn1_f16_kernel (D = 3): the filter recast as a dense 16-bit GEMM tile (north_star: "MFMA only if the
// pairwise-distance inner product is recast as a dense 16-bit GEMM and rocprof shows it beating the
// LDS-tiled path" -- it does: profiles/).
//
//   One v_mfma_f32_32x32x16_f16 produces 1024 filter values (32 candidates x 32 queries) in the time
//   the f32 form needs for 256.  Precision is recovered by 2-way fp16 splits (hi + lo ~ 22 bits) of
//   coordinates centred on the cloud's mean mu and pre-scaled by a power of two s so that |c~| = |s (c - mu)| < 2^7:
//     K slot : 0      1      2      3      4      5      6      7    | 8      9  10 11  12     13     14    15
//     A (c~) : chx    chx    clx    chy    chy    cly    chz    chz  | clz    n1 n2 n3  clx    cly    clz   0
//     B (qm~): qhx    qlx    qhx    qhy    qly    qhy    qhz    qlz  | qhz    1  1  1   qlx    qly    qlz   0
//   with qm~ = -2 s (q - mu) and n1+n2+n3 the 3-way split of the Float32 |c~|^2, so
//     D[i][j] = |c~_i|^2 + qm~_j . c~_i  =  s^2 (|c'_i|^2 - 2 q'_j . c'_i)     up to
//     |err| <= beta (|c~|^2 + |q~|^2) + floor, beta = 2^-18  (with |c~| <= 1: 2^-20 (4 + 2 S), S = sum_d |qm~_d|; measured
//     max 2^-23.2 (3 + S), tools/test_f16_filter.hip).  The candidate's share is folded into its norm and the band is
//     relative to the running minimum: see make_pieces / kBandB1 / kBandA below.
//   Queries with |qm~| beyond the fp16 range take the exact path.
//   Lane l of a wave holds query l&31 and the 16 candidate rows (r&3)+8(r>>2)+4(l>>5) of every
//   32-candidate block; the two half-waves are merged through the per-query LDS slot.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float chamfer_loss_from_sums(double sa, double sb, int N, int M, int D,
                                                        long long Bg, float w1, float w2);

// The tail of a launch with fused finalisation, run by the FIRST WAVE of every block after the block's sum `tot` has
// arrived in its lane 0: publish the partial (8-byte agent-scope store -> drain -> relaxed ticket), and the last arriver
// reduces all partials in a fixed order and writes the sums / the loss.  That reduction is the launch's tail -- every
// other CU is idle by then --, so it is one wave, all its loads in flight together (four per lane and direction), DPP
// double adds, no barrier, 32-bit index math (the block-wide version with two barrier rounds and 64-bit divisions took
// 6.0 k cycles from ticket to end at C2; this one 2.3 k).  `slot`: this block's entry; rows of `stride` entries per
// (direction, cloud), the first tiles_x / tiles_y of a row are valid.
// MEMORY-ORDER NOTE (applies to this hand-off and to the fused split merge further down).  The protocol is
//   producer:  agent-scope RELAXED atomic stores of the data (sc1: they go THROUGH the XCD's L2 to memory) -> s_waitcnt vmcnt(0) (the
//              stores have left the CU and are acknowledged) -> relaxed agent-scope fetch_add on the counter
//   consumer:  the last arriver's relaxed agent-scope atomic LOADS of the data (sc1: served from memory / the coherent fabric,
//              never from a stale line of its own XCD's L2)
// i.e. ordering by completion (waitcnt) + coherence by the access kind, not by release / acquire fences.  It is outside the HIP /
// LLVM memory model on purpose: an agent-scope RELEASE on gfx942 / gfx950 is `buffer_wbl2 sc1` -- a write-back of the WHOLE L2 of
// the XCD -- and was measured at 32 -> 94 us on the split runs (round 5); with the pruning scratch of round 6 dirty in L2 (21 MB per
// launch) it would cost more.  The `asm volatile("s_waitcnt vmcnt(0)" ::: "memory")` is both the hardware wait and the compiler
// barrier (no store may sink below it, no load of the counter may rise above it).  The assumption -- sc1 stores are visible to sc1
// loads of every XCD once vmcnt has drained -- is MI355X_MICROARCH.md's "valid forms" table; the library is built for gfx950 only:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "chamfer.hip: the relaxed-atomic + s_waitcnt hand-off between blocks is validated on gfx950 only (see the note above)"
#endif
struct FinalizeArgs {
    unsigned long long *pp;
    unsigned int *ticket;
    unsigned int nvalid;
    int B, stride, tiles_x, tiles_y;
    double *sums_out;
    float *loss_out;
    int N, M;
    long long Bg;
    float w1, w2;
};
__device__ __forceinline__ int fused_finalize_wave0(const FinalizeArgs &f, size_t slot, double tot, int lane) {
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(&f.pp[slot], __builtin_bit_cast(unsigned long long, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int old = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old == f.nvalid - 1;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return 0;
    // lane l sums entries l, l + 64, ... of a direction in order (four of each direction in flight), then the DPP tree
    double a[2] = {0.0, 0.0};
    const unsigned int n0 = (unsigned int)f.B * (unsigned int)f.tiles_x, n1 = (unsigned int)f.B * (unsigned int)f.tiles_y;  // (< 2^30: check_shapes)
    for (unsigned int k0 = lane; k0 < n0 || k0 < n1; k0 += 64 * 4) {
        unsigned long long v[2][4];
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned int nt = dd ? f.tiles_y : f.tiles_x;
                const unsigned int k = k0 + 64 * u, kc = k < (dd ? n1 : n0) ? k : 0;
                // (directions of equal tile counts -- the usual case -- : rows are dense, no division)
                const size_t e = nt == (unsigned int)f.stride ? (size_t)dd * f.B * f.stride + kc : ((size_t)(dd * f.B) + kc / nt) * f.stride + kc % nt;
                v[dd][u] = __hip_atomic_load(&f.pp[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + 64 * u < (dd ? n1 : n0)) a[dd] += __builtin_bit_cast(double, v[dd][u]);
    }
    const double t0 = wave_sum_l63_f64(a[0]), t1 = wave_sum_l63_f64(a[1]);
    if (lane == 63) {
        if (f.sums_out) { f.sums_out[0] = t0; f.sums_out[1] = t1; }
        if (f.loss_out) *f.loss_out = chamfer_loss_from_sums(t0, t1, f.N, f.M, 3, f.Bg, f.w1, f.w2);
        __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for reuse
    }
    return 1;
}
constexpr int kHThreads = 1024;   // 16 waves share one LDS image: 1 block per CU, 4 waves per SIMD
#ifndef FX3D_HLT
#define FX3D_HLT 2
#endif
constexpr int kHLT = FX3D_HLT;    // 32-candidate blocks per lane tile (lane sees 16 rows of each)
#ifndef FX3D_HFIFO
#define FX3D_HFIFO 5
#endif
constexpr int kHFifo = FX3D_HFIFO;  // minima tracked per lane and pass (round 6: of 32-candidate BLOCKS, five of them -- same-box, C2's shape:
                                    // lane tiles x 4: uniform 42.0 us kernel, clusters A != B 69, lattice A != B 70; blocks x 4: 38.1 / 93 / 68;
                                    // blocks x 5: 38.5 / 77 / 55; blocks x 6: 38.9 / 78 / 52).  Round 5 (lane tiles): FOUR instead of three -- a query
                                    // whose band holds four lane tiles no longer sends its wave into the retry pass (the reference harness's
                                    // collinear A == B input at n = 16384: 68.4 -> 55.0 us, a 1/16 lattice A == B at C2's shape 109 -> 88;
                                    // uniform C2 unchanged, 51.0 -> 50.5 .. 50.9 on one box: the extra v_med3 per lane tile hides under the MFMAs);
                                    // five / six cost uniform C2 0.5 - 1.4 us and make the A != B lattice slower (99 -> 122 / 129 us)
constexpr int kHChunkMax = 4096;  // 32 B per candidate => 128 KiB
#ifndef FX3D_HRUNS_FROM
#define FX3D_HRUNS_FROM 16
#endif
constexpr int kHRunsFrom = FX3D_HRUNS_FROM;  // slow queries in a wave from which its retry pass enqueues runs instead of lane tiles
constexpr int kScanU = 2;  // runs of four candidates a lane has in flight in the exact scan of ONE slow query (3 measured equal; 4 spills)
#ifndef FX3D_HRUNCAP
#define FX3D_HRUNCAP 384
#endif
constexpr int kHRunCap = FX3D_HRUNCAP;        // the retry pass's list of RUN items (four consecutive candidates of one query each): a 32-candidate block
                                    // appends at most 256, the list is drained when fewer are free
constexpr int kHItemCap = 64 * kHFifo > kHRunCap ? 64 * kHFifo : kHRunCap;  // (the FIFO path never overflows its 64 * kHFifo)
constexpr int kHFarCap = 64;      // far candidates kept on the exact side list; more: the chunk falls back to exact scans
constexpr int kHTail = 64;        // a cloud of up to kHChunkMax + kHTail points stays one LDS image: the last <= 64 candidates are
                                  // compared exactly by every query (N = M = 4097 was 2.1 x N = M = 4096: two half-empty chunks,
                                  // three rounds of blocks), and <= 64 queries beyond a block's passes are one more pass of one wave
#ifndef FX3D_HBLKTRACK
#define FX3D_HBLKTRACK 1
#endif
// (round 6) the minima are tracked per 32-candidate BLOCK instead of per lane tile of two: an item of the exact phase is then a
// lane's 16 rows of one block (four runs of four candidates) instead of 32 rows -- the phase's VALU work, the SIMDs' busiest stretch,
// halves for 2.5 more VALU per MFMA in the main loop, which pruning has made short.
constexpr bool kHBlkTrack = FX3D_HBLKTRACK != 0;
constexpr int kHIdBits = kHBlkTrack ? 7 : 6;
static_assert(kHChunkMax / (32 * kHLT) <= 64, "lane-tile ids live in the six (block ids: seven) low mantissa bits of the tracked keys");
// Tracking keys: a lane tile's (block's) minimum with its id in its kHIdBits low mantissa bits (one v_and_or): |key - t| < 2^-17 |t|
// (2^-16).  kKeyUp turns a key into an upper bound of the value it came from (and a threshold on values into one on keys).
constexpr float kKeyUp = kHBlkTrack ? 0x1.2p-16f : 0x1.2p-17f;
constexpr float kPadF16 = 65504.0f;  // K slot 15: padding / far candidates get 65504 x 65504 = 4.3e9, finite and above every real
                                     // filter value (|t| <= 3 2^14 (1 + beta) + 128 S, S < 3e4): no +Inf in the image, no NaN keys
// (PRUNE) the cells t = ux | uy << 3 | uz << 6 of the 8 x 8 x 8 sorting grid along a 3-D Hilbert curve (Skilling's transpose form, generated
// offline): consecutive cells share a face, so 64 consecutive image rows -- a lane tile -- and 32 consecutive queries -- a wave -- are compact
// in space even where a surface leaves most cells empty.  (Morton order jumps across the grid between siblings: lane tiles straddling a
// jump had boxes half the cloud wide; uniform clouds ran 35 % of their lane tiles, 22 % along this curve.)
__device__ const unsigned short kHilbertCell[512] = {
    0, 1, 65, 64, 72, 73, 9, 8, 16, 24, 88, 80, 81, 89, 25, 17, 18, 26, 27, 19, 83, 91, 90, 82, 74, 10, 11, 75, 67, 3, 2, 66,
    130, 194, 195, 131, 139, 203, 202, 138, 146, 154, 155, 147, 211, 219, 218, 210, 209, 217, 153, 145, 144, 152, 216, 208, 200, 201, 137, 136, 128, 129, 193, 192,
    256, 257, 265, 264, 328, 329, 321, 320, 384, 448, 456, 392, 393, 457, 449, 385, 386, 450, 451, 387, 395, 459, 458, 394, 330, 322, 323, 331, 267, 259, 258, 266,
    274, 282, 283, 275, 339, 347, 346, 338, 402, 466, 467, 403, 411, 475, 474, 410, 409, 473, 465, 401, 400, 464, 472, 408, 344, 345, 337, 336, 272, 273, 281, 280,
    288, 296, 297, 289, 353, 361, 360, 352, 416, 480, 481, 417, 425, 489, 488, 424, 432, 496, 504, 440, 441, 505, 497, 433, 369, 368, 376, 377, 313, 312, 304, 305,
    306, 307, 315, 314, 378, 379, 371, 370, 434, 498, 506, 442, 443, 507, 499, 435, 427, 491, 490, 426, 418, 482, 483, 419, 355, 363, 362, 354, 290, 298, 299, 291,
    227, 235, 171, 163, 162, 170, 234, 226, 225, 224, 160, 161, 169, 168, 232, 233, 241, 240, 248, 249, 185, 184, 176, 177, 178, 242, 250, 186, 187, 251, 243, 179,
    115, 51, 59, 123, 122, 58, 50, 114, 113, 112, 120, 121, 57, 56, 48, 49, 41, 40, 104, 105, 97, 96, 32, 33, 34, 42, 106, 98, 99, 107, 43, 35,
    36, 44, 108, 100, 101, 109, 45, 37, 38, 39, 103, 102, 110, 111, 47, 46, 54, 55, 63, 62, 126, 127, 119, 118, 117, 53, 61, 125, 124, 60, 52, 116,
    180, 244, 252, 188, 189, 253, 245, 181, 182, 183, 191, 190, 254, 255, 247, 246, 238, 239, 175, 174, 166, 167, 231, 230, 229, 237, 173, 165, 164, 172, 236, 228,
    292, 300, 301, 293, 357, 365, 364, 356, 420, 484, 485, 421, 429, 493, 492, 428, 436, 500, 508, 444, 445, 509, 501, 437, 373, 372, 380, 381, 317, 316, 308, 309,
    310, 311, 319, 318, 382, 383, 375, 374, 438, 502, 510, 446, 447, 511, 503, 439, 431, 495, 494, 430, 422, 486, 487, 423, 359, 367, 366, 358, 294, 302, 303, 295,
    287, 286, 278, 279, 343, 342, 350, 351, 415, 479, 471, 407, 406, 470, 478, 414, 413, 477, 476, 412, 404, 468, 469, 405, 341, 349, 348, 340, 276, 284, 285, 277,
    269, 261, 260, 268, 332, 324, 325, 333, 397, 461, 460, 396, 388, 452, 453, 389, 390, 454, 462, 398, 399, 463, 455, 391, 327, 326, 334, 335, 271, 270, 262, 263,
    199, 198, 134, 135, 143, 142, 206, 207, 215, 223, 159, 151, 150, 158, 222, 214, 213, 221, 220, 212, 148, 156, 157, 149, 141, 205, 204, 140, 132, 196, 197, 133,
    69, 5, 4, 68, 76, 12, 13, 77, 85, 93, 92, 84, 20, 28, 29, 21, 22, 30, 94, 86, 87, 95, 31, 23, 15, 14, 78, 79, 71, 70, 6, 7,
};
constexpr int kHGroupsMax = 8 * 16 + 2 + 1;  // (PRUNE) 32-query groups of a block's window: up to eight passes + the folded remainder; + one slot: the query cloud's box
constexpr size_t kHScratchBytes = (kHThreads / 64) * (32 * 8 + kHItemCap * 2 + 32 * 3 * 4) + 64 * 32;  // + 2 pad blocks

// plain v_min_f32 (fminf() also emits a canonicalising v_max in IEEE mode; the filter values are never
// signalling NaNs, and a NaN filter value only sends the query down the exact path)
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ void split2h(float v, _Float16 &h, _Float16 &l) {
    h = (_Float16)v;
    l = (_Float16)(v - (float)h);
}

// fp16 image pieces of one candidate: c~ = s (c - mu)
// Error model of the fp16-split filter value t^ of candidate c for query q (scaled units, any magnitudes):
//   |t^ - (|c~|^2 - 2 q~.c~)| <= beta (|c~|^2 + |q~|^2) + floor,   beta = 2^-18
// (2-way splits 3 2^-22 |c~_d||m_d|, fp32 accumulation of 16 exact products 2^-20 (2|c~|^2 + |q~|^2), norm split;
// >= the absolute bound 2^-20 (4 + 2S) used while |c~| <= 1, which tools/test_f16_filter.hip showed 9x pessimistic).
// The candidate's share is folded into its norm (x (1 + kBetaC): the image holds UPPER bounds U_c), so that a far
// point's large norm inflates only its own value.  The nearest candidate c* then satisfies
//   U_c* <= Umin + 2 beta |c~*|^2 + 2 beta |q~|^2,   |c~*|^2 <= (|q~| + d*)^2 <= (1 + e)|q~|^2 + (1 + 1/e) d*^2  (any e > 0),
//   d*^2 <= Umin + (1 + beta)|q~|^2;   with e = 1/8:   U_c* <= Umin (1 + 18 beta) + 22.3 beta |q~|^2 + floor.
// (Umin ~ -|q~|^2 + d*^2 and d* << |q~| for most queries: the band is ~4.3 beta |q~|^2 there, e = 1 would give 6.)
// The oracle's own rounding (6u of the distances) adds 2^-20 (Umin + 2|q~|^2): kBandB1, kBandA below.
constexpr float kBetaC = 0x1.1p-18f;                       // beta (+6 %: rounding of n (1 + beta), subnormal share 2^-26 sqrt(3))
constexpr float kBandB1 = 1.0f + 18.0f * kBetaC + 0x1p-20f;
constexpr float kBandA = 22.3f * kBetaC + 0x1p-19f;
__device__ __forceinline__ void make_pieces(float cx, float cy, float cz, h8 &p0, h8 &p1) {
    _Float16 hx, lx, hy, ly, hz, lz, n1, n2, n3;
    split2h(cx, hx, lx); split2h(cy, hy, ly); split2h(cz, hz, lz);
    const float n0 = ((cx * cx) + (cy * cy)) + (cz * cz);
    const float n = n0 + kBetaC * n0;
    n1 = (_Float16)n;
    const float r1 = n - (float)n1;
    n2 = (_Float16)r1;
    n3 = (_Float16)(r1 - (float)n2);
    const _Float16 z = (_Float16)0.0f;
    p0 = h8{hx, hx, lx, hy, hy, ly, hz, hz};
    p1 = h8{lz, n1, n2, n3, lx, ly, lz, z};
}

// SPLITFUSE: the instantiation of candidate-split runs with the merge fused into the kernel's tail (a template parameter, not a
// run-time flag: with the merge code inside, the one-chunk kernel -- the headline path -- spilled a register: 1 MB of scratch
// writes per launch at C2, where that code never runs)
template <bool WANT_IDX, bool SPLITFUSE = false, bool PRUNE = false>
__global__ __launch_bounds__(kHThreads, 4) void nn1_f16_kernel(Nn1Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * (kHThreads / 64) + 16];  // per wave: min, max, sum, sum of squares (rows padded to 4 floats); + the block's result (+ the bounding box: PRUNE)
    __shared__ int nfar[2];                          // candidates of the chunk beyond the robust range (chunks alternate) ...
    __shared__ unsigned short farlist[kHFarCap];     // ... their indices within the chunk: compared exactly by every query
    constexpr int QB = (kHThreads / 64) * 32;  // queries per tile pass

    const int L = blockIdx.x;
    const int per_cloud = p.tiles * p.nsplit;
    // >= 8 clouds: all blocks of a cloud get ids with equal L%8, i.e. one XCD and one L2 (block L runs on
    // XCD L%8).  Fewer clouds than XCDs: plain order, so that a cloud's blocks spread over every XCD.
    const bool by_xcd = 2 * p.B >= 8;
    const int slot = by_xcd ? L >> 3 : L;
    const int c = by_xcd ? (slot / per_cloud) * 8 + (L & 7) : slot / per_cloud;
    const int tile = (slot % per_cloud) / p.nsplit;
    const int split = (slot % per_cloud) % p.nsplit;
    if (c >= 2 * p.B) return;
    const int dir = c >= p.B ? 1 : 0;
    const int b = dir ? c - p.B : c;
    const int NQ = dir ? p.M : p.N;
    const int NC = dir ? p.N : p.M;
    if (tile >= (dir ? p.tiles_y : p.tiles_x)) return;
    if ((long long)split * p.chunk >= NC) return;  // this direction has fewer chunks than splits
    const int ctail = (p.tail > 0 && p.nsplit == 1 && NC > p.chunk && NC - p.chunk <= p.tail) ? NC - p.chunk : 0;
    const int NCm = NC - ctail;                    // candidates that go through the filter (the chunk loop)
    const float *__restrict__ qb = (dir ? p.y : p.x) + (size_t)b * NQ * 3;
    const float *__restrict__ cb = (dir ? p.x : p.y) + (size_t)b * NC * 3;
    int32_t *idx_out = dir ? p.idx_y : p.idx_x;
    float *dmin_out = dir ? p.dmin_y : p.dmin_x;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: everything derived from it lives in scalar registers
    const int jq = lane & 31, hh = lane >> 5;
    const int CH = p.chunk;
    const int tpb = dir ? p.tpb_y : p.tpb;
    h8 *imgp = reinterpret_cast<h8 *>(lds);  // piece (blk, half, row) at (blk*2 + half)*32 + row, 16 B each
    float4 *imgf = reinterpret_cast<float4 *>(lds);
    unsigned long long *wres = reinterpret_cast<unsigned long long *>(lds + 8 * (CH + 64));  // image + 2 pad blocks
    unsigned short *witems = reinterpret_cast<unsigned short *>(wres + (kHThreads / 64) * 32);  // items: (query << 7) | (half << 6) | lane tile
    float *wq = reinterpret_cast<float *>(witems + (kHThreads / 64) * kHItemCap);
    unsigned long long *qres = wres + wv * 32;
    unsigned short *items = witems + wv * kHItemCap;
    float *qtab = wq + wv * 96;
    const bool vec = (reinterpret_cast<uintptr_t>(cb) & 15) == 0;
    const bool one_shot = NCm <= CH;
    if (tid == 0) { nfar[0] = 0; nfar[1] = 0; }  // (ordered before their first use by the barrier of the bounding-box pass)
    float qpre[3];  // this lane's query of the coming tile pass (the load's latency hides behind the prologue)
    {
        const int q0i = (tile * tpb) * QB + wv * 32 + jq;
        const int qc0 = q0i < NQ ? q0i : NQ - 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) qpre[d] = qb[(size_t)qc0 * 3 + d];
    }
    if constexpr (PRUNE) {  // the sort's counters (in the waves' scratch, idle until the passes) and the query groups' boxes: the cloud pass's barrier orders this
        unsigned int *zc = reinterpret_cast<unsigned int *>(wres);
        for (int i = tid; i < 1040 + 16 * 256; i += kHThreads) zc[i] = 0u;
        int *gb = reinterpret_cast<int *>(wq + (kHThreads / 64) * 96) + 64 * 8;
        for (int i = tid; i < kHGroupsMax * 8; i += kHThreads) gb[i] = (i & 4) ? (int)0x80000000 : 0x7fffffff;  // lo keys: +max, hi keys: -max
    }
    FX3D_PROBE_MARK(0);

    // ---- bounding box and mean -> centre mu = the MEAN (a stray far point moves the box centre, hardly the mean), largest
    //      |c - mu| cinf, power-of-two scale sc with cinf*sc in [64,128): seven binades of fp16 range above 1, so that a bulk
    //      much smaller than the farthest point keeps its fp16 pieces out of the subnormals; |c~|^2 < 3 * 2^14 fits fp16 ----
    // Scope of the statistics below (centre, extent, spread -> the image's centre and scale): the whole cloud -- or, in a
    // candidate-split run whose block takes ONE chunk, that chunk alone (round 5: any centre is correct and the scale only has to
    // cover the candidates of THIS block's image; C3's blocks walked all 5000 points for an image of 1728, and read their chunk
    // a second time for the image: it is parked in LDS during the pass now, as in one-chunk runs)
    const bool own_chunk_only = p.nsplit > 1 && (long long)split * CH + (long long)p.nsplit * CH >= NCm;
    const int s_lo = own_chunk_only ? split * CH : 0;
    const int SN = own_chunk_only ? (NCm - s_lo < CH ? NCm - s_lo : CH) : NC;
    const float *__restrict__ sb = cb + (size_t)s_lo * 3;
    const bool parked = one_shot || own_chunk_only;  // the raw points of the block's (only) chunk sit in their image slots
    float mu[3], cinf = 0.0f, varmax = 0.0f;
    float bxlo[3] = {0.f, 0.f, 0.f}, bxhi[3] = {0.f, 0.f, 0.f};  // (PRUNE) the cloud's bounding box: the frame of the sorting grid
    bool allfin = true;
    {
        // thread t takes points t, t + 1024, ...: 12-byte loads, consecutive lanes on consecutive points (coalesced, and the
        // 16-byte LDS slots of a wave's points are consecutive: no bank conflicts when they are parked and converted)
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, sm3[3] = {0.f, 0.f, 0.f};
        float sqt = 0.0f;  // second moment (all three coordinates) of a SAMPLE about the cloud's first point -- the spread: wave w
                           // takes its points of every fourth sweep (64-point runs all over the cloud, ~SN/4 points)
        const float pil[3] = {sb[0], sb[1], sb[2]};
        const int nsweep = (SN + kHThreads - 1) / kHThreads;
        for (int i0 = 0; i0 < nsweep; i0 += 4) {
            P3 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = (i0 + e) * kHThreads + tid;
                v[e] = *reinterpret_cast<const P3 *>(sb + (size_t)(pt < SN ? pt : SN - 1) * 3);  // (clamped: always valid)
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = (i0 + e) * kHThreads + tid;
                mn[0] = fminf(mn[0], v[e].x); mx[0] = fmaxf(mx[0], v[e].x);
                mn[1] = fminf(mn[1], v[e].y); mx[1] = fmaxf(mx[1], v[e].y);
                mn[2] = fminf(mn[2], v[e].z); mx[2] = fmaxf(mx[2], v[e].z);
                if (pt < SN) {
                    sm3[0] = sm3[0] + v[e].x; sm3[1] = sm3[1] + v[e].y; sm3[2] = sm3[2] + v[e].z;
                    if (parked) imgf[((pt >> 5) * 2) * 32 + (pt & 31)] = float4{v[e].x, v[e].y, v[e].z, 0.0f};  // parked in its own first piece
                }
                if (((i0 + e) & 3) == (wv & 3) && pt < SN) {  // (the first condition is wave-uniform)
                    sqt = __builtin_fmaf(v[e].x - pil[0], v[e].x - pil[0], sqt); sqt = __builtin_fmaf(v[e].y - pil[1], v[e].y - pil[1], sqt);
                    sqt = __builtin_fmaf(v[e].z - pil[2], v[e].z - pil[2], sqt);
                }
            }
        }
        FX3D_PROBE_MARK(5);
        {
            float4 lo4, hi4, st4;
            lo4.x = wave_min_l63(mn[0]); lo4.y = wave_min_l63(mn[1]); lo4.z = wave_min_l63(mn[2]); lo4.w = 0.0f;
            hi4.x = wave_max_l63(mx[0]); hi4.y = wave_max_l63(mx[1]); hi4.z = wave_max_l63(mx[2]); hi4.w = 0.0f;
            st4.x = wave_sum_l63(sm3[0]); st4.y = wave_sum_l63(sm3[1]); st4.z = wave_sum_l63(sm3[2]); st4.w = wave_sum_l63(sqt);
            if (lane == 63) {
                float4 *r4 = reinterpret_cast<float4 *>(red);
                r4[wv * 4] = lo4; r4[wv * 4 + 1] = hi4; r4[wv * 4 + 2] = st4;
            }
        }
        FX3D_PROBE_MARK(9);
        __syncthreads();
        FX3D_PROBE_MARK(10);
        static_assert(kHThreads / 64 == 16, "the cross-wave reduction below: one wave's row of 16 lanes, one lane per wave");
        if (wv == 0) {
            // cross-wave reduction by the first wave alone: lane l of each row reads wave l's four statistics (16-byte reads),
            // four DPP steps reduce over the row; the result goes back through LDS (every other wave waits at the barrier
            // instead of repeating 64 LDS reads and 190 operations)
            const float4 *r4 = reinterpret_cast<const float4 *>(red);
            const int w = lane & 15;
            float4 lo4 = r4[w * 4], hi4 = r4[w * 4 + 1], st4 = r4[w * 4 + 2];
#define NN1_ROW_MIN(v) v = fkey_inv(row_mm_key<false>(fkey(v)));
#define NN1_ROW_MAX(v) v = fkey_inv(row_mm_key<true>(fkey(v)));
#define NN1_ROW_SUM(v) v = v + dpp_mov<0xB1>(v); v = v + dpp_mov<0x4E>(v); v = v + dpp_mov<0x141>(v); v = v + dpp_mov<0x140>(v);
            NN1_ROW_MIN(lo4.x) NN1_ROW_MIN(lo4.y) NN1_ROW_MIN(lo4.z)
            NN1_ROW_MAX(hi4.x) NN1_ROW_MAX(hi4.y) NN1_ROW_MAX(hi4.z)
            NN1_ROW_SUM(st4.x) NN1_ROW_SUM(st4.y) NN1_ROW_SUM(st4.z) NN1_ROW_SUM(st4.w)
#undef NN1_ROW_MIN
#undef NN1_ROW_MAX
#undef NN1_ROW_SUM
            const float lo3[3] = {lo4.x, lo4.y, lo4.z}, hi3[3] = {hi4.x, hi4.y, hi4.z}, st3[3] = {st4.x, st4.y, st4.z};
            // total variance about the mean from the sampled second moment about the first point (a data point: no cancellation for
            // clouds far from the origin); the sample holds ~SN/4 points (the gate below is a heuristic: any estimate is correct)
            float m3[3], ci = 0.0f, vm = st4.w / fmaxf(0.25f * (float)SN, 1.0f);
            bool fin = true;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float lo = lo3[d], hi = hi3[d], st = st3[d];
                m3[d] = fminf(fmaxf(st / (float)SN, lo), hi);  // (any centre is correct)
                const float off = st / (float)SN - pil[d];
                vm = vm - off * off;
                ci = fmaxf(ci, fmaxf(hi - m3[d], m3[d] - lo));
                // a NaN or +-Inf coordinate makes the coordinate sum non-finite (fminf / fmaxf above skip NaNs)
                fin = fin && fabsf(st) < INFINITY;
            }
            if (lane == 0) {
                float4 *o4 = reinterpret_cast<float4 *>(red) + 4 * (kHThreads / 64);
                o4[0] = float4{m3[0], m3[1], m3[2], ci * 1.000001f};
                o4[1] = float4{vm, fin ? 1.0f : 0.0f, 0.0f, 0.0f};
                if (PRUNE) { o4[2] = float4{lo3[0], lo3[1], lo3[2], 0.0f}; o4[3] = float4{hi3[0], hi3[1], hi3[2], 0.0f}; }
            }
        }
        __syncthreads();
        {
            const float4 *o4 = reinterpret_cast<const float4 *>(red) + 4 * (kHThreads / 64);
            const float4 a = o4[0], b2 = o4[1];
            mu[0] = a.x; mu[1] = a.y; mu[2] = a.z; cinf = a.w;
            varmax = b2.x; allfin = b2.y != 0.0f;
            if (PRUNE) {
                const float4 l4 = o4[2], h4 = o4[3];
                bxlo[0] = l4.x; bxlo[1] = l4.y; bxlo[2] = l4.z; bxhi[0] = h4.x; bxhi[1] = h4.y; bxhi[2] = h4.z;
            }
        }
    }
    // ---- robust range: a few points far from the bulk must not set the scale (the bulk would sink below fp16's resolution and
    //      the whole cloud fall back to exact scans: one point 10^5 x out cost 11 x the uniform time in round 1).
    //      rng = min(cinf, 16 x the mean max-norm deviation from mu): clouds without outliers keep rng = cinf (uniform box:
    //      16 x 0.375 of the half width; Gaussian: 21 sigma); candidates beyond the range ("far") are left out of the filter
    //      (norm = +inf) and compared exactly by every query through a side list of at most kHFarCap entries per chunk. ----
    float rng = cinf;
    // (clean clouds never get here -- uniform boxes have cinf^2 = 3 var, Gaussians of a million points 30 var --; a round costs
    //  two barriers, ~1.3 us at C2; all three run only when the first finds the bulk kRobustHarm x below the farthest point)
    if (allfin && cinf < 1.0e16f && 3.0f * cinf * cinf > kRobustGate * varmax) {  // (varmax: the TOTAL variance of the three coordinates)
        const float4 r = robust_range3<kHThreads, false>(sb, SN, parked, imgf, red, mu[0], mu[1], mu[2], cinf);
        mu[0] = r.x; mu[1] = r.y; mu[2] = r.z; rng = r.w;
    }
    // not sane (non-finite or huge coordinates): the filter is unusable, every query of the block scans every lane
    // tile exactly, in the isless order (a finite cloud with cinf < 1e16 never produces an infinite or NaN distance
    // to a query inside the fp16 range)
    const bool sane = allfin && cinf < 1.0e16f;  // usable queries lie within 234 cinf of the centre: 3 (235 cinf)^2 stays finite
    float sc = 1.0f;
    if (sane && rng > 1.0e-30f) {
        int e;
        (void)frexpf(rng, &e);  // rng = m 2^e, m in [0.5,1)
        sc = ldexpf(1.0f, 7 - e);
    }
    // (block-uniform values the compiler cannot prove uniform -- they came through LDS --: into scalar registers, the main
    //  loop needs the vector ones; a kernel that spills pays ~1.2 us per launch for the scratch set-up)
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    mu[0] = uni(mu[0]); mu[1] = uni(mu[1]); mu[2] = uni(mu[2]); sc = uni(sc); rng = uni(rng); cinf = uni(cinf);
    // pieces of candidate `pt` (index within the chunk); a far one leaves the filter (t = +inf) for the side list
    const bool has_far = sane && rng < cinf;  // (uniform) clean clouds skip the test below
    int fslot = 0;                             // nfar[fslot] counts this chunk's far candidates
    auto pieces = [&](float x, float y, float z, int pt, h8 &p0, h8 &p1) {
        const float sx = (x - mu[0]) * sc, sy = (y - mu[1]) * sc, sz = (z - mu[2]) * sc;
        if (!has_far) { make_pieces(sx, sy, sz, p0, p1); return; }
        const bool far = !(fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz)) < 128.0f);
        make_pieces(far ? 0.f : sx, far ? 0.f : sy, far ? 0.f : sz, p0, p1);
        if (far) {
            p1[7] = (_Float16)kPadF16;
            const int f = atomicAdd(&nfar[fslot], 1);
            if (f < kHFarCap) farlist[f] = (unsigned short)pt;
        }
    };

    // ---- (PRUNE, round 6) SPATIAL ORDER.  tools/ubench_overlap.hip (profiles/r06_ubench_overlap.txt) shows the main loop at the
    //      SIMD's VALU-issue bound for its instruction mix -- no schedule of the same pairs is faster -- so the pairs have to go:
    //      the candidates are laid into the image along a Hilbert curve through an 8 x 8 x 8 grid over the cloud's bounding box (counting sort:
    //      LDS integer atomics, one scan), every lane tile (64 consecutive image rows) gets a bounding box, the block takes its
    //      queries as a WINDOW of the query cloud in the same order (32 consecutive ones per wave: a compact box), and a wave runs
    //      the filter only over the lane tiles whose box can hold a nearest neighbour of one of its queries (main loop below).
    //      Results do not change: a skipped lane tile holds no candidate as near as the one already found (bounds below), ties
    //      included, and everything that decides a result -- the exact distances, the 64-bit (distance, ORIGINAL index) minima --
    //      is as before.  The image order of candidates inside a cell is the arrival order of LDS atomics (it only moves a
    //      candidate between lane tiles); the order of the QUERIES is made unique (cell, then index), because it decides which
    //      block and lane a query's term of the loss is summed in.  The rows (x, y, z, original index) of both orders go to the
    //      block's scratch: the exact phase reads candidates by image position from there, the passes read their queries. ----
    [[maybe_unused]] float4 *cs = nullptr, *qsw = nullptr;
    [[maybe_unused]] bool sorted_c = false;
    [[maybe_unused]] int qwlen = 0;
    [[maybe_unused]] float4 *boxes = nullptr;
    if constexpr (PRUNE) {
        cs = p.pscr + (size_t)blockIdx.x * p.pscr_stride;
        qsw = cs + kHChunkMax;
        boxes = reinterpret_cast<float4 *>(wq + (kHThreads / 64) * 96);  // [64 lane tiles] x (lo, hi): 2 KiB behind the waves' scratch
        // (the waves' scratch is idle here: 20.5 of its 22 KiB hold the counters)  Cells are numbered t = ux | uy << 3 | uz << 6; the
        // SCANS walk them along the Hilbert curve (kHilbertCell), so the image order is the curve's order of the cells at two shifts per point.
        unsigned int *ccnt = reinterpret_cast<unsigned int *>(wres);  // [512] candidates per cell -> first image row of the cell
        unsigned int *qtot = ccnt + 520;                               // [512] queries per cell -> first rank of the cell
        unsigned int *qwh = qtot + 520;                                // [16 waves][256] per-wave counts, two 16-bit cells per word -> counts of the waves before
        const bool last_t = tile == (dir ? p.tiles_y : p.tiles_x) - 1;
        const int qw0 = tile * tpb * QB;
        const int qw1 = last_t || qw0 + tpb * QB > NQ ? NQ : qw0 + tpb * QB;
        qwlen = qw1 - qw0;
        const int cnt = NCm;  // (PRUNE launches are one-chunk plans without a tail)
        const bool do_sort = sane && cnt >= 512;
        // The queries' order must not depend on timing (it decides in which block and lane a query's term of the loss is summed):
        // wave w counts the queries 256 w .. 256 w + 255 in counters of its own -- returning LDS atomics of ONE wave are served in
        // program order, lanes of one instruction in the hardware's fixed order --, a query's rank is (queries in cells before its
        // cell) + (queries of its cell in the waves before) + (the value its own atomic returned).
        // The queries are ordered in a grid over THEIR OWN bounding box (the clouds of a pair need not overlap, nor be of one size: in the
        // candidates' frame a chair's points fell outside an aeroplane's flat box and crowded into its border cells), found by the
        // block while the candidates are counted: integer-key min / max over the wave (DPP), 6 LDS atomics per wave.
        const bool sorted_q = do_sort && NQ <= 4 * kHThreads;
        P3 qv[4];
        if (sorted_q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int qp = 256 * wv + 64 * e + lane;
                qv[e] = *reinterpret_cast<const P3 *>(qb + (size_t)(qp < NQ ? qp : NQ - 1) * 3);
            }
        }
        FX3D_PROBE_MARK2(0);
        int *gbox = reinterpret_cast<int *>(boxes + 64 * 2);  // [groups of 32 window rows] x (lo keys, -, hi keys, -)
        auto group_box = [&](int wrow, float x, float y, float z) {  // a query of the window: into the box of its group (image frame)
            int *g = gbox + (wrow >> 5) * 8;
            atomicMin(g + 0, fkey((x - mu[0]) * sc)); atomicMin(g + 1, fkey((y - mu[1]) * sc)); atomicMin(g + 2, fkey((z - mu[2]) * sc));
            atomicMax(g + 4, fkey((x - mu[0]) * sc)); atomicMax(g + 5, fkey((y - mu[1]) * sc)); atomicMax(g + 6, fkey((z - mu[2]) * sc));
        };
        if (do_sort) {
            float4 cv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = e * kHThreads + tid;
                cv[e] = float4{0.f, 0.f, 0.f, 0.f};
                if (pt < cnt) cv[e] = imgf[((pt >> 5) * 2) * 32 + (pt & 31)];  // parked by this thread
            }
            FX3D_PROBE_MARK2(1);
            float inv[3], off[3], cfl[3], cfh[3];  // (cfl, cfh: the candidates' grid frame)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                // the grid spans the cloud's box, clipped to the filter's range about the centre (|c~| < 128): far outliers -- they
                // stay out of the filter, on the side list -- fall into border cells instead of squeezing the bulk into one cell
                // (with far outliers rng = 16 x the bulk's mean max-norm deviation -- 12 half widths of a uniform box, 21 sigma of a
                //  Gaussian --: an eighth of it about the centre holds the bulk)
                const float fr = has_far ? 0.125f * rng : 128.0f / sc;
                const float flo = fmaxf(bxlo[d], mu[d] - fr), fhi = fminf(bxhi[d], mu[d] + fr);
                inv[d] = fhi > flo ? 8.0f / (fhi - flo) : 0.0f;
                off[d] = -flo * inv[d];
                cfl[d] = flo; cfh[d] = fhi;
            }
            auto cell = [&](float x, float y, float z) -> unsigned int {  // the 8 x 8 x 8 cell of a point (outside the box: a border cell)
                const unsigned int ux = (unsigned int)(int)fminf(fmaxf(__builtin_fmaf(x, inv[0], off[0]), 0.0f), 7.0f);
                const unsigned int uy = (unsigned int)(int)fminf(fmaxf(__builtin_fmaf(y, inv[1], off[1]), 0.0f), 7.0f);
                const unsigned int uz = (unsigned int)(int)fminf(fmaxf(__builtin_fmaf(z, inv[2], off[2]), 0.0f), 7.0f);
                return ux | (uy << 3) | (uz << 6);
            };

            unsigned int ccr[4], qcr[4];  // (cell << 16) | rank among the cell's points (candidates: arrival; queries: within the wave)
            // Lanes of one instruction that share a cell serialise on its LDS address (a cloud that is a point, a tight cluster beside
            // far outliers, queries outside the candidates' box that all clamp into one corner cell: 4096 same-address atomics cost a
            // block 30 k cycles): up to two crowded cells per instruction -- the first uncounted lane's, if eight or more lanes share
            // it -- are counted by ONE atomic each, their lanes ranked in lane order.
            auto count_cells = [&](bool valid, unsigned int t, auto &&add) -> unsigned int {  // add(cell, n) -> the count before; returns this lane's rank
                unsigned int rank = 0u;
                bool mine = valid;
                // crowded?  eight or more lanes whose neighbour lane sits in the same cell (uniform data: none) -- then the cells of the
                // instruction are taken one by one, a cell with four or more lanes by ONE atomic (ranks in lane order)
                const unsigned int tn = (unsigned int)__builtin_amdgcn_update_dpp((int)~t, (int)t, 0x111, 0xF, 0xF, false);  // row_shr:1
                if (__builtin_popcountll(__ballot(valid && t == tn)) >= 8) {
                    unsigned long long rem = __ballot(valid);
                    for (int it = 0; it < 12 && rem; ++it) {
                        const unsigned int t0 = (unsigned int)__builtin_amdgcn_readlane((int)t, __builtin_ctzll(rem));
                        const unsigned long long grp = __ballot(mine && t == t0);
                        rem &= ~grp;
                        if (__builtin_popcountll(grp) < 4) continue;
                        unsigned int base = 0u;
                        if ((int)lane == __builtin_ctzll(grp)) base = add(t0, (unsigned int)__builtin_popcountll(grp));
                        base = (unsigned int)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(grp));
                        if (mine && t == t0) {
                            rank = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(grp >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)grp, 0u));
                            mine = false;
                        }
                    }
                }
                if (mine) rank = add(t, 1u);
                return rank;
            };
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = e * kHThreads + tid;
                const unsigned int t = pt < cnt ? cell(cv[e].x, cv[e].y, cv[e].z) : 0u;
                ccr[e] = (t << 16) | count_cells(pt < cnt, t, [&](unsigned int c, unsigned int n) { return atomicAdd(&ccnt[c], n); });
            }
            int *qbb = reinterpret_cast<int *>(boxes + 64 * 2 + kHGroupsMax * 2 - 2);  // the query cloud's box as keys: the spare group slot (initialised with the groups')
            if (sorted_q) {
                int kl[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, kh[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qp = 256 * wv + 64 * e + lane;
                    if (qp < NQ) {
                        const int k3[3] = {fkey(qv[e].x), fkey(qv[e].y), fkey(qv[e].z)};
#pragma unroll
                        for (int d = 0; d < 3; ++d) { kl[d] = kl[d] < k3[d] ? kl[d] : k3[d]; kh[d] = kh[d] > k3[d] ? kh[d] : k3[d]; }
                    }
                }
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    int a = row_mm_key<false>(kl[d]), b2 = row_mm_key<true>(kh[d]);
                    a = imm<false>(a, dpp_keepi<0x142, 0xA>(a)); a = imm<false>(a, dpp_keepi<0x143, 0xC>(a));
                    b2 = imm<true>(b2, dpp_keepi<0x142, 0xA>(b2)); b2 = imm<true>(b2, dpp_keepi<0x143, 0xC>(b2));
                    if (lane == 63) { atomicMin(qbb + d, a); atomicMax(qbb + 4 + d, b2); }
                }
            }
            FX3D_PROBE_MARK2(2);
            __syncthreads();
            FX3D_PROBE_MARK2(3);
            float qinv[3] = {0.f, 0.f, 0.f}, qoff[3] = {0.f, 0.f, 0.f};
            if (sorted_q) {
                // ... clipped to the candidates' frame grown by its largest extent on every side: queries farther out than that (a stray
                // far point would squeeze all others into one cell) fall into border cells
                const float ext = fmaxf(fmaxf(cfh[0] - cfl[0], cfh[1] - cfl[1]), cfh[2] - cfl[2]);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    // (a query box more than four times the candidates' extent in this dimension -- a far outlier among the queries --: grown by a
                    //  quarter only, so that the bulk, which then sits where the candidates are, is not squeezed into the middle cells)
                    const float ql = fkey_inv(qbb[d]), qh = fkey_inv(qbb[4 + d]);
                    const float grow = qh - ql > 4.0f * ext ? 0.25f * ext : ext;
                    const float lo = fmaxf(ql, cfl[d] - grow), hi = fminf(qh, cfh[d] + grow);  // (fmaxf / fminf drop a NaN bound)
                    qinv[d] = hi > lo && hi - lo < INFINITY ? 8.0f / (hi - lo) : 0.0f;  // (no extent, or not finite: one cell in this dimension)
                    qoff[d] = qinv[d] != 0.0f ? -lo * qinv[d] : 0.0f;
                }
            }
            auto qcell = [&](float x, float y, float z) -> unsigned int {
                const unsigned int ux = (unsigned int)(int)fminf(fmaxf(__builtin_fmaf(x, qinv[0], qoff[0]), 0.0f), 7.0f);
                const unsigned int uy = (unsigned int)(int)fminf(fmaxf(__builtin_fmaf(y, qinv[1], qoff[1]), 0.0f), 7.0f);
                const unsigned int uz = (unsigned int)(int)fminf(fmaxf(__builtin_fmaf(z, qinv[2], qoff[2]), 0.0f), 7.0f);
                return ux | (uy << 3) | (uz << 6);
            };
            // exclusive scan of a count array along the Hilbert curve by one wave (eight consecutive cells of the curve per lane)
            auto curve_scan = [&](unsigned int *a) {
                unsigned int tc[8];
                {
                    const uint4 h = *reinterpret_cast<const uint4 *>(kHilbertCell + 8 * lane);
                    tc[0] = h.x & 0xffffu; tc[1] = h.x >> 16; tc[2] = h.y & 0xffffu; tc[3] = h.y >> 16;
                    tc[4] = h.z & 0xffffu; tc[5] = h.z >> 16; tc[6] = h.w & 0xffffu; tc[7] = h.w >> 16;
                }
                unsigned int c8[8], tot = 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) { c8[k] = a[tc[k]]; tot += c8[k]; }
                unsigned int inc = tot;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned int v = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += v;
                }
                unsigned int run = inc - tot;
#pragma unroll
                for (int k = 0; k < 8; ++k) { a[tc[k]] = run; run += c8[k]; }
            };
            if (wv == 0) curve_scan(ccnt);  // the candidates' cells -> first image rows
            if (sorted_q) {                  // the queries' cells, counted per wave
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qp = 256 * wv + 64 * e + lane;
                    const unsigned int t = qp < NQ ? qcell(qv[e].x, qv[e].y, qv[e].z) : 0u;
                    qcr[e] = (t << 16) | count_cells(qp < NQ, t, [&](unsigned int c, unsigned int n) {
                                 const unsigned int sh = 16u * (c & 1u);
                                 const unsigned int old = atomicAdd(&qwh[wv * 256 + (c >> 1)], n << sh);
                                 atomicAdd(&qtot[c], n);
                                 return (old >> sh) & 0xffffu;
                             });
                }
            }
            FX3D_PROBE_MARK2(4);
            __syncthreads();
            FX3D_PROBE_MARK2(5);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = e * kHThreads + tid;
                if (pt < cnt) {
                    const int pos = (int)(ccnt[ccr[e] >> 16] + (ccr[e] & 0xffffu));
                    imgf[((pos >> 5) * 2) * 32 + (pos & 31)] = float4{cv[e].x, cv[e].y, cv[e].z, __builtin_bit_cast(float, pt)};
                }
            }
            sorted_c = true;
            if (sorted_q) {
                if (wv == 1) {
                    curve_scan(qtot);       // -> first rank of every cell
                } else if (wv >= 2 && wv < 6) {  // a pair of cells per thread: counts of the waves before, wave by wave
                    const int w2 = tid - 128;
                    unsigned int run = 0u;
#pragma unroll
                    for (int w = 0; w < 16; ++w) {
                        const unsigned int v = qwh[w * 256 + w2];
                        qwh[w * 256 + w2] = run;
                        run += v;  // (two 16-bit sums of at most 4096: no carry between them)
                    }
                }
            }
            FX3D_PROBE_MARK2(6);
            __syncthreads();  // (the image rows are in place: the staging below converts them where they lie)
            FX3D_PROBE_MARK2(7);
            if (sorted_q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qp = 256 * wv + 64 * e + lane;
                    if (qp < NQ) {
                        const unsigned int t = qcr[e] >> 16;
                        const int rank = (int)(qtot[t] + ((qwh[wv * 256 + (t >> 1)] >> (16u * (t & 1u))) & 0xffffu) + (qcr[e] & 0xffffu));
                        if (rank >= qw0 && rank < qw1) {
                            qsw[rank - qw0] = float4{qv[e].x, qv[e].y, qv[e].z, __builtin_bit_cast(float, qp)};
                            group_box(rank - qw0, qv[e].x, qv[e].y, qv[e].z);
                        }
                    }
                }
            }
        }
        if (!sorted_q) {  // index order: the block's window of the query cloud as it lies in memory
            // (the 32 rows of a group are 32 consecutive lanes here: their box by four DPP row steps, two lanes per group go to LDS --
            //  32 lanes on one LDS address would serialise)
            for (int q0 = qw0 + wv * 64; q0 < qw1; q0 += kHThreads) {
                const int qp = q0 + lane;
                const P3 r = *reinterpret_cast<const P3 *>(qb + (size_t)(qp < qw1 ? qp : qw1 - 1) * 3);  // (lanes past the window repeat its last row)
                if (qp < qw1) qsw[qp - qw0] = float4{r.x, r.y, r.z, __builtin_bit_cast(float, qp)};
                const int kx = fkey((r.x - mu[0]) * sc), ky = fkey((r.y - mu[1]) * sc), kz = fkey((r.z - mu[2]) * sc);
                const int lx = row_mm_key<false>(kx), ly = row_mm_key<false>(ky), lz = row_mm_key<false>(kz);
                const int hx = row_mm_key<true>(kx), hy = row_mm_key<true>(ky), hz = row_mm_key<true>(kz);
                if ((lane & 15) == 0 && qp < qw1) {
                    int *g = gbox + ((qp - qw0) >> 5) * 8;
                    atomicMin(g + 0, lx); atomicMin(g + 1, ly); atomicMin(g + 2, lz);
                    atomicMax(g + 4, hx); atomicMax(g + 5, hy); atomicMax(g + 6, hz);
                }
            }
        }
        // (the barrier behind the image staging below orders these stores -- and the image's -- before their readers)
        FX3D_PROBE_MARK2(8);
    }
    FX3D_PROBE_MARK(1);

    // Largest scaled norm^2 of a candidate inside the filter (|c~|_inf <= min(cinf sc, 128)): the far-query form of the band below
    const float cm2 = 3.0f * (fminf(cinf * sc, 128.0f) * fminf(cinf * sc, 128.0f));
    float qr[3], da = 0.0f;  // band: a tile qualifies while its minimum <= best * kBandB1 + da
    float da_far = 0.0f;     // ... or <= best (1 + 2^-20) + da_far, whichever is lower (round 4)
    [[maybe_unused]] float dpr = INFINITY;  // (PRUNE) scaled squared distance to the best candidate found <= its filter value + dpr
    bool qfin = true;        // this lane's query has finite coordinates
    int qi = 0;
    bool qok = true;
    h8 bq;
    double acc = 0.0;

    const int jfirst = split * CH, jstep = p.nsplit * CH;
    for (int j0 = jfirst; j0 < NCm; j0 += jstep) {
        const int cnt = (NCm - j0) < CH ? (NCm - j0) : CH;
        const int cnt_pad = (cnt + 32 * kHLT - 1) / (32 * kHLT) * (32 * kHLT);
        if (j0 > jfirst) __syncthreads();
        // ---- stage the fp16 split image ------------------------------------------------------------------
        if constexpr (PRUNE) {
// (PRUNE) the 16 threads of a DPP row convert the 64 image rows of ONE lane tile where the sort laid them (or where the cloud
            // pass parked them): row k 16 + (lane of the row) in step k -- 256 contiguous bytes per row and LDS access --, write them to
            // the block's scratch, and reduce the tile's box (image frame) with four row steps
            for (int tl = tid >> 4; tl * 64 < cnt_pad + 64; tl += kHThreads / 16) {
                float4 r4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pt = tl * 64 + k * 16 + (tid & 15);
                    r4[k] = float4{0.f, 0.f, 0.f, 0.f};
                    if (pt < cnt) r4[k] = imgf[((pt >> 5) * 2) * 32 + (pt & 31)];
                }
                float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pt = tl * 64 + k * 16 + (tid & 15), i0 = ((pt >> 5) * 2) * 32 + (pt & 31);
                    h8 p0, p1;
                    if (pt < cnt) {
                        pieces(r4[k].x, r4[k].y, r4[k].z, pt, p0, p1);
                        const int id = sorted_c ? __builtin_bit_cast(int, r4[k].w) : j0 + pt;
                        cs[pt] = float4{r4[k].x, r4[k].y, r4[k].z, __builtin_bit_cast(float, id)};
                        const float b3[3] = {(r4[k].x - mu[0]) * sc, (r4[k].y - mu[1]) * sc, (r4[k].z - mu[2]) * sc};
#pragma unroll
                        for (int d = 0; d < 3; ++d) { lo3[d] = vmin(lo3[d], b3[d]); hi3[d] = -vmin(-hi3[d], -b3[d]); }
                    } else {
                        make_pieces(0.f, 0.f, 0.f, p0, p1);
                        p1[7] = (_Float16)kPadF16;
                    }
                    imgp[i0] = p0;
                    imgp[i0 + 32] = p1;
                }
                float bl[3], bh[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) { bl[d] = fkey_inv(row_mm_key<false>(fkey(lo3[d]))); bh[d] = fkey_inv(row_mm_key<true>(fkey(hi3[d]))); }
                if ((tid & 15) == 0 && tl < 64) {
                    boxes[tl * 2] = float4{bl[0], bl[1], bl[2], 0.0f};
                    boxes[tl * 2 + 1] = float4{bh[0], bh[1], bh[2], 0.0f};
                }
            }
        } else
        for (int pt = tid; pt < cnt_pad + 64; pt += kHThreads) {
            const int i0 = ((pt >> 5) * 2) * 32 + (pt & 31);
            h8 p0, p1;
            if (pt < cnt) {
                if (parked) {
                    const float4 r = imgf[i0];  // parked by this thread in the bounding-box pass
                    pieces(r.x, r.y, r.z, pt, p0, p1);
                } else {
                    const P3 r = *reinterpret_cast<const P3 *>(cb + (size_t)(j0 + pt) * 3);
                    pieces(r.x, r.y, r.z, pt, p0, p1);
                }
            } else {  // padding: t = 4.3e9 (K slot 15), above every real value; the exact phase skips rows >= cnt
                make_pieces(0.f, 0.f, 0.f, p0, p1);
                p1[7] = (_Float16)kPadF16;
            }
            imgp[i0] = p0;
            imgp[i0 + 32] = p1;
        }
        __syncthreads();
        FX3D_PROBE_MARK(j0 == jfirst ? 2 : 6);
        FX3D_PROBE_MARKW(0);

        const int nf = nfar[fslot];          // far candidates of this chunk (valid after the barrier above)
        if (tid == 0) nfar[fslot ^ 1] = 0;   // the next chunk's counter (its staging starts behind the barrier at the loop top)
        fslot ^= 1;
        const bool far_ok = nf <= kHFarCap;  // more than the side list holds: this chunk's filter is not used
        // (the last tile of a direction goes on past tpb passes while queries are left: the planner folds a remainder of <= 64
        //  queries into it instead of giving them a block of their own)
        const bool last_tile = tile == (dir ? p.tiles_y : p.tiles_x) - 1;
        for (int tp = 0; tp < tpb || (last_tile && one_shot); ++tp) {
            if ((tile * tpb + tp) * QB >= NQ) break;  // uniform
            if constexpr (PRUNE) {
                if (tp * QB + wv * 32 >= qwlen) continue;  // (wave-uniform) no query of the window left for this wave: nothing to do, nothing to write
            }
            if (j0 == jfirst) {
                qi = (tile * tpb + tp) * QB + wv * 32 + jq;
                if constexpr (PRUNE) {  // row (tp QB + wave's 32 + jq) of the block's window of the query cloud, in processing order
                    const int wpos = tp * QB + wv * 32 + jq;
                    const float4 q4 = qsw[wpos < qwlen ? wpos : qwlen - 1];  // (lanes past the window: a valid query, no output)
                    qr[0] = q4.x; qr[1] = q4.y; qr[2] = q4.z;
                    qi = wpos < qwlen ? __builtin_bit_cast(int, q4.w) : NQ;
                } else if (tp == 0) {  // requested before the bounding-box pass
#pragma unroll
                    for (int d = 0; d < 3; ++d) qr[d] = qpre[d];
                } else {        // (not prefetched behind the previous pass: three registers live across the main loop would spill)
                    const int qc1 = qi < NQ ? qi : NQ - 1;
#pragma unroll
                    for (int d = 0; d < 3; ++d) qr[d] = qb[(size_t)qc1 * 3 + d];
                }
                float m0 = -2.0f * ((qr[0] - mu[0]) * sc), m1 = -2.0f * ((qr[1] - mu[1]) * sc), m2 = -2.0f * ((qr[2] - mu[2]) * sc);
                float S = (fabsf(m0) + fabsf(m1)) + fabsf(m2);
                // A query far outside the candidate cloud (|qm~| beyond the fp16 range: clouds of very different extent, e.g. a unit
                // teapot against a ModelNet table in millimetres) gets a power-of-two scale sq of its own: the operand holds
                // sq qm~ and sq in the three norm slots, the accumulator is sq (|c~|^2 + qm~ . c~) -- every comparison of this
                // query (tile minima, band, keys) is in its scaled unit, all relative error terms are unchanged (products of
                // fp16 values with a power of two are exact).  Without it such queries scanned the whole chunk exactly:
                // 690 us instead of ~100 for C2's shape with an extent ratio of 250.  (The padding slot stays unscaled: 4.3e9.)
                float sq = 1.0f, fl_s = 0.0f;
                if (S >= 3.0e4f && S < 0x1p28f) {
                    int e;
                    (void)frexpf(S, &e);          // S = f 2^e, f in [0.5, 1)
                    sq = ldexpf(1.0f, 14 - e);    // sq S in [2^13, 2^14); sq >= 2^-14: a normal fp16
                    m0 *= sq; m1 *= sq; m2 *= sq; S *= sq;
                    fl_s = 0x1p-15f;              // fp16 subnormal quantum of a scaled-down small component x |c~_d| <= 2^-25 x 3 x 128
                }
                qok = S < 3.0e4f;  // inside the fp16 range (also false for NaN)
                qfin = fabsf(qr[0]) + fabsf(qr[1]) + fabsf(qr[2]) < INFINITY;
                const float qn = 0.25f * ((m0 * m0 + m1 * m1) + m2 * m2) / sq;  // sq |q~|^2: the band in the query's unit
                da = kBandA * qn + 0x1p-24f * (S + 4.0f) + fl_s;
                // Round 4, queries OUTSIDE the candidate cloud (|q~| >~ Cmax: clouds of different extent -- a unit teapot against a
                // table in millimetres).  The filter's error is really beta (|c~|^2 + |q~||c~|) + floor (tools/test_f16_filter.hip:
                // max 2^-20.6 over magnitudes up to the scaled far queries'; beta = 2^-18 keeps 6 x head-room); the form above
                // charges the cross term as |q~|^2.  With |c~| <= Cmax for every candidate in the filter:
                //   U_c >= t_c - beta |q~| Cmax - fl,   U_c <= t_c + 2.07 beta Cmax^2 + beta |q~| Cmax + fl
                //   => U_c* <= Umin (1 + 2^-20) + 2.1 beta Cmax^2 + 2 beta |q~| Cmax + 2^-19 |q~|^2 + floor
                // (the 2^-20 terms: the oracle's own rounding of the two distances, as in kBandB1 / kBandA).  For |q~| = R Cmax the
                // band is (2.1 + 2 R + 0.5 R^2) beta Cmax^2 instead of 4.3 R^2: 6 x narrower at R = 12, 8.5 x at R = 250 -- what is
                // left is the reference's own Float32 rounding of distances that large.  Both are valid; the lower one is used.
                {
                    const float c2 = sq * cm2;                     // in the query's unit, like qn = sq |q~|^2
                    da_far = 2.1f * kBetaC * c2 + 2.0f * kBetaC * sqrtf(qn * c2) + 0x1p-19f * qn + 0x1p-24f * (S + 4.0f) + fl_s;
                }
                if constexpr (PRUNE) {
                    // |q~ - c~|^2 = |q~|^2 + t(c) and t(c) <= (its filter value) + beta (|c~|^2 + |q~|^2) + floor (the filter's error
                    // model above), |c~|^2 <= cm2: twice that error is added.  Queries with a scale of their own (far outside the
                    // cloud) or outside the fp16 range do not bound anything: +Inf keeps every lane tile of their wave.
                    dpr = qok && sq == 1.0f ? qn + 2.0f * kBetaC * (cm2 + qn) + 0x1p-23f * (S + 4.0f) : INFINITY;
                }
                _Float16 hx, lx, hy, ly, hz, lz;
                split2h(qok ? m0 : 0.f, hx, lx); split2h(qok ? m1 : 0.f, hy, ly); split2h(qok ? m2 : 0.f, hz, lz);
                const _Float16 one = (_Float16)sq, pad = (_Float16)kPadF16;
                bq = hh == 0 ? h8{hx, lx, hx, hy, ly, hy, hz, lz} : h8{hz, one, one, one, lx, ly, lz, pad};
                if (hh == 0) {
                    qres[jq] = ~0ull;
                    qtab[jq * 3 + 0] = qr[0]; qtab[jq * 3 + 1] = qr[1]; qtab[jq * 3 + 2] = qr[2];
                }
            }

            // the kHFifo smallest lane-tile minima of this lane, as keys (tile id in the six low mantissa bits): fk[0] <= fk[1] <= ...
            // Four VALU operations per lane tile (v_and_or, 2 x v_med3, v_min) -- the FIFO of the first two rounds (threshold fma,
            // compare, five selects / shifts, min: nine) is gone, and "may the lane have missed a tile" is exact now (kc in band).
            float tm = INFINITY, fk[kHFifo];
#pragma unroll
            for (int s = 0; s < kHFifo; ++s) fk[s] = INFINITY;
            const unsigned int keymask = ~((1u << kHIdBits) - 1u);

            // ---- main loop, software-pipelined by TWO 32-candidate blocks ---------------------------------------
            // Block b's MFMA is issued two steps before its 16 accumulators are folded (three accumulator sets in rotation):
            // with one step of distance the fold of a block sat right behind its own MFMA's latency (the compiler filled
            // the gap with s_nop 6), and tools/ubench_overlap.hip puts a fold of the block before last 5 cycles per tile
            // under a fold of the last one at this instruction mix.  Six blocks (three lane tiles) per iteration keep every
            // register index static.  (The image carries two padding blocks behind cnt_pad, so the operand prefetch never
            // needs a clamp and every ds_read_b128 is base + immediate offset: no address VALU in the loop.)
            static_assert(kHLT == 2, "the rotation below is written for lane tiles of two blocks");
            const int nblk = cnt_pad / 32;  // even
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
#define NN1_FOLD(ACC, FIRST)                                                                                         \
            {                                                                                                        \
                const float t0 = min3f(ACC[0], ACC[1], ACC[2]), t1 = min3f(ACC[3], ACC[4], ACC[5]);                  \
                const float t2 = min3f(ACC[6], ACC[7], ACC[8]), t3 = min3f(ACC[9], ACC[10], ACC[11]);                \
                const float t4 = min3f(ACC[12], ACC[13], ACC[14]);                                                   \
                const float t5 = min3f(t0, t1, t2), t6 = min3f(t3, t4, ACC[15]);                                     \
                tm = (FIRST) ? vmin(t5, t6) : min3f(tm, t5, t6); /* first block of the lane tile restarts tm */      \
            }
#define NN1_TRACK_ID(LT)                                                                                             \
            {                                                                                                        \
                float key;                                                                                           \
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(tm), "v"(keymask), "s"(LT));                     \
                _Pragma("unroll") for (int s_ = kHFifo - 1; s_ > 0; --s_) fk[s_] = __builtin_amdgcn_fmed3f(fk[s_ - 1], fk[s_], key);    \
                fk[0] = vmin(fk[0], key);                                                                            \
            }
#define NN1_TRACK() { NN1_TRACK_ID(lt) ++lt; }
            // closing a fold: per lane tile (the odd block's fold carries on from the even one's and is tracked), or per block
#define NN1_CLOSE(ACC, ODD, LT)                                                                                      \
            if constexpr (kHBlkTrack) { NN1_FOLD(ACC, true) NN1_TRACK_ID(2 * (LT) + (ODD)) }                           \
            else { NN1_FOLD(ACC, !(ODD)) if (ODD) NN1_TRACK_ID(LT) }
            // one step: issue the next block into ISSUE, fold FOLD (issued two steps ago); an odd fold closes its lane tile.
            // (round 4) The fold and the tracking run at raised wave priority, the MFMA issue at the base one: among the four
            // waves of a SIMD the ones with VALU work go first and the matrix pipe drains the others' MFMAs underneath
            // (same-box A/B, three variants: bench.py 52.3 -> 51.6 us per step, kernel -1.0 ... -1.5 us; priority on the MFMA
            // issue instead is what costs: knn.hip)
#define NN1_STEP(ISSUE, FOLD, ODD, OFF)                                                                              \
            __builtin_amdgcn_s_setprio(0);                                                                           \
            ISSUE = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, bq, zero, 0, 0, 0);                                   \
            an = pa[(OFF) * 64];                                                                                     \
            __builtin_amdgcn_s_setprio(1);                                                                           \
            NN1_CLOSE(FOLD, ODD, lt)                                                                                 \
            if (ODD) ++lt;
            if constexpr (!PRUNE) {
            const h8 *pa = imgp + hh * 32 + jq;
            f32x16 acc0, acc1, acc2;
            h8 an = pa[0];                     // operand of the next block to issue (one ds_read_b128 in flight per step)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, bq, zero, 0, 0, 0); an = pa[1 * 64];
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, bq, zero, 0, 0, 0); an = pa[2 * 64];
            pa += 3 * 64;                      // -> the operand one block past the next issue
            int lt = 0;                        // lane tile of the next fold
            int nb = 2;                        // next block to issue (even); blocks nb - 2, nb - 1 are in acc0, acc1
            for (; nb + 6 <= nblk; nb += 6) {
                NN1_STEP(acc2, acc0, 0, 0) NN1_STEP(acc0, acc1, 1, 1) NN1_STEP(acc1, acc2, 0, 2)
                NN1_STEP(acc2, acc0, 1, 3) NN1_STEP(acc0, acc1, 0, 4) NN1_STEP(acc1, acc2, 1, 5)
                pa += 6 * 64;
            }
            // the last 0, 2 or 4 blocks, then the two folds still pending
            if (nblk - nb == 4) {
                NN1_STEP(acc2, acc0, 0, 0) NN1_STEP(acc0, acc1, 1, 1) NN1_STEP(acc1, acc2, 0, 2) NN1_STEP(acc2, acc0, 1, 3)
                NN1_CLOSE(acc1, 0, lt) NN1_CLOSE(acc2, 1, lt)
            } else if (nblk - nb == 2) {
                NN1_STEP(acc2, acc0, 0, 0) NN1_STEP(acc0, acc1, 1, 1)
                NN1_CLOSE(acc2, 0, lt) NN1_CLOSE(acc0, 1, lt)
            } else {
                NN1_CLOSE(acc0, 0, lt) NN1_CLOSE(acc1, 1, lt)
            }
            } else {
            // ---- (PRUNE) the same pipeline over a LIST of lane tiles, in two phases ----------------------------------------------
            // Lane l tests lane tile l: squared distance bd between the tile's box and the box of the wave's 32 queries (image frame).
            // Phase 0 runs the tiles whose box meets the queries' box (or, if none does, the nearest one); that gives every query an
            // upper bound of its nearest distance, and phase 1 runs the tiles within the largest of them:
            //   a candidate c' as near as the best one found (or tied with it) has  |q~ - c'~|^2 <= (fk[0] + dpr) (1 + 2^-20)  in real
            //   arithmetic (the oracle's Float32 distance of either differs from the real one by < 2^-21 relative), the image-frame
            //   coordinates (x - mu) sc are off by <= 2^-24 of their magnitude -- covered by (1 + 2^-9) and + 2^-8 on the bound (a wide
            //   margin: 0.2 % of a radius that prunes ~3/4 of the tiles) --, and bd <= |q~ - c'~|^2 for every c' of the tile.
            // A skipped tile therefore holds neither the nearest candidate nor a tie of it, for any query of the wave; the tracked keys,
            // the band and the exact phase work on the tiles that ran.  The list is consumed three lane tiles (six blocks) per
            // iteration -- the rotation of three accumulator sets keeps static register names --, padded with the image's two padding
            // blocks (lane tile `nlt`: keys of 4.3e9 that no band reaches).
            const int nltp = nblk / kHLT;      // lane tiles of the chunk; index nltp = the padding blocks
            float bd = INFINITY;
            {
                // the box of the wave's 32 queries (group tp 16 + wave of the window): gathered when the window was laid out
                const int4 *gq = reinterpret_cast<const int4 *>(boxes + 64 * 2) + (tp * (kHThreads / 64) + wv) * 2;
                const int4 gl = gq[0], gh = gq[1];
                const float qlx = fkey_inv(gl.x), qly = fkey_inv(gl.y), qlz = fkey_inv(gl.z);
                const float qhx = fkey_inv(gh.x), qhy = fkey_inv(gh.y), qhz = fkey_inv(gh.z);
                if (lane < nltp) {
                    const float4 lo = boxes[lane * 2], hi = boxes[lane * 2 + 1];
                    const float gx = fmaxf(fmaxf(lo.x - qhx, qlx - hi.x), 0.0f), gy = fmaxf(fmaxf(lo.y - qhy, qly - hi.y), 0.0f);
                    const float gz = fmaxf(fmaxf(lo.z - qhz, qlz - hi.z), 0.0f);
                    bd = (gx * gx + gy * gy) + gz * gz;  // (NaN for a non-finite query: its wave keeps every tile, below)
                }
            }
            const unsigned long long all_lt = nltp >= 64 ? ~0ull : (1ull << nltp) - 1ull;
            unsigned long long todo = all_lt, done = 0ull;
            if (sorted_c) {
                todo = __ballot(bd == 0.0f);
                if (!todo) {
                    const float bmin = wave_min_f(bd);
                    todo = bmin < INFINITY ? __ballot(bd == bmin) : all_lt;
                }
            }
            const h8 *pbase = imgp + hh * 32 + jq;
            for (int phase = 0; phase < 2; ++phase) {
                if (phase == 1) {
                    if (done == all_lt) break;
                    // largest upper bound of a nearest squared distance among the wave's queries (+Inf / NaN: every tile)
                    const float kb = fk[0];
                    const float bb = __builtin_fmaf(fabsf(kb), kKeyUp, kb);
                    const float dj = fminf(bb, __shfl_xor(bb, 32, 64)) + dpr;
                    const float ucut = wave_max_f(__builtin_fmaf(dj, 1.0f + 0x1p-9f, 0x1p-8f));
                    const bool keep = !(bd > ucut) || !(dj == dj);
                    todo = (__ballot(!(dj == dj)) ? all_lt : __ballot(keep)) & all_lt & ~done;
                    if (!todo) break;
                }
                done |= todo;
                unsigned long long mk = todo;
                int nrem = __builtin_popcountll(mk) - 1;  // lane tiles of the list behind the one in flight
#define NN1_NEXT(T) { T = nltp; if (mk) { T = __builtin_ctzll(mk); mk &= mk - 1ull; } }
                int ta, tb;
                NN1_NEXT(ta) NN1_NEXT(tb)
                f32x16 acc0, acc1, acc2;
                const h8 *pn = pbase + ta * (kHLT * 64);
                h8 an = pn[0];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, bq, zero, 0, 0, 0); an = pn[64];
                pn = pbase + tb * (kHLT * 64);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, bq, zero, 0, 0, 0); an = pn[0];
                // one step: issue the block whose operand is in `an`, fetch the operand NEXT, fold FOLD (issued two steps ago)
#define NN1_PSTEP(ISSUE, FOLD, ODD, LT, NEXT)                                                                         \
                __builtin_amdgcn_s_setprio(0);                                                                       \
                ISSUE = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, bq, zero, 0, 0, 0);                               \
                an = NEXT;                                                                                           \
                __builtin_amdgcn_s_setprio(1);                                                                       \
                NN1_CLOSE(FOLD, ODD, LT)
                // invariant: lane tile ta is in flight (acc0, acc1), tb is the next one (the padding tile if the list is exhausted),
                // `an` holds tb's first operand, pn -> tb
                while (nrem >= 3) {
                    int tc, td, te;
                    NN1_NEXT(tc) NN1_NEXT(td) NN1_NEXT(te)
                    nrem -= 3;
                    const h8 *pc = pbase + tc * (kHLT * 64), *pd = pbase + td * (kHLT * 64);
                    NN1_PSTEP(acc2, acc0, 0, ta, pn[64])
                    NN1_PSTEP(acc0, acc1, 1, ta, pc[0])
                    pn = pbase + te * (kHLT * 64);
                    NN1_PSTEP(acc1, acc2, 0, tb, pc[64])
                    NN1_PSTEP(acc2, acc0, 1, tb, pd[0])
                    NN1_PSTEP(acc0, acc1, 0, tc, pd[64])
                    NN1_PSTEP(acc1, acc2, 1, tc, pn[0])
                    ta = td; tb = te;
                }
                // the last 0, 1 or 2 lane tiles, then the two folds still pending
                if (nrem == 2) {
                    int tc;
                    NN1_NEXT(tc)
                    const h8 *pc = pbase + tc * (kHLT * 64);
                    NN1_PSTEP(acc2, acc0, 0, ta, pn[64])
                    NN1_PSTEP(acc0, acc1, 1, ta, pc[0])
                    NN1_PSTEP(acc1, acc2, 0, tb, pc[64])
                    NN1_PSTEP(acc2, acc0, 1, tb, pc[64])
                    NN1_CLOSE(acc1, 0, tc) NN1_CLOSE(acc2, 1, tc)
                } else if (nrem == 1) {
                    NN1_PSTEP(acc2, acc0, 0, ta, pn[64])
                    NN1_PSTEP(acc0, acc1, 1, ta, pn[64])
                    NN1_CLOSE(acc2, 0, tb) NN1_CLOSE(acc0, 1, tb)
                } else {
                    NN1_CLOSE(acc0, 0, ta) NN1_CLOSE(acc1, 1, ta)
                }
#undef NN1_PSTEP
#undef NN1_NEXT
                __builtin_amdgcn_s_setprio(0);
#ifdef FX3D_PROBE_COUNT
                if ((threadIdx.x & 63) == 0) atomicAdd(&g_probe[4095 * 16 + 6 + phase], (unsigned long long)__builtin_popcountll(todo));
#endif
            }
            }
#undef NN1_STEP
#undef NN1_CLOSE
#undef NN1_TRACK
#undef NN1_TRACK_ID
#undef NN1_FOLD
            __builtin_amdgcn_s_setprio(0);
            float ft[kHFifo];
#pragma unroll
            for (int s = 0; s < kHFifo; ++s) ft[s] = fk[s];
            const float ka = fk[0];
            // this lane's smallest tile minimum is <= best (an upper bound of it: the key of the smallest VALUE is >= ka)
            const float best = __builtin_fmaf(fabsf(ka), kKeyUp, ka);
            FX3D_PROBE_MARK(tp == 0 ? 3 : 7);
            FX3D_PROBE_MARKW(tp == 0 ? 1 : 3);

            // ---- exact phase, wave-cooperative ----------------------------------------------------------------
            {
                // (the lane index through an opaque zero: otherwise every lane-derived address and mask of this phase is hoisted
                //  out of the pass / chunk loops and stays live across the main loop -- ~40 vector registers -- and the kernel spills)
                int opq;
                asm volatile("s_mov_b32 %0, 0" : "=s"(opq));
                const int lane = (int)(threadIdx.x & 63) + opq, jq = lane & 31, hh = lane >> 5;
                const float m = fminf(best, __shfl_xor(best, 32, 64));
                const float thr1 = fminf(__builtin_fmaf(m, kBandB1, da), __builtin_fmaf(m, 1.0f + 0x1p-20f, da_far));  // on tile minima (m >= the true minimum)
                const float thr1k = __builtin_fmaf(fabsf(thr1), kKeyUp, thr1);      // on keys: t <= thr1  =>  key(t) <= thr1k
                const bool usable = sane && far_ok && qok && m < INFINITY;  // filter meaningful for this query
                const bool slow = !usable || !(fk[kHFifo - 1] > thr1k);  // one more lane tile than the FIFO holds may lie within the band
                // Common case (no slow lane in the wave): the items are the FIFO entries within the band.
                // Rare case (degenerate / near-tied data, unusable filter): the wave re-runs its filter pass
                // with the now known threshold and enqueues exactly the lane tiles within the band (every
                // tile for lanes whose filter is unusable), draining the list whenever it is full.
                // A wave with ONE or TWO such queries does not re-run its pass (one wave's ~7 us would be the launch's tail: the
                // 256 blocks are one round): after the FIFO items it scans the chunk exactly for each of them (every lane tile of
                // both halves as items: ~2 us per query).  More slow queries (degenerate data): the retry pass.
                const unsigned long long sb = __ballot(slow);
                unsigned int qslow = (unsigned int)sb | (unsigned int)(sb >> 32);  // (a query's two half-wave lanes share bit jq)
                const bool retry = __builtin_popcount(qslow) > 2;
                bool fifo_done = false;
                // a NaN distance needs a non-finite cloud or a non-finite query (wave-uniform switch of the task code)
                const bool nonfinite = !sane || __ballot(!qfin) != 0;
                const int nlt = nblk / kHLT;
                int lt2 = 0;  // (retry: the next 32-candidate BLOCK of the pass, nblk of them)
                // (retry) tightly clustered clouds put most queries of a wave over the FIFO: RUN items; a few slow queries
                // (ties on a lattice, duplicates): lane-tile items, whose filter pass is cheaper (one ballot per 64 candidates)
                const bool runs = retry && __builtin_popcount(qslow) >= kHRunsFrom;
#ifdef FX3D_PROBE_COUNT  // (with FX3D_PROBE; the atomics distort the stamps: counters and stamps in separate builds)
                if ((threadIdx.x & 63) == 0) {  // wave-passes, with slow queries, retried, with run items, slow queries, unusable lanes
                    unsigned long long *pc = &g_probe[4095 * 16];
                    atomicAdd(&pc[0], 1ull); atomicAdd(&pc[1], qslow ? 1ull : 0ull); atomicAdd(&pc[2], retry ? 1ull : 0ull);
                    atomicAdd(&pc[3], runs ? 1ull : 0ull); atomicAdd(&pc[4], (unsigned long long)__builtin_popcount(qslow));
                    atomicAdd(&pc[5], (unsigned long long)__builtin_popcountll(__ballot(!usable)));
                }
#endif
                const int lt2_end = runs ? nblk : nlt;
                do {
                    int nitems = 0;
                    if (!retry && fifo_done) {
                        // one slow query against the rows of its slow half-wave lane(s), exactly: a lane (jq, hh) sees the runs of
                        // four rows 8 g + 4 hh of every 32-candidate block, so the scan takes every second run (both halves slow:
                        // all of them); two runs per lane in flight, lane-local minimum (lowest index on ties: a lane's runs
                        // ascend), one LDS atomic per lane
                        const int qsl = __builtin_ctz(qslow);
                        qslow &= qslow - 1;
                        const float qq[3] = {qtab[qsl * 3], qtab[qsl * 3 + 1], qtab[qsl * 3 + 2]};
                        unsigned long long kbest = ~0ull;
                        for (int h = 0; h < 2; ++h) {
                            if (!((sb >> (32 * h + qsl)) & 1ull)) continue;  // (wave-uniform)
                            for (int r0 = 4 * (2 * lane + h); r0 < cnt; r0 += kScanU * 512) {
                                float cx[kScanU][4], cy[kScanU][4], cz[kScanU][4];
                                [[maybe_unused]] int cid[kScanU][4];  // (PRUNE) original indices of the rows
#pragma unroll
                                for (int u = 0; u < kScanU; ++u) {
                                    const int jl0 = r0 + 512 * u;
                                    if constexpr (PRUNE) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) {
                                            const float4 v = cs[jl0 + r < cnt ? jl0 + r : cnt - 1];
                                            cx[u][r] = v.x; cy[u][r] = v.y; cz[u][r] = v.z; cid[u][r] = __builtin_bit_cast(int, v.w);
                                        }
                                    } else if (vec && jl0 + 4 <= cnt) {
                                        load4pts(cb, j0 + jl0, cx[u], cy[u], cz[u]);
                                    } else {
#pragma unroll
                                        for (int r = 0; r < 4; ++r) {
                                            const int jc = jl0 + r < cnt ? jl0 + r : cnt - 1;
                                            const float *src = cb + (size_t)(j0 + jc) * 3;
                                            cx[u][r] = src[0]; cy[u][r] = src[1]; cz[u][r] = src[2];
                                        }
                                    }
                                }
#pragma unroll
                                for (int u = 0; u < kScanU; ++u)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const int jc = r0 + 512 * u + r;
                                        const float cc3[3] = {cx[u][r], cy[u][r], cz[u][r]};
                                        const unsigned int cidx = PRUNE ? (unsigned int)cid[u][r] : (unsigned int)(j0 + jc);
                                        const unsigned long long key = ((unsigned long long)dist_key(sqd<3>(qq, cc3)) << 32) | cidx;
                                        if (jc < cnt && key < kbest) kbest = key;
                                    }
                            }
                        }
                        atomicMin(&qres[qsl], kbest);
                        lt2 = qslow ? 0 : nlt;
                    } else if (!retry) {
#pragma unroll
                        for (int s = 0; s < kHFifo; ++s) {
                            const bool qual = ft[s] <= thr1k;  // (finite: the key's low bits are its lane tile)
                            const unsigned long long bal = __ballot(qual);
                            if (bal) {
                                const int pos = nitems + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0));
                                if (qual) items[pos] = (unsigned short)(((unsigned int)jq << (kHIdBits + 1)) | ((unsigned int)hh << kHIdBits) |
                                                                        (__builtin_bit_cast(unsigned int, ft[s]) & ((1u << kHIdBits) - 1u)));
                                nitems += __builtin_popcountll(bal);  // <= 64 * kHFifo == kHItemCap
                            }
                        }
                        fifo_done = true;
                        lt2 = qslow ? 0 : nlt;
                    } else {
                        // (round 5) RUN-granular items: a lane's 16 rows of a block are four runs of four consecutive candidates
                        // (rows 4g .. 4g + 3 = candidates 8g + 4hh + 0..3); a run is an item when its minimum lies within the band.
                        // Lane-tile items (32 candidates each, as the FIFO path has them) made tightly clustered clouds evaluate
                        // more than half of all pairs exactly: ~100 candidates of 4096 within the band, one in every other lane tile.
                        const h8 *pb = imgp + hh * 32 + jq;
                        if (!runs) {
#pragma unroll 1
                            for (; lt2 < nlt && nitems <= kHItemCap - 64 * kHLT; ++lt2) {
                                float t2 = INFINITY;
#pragma unroll
                                for (int bb = 0; bb < kHLT; ++bb) {
                                    const f32x16 av = __builtin_amdgcn_mfma_f32_32x32x16_f16(pb[(lt2 * kHLT + bb) * 64], bq, zero, 0, 0, 0);
                                    if (kHBlkTrack) t2 = INFINITY;   // items are blocks
#pragma unroll
                                    for (int r = 0; r < 16; r += 2) t2 = min3f(t2, av[r], av[r + 1]);
                                    if (kHBlkTrack || bb == kHLT - 1) {
                                        const bool qual = !usable || t2 <= thr1;
                                        const unsigned long long bal = __ballot(qual);
                                        if (bal) {
                                            const int pos = nitems + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0));
                                            const unsigned int id = kHBlkTrack ? (unsigned int)(lt2 * kHLT + bb) : (unsigned int)lt2;
                                            if (qual) items[pos] = (unsigned short)(((unsigned int)jq << (kHIdBits + 1)) | ((unsigned int)hh << kHIdBits) | id);
                                            nitems += __builtin_popcountll(bal);
                                        }
                                    }
                                }
                            }
                        } else
#pragma unroll 1
                        for (; lt2 < nblk && nitems <= kHItemCap - 256; ++lt2) {
                            const f32x16 av = __builtin_amdgcn_mfma_f32_32x32x16_f16(pb[lt2 * 64], bq, zero, 0, 0, 0);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const float mg = vmin(min3f(av[4 * g], av[4 * g + 1], av[4 * g + 2]), av[4 * g + 3]);
                                const bool qual = !usable || mg <= thr1;
                                const unsigned long long bal = __ballot(qual);
                                if (bal) {
                                    const int pos = nitems + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0));
                                    if (qual) items[pos] = (unsigned short)(((unsigned int)jq << 10) | ((unsigned int)hh << 9) | ((unsigned int)lt2 << 2) | (unsigned int)g);
                                    nitems += __builtin_popcountll(bal);
                                }
                            }
                        }
                    }
                    // all 64 lanes share the (item, run-of-4-candidates) tasks
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                    if (!runs) {
                    constexpr int kRunsPerItem = kHBlkTrack ? 4 : kHLT * 4;
                    const int ntask = nitems * kRunsPerItem;
                    for (int t0 = 0; t0 < ntask; t0 += 64) {
                        const int t = t0 + lane;
                        if (t < ntask) {
                            const unsigned int it = items[t / kRunsPerItem];
                            const int run = t % kRunsPerItem;
                            const int qs = it >> (kHIdBits + 1), ih = (it >> kHIdBits) & 1, tl = it & ((1u << kHIdBits) - 1u);
                            const int jl0 = kHBlkTrack ? tl * 32 + 8 * run + 4 * ih : (tl * kHLT + (run >> 2)) * 32 + 8 * (run & 3) + 4 * ih;
                            float cx[4], cy[4], cz[4];
                            [[maybe_unused]] int cid[4];
                            if constexpr (PRUNE) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    // (round 6: no clamp -- rows up to cnt_pad <= kHChunkMax lie inside the block's scratch, the ones past cnt hold
                                    //  whatever they held and are masked below -- and the key form chosen once per wave: the exact phase issues
                                    //  ~700 VALU per wave and pass, the SIMDs' busiest stretch; same-box A/B 43.0 -> 41.5 us kernel)
                                    const float4 v = cs[jl0 + r];
                                    cx[r] = v.x; cy[r] = v.y; cz[r] = v.z; cid[r] = __builtin_bit_cast(int, v.w);
                                }
                            } else if (vec && jl0 + 4 <= cnt) {
                                load4pts(cb, j0 + jl0, cx, cy, cz);
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int jc = jl0 + r < cnt ? jl0 + r : cnt - 1;
                                    const float *src = cb + (size_t)(j0 + jc) * 3;
                                    cx[r] = src[0]; cy[r] = src[1]; cz[r] = src[2];
                                }
                            }
                            const float qq[3] = {qtab[qs * 3], qtab[qs * 3 + 1], qtab[qs * 3 + 2]};
                            unsigned int kb;
                            int ib = j0 + jl0;  // an all-+Inf run still names a real candidate (its first)
                            if constexpr (PRUNE) {  // the rows' original indices do not ascend: the run's minimum on full (distance, index) keys
                                unsigned long long k64 = ~0ull;
                                if (!nonfinite) {  // (wave-uniform) distances are >= 0 or +Inf: their bits are the keys
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float cc3[3] = {cx[r], cy[r], cz[r]};
                                        const unsigned long long key = ((unsigned long long)__builtin_bit_cast(unsigned int, sqd<3>(qq, cc3)) << 32) | (unsigned int)cid[r];
                                        if (jl0 + r < cnt && key < k64) k64 = key;
                                    }
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float cc3[3] = {cx[r], cy[r], cz[r]};
                                        const unsigned long long key = ((unsigned long long)dist_key(sqd<3>(qq, cc3)) << 32) | (unsigned int)cid[r];
                                        if (jl0 + r < cnt && key < k64) k64 = key;
                                    }
                                }
                                kb = (unsigned int)(k64 >> 32); ib = (int)(unsigned int)k64;
                            } else
                            if (!nonfinite) {   // distances are >= 0 or +Inf: float order == key order
                                float db = INFINITY;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float cc3[3] = {cx[r], cy[r], cz[r]};
                                    const float dd = sqd<3>(qq, cc3);
                                    if (jl0 + r < cnt && dd < db) { db = dd; ib = j0 + jl0 + r; }
                                }
                                kb = __builtin_bit_cast(unsigned int, db);
                            } else {            // NaN distances possible: canonical keys, NaN after +Inf
                                kb = 0xffffffffu;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float cc3[3] = {cx[r], cy[r], cz[r]};
                                    const unsigned int kd = dist_key(sqd<3>(qq, cc3));
                                    if (jl0 + r < cnt && kd < kb) { kb = kd; ib = j0 + jl0 + r; }
                                }
                            }
                            if (jl0 < cnt) atomicMin(&qres[qs], ((unsigned long long)kb << 32) | (unsigned int)ib);
                        }
                    }
                    } else {
                        // run items: one task per item, a loop of its own with plain 4-byte loads and canonical keys (the loop above keeps
                        // the instruction stream it had before the run items existed: sharing it through selects on `runs` cost C2 + 0.7 us
                        // per step on the same box, sharing the task body -- lambda or macro -- lost its 16-byte loads: + 6.7 us; here the
                        // loads are not what bounds the tasks: knocked out, the dense case keeps its time)
#pragma unroll 1
                        for (int t0 = 0; t0 < nitems; t0 += 64) {
                            const int t = t0 + lane;
                            if (t < nitems) {
                                const unsigned int it = items[t];
                                const int qs = (int)(it >> 10);
                                const int jl0 = (int)((it >> 2) & 127u) * 32 + 8 * (int)(it & 3u) + 4 * (int)((it >> 9) & 1u);
                                const float qq[3] = {qtab[qs * 3], qtab[qs * 3 + 1], qtab[qs * 3 + 2]};
                                unsigned long long kbest = ~0ull;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int jc = jl0 + r < cnt ? jl0 + r : cnt - 1;
                                    float cc3[3];
                                    unsigned int cidx = (unsigned int)(j0 + jc);
                                    if constexpr (PRUNE) {
                                        const float4 v = cs[jc];
                                        cc3[0] = v.x; cc3[1] = v.y; cc3[2] = v.z; cidx = __builtin_bit_cast(unsigned int, v.w);
                                    } else {
                                        const float *src = cb + (size_t)(j0 + jc) * 3;
                                        cc3[0] = src[0]; cc3[1] = src[1]; cc3[2] = src[2];
                                    }
                                    const unsigned long long key = ((unsigned long long)dist_key(sqd<3>(qq, cc3)) << 32) | cidx;
                                    if (jl0 + r < cnt && key < kbest) kbest = key;
                                }
                                if (jl0 < cnt) atomicMin(&qres[qs], kbest);
                            }
                        }
                    }
                } while (lt2 < lt2_end);
                // the far candidates (outside the filter): every query of the wave against each of them, exactly
                if (far_ok) {
                    for (int t = lane; t < 32 * nf; t += 64) {
                        const int qs = t & 31, jc = farlist[t >> 5];
                        const float qq[3] = {qtab[qs * 3], qtab[qs * 3 + 1], qtab[qs * 3 + 2]};
                        float cc3[3];
                        unsigned int cidx = (unsigned int)(j0 + jc);
                        if constexpr (PRUNE) {
                            const float4 v = cs[jc];
                            cc3[0] = v.x; cc3[1] = v.y; cc3[2] = v.z; cidx = __builtin_bit_cast(unsigned int, v.w);
                        } else {
                            const float *src = cb + (size_t)(j0 + jc) * 3;
                            cc3[0] = src[0]; cc3[1] = src[1]; cc3[2] = src[2];
                        }
                        const float dd = sqd<3>(qq, cc3);
                        const unsigned int kd = nonfinite ? dist_key(dd) : __builtin_bit_cast(unsigned int, dd);
                        atomicMin(&qres[qs], ((unsigned long long)kd << 32) | cidx);
                    }
                }
                // the cloud's tail beyond the LDS image (<= kHTail candidates): every query of the wave against each, exactly
                if (ctail && j0 + jstep >= NCm) {
                    for (int t = lane; t < 32 * ctail; t += 64) {
                        const int qs = t & 31, jc = NCm + (t >> 5);
                        const float qq[3] = {qtab[qs * 3], qtab[qs * 3 + 1], qtab[qs * 3 + 2]};
                        const float *src = cb + (size_t)jc * 3;
                        const float cc3[3] = {src[0], src[1], src[2]};
                        atomicMin(&qres[qs], ((unsigned long long)dist_key(sqd<3>(qq, cc3)) << 32) | (unsigned int)jc);
                    }
                }
            }
            FX3D_PROBE_MARK(tp == 0 ? 4 : 8);
            FX3D_PROBE_MARKW(tp == 0 ? 2 : 4);

            if (j0 + jstep >= NCm) {  // last chunk of this block: results of this tile pass
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                if (hh == 0) {
                    const unsigned long long r = qres[jq];
                    const float dd = __builtin_bit_cast(float, (unsigned int)(r >> 32));
                    const int ii = (int)(unsigned int)r;
                    if (p.nsplit > 1) {  // this chunk subset's row; the finalize kernel (or, fused, the tile's last subset) takes the
                                         // minimum over the subsets.  Fused: a write-through (agent-scope) store -- the reader may sit
                                         // on another XCD, behind another L2; a device-wide fence instead costs a whole L2 write-back
                                         // per block (measured: 32 -> 94 us)
                        unsigned long long *gp = &p.gres[((size_t)split * 2 * p.B + c) * p.qstride + qi];
                        if (qi < NQ) {
                            if (SPLITFUSE) __hip_atomic_store(gp, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else *gp = r;
                        }
                    } else if (qi < NQ) {
                        if (WANT_IDX && idx_out) idx_out[(size_t)b * NQ + qi] = ii;
                        if (dmin_out) dmin_out[(size_t)b * NQ + qi] = dd;
                        acc += (double)dd;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    FX3D_PROBE_MARK(11);
    if (SPLITFUSE && p.nsplit > 1) {
        // ---- candidate-split run, merge fused (round 5): the chunk subsets of one query tile arrive at a counter of their own
        //      (a spare word of the launch's ticket slot: zero between launches, the last arriver returns it to zero); the last one
        //      takes the 64-bit minimum over the subsets' rows -- (distance bits, index): `isless`, then the lowest index --, writes
        //      the outputs and is the only one of them to deliver a partial.  One launch instead of two (the second one was a
        //      dependent launch of ~6 us behind a 26 us kernel at C3's shape, 8 x 5000 x 5000).
        __shared__ int s_last;
        FX3D_PROBE_MARK2(10);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores of its gres rows have left the CU
        FX3D_PROBE_MARK2(11);
        // (tools/nn1_probe at C3's shape: thread 0 waits ~14 k cycles here -- for the block's slower WAVES at the barrier below, not
        //  for the stores: a SIMD serves its oldest wave first, wave 0 is done ~20 k cycles before the last one at C2 too --, then
        //  counter 1.2 k, merge 4.6 k, finalisation 4.9 k cycles)
        __syncthreads();
        FX3D_PROBE_MARK2(12);
        const int ns = (NC + CH - 1) / CH < p.nsplit ? (NC + CH - 1) / CH : p.nsplit;  // subsets that exist for this direction
        if (tid == 0) {
            const unsigned int t = (unsigned int)c * (unsigned int)p.tiles + (unsigned int)tile;
            unsigned int *ctr = p.ticket + (t / 15u) * 16u + 1u + t % 15u;
            const unsigned int old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned int)ns - 1u;
            if (s_last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            FX3D_PROBE_MARK2(13);
        }
        __syncthreads();
        if (!s_last) return;  // (block-uniform)
        FX3D_PROBE_MARK(6);
        acc = 0.0;
        const int qend = (tile + 1) * tpb * QB < NQ ? (tile + 1) * tpb * QB : NQ;
        // (every row of every query of this thread requested at once: these loads go to memory -- another XCD's block wrote them --
        //  and one after the other they were three dependent round trips per query, 23 k cycles for the tile's last block)
        // (round 6: one query per thread and sweep, sixteen subsets in flight -- a tile has at most 1024 queries per sweep anyway, and
        //  the fit iteration's plan, one mesh of 5000 samples against another, has twelve subsets: the four beyond the eighth were
        //  four more dependent round trips to memory)
        constexpr int kQ = 1, kS = 16;  // queries per thread and sweep x subsets in flight
        for (int q0 = tile * tpb * QB + tid; q0 < qend; q0 += kQ * kHThreads) {
          unsigned long long rr[kQ][kS];
#pragma unroll
          for (int u = 0; u < kQ; ++u)
#pragma unroll
            for (int sp = 0; sp < kS; ++sp) {
                const int q = q0 + u * kHThreads < qend ? q0 + u * kHThreads : q0;
                // (UNCONDITIONAL loads, a subset beyond the last re-reads the last one's row: `sp < ns ? load : ~0` compiled to a branch
                //  around every load with a wait behind it -- twelve dependent round trips to memory, 8.8 k cycles at the fit
                //  iteration's shape, tools/nn1_probe "tail detail")
                const int spc = sp < ns ? sp : ns - 1;
                rr[u][sp] = __hip_atomic_load(&p.gres[((size_t)spc * 2 * p.B + c) * p.qstride + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#ifdef FX3D_PROBE
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          FX3D_PROBE_MARK2(14);
#endif
#pragma unroll
          for (int u = 0; u < kQ; ++u) {
            const int q = q0 + u * kHThreads;
            if (q >= qend) break;
            unsigned long long r = rr[u][0];
#pragma unroll
            for (int sp = 1; sp < kS; ++sp) r = rr[u][sp] < r ? rr[u][sp] : r;
            for (int sp = kS; sp < ns; ++sp) {  // (more than sixteen subsets: the rest one by one)
                const unsigned long long o = __hip_atomic_load(&p.gres[((size_t)sp * 2 * p.B + c) * p.qstride + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r = o < r ? o : r;
            }
            const float dd = __builtin_bit_cast(float, (unsigned int)(r >> 32));
            if (idx_out) idx_out[(size_t)b * NQ + q] = (int)(unsigned int)r;
            if (dmin_out) dmin_out[(size_t)b * NQ + q] = dd;
            acc += (double)dd;
          }
        }
        FX3D_PROBE_MARK(7);
    }
    if (p.partials) {
        __shared__ double sm[kHThreads / 64];
        const double tot = block_sum<kHThreads>(acc, sm);
        FX3D_PROBE_MARK(8);
        if (!p.ticket) {
            if (tid == 0) p.partials[(size_t)c * p.tiles + tile] = tot;
        } else {
            // ---- fused finalisation: placement-independent hand-off through 8-byte agent-scope atomics
            //      (write-through store -> drain -> relaxed ticket; the last arriver reads with agent-scope
            //      loads), MI355X_MICROARCH.md "valid forms".  Fixed summation order => deterministic.
            // (only the first wave goes on: the block's sum is in its lane 0)
            if (wv == 0) {
                const FinalizeArgs fa{reinterpret_cast<unsigned long long *>(p.partials), p.ticket, p.nvalid, p.B, p.tiles, p.tiles_x, p.tiles_y,
                                      p.sums_out, p.loss_out, p.N, p.M, p.Bg, p.w1, p.w2};
                const int was_last = fused_finalize_wave0(fa, (size_t)c * p.tiles + tile, tot, lane);
#ifdef FX3D_PROBE
                if (tid == 0 && blockIdx.x < 4096) { g_probe[blockIdx.x * 16 + 15] = was_last; g_probe[blockIdx.x * 16 + 9] = __builtin_readcyclecounter(); }
#else
                (void)was_last;
#endif
            }
        }
    }
    FX3D_PROBE_MARK(12);
}

// Unpack the per-query (d, index) slots of a candidate-split run, write the outputs and the per-block
// partial sums (layout: tiles = ceil(max(N,M)/256) blocks per cloud).
__global__ __launch_bounds__(kThreads) void nn1_split_finalize_kernel(Nn1Params p, int tiles_f) {
    const int c = blockIdx.y;
    const int dir = c >= p.B ? 1 : 0;
    const int b = dir ? c - p.B : c;
    const int NQ = dir ? p.M : p.N;
    const int qi = blockIdx.x * kThreads + threadIdx.x;
    double acc = 0.0;
    if (qi < NQ) {
        // the chunk subsets that exist for this direction (nn1_f16_kernel: `split * chunk >= NC` blocks return at once)
        const int NC = dir ? p.N : p.M;
        const int ns = (NC + p.chunk - 1) / p.chunk < p.nsplit ? (NC + p.chunk - 1) / p.chunk : p.nsplit;
        unsigned long long r = p.gres[(size_t)c * p.qstride + qi];
        for (int sp = 1; sp < ns; sp += 4) {  // (distance bits, index): the 64-bit minimum is the `isless` + lowest-index winner
            unsigned long long o[4];          // four rows in flight (a row index past the end re-reads row 0)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = p.gres[((size_t)(sp + e < ns ? sp + e : 0) * 2 * p.B + c) * p.qstride + qi];
#pragma unroll
            for (int e = 0; e < 4; ++e) r = o[e] < r ? o[e] : r;
        }
        const float dd = __builtin_bit_cast(float, (unsigned int)(r >> 32));
        int32_t *idx_out = dir ? p.idx_y : p.idx_x;
        float *dmin_out = dir ? p.dmin_y : p.dmin_x;
        if (idx_out) idx_out[(size_t)b * NQ + qi] = (int)(unsigned int)r;
        if (dmin_out) dmin_out[(size_t)b * NQ + qi] = dd;
        acc = (double)dd;
    }
    if (p.partials) {
        __shared__ double sm[kThreads / 64];
        const double tot = block_sum<kThreads>(acc, sm);
        if (!p.ticket) {
            if (threadIdx.x == 0) p.partials[(size_t)c * tiles_f + blockIdx.x] = tot;
            return;
        }
        // fused finalisation (as in nn1_f16_kernel): the last block to arrive sums the partials and writes the sums / the loss --
        // one launch less per split run; by its first wave alone (fused_finalize_wave0)
        if (threadIdx.x < 64) {
            const FinalizeArgs fa{reinterpret_cast<unsigned long long *>(p.partials), p.ticket, p.nvalid, p.B, tiles_f, p.tiles_x, p.tiles_y,
                                  p.sums_out, p.loss_out, p.N, p.M, p.Bg, p.w1, p.w2};
            (void)fused_finalize_wave0(fa, (size_t)c * tiles_f + blockIdx.x, tot, (int)threadIdx.x);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// nn1_tiny_kernel (D = 3, round 4): the exact loop for SMALL problems -- 2 B N M below a few ten million pair evaluations
// (C1, the reference harness's n <= 4096: benchmarks/metrics.jl:40) -- where nn1_f16_kernel's per-cloud statistics, image
// and 1024-thread blocks are all overhead (C1 17.6 us, n = 64: 13.7 us; an empty launch is ~6 us).  No statistics, no
// image, no filter, no LDS staging, no barrier before the arithmetic: a 256-thread block owns 16 R queries and ALL
// candidates of their cloud.  Lane l of wave w holds query l & 15 (+ 16 r) and works on candidate slice s = 4 w + (l >> 4)
// of 16 contiguous slices.  The slice is consumed in groups of 16 candidates: lane l LOADS candidate (l & 15) of the group
// (one 12-byte global load per lane and 16 pairs; consecutive lanes, consecutive points) and every lane of the 16-lane
// row reads it through the DPP row broadcast of the subtraction itself (v_subrev_f32_dpp row_newbcast:i: no move, no LDS):
// 8 VALU per pair for the oracle's unfused ((dx dx) + dy dy) + dz dz, 8 v_min3 per group of 16, 3 for (best, best group)
// with a strict `<`.  The winning group is re-scanned for the FIRST candidate that attains the minimum, so a lane's
// result is its slice's (distance, lowest index); the 16 slices of a query meet in a 64-bit LDS atomicMin on
// (distance bits << 32 | index) -- the oracle's order (isless, then the lower index; no NaN passes `<`, a query without
// any distance below +Inf takes nn1_scan_isless).  Loss: per-block Float64 partial + nn1_f16_kernel's fused finalisation.
constexpr int kTyThreads = 256;
constexpr int kTyQ = 16;                      // queries per 16-lane row
constexpr int kTySl = 16;                     // candidate slices per block (4 per wave, one per DPP row)
constexpr int kTyG = 16;                      // candidates per group (one per lane of a row)

template <int I>
__device__ __forceinline__ float row_bcast(float v) {  // lane (l & ~15) + I of every 16-lane row, read by the consuming instruction's DPP operand
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + I, 0xF, 0xF, false));
}
template <int I>
__device__ __forceinline__ float tiny_d(const float (&q)[3], float cx, float cy, float cz) {
    const float t0 = q[0] - row_bcast<I>(cx), t1 = q[1] - row_bcast<I>(cy), t2 = q[2] - row_bcast<I>(cz);
    return ((t0 * t0) + (t1 * t1)) + (t2 * t2);
}
__device__ __forceinline__ void tiny_group(const float (&q)[3], float cx, float cy, float cz, float (&d)[kTyG]) {
    d[0] = tiny_d<0>(q, cx, cy, cz); d[1] = tiny_d<1>(q, cx, cy, cz); d[2] = tiny_d<2>(q, cx, cy, cz); d[3] = tiny_d<3>(q, cx, cy, cz);
    d[4] = tiny_d<4>(q, cx, cy, cz); d[5] = tiny_d<5>(q, cx, cy, cz); d[6] = tiny_d<6>(q, cx, cy, cz); d[7] = tiny_d<7>(q, cx, cy, cz);
    d[8] = tiny_d<8>(q, cx, cy, cz); d[9] = tiny_d<9>(q, cx, cy, cz); d[10] = tiny_d<10>(q, cx, cy, cz); d[11] = tiny_d<11>(q, cx, cy, cz);
    d[12] = tiny_d<12>(q, cx, cy, cz); d[13] = tiny_d<13>(q, cx, cy, cz); d[14] = tiny_d<14>(q, cx, cy, cz); d[15] = tiny_d<15>(q, cx, cy, cz);
}

// kTyPF = groups in flight per lane (4; 1 for clouds of <= 256 candidates: a quarter of the code -- the smallest launches are
// eight blocks on eight cold instruction caches).
template <int R, int kTyPF, bool WANT_IDX>
__global__ __launch_bounds__(kTyThreads) void nn1_tiny_kernel(Nn1Params p) {
    __shared__ unsigned long long slot[2][kTyQ * R];
    const int tiles = p.tiles;
    const int c = blockIdx.x / tiles, btile = blockIdx.x - c * tiles;   // cloud id in [0, 2B): dir = c / B
    const int dir = c >= p.B ? 1 : 0;
    const int b = dir ? c - p.B : c;
    const int NQ = dir ? p.M : p.N, NC = dir ? p.N : p.M;
    if (btile >= (dir ? p.tiles_y : p.tiles_x)) return;
    const float *__restrict__ qb = (dir ? p.y : p.x) + (size_t)b * NQ * 3;
    const float *__restrict__ cb = (dir ? p.x : p.y) + (size_t)b * NC * 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 15, s = wave * 4 + (lane >> 4);
    // slice s = candidates [s L, (s + 1) L), L a multiple of 16; this lane loads candidate jl + 16 g of group g
    const int L = ((NC + kTySl - 1) / kTySl + kTyG - 1) / kTyG * kTyG;
    const int ngroups = L / kTyG;
    const int jl = s * L + ql;
    auto load_group = [&](int g, float &cx, float &cy, float &cz) {
        const int j = jl + kTyG * g;
        const bool ok = g < ngroups && j < NC;
        const P3 t = *reinterpret_cast<const P3 *>(cb + 3ll * (ok ? j : 0));
        cx = ok ? t.x : INFINITY; cy = ok ? t.y : INFINITY; cz = ok ? t.z : INFINITY;   // padding at +Inf: never below anything
    };
    float cur[kTyPF][3], nxt[kTyPF][3];
#pragma unroll
    for (int u = 0; u < kTyPF; ++u) load_group(u, cur[u][0], cur[u][1], cur[u][2]);
    const bool resident = ngroups <= kTyPF;   // the lane's share of the cloud stays in registers across the block's query tiles
    if (tid < 2 * kTyQ * R) (&slot[0][0])[tid] = ~0ull;
    __syncthreads();

    // a block takes p.tpb consecutive query tiles of its cloud (one partial sum, one arrival at the ticket per BLOCK: with a
    // block per tile the 1024+ same-address atomics and the last arriver's pass over as many partials were the launch's tail)
    const int raw_tiles = (NQ + kTyQ * R - 1) / (kTyQ * R);
    int32_t *idx_out = dir ? p.idx_y : p.idx_x;
    float *dmin_out = dir ? p.dmin_y : p.dmin_x;
    double acc = 0.0;
    auto load_queries = [&](int tile, float (&qq)[R][3]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = tile * (kTyQ * R) + r * kTyQ + ql;
            const P3 t = *reinterpret_cast<const P3 *>(qb + 3ll * (qi < NQ ? qi : NQ - 1));
            qq[r][0] = t.x; qq[r][1] = t.y; qq[r][2] = t.z;
        }
    };
    float qn[R][3];
    load_queries(btile * p.tpb, qn);
    for (int tt = 0; tt < p.tpb; ++tt) {
        const int tile = btile * p.tpb + tt;
        if (tile >= raw_tiles) break;   // (block-uniform)
        float q[R][3];
#pragma unroll
        for (int r = 0; r < R; ++r) { q[r][0] = qn[r][0]; q[r][1] = qn[r][1]; q[r][2] = qn[r][2]; }
        if (tt + 1 < p.tpb) load_queries(tile + 1, qn);   // the next tile's queries travel while this one is evaluated
        if (!resident && tt > 0) {
#pragma unroll
            for (int u = 0; u < kTyPF; ++u) load_group(u, cur[u][0], cur[u][1], cur[u][2]);
        }
        float best[R];
        int bg[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { best[r] = INFINITY; bg[r] = 0; }
        for (int g0 = 0; g0 < ngroups; g0 += kTyPF) {
            if (g0 + kTyPF < ngroups) {
#pragma unroll
                for (int u = 0; u < kTyPF; ++u) load_group(g0 + kTyPF + u, nxt[u][0], nxt[u][1], nxt[u][2]);
            }
#pragma unroll
            for (int u = 0; u < kTyPF; ++u) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float d[kTyG];
                    tiny_group(q[r], cur[u][0], cur[u][1], cur[u][2], d);
                    float m = min3f(d[0], d[1], d[2]);
                    m = min3f(m, d[3], d[4]);
                    m = min3f(m, d[5], d[6]);
                    m = min3f(m, d[7], d[8]);
                    m = min3f(m, d[9], d[10]);
                    m = min3f(m, d[11], d[12]);
                    m = min3f(m, d[13], d[14]);
                    m = __builtin_fminf(m, d[15]);
                    const bool better = m < best[r];  // strict: the first group that holds the lane's minimum
                    best[r] = better ? m : best[r];
                    if (WANT_IDX) bg[r] = better ? g0 + u : bg[r];
                }
            }
            if (g0 + kTyPF < ngroups) {
#pragma unroll
                for (int u = 0; u < kTyPF; ++u) { cur[u][0] = nxt[u][0]; cur[u][1] = nxt[u][1]; cur[u][2] = nxt[u][2]; }
            }
        }
        unsigned long long *sl = slot[tt & 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int bi = 0;
            if (WANT_IDX) {
                // the FIRST candidate of the lane's winning group that attains the minimum.  The winning groups differ lane by
                // lane, so every lane reads the 16 candidates of its own (16 loads in flight, L1 / L2 hits)
                float d[kTyG];
#pragma unroll
                for (int i = 0; i < kTyG; ++i) {
                    const int j = s * L + kTyG * bg[r] + i;
                    const P3 t = *reinterpret_cast<const P3 *>(cb + 3ll * (j < NC ? j : 0));
                    const float t0 = q[r][0] - t.x, t1 = q[r][1] - t.y, t2 = q[r][2] - t.z;
                    d[i] = j < NC ? ((t0 * t0) + (t1 * t1)) + (t2 * t2) : INFINITY;
                }
#pragma unroll
                for (int i = kTyG - 1; i >= 0; --i)
                    if (d[i] == best[r]) bi = s * L + kTyG * bg[r] + i;
            }
            if (best[r] < INFINITY)
                atomicMin(&sl[r * kTyQ + ql], ((unsigned long long)__builtin_bit_cast(unsigned int, best[r]) << 32) | (unsigned int)bi);
        }
        __syncthreads();
        // the first wave writes the tile's results (16 R <= 32 queries) and returns the slots to "empty"; the other buffer
        // takes the next tile's minima meanwhile (this buffer is used again two tiles on, behind the next barrier)
        const int qidx = tile * (kTyQ * R) + tid;
        if (tid < kTyQ * R) {
            const unsigned long long k = sl[tid];
            sl[tid] = ~0ull;
            if (qidx < NQ) {
                float dd = __builtin_bit_cast(float, (unsigned int)(k >> 32));
                int ii = (int)(unsigned int)k;
                if (k == ~0ull) {  // nothing below +Inf (non-finite or overflowing coordinates): the exact scan in isless order
                    const P3 t = *reinterpret_cast<const P3 *>(qb + 3ll * qidx);
                    const float qq[3] = {t.x, t.y, t.z};
                    nn1_scan_isless<3>(qq, cb, NC, dd, ii);
                }
                if (WANT_IDX && idx_out) idx_out[(size_t)b * NQ + qidx] = ii;
                if (dmin_out) dmin_out[(size_t)b * NQ + qidx] = dd;
                acc += (double)dd;
            }
        }
    }
    if (tid >= 64 || !p.partials) return;
    const double tot = __shfl(wave_sum_l63_f64(acc), 63);   // fixed order: deterministic
    if (!p.ticket) {
        if (tid == 0) p.partials[(size_t)c * tiles + btile] = tot;
        return;
    }
    const FinalizeArgs fa{reinterpret_cast<unsigned long long *>(p.partials), p.ticket, p.nvalid, p.B, tiles, p.tiles_x, p.tiles_y,
                          p.sums_out, p.loss_out, p.N, p.M, p.Bg, p.w1, p.w2};
    (void)fused_finalize_wave0(fa, (size_t)c * tiles + btile, tot, tid);
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
        float best = 0.0f;
        int bi = 0;
        const float *a = qb + (size_t)i * D;
        for (int j = 0; j < NC; ++j) {
            const float *cc = cb + (size_t)j * D;
            float s = 0.0f;
            for (int d = 0; d < D; ++d) { float t = a[d] - cc[d]; s = s + t * t; }
            if (j == 0 || fless(s, best)) { best = s; bi = j; }
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

__global__ void chamfer_loss_many_kernel(const double *sums, int count, int N, int M, int D, long long Bg,
                                         float w1, float w2, float *loss) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) loss[i] = chamfer_loss_from_sums(sums[2 * i], sums[2 * i + 1], N, M, D, Bg, w1, w2);
}

struct Plan {
    int R, tiles_x, tiles_y, tiles, chunk, grid;
    size_t lds_bytes;
    int variant;  // 0 = exact hot loop (D = 2, and D = 3 under FX3D_NN1_VARIANT=0), 3 = fp16-split MFMA filter + exact re-scan,
                  // 4 = nn1_tiny_kernel (D = 3, small problems: exact, no per-cloud statistics / image)
    int threads, tpb, tpb_y;  // tpb: passes per block of the x -> y direction, tpb_y: of y -> x
    int nsplit;  // fp16 variant: chunk subsets per query tile (multi-chunk clouds with too few blocks)
    int tail;    // fp16 variant: kHTail when clouds of chunk + (1 .. kHTail) points run as one chunk + an exact tail
};

// option nn1_variant = 0 selects the exact VALU loop for D = 3 (A/B measurements; the f32 VALU / MFMA filter variants
// of round 1 are in the history: DESIGN.md 3.1 "ladder").
int nn1_variant() { return opt(OPT_NN1_VARIANT) == 0 ? 0 : 3; }

Plan make_plan(int N, int M, int B, int D, bool allow_split = true) {
    Plan pl{};
    const long long work = (long long)B * ((long long)N + M);  // total queries, both directions
    pl.variant = D == 3 ? nn1_variant() : 0;
    pl.threads = pl.variant == 3 ? kHThreads : kThreads;
    // exact loop: R queries per thread -- enough blocks to fill 256 CUs x ~2 blocks, but as much register
    // blocking (LDS-read amortisation, ILP) as the problem size allows.
    int R = 4;
    while (R > 1 && work / (kThreads * R) < 512) R >>= 1;
    pl.R = R;
    pl.tpb = pl.tpb_y = 1;
    const int maxc0 = N > M ? N : M;
    const int clouds8 = (2 * B + 7) / 8;
    pl.nsplit = 1;
    if (pl.variant == 3 && 2ll * B * (long long)N * M <= 1000000ll * opt(OPT_NN1_TINY_MPAIRS)) {
        // small problem: the exact kernel without statistics / image (C1, the reference harness's n <= 1024).  Two queries per
        // lane once 16-query blocks would be more than two rounds of the chip
        pl.variant = 4;
        pl.threads = kTyThreads;
        const int cus = device_cus();
        pl.R = work / kTyQ > 2ll * cus ? 2 : 1;
        const int per = kTyQ * pl.R;
        const int rx = (N + per - 1) / per, ry = (M + per - 1) / per;   // query tiles per cloud and direction
        int tpb = 1;                                                     // consecutive tiles per block: at most ~2 blocks per CU
        while ((long long)B * ((rx + tpb - 1) / tpb + (ry + tpb - 1) / tpb) > 2ll * cus && tpb < (rx > ry ? rx : ry)) tpb *= 2;
        pl.tpb = pl.tpb_y = tpb;
        pl.tiles_x = (rx + tpb - 1) / tpb;
        pl.tiles_y = (ry + tpb - 1) / tpb;
        pl.tiles = pl.tiles_x > pl.tiles_y ? pl.tiles_x : pl.tiles_y;
        pl.chunk = 0;
        pl.lds_bytes = 0;
        pl.grid = 2 * B * pl.tiles;
        return pl;
    }
    if (pl.variant == 0) {
        const int per_block = kThreads * R;
        pl.tiles_x = (N + per_block - 1) / per_block;
        pl.tiles_y = (M + per_block - 1) / per_block;
        pl.tiles = pl.tiles_x > pl.tiles_y ? pl.tiles_x : pl.tiles_y;
        int chunk = (maxc0 + kTile - 1) / kTile * kTile;
        if (chunk > kChunkMax) chunk = kChunkMax;
        pl.chunk = chunk;
        pl.lds_bytes = (size_t)chunk * (D <= 3 ? D : 0) * sizeof(float);
        pl.grid = clouds8 * 8 * pl.tiles;
        return pl;
    }
    // nn1_f16_kernel: choose (candidate chunk size, chunks per block, 512-query passes per block) by a small
    // cost model in microseconds, measured at C2 (tools/nn1_probe.hip): bounding box 2.8 per 4096 candidates of
    // the cloud, image 5.0 per 4096 of the chunk, one pass (filter + exact) 9.7 per 4096, 256 resident blocks.
    // A block either walks all chunks serially (one pass per block: the per-query slot lives in LDS) or takes
    // ONE chunk of a split run (any number of passes; the subsets' rows merge in the same launch since round 5).  A split
    // plan is charged 8 us: no longer a second launch, a FITTED constant -- swept 8 / 5 / 3 over tools/nn1_shapes_time.py's
    // shapes on one box, only 8 x 8192 x 8192 changes plan, and the lower charges pick the slower one (62 us against 55.5:
    // the model under-prices 2048-candidate images run four passes each).  Few large clouds want many small chunks, many
    // small clouds want passes.
    const int cmax = kHChunkMax, gran = 32 * kHLT;
    const int ncu = device_cus();  // blocks resident at once: one per CU (256 on an MI355X in SPX mode)
    // a larger cloud of at most cmax + kHTail points is planned (and run) as ONE chunk of cmax with an exact tail
    int maxc = maxc0, b_chunk = 0, b_tpb = 1, b_split = 1;
    auto search = [&](int mc) {  // mc: the candidates that go through LDS images
        maxc = mc;
        const int cminc = (maxc + cmax - 1) / cmax;
        double best = 1e30;
        b_chunk = (maxc + gran - 1) / gran * gran < cmax ? (maxc + gran - 1) / gran * gran : cmax; b_tpb = 1; b_split = 1;
        for (int nch = cminc; nch <= cminc * 8 && nch <= 64; ++nch) {
            int ch = ((maxc + nch - 1) / nch + gran - 1) / gran * gran;
            if (ch > cmax) continue;
            const int anch = (maxc + ch - 1) / ch;
            for (int split = 0; split < 2; ++split) {
                if (split && (!allow_split || anch == 1 || opt(OPT_NN1_NOSPLIT))) continue;
                for (int tpb = 1; tpb <= 8; tpb *= 2) {
                    if (!split && anch > 1 && tpb > 1) continue;
                    const long long tiles = ((long long)maxc + 512 * tpb - 1) / (512 * tpb);
                    const long long blocks = 2ll * B * tiles * (split ? anch : 1);
                    const double per_chunk = 5.0 * ch / 4096.0 + 0.5 + tpb * (9.7 * ch / 4096.0 + 0.8);
                    const double t_block = 2.8 * maxc / 4096.0 + (split ? 1 : anch) * per_chunk;
                    const double rounds = (double)((blocks + ncu - 1) / ncu);
                    const double t = rounds * t_block + (split ? 8.0 : 0.0);
                    if (t < best - 1e-9) { best = t; b_chunk = ch; b_tpb = tpb; b_split = split ? anch : 1; }
                }
            }
        }
    };
    // a cloud of cmax + (1 .. kHTail) points: planned as cmax; kept if that plan is ONE chunk of cmax per block (then the kernel
    // runs the tail exactly), otherwise planned again at its true size
    pl.tail = 0;
    if (maxc0 > cmax && maxc0 - cmax <= kHTail) {
        search(cmax);
        if (b_split == 1 && b_chunk == cmax) pl.tail = kHTail;
    }
    if (!pl.tail) search(maxc0);
    pl.chunk = b_chunk;
    pl.tpb = pl.tpb_y = b_tpb;
    pl.nsplit = b_split;
    pl.lds_bytes = (size_t)pl.chunk * 8 * sizeof(float) + kHScratchBytes;
    if (N != M && b_split == 1 && maxc <= b_chunk) {
        // clouds of different sizes, one chunk each: the direction whose CANDIDATES are the large cloud has few, heavy blocks
        // (N = 4096 against M = 1024 at B = 32: 32 blocks as long as C2's on 32 CUs while the rest of the chip idles) -- the
        // passes per block are chosen per direction: t = the slower direction's block, or the chip's throughput if the
        // blocks of both do not fit at once (same unit costs as above)
        double bt = 1e30;
        for (int ta = 1; ta <= 8; ta *= 2)
            for (int tb = 1; tb <= 8; tb *= 2) {
                const double blk_a = 2.8 * M / 4096.0 + 5.0 * M / 4096.0 + 0.5 + ta * (9.7 * M / 4096.0 + 0.8);  // x -> y: candidates y
                const double blk_b = 2.8 * N / 4096.0 + 5.0 * N / 4096.0 + 0.5 + tb * (9.7 * N / 4096.0 + 0.8);  // y -> x: candidates x
                const double na = (double)B * ((N + 512 * ta - 1) / (512 * ta)), nb = (double)B * ((M + 512 * tb - 1) / (512 * tb));
                const double thr = (na * blk_a + nb * blk_b) / (double)ncu;
                double t = blk_a > blk_b ? blk_a : blk_b;
                t = t > thr ? t : thr;
                // the grid has max(tiles) slots per (cloud, direction): more than one round of them delays the heavy direction's blocks
                const long long tmax = (N + 512 * ta - 1) / (512 * ta) > (M + 512 * tb - 1) / (512 * tb) ? (N + 512 * ta - 1) / (512 * ta) : (M + 512 * tb - 1) / (512 * tb);
                t += 1.0 * (double)((2ll * B * tmax + ncu - 1) / ncu - 1);
                if (t < bt - 1e-9) { bt = t; pl.tpb = ta; pl.tpb_y = tb; }
            }
    }
    // one-chunk plans: a remainder of <= kHTail queries beyond a direction's last full tile is one more pass of that tile's block
    // (one wave busy for ~6 us) instead of a block of its own (prologue + a pass: a second round of blocks at N = 4097)
    const bool fold = b_split == 1 && maxc <= b_chunk;
    auto ntiles = [&](int nq, int tpb) {
        const int per = 512 * tpb;
        int t = (nq + per - 1) / per;
        if (fold && t > 1 && nq - (t - 1) * per <= kHTail) --t;
        return t;
    };
    pl.tiles_x = ntiles(N, pl.tpb);
    pl.tiles_y = ntiles(M, pl.tpb_y);
    pl.tiles = pl.tiles_x > pl.tiles_y ? pl.tiles_x : pl.tiles_y;
    pl.grid = clouds8 * 8 * pl.tiles * pl.nsplit;
    if (2 * B < 8) pl.grid = 2 * B * pl.tiles * pl.nsplit;  // plain block order (see kernel)
    return pl;
}

size_t partials_count(const Plan &pl, int B) { return (size_t)2 * B * pl.tiles; }

// Spatial pruning (nn1_f16_kernel<.., PRUNE>): one-chunk plans of the fp16 kernel without split or tail, clouds of at least 1024
// points.  Rows of scratch per block: the candidate cloud in image order + the block's window of the query cloud.
constexpr size_t kHBoxBytes = 64 * 2 * sizeof(float4) + kHGroupsMax * 32;  // LDS: the lane tiles' boxes + the query groups' boxes
int prune_rows_per_block(const Plan &pl, int N, int M, int D) {
    const int maxc = N > M ? N : M;
    if (D != 3 || pl.variant != 3 || pl.nsplit != 1 || pl.tail != 0 || maxc > pl.chunk || maxc < 1024 || !opt(OPT_NN1_PRUNE)) return 0;
    if ((pl.tpb > pl.tpb_y ? pl.tpb : pl.tpb_y) * 16 + 2 > kHGroupsMax - 1) return 0;  // (the query groups' boxes: LDS for eight passes per block)
    // one pass per block does not repay the sort (measured at B = 8 .. 16 x 4096: 36 against 30 us; two passes -- C2 -- 46 against 53)
    if ((pl.tpb < pl.tpb_y ? pl.tpb : pl.tpb_y) < 2) return 0;
    return kHChunkMax + (pl.tpb > pl.tpb_y ? pl.tpb : pl.tpb_y) * 512 + kHTail;
}
size_t prune_scratch_bytes(const Plan &pl, int N, int M, int D) {
    return (size_t)prune_rows_per_block(pl, N, M, D) * pl.grid * sizeof(float4);
}

template <int DIM, bool WANT_IDX>
fx3d_status launch_small(const Nn1Params &p, const Plan &pl, hipStream_t st) {
    if (pl.variant == 4) {
        if (DIM == 3) {
            const bool small = (p.N > p.M ? p.N : p.M) <= kTySl * kTyG;   // both directions' candidates are one group per slice
            if (pl.R == 2 && small) hipLaunchKernelGGL((nn1_tiny_kernel<2, 1, WANT_IDX>), dim3(pl.grid), dim3(kTyThreads), 0, st, p);
            else if (pl.R == 2) hipLaunchKernelGGL((nn1_tiny_kernel<2, 4, WANT_IDX>), dim3(pl.grid), dim3(kTyThreads), 0, st, p);
            else if (small) hipLaunchKernelGGL((nn1_tiny_kernel<1, 1, WANT_IDX>), dim3(pl.grid), dim3(kTyThreads), 0, st, p);
            else hipLaunchKernelGGL((nn1_tiny_kernel<1, 4, WANT_IDX>), dim3(pl.grid), dim3(kTyThreads), 0, st, p);
        }
        FX3D_LAUNCH_CHECK();
        return FX3D_OK;
    }
    if (pl.variant == 3) {
        if (DIM == 3) {
            // > 64 KiB of dynamic LDS needs an explicit opt-in (static LDS of the kernel: < 1 KiB)
            const void *kfn = p.pscr ? reinterpret_cast<const void *>(&nn1_f16_kernel<WANT_IDX, false, true>)
                              : p.fuse_split ? reinterpret_cast<const void *>(&nn1_f16_kernel<WANT_IDX, true>)
                                             : reinterpret_cast<const void *>(&nn1_f16_kernel<WANT_IDX, false>);
            const fx3d_status arc = ensure_dynamic_lds(kfn, (int)(kHChunkMax * 32 + kHScratchBytes + (p.pscr ? kHBoxBytes : 0)), "nn1_f16_kernel");
            if (arc != FX3D_OK) return arc;
            if (p.pscr) hipLaunchKernelGGL((nn1_f16_kernel<WANT_IDX, false, true>), dim3(pl.grid), dim3(kHThreads), pl.lds_bytes + kHBoxBytes, st, p);
            else if (p.fuse_split) hipLaunchKernelGGL((nn1_f16_kernel<WANT_IDX, true>), dim3(pl.grid), dim3(kHThreads), pl.lds_bytes, st, p);
            else hipLaunchKernelGGL((nn1_f16_kernel<WANT_IDX, false>), dim3(pl.grid), dim3(kHThreads), pl.lds_bytes, st, p);
        }
        FX3D_LAUNCH_CHECK();
        return FX3D_OK;
    }
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

constexpr long long kSplitFuseMax = 255;  // tile counters of a fused split run: the 15 spare words of each of a ticket slot's 17 lines
struct Fused {
    unsigned int *ticket;
    unsigned int nvalid;
    double *sums_out;
    float *loss_out;
    float w1, w2;
    long long Bg;
};

fx3d_status run_nn1(const float *x, int N, const float *y, int M, int B, int D, int32_t *idx_x,
                    int32_t *idx_y, float *dmin_x, float *dmin_y, double *partials,
                    const Plan &pl, hipStream_t st, const Fused *fu = nullptr,
                    unsigned long long *gres = nullptr, int qstride = 0, float4 *pscr = nullptr) {
    Nn1Params p{};
    p.pscr = pscr;
    p.pscr_stride = pscr ? prune_rows_per_block(pl, N, M, D) : 0;
    if (!p.pscr_stride) p.pscr = nullptr;
    p.nsplit = gres ? pl.nsplit : 1;
    p.gres = gres;
    p.qstride = qstride;
    p.fuse_split = gres && fu && fu->ticket ? 1 : 0;  // (the caller checked that the tile counters fit the ticket slot)
    if (fu) {
        p.ticket = fu->ticket; p.nvalid = fu->nvalid; p.sums_out = fu->sums_out; p.loss_out = fu->loss_out;
        p.w1 = fu->w1; p.w2 = fu->w2; p.Bg = fu->Bg;
    }
    p.x = x; p.y = y; p.N = N; p.M = M; p.B = B;
    p.idx_x = idx_x; p.idx_y = idx_y; p.dmin_x = dmin_x; p.dmin_y = dmin_y;
    p.partials = partials;
    p.tiles = pl.tiles; p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.chunk = pl.chunk;
    p.tpb = pl.tpb; p.tpb_y = pl.tpb_y;
    p.tail = pl.tail;
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

// Library-owned arrival counters for the fused finalisation: zeroed once at allocation, every launch returns its
// counter to zero.  EAGER launches take one of kTickets slots round robin (two launches share a slot only if more than
// kTickets launches are simultaneously in flight).  A launch recorded by a STREAM CAPTURE bakes its slot's address into
// the graph for good, so it gets a slot of its own that is never handed out again: replays of the graph (ordered on
// their stream) are its only users, and no eager launch 1024 k launches later can share the counter.
namespace fx3d {
static constexpr int kTickets = 1024;      // eager, round robin
static constexpr int kCapChunk = 4096;     // capture-owned slots per allocation
// A slot is kTicketStride words: the arrival counter itself at [0] and, 64 bytes apart, the 16 first-level counters of the
// two-level arrival (fx3d_common.h: ticket_arrive_last) that launches of many blocks use.
unsigned int *ticket_slot(fx3d_status *rc, hipStream_t st) {
    static std::mutex mu;
    static std::atomic<unsigned int *> pools[64];
    // capture-owned slots: the chunk in use and a spare.  Allocation + zeroing are not legal while a stream captures
    // (hipMemset synchronises), so both happen on EAGER calls: the first call of the process sets up the chunk, an
    // eager call that sees it more than half used sets up the spare.  Chunks are never freed: graphs hold their addresses.
    static unsigned int *cap_chunk[64] = {nullptr}, *cap_spare[64] = {nullptr};
    static std::atomic<int> cap_used[64];
    static std::atomic<unsigned int> next{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *rc = FX3D_ERR_HIP; return nullptr; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
    auto fresh = [&](size_t n) -> unsigned int * {
        unsigned int *pnew = nullptr;
        if (hipMalloc(&pnew, n * sizeof(unsigned int)) != hipSuccess || hipMemset(pnew, 0, n * sizeof(unsigned int)) != hipSuccess) {
            set_error("ticket pool allocation failed");
            *rc = FX3D_ERR_OOM;
            return nullptr;
        }
        return pnew;
    };
    if (cs == hipStreamCaptureStatusActive) {
        std::lock_guard<std::mutex> lk(mu);
        if (cap_chunk[dev] && cap_used[dev].load() == kCapChunk && cap_spare[dev]) {
            cap_chunk[dev] = cap_spare[dev];
            cap_spare[dev] = nullptr;
            cap_used[dev].store(0);
        }
        if (!cap_chunk[dev] || cap_used[dev].load() == kCapChunk) {
            set_error("no capture-owned arrival counter left: run the captured sequence once eagerly before capturing "
                      "(the library sets its counters up on eager calls)");
            *rc = FX3D_ERR_HIP;
            return nullptr;
        }
        return cap_chunk[dev] + (size_t)cap_used[dev].fetch_add(1) * kTicketStride;
    }
    unsigned int *pool = pools[dev].load(std::memory_order_acquire);
    if (!pool || !cap_chunk[dev] || (cap_used[dev].load() > kCapChunk / 2 && !cap_spare[dev])) {
        std::lock_guard<std::mutex> lk(mu);
        pool = pools[dev].load(std::memory_order_relaxed);
        if (!pool) {
            pool = fresh((size_t)kTickets * kTicketStride);
            if (!pool) return nullptr;
            pools[dev].store(pool, std::memory_order_release);
        }
        if (!cap_chunk[dev]) {
            cap_chunk[dev] = fresh((size_t)kCapChunk * kTicketStride);
            if (!cap_chunk[dev]) return nullptr;
            cap_used[dev].store(0);
        } else if (cap_used[dev].load() > kCapChunk / 2 && !cap_spare[dev]) {
            cap_spare[dev] = fresh((size_t)kCapChunk * kTicketStride);
            if (!cap_spare[dev]) return nullptr;
        }
    }
    return pool + (size_t)(next.fetch_add(1) % kTickets) * kTicketStride;
}
}  // namespace fx3d

extern "C" {

fx3d_status fx3d_nn1(const float *x, int32_t N, const float *y, int32_t M, int32_t B, int32_t D,
                     int32_t *idx_x, int32_t *idx_y, float *dmin_x, float *dmin_y,
                     fx3d_stream_t s) {
    fx3d_status rc = check_shapes("fx3d_nn1", x, N, y, M, B, D);
    if (rc) return rc;
    Plan pl = make_plan(N, M, B, D);
    if (pl.nsplit > 1) {  // no scratch at this entry point: plan without the split option
        pl = make_plan(N, M, B, D, false);
    }
    return run_nn1(x, N, y, M, B, D, idx_x, idx_y, dmin_x, dmin_y, nullptr, pl, as_stream(s));
}

fx3d_status fx3d_nn1_plan_describe(int32_t N, int32_t M, int32_t B, int32_t D, char *buf, size_t n) {
    FX3D_REQUIRE(buf && n > 0, "fx3d_nn1_plan_describe: null buffer");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0, "fx3d_nn1_plan_describe: empty problem");
    const Plan pl = make_plan(N, M, B, D);
    snprintf(buf, n, "variant=%d threads=%d chunk=%d nsplit=%d tpb=%d tpb_y=%d tiles_x=%d tiles_y=%d grid=%d tail=%d lds=%zu",
             pl.variant, pl.threads, pl.chunk, pl.nsplit, pl.tpb, pl.tpb_y, pl.tiles_x, pl.tiles_y, pl.grid, pl.tail, pl.lds_bytes);
    return FX3D_OK;
}

fx3d_status fx3d_chamfer_workspace_bytes(int32_t N, int32_t M, int32_t B, int32_t D, size_t *bytes) {
    FX3D_REQUIRE(bytes, "fx3d_chamfer_workspace_bytes: null output");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0, "fx3d_chamfer_workspace_bytes: empty input");
    const Plan pl = make_plan(N, M, B, D);
    int tiles, tx, ty;
    partial_layout(pl, N, M, D, &tiles, &tx, &ty);
    *bytes = ((size_t)2 * B * tiles + 2) * sizeof(double);
    if (const size_t ps = prune_scratch_bytes(pl, N, M, D)) *bytes = ((*bytes + 255) & ~(size_t)255) + ps;  // (a smaller workspace still runs: without pruning)
    if (pl.nsplit > 1) {  // split run: 256-query finalize tiles + the per-query merge slots
        const int maxq = N > M ? N : M;
        const int tiles_f = (maxq + kThreads - 1) / kThreads;
        *bytes = ((size_t)2 * B * tiles_f + 2) * sizeof(double) + (size_t)pl.nsplit * 2 * B * maxq * sizeof(unsigned long long);
    }
    return FX3D_OK;
}


}  // extern "C"

namespace fx3d {
fx3d_status chamfer_check_shapes(const char *fn, const void *x, int N, const void *y, int M, int B, int D) { return check_shapes(fn, x, N, y, M, B, D); }
}

extern "C" {

static fx3d_status chamfer_common(const float *x, int N, const float *y, int M, int B, int D,
                                  double *sums_dev, float *loss_dev, long long Bg, float w1,
                                  float w2, int32_t *idx_x, int32_t *idx_y, void *ws,
                                  size_t ws_bytes, hipStream_t st, const char *fn) {
    fx3d_status rc = check_shapes(fn, x, N, y, M, B, D);
    if (rc) return rc;
    const Plan pl = make_plan(N, M, B, D);
    int tiles, tx, ty;
    partial_layout(pl, N, M, D, &tiles, &tx, &ty);
    size_t need = ((size_t)2 * B * tiles + 2) * sizeof(double);
    const int maxq = N > M ? N : M;
    const int tiles_f = (maxq + kThreads - 1) / kThreads;
    if (pl.nsplit > 1)
        need = ((size_t)2 * B * tiles_f + 2) * sizeof(double) + (size_t)pl.nsplit * 2 * B * maxq * sizeof(unsigned long long);
    if (!ws || ws_bytes < need) {
        set_error("%s: workspace too small (%zu < %zu bytes)", fn, ws ? ws_bytes : (size_t)0, need);
        return FX3D_ERR_WORKSPACE;
    }
    double *partials = reinterpret_cast<double *>(ws);
    if (pl.nsplit > 1) {
        // few large clouds: chunk subsets run in parallel blocks, each stores its per-query result row in gres (plain
        // stores: no memset node, no atomics), the finalize kernel merges the rows
        unsigned long long *gres = reinterpret_cast<unsigned long long *>(partials + (size_t)2 * B * tiles_f + 2);
        if (2ll * B * pl.tiles <= kSplitFuseMax) {
            // one launch: the last chunk subset of every query tile merges (tile counters in the ticket slot's spare words), the
            // last of those blocks reduces the partials (layout of the one-chunk path: pl.tiles entries per (direction, cloud))
            fx3d_status trc = FX3D_OK;
            unsigned int *ticket = ticket_slot(&trc, st);
            if (!ticket) return trc;
            Fused fu{ticket, (unsigned int)((long long)B * pl.tiles_x + (long long)B * pl.tiles_y),
                     sums_dev ? sums_dev : partials + (size_t)2 * B * tiles_f, loss_dev, w1, w2, Bg};
            return run_nn1(x, N, y, M, B, D, idx_x, idx_y, nullptr, nullptr, partials, pl, st, &fu, gres, maxq);
        }
        rc = run_nn1(x, N, y, M, B, D, nullptr, nullptr, nullptr, nullptr, nullptr, pl, st, nullptr, gres, maxq);
        if (rc) return rc;
        Nn1Params fp{};
        fp.N = N; fp.M = M; fp.B = B; fp.idx_x = idx_x; fp.idx_y = idx_y; fp.partials = partials;
        fp.gres = gres; fp.qstride = maxq; fp.nsplit = pl.nsplit; fp.chunk = pl.chunk;
        // (split runs exist for D == 3 only) the last block of the unpack kernel reduces the partials: no finalize launch
        fx3d_status trc = FX3D_OK;
        unsigned int *ticket = ticket_slot(&trc, st);
        if (!ticket) return trc;
        fp.ticket = ticket; fp.nvalid = (unsigned int)((long long)tiles_f * 2 * B);
        fp.tiles_x = (N + kThreads - 1) / kThreads; fp.tiles_y = (M + kThreads - 1) / kThreads;
        fp.sums_out = sums_dev ? sums_dev : partials + (size_t)2 * B * tiles_f;
        fp.loss_out = loss_dev; fp.w1 = w1; fp.w2 = w2; fp.Bg = Bg;
        hipLaunchKernelGGL(nn1_split_finalize_kernel, dim3(tiles_f, 2 * B), dim3(kThreads), 0, st, fp, tiles_f);
        FX3D_LAUNCH_CHECK();
        return FX3D_OK;
    }
    if ((pl.variant == 3 || pl.variant == 4) && D == 3) {  // one launch: the last block reduces the partials
        fx3d_status trc = FX3D_OK;
        unsigned int *ticket = ticket_slot(&trc, st);
        if (!ticket) return trc;
        Fused fu{ticket, (unsigned int)((long long)B * tx + (long long)B * ty),
                 sums_dev ? sums_dev : partials + (size_t)2 * B * tiles, loss_dev, w1, w2, Bg};
        // spatial pruning when the workspace holds the blocks' scratch behind the partial sums (fx3d_chamfer_workspace_bytes asks for it)
        float4 *pscr = nullptr;
        const size_t poff = (need + 255) & ~(size_t)255, ps = prune_scratch_bytes(pl, N, M, D);
        if (ps && ws_bytes >= poff + ps) pscr = reinterpret_cast<float4 *>(static_cast<char *>(ws) + poff);
        return run_nn1(x, N, y, M, B, D, idx_x, idx_y, nullptr, nullptr, partials, pl, st, &fu, nullptr, 0, pscr);
    }
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

}  // extern "C"

namespace fx3d {  // the forward driver, for the value-and-gradient entry point in chamfer_bwd.hip
fx3d_status chamfer_forward(const float *x, int N, const float *y, int M, int B, int D, float *loss_dev, long long Bg, float w1,
                            float w2, int32_t *idx_x, int32_t *idx_y, void *ws, size_t ws_bytes, hipStream_t st, const char *fn) {
    return chamfer_common(x, N, y, M, B, D, nullptr, loss_dev, Bg, w1, w2, idx_x, idx_y, ws, ws_bytes, st, fn);
}
}

extern "C" {

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

fx3d_status fx3d_chamfer_finalize_many(const double *sums_dev, int32_t count, int32_t N, int32_t M, int64_t B_global,
                                       int32_t D, float w1, float w2, float *losses_dev, fx3d_stream_t s) {
    FX3D_REQUIRE(sums_dev && losses_dev, "fx3d_chamfer_finalize_many: null pointer");
    FX3D_REQUIRE(count > 0 && N > 0 && M > 0 && B_global > 0 && D > 0, "fx3d_chamfer_finalize_many: bad sizes");
    hipLaunchKernelGGL(chamfer_loss_many_kernel, dim3((count + 63) / 64), dim3(64), 0, as_stream(s), sums_dev, count, N, M,
                       D, (long long)B_global, w1, w2, losses_dev);
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

}  // extern "C"
