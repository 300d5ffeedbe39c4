// Adjoint of _chamfer_distance (Zygote through src/metrics/pcloud.jl:45-48 with the nearest-neighbour indices constant, :45
// `@ignore`), of chamfer_distance(m1::TriMesh, m2::TriMesh, n) (src/metrics/mesh.jl:34-44) w.r.t. the meshes' vertices, and the
// value-and-gradient entry point (benchmarks/metrics.jl:24-38 "total", examples/fit_mesh.jl:106-110).
//
// Round 5: NO float atomics on the default path.  The gradient row of point i of one side is
//     side x:  g[i] = (0 + ca (x_i - y[ix[i]]))  -  sum over { j : iy[j] == i }, j ASCENDING, of  cb (y_j - x_i)
//     side y:  g[j] = (0 - sum over { i : ix[i] == j }, i ASCENDING, of ca (x_i - y_j))  +  cb (y_j - x[iy[j]])
// -- exactly the order in which oracle/flux3d_oracle.c: fx3d_oracle_chamfer_bwd (two loops per batch element) reaches the
// row, so the result is the oracle's bit for bit and the same from run to run.  A block owns up to 4096 rows of one
// (cloud, side); per round of 4096 rows of the OTHER side it builds the inverse lists in LDS with a counting sort of their
// indices (integer LDS atomics only: counts, a block scan, cursor placement), puts every list into ascending order (up to eight
// entries: a sorting network in the owner's registers; longer ones: a wave sorts the list in place through a 4096-bit bitmap
// and walks it with lane-parallel gathers and an ordered readlane chain), and the owner of a row subtracts its entries in
// order.  D = 3: accumulators in registers, 12-byte row loads; other D: the row of g itself is the accumulator.
#include <algorithm>

#include "fx3d_common.h"
#include "sample_gather.h"
#include "mesh_reg.h"

using namespace fx3d;

namespace {

constexpr int kThreads = 256;

// Scatter form with GLOBAL float atomics (option bwd_global_atomics; arrival order decides the last bit), two ordered passes:
//   own pass     : gx[i] = ca (x_i - y[ix[i]]),  gy[j] = cb (y_j - x[iy[j]])        plain coalesced stores
//   scatter pass : gy[ix[i]] -= ca (x_i - y[ix[i]]),  gx[iy[j]] -= cb (y_j - x[iy[j]])   float atomics
template <bool SCATTER>
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
                if (SCATTER) atomicAdd(&gyb[(size_t)j * D + d], -t);
                else gxb[(size_t)i * D + d] = t;
            }
        } else {
            const int j = r - N, i = idx_y[(size_t)b * M + j];
            for (int d = 0; d < D; ++d) {
                const float t = cb * (yb[(size_t)j * D + d] - xb[(size_t)i * D + d]);
                if (SCATTER) atomicAdd(&gxb[(size_t)i * D + d], -t);
                else gyb[(size_t)j * D + d] = t;
            }
        }
    }
}

// ---- the gather form ---------------------------------------------------------------------------------------------------------
#ifndef FX3D_BG_THREADS
#define FX3D_BG_THREADS 1024
#endif
constexpr int kBgThreads = FX3D_BG_THREADS;
constexpr int kBgRows = 4096;    // rows of g one block owns at most
constexpr int kBgPer = kBgRows / kBgThreads;  // rows / scan slots / other-side rows per thread and round
constexpr int kBgChunk = 4096;   // rows of the other side bucketed per round (their round-local ids fit 16 bits)
constexpr int kBgSmall = 8;      // lists up to this length are ordered in the owner's registers (one in 10^6 of uniform data's is longer)
constexpr int kBgSortWaves = kBgThreads / 64;  // every wave takes rows with longer lists (one 4096-bit bitmap each)
constexpr int kBgBigW = 256;     // long rows whose own point and index travel through LDS (the others re-read them from memory)
#ifndef FX3D_BG_OCC
#define FX3D_BG_OCC 4   // waves per SIMD the kernels are compiled for.  4 = one 1024-thread block per CU with up to 128 registers; 8 (two
#endif                  // blocks per CU, 64 registers) spills and measured 41 - 51 us against 37 at B = 256 x 4096 in every form tried

#ifndef FX3D_BG_STOP
#define FX3D_BG_STOP 0   // timing experiments only: leave after phase n
#endif
#define BG_STOP(n) do { if (FX3D_BG_STOP == (n)) return; } while (0)
#ifdef FX3D_BG_PROBE   // phase stamps of every block (s_memtime, 100 MHz) for tools/chamfer_bwd_probe.py; never in the shipped build
__device__ unsigned long long g_bg_probe[1024 * 8];
#define BG_STAMP(n) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_bg_probe[blockIdx.x * 8 + (n)] = __builtin_readcyclecounter(); } while (0)
#else
#define BG_STAMP(n) do { } while (0)
#endif

// 38 KB
struct BgLds {
    unsigned int cnt2[kBgRows / 2];     // entries of own row lr in this round, two 16-bit counters per word (<= 4096 each):
                                        // counted up, then counted DOWN by the placement -- zero again for the next round
    unsigned short start[kBgRows + 4];  // first list slot of row lr; start[lr + 1] - start[lr] = its length
    unsigned short list[kBgChunk];      // round-local ids of the other side's rows, list by list
    unsigned int bitmap[kBgSortWaves][kBgChunk / 32];   // in-place ordering of a long list
    unsigned short big[kBgChunk / (kBgSmall + 1) + 1];  // rows with long lists (more cannot exist in one round)
    float4 bigw[kBgBigW];                               // D = 3: (own point, index of its neighbour as bits) of the first long rows
    __attribute__((aligned(16))) unsigned int wsum[kBgThreads / 64];
    unsigned int nbig;
};

__device__ __forceinline__ void ce(unsigned int &a, unsigned int &b) {
    const unsigned int lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}
// all LDS traffic this wave issued so far is complete and visible to its other lanes
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// inclusive wave64 prefix sum on the VALU (DPP row shifts + row broadcasts; __shfl_up would be six ds_bpermute round trips)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned int dpp_zero(unsigned int v) {  // lanes without a source / outside ROWMASK read 0
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ unsigned int wave_scan_incl(unsigned int v) {
    v += dpp_zero<0x111, 0xF>(v);  // row_shr:1
    v += dpp_zero<0x112, 0xF>(v);  // row_shr:2
    v += dpp_zero<0x114, 0xF>(v);  // row_shr:4
    v += dpp_zero<0x118, 0xF>(v);  // row_shr:8
    v += dpp_zero<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
    v += dpp_zero<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ float lane_val(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

struct SampledSide {
    const int32_t *faces;     // (3, Fmax, B) mesh-local
    const int32_t *face_idx;  // (n, B) the draws
    const float *r1, *r2;
    float *gverts;            // (3, Vmax, B), added to
    int Vmax, Fmax;
};

struct BgJob {
    const float *own, *oth;               // this side's rows (R x D) / the other side's (S x D) of one cloud pair
    const int32_t *idx_own, *idx_oth;     // own row -> other row / other row -> own row
    int R, S, D, r0, r1, side, b;         // the block owns rows [r0, r1) of side `side` (0: x, 1: y) of cloud pair b
    float c_own, c_oth;
    float *g;                             // MODE 0 / 1: this (cloud, side)'s gradient rows (R x D): result, and the accumulator
                                          // between the rounds of an other side beyond kBgChunk rows
    P3 *part;                             // MODE 2: the accumulator between rounds, in LDS
    SampledSide smp;                      // MODE 2: where a finished row goes
};

// MODE 2: row i of the gradient w.r.t. the sampled points -> the three vertices of sample i's face with the barycentric
// weights of its draw (sample_bwd_kernel's arithmetic, src/transforms/mesh_func.jl:64-75); a face is drawn many times: float atomics
__device__ __forceinline__ void bg_scatter_sample(const BgJob &J, int i, P3 a) {
    const SampledSide &S = J.smp;
    const size_t k = (size_t)J.b * J.R + i;
    const int32_t *fc = S.faces + ((size_t)J.b * S.Fmax + S.face_idx[k]) * 3;
    const int f3[3] = {fc[0], fc[1], fc[2]};
    const float uu = sqrtf(S.r1[k]), v = S.r2[k];
    const float wt[3] = {1.0f - uu, uu * (1.0f - v), uu * v};
    const float gr[3] = {a.x, a.y, a.z};
    float *gb = S.gverts + (size_t)J.b * S.Vmax * 3;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int d = 0; d < 3; ++d) atomicAdd(&gb[3ll * f3[t] + d], wt[t] * gr[d]);
}

// MODE 0: D = 3, rows stored to g;  1: any D, rows accumulated in g (no LDS image: gathers from L2);  2: D = 3, rows scattered
// onto the sampled faces' vertices
template <int MODE>
__device__ __forceinline__ void bg_rows(BgLds &L, const BgJob &J) {
    constexpr bool D3 = MODE != 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the long-row loop and its readlane chain must not be "divergent")
    const int nrows = J.r1 - J.r0, D = J.D;
    const int rpt = (nrows + kBgThreads - 1) / kBgThreads;  // 1 .. 4 rows per thread, kBgThreads apart: a wave's rows are contiguous
                                                            // (12-byte loads / stores of consecutive lanes coalesce; 48 bytes apart they are 64 transactions)
    const float *__restrict__ own = J.own;
    const float *__restrict__ oth = J.oth;
    const float c_own = J.c_own, c_oth = J.c_oth;
    const bool own_first = J.side == 0;  // side x: the own term opens the row (the oracle's first loop), side y: it closes it
    // requested before anything else: this thread's rows and their indices
    int jown[kBgPer];
    float wx[kBgPer], wy[kBgPer], wz[kBgPer];  // (plain scalar arrays: an array of the packed 12-byte P3 is not promoted to registers -- it lived in scratch)
#pragma unroll
    for (int u = 0; u < kBgPer; ++u) {
        const bool mine = u < rpt && tid + u * kBgThreads < nrows;
        jown[u] = mine ? J.idx_own[J.r0 + tid + u * kBgThreads] : 0;
        wx[u] = 0.0f; wy[u] = 0.0f; wz[u] = 0.0f;
        if (D3 && mine) {
            const P3 t = *reinterpret_cast<const P3 *>(own + (size_t)(J.r0 + tid + u * kBgThreads) * 3);
            wx[u] = t.x; wy[u] = t.y; wz[u] = t.z;
        }
    }
    // (round 6) the own term's rows of the other side, requested as soon as the indices are here and consumed in phase (5): the request
    // used to sit inside phase (5)'s per-row branch with a full wait behind it -- one more dependent trip to the L2 per row.
    // UNCONDITIONAL loads throughout this function (a thread without a row reads row 0): `cond ? load : 0` compiles to a branch
    // around the load and `s_waitcnt vmcnt(0)` behind it, i.e. the loads of a group leave one after the other.
    float ojx[kBgPer], ojy[kBgPer], ojz[kBgPer];
#pragma unroll
    for (int u = 0; u < kBgPer; ++u) {
        ojx[u] = 0.0f; ojy[u] = 0.0f; ojz[u] = 0.0f;
        if (D3) {
            const P3 t = *reinterpret_cast<const P3 *>(oth + (size_t)jown[u] * 3);
            ojx[u] = t.x; ojy[u] = t.y; ojz[u] = t.z;
        }
    }
    BG_STAMP(0);
#pragma unroll
    for (int q = 0; q < kBgPer / 2; ++q) L.cnt2[kBgPer / 2 * tid + q] = 0u;  // (the placement of every round leaves the counters at zero)

    for (int c0 = 0; c0 < J.S; c0 += kBgChunk) {
        const int cn = J.S - c0 < kBgChunk ? J.S - c0 : kBgChunk;
        const bool first = c0 == 0, last = c0 + kBgChunk >= J.S;
        // (1) this round's rows of the other side -> LDS image; counts: which of my rows does other row c0 + jl point at
        if (tid == 0) L.nbig = 0u;
        int il[kBgPer];
#pragma unroll
        for (int v = 0; v < kBgPer; ++v) {
            const int jl = tid + v * kBgThreads;
            il[v] = J.idx_oth[c0 + (jl < cn ? jl : 0)];
        }
        static_assert(kBgPer == 4, "the pin below names four values");
        // (an empty asm that names all four: the optimiser otherwise sinks every load back into the branch of its `jl < cn` select,
        //  with a full wait behind it -- four dependent trips to memory instead of one)
        asm volatile("" : "+v"(il[0]), "+v"(il[1]), "+v"(il[2]), "+v"(il[3]));
#pragma unroll
        for (int v = 0; v < kBgPer; ++v) il[v] = tid + v * kBgThreads < cn ? il[v] - J.r0 : -1;
        __syncthreads();
        BG_STAMP(1);
        BG_STOP(1);
#pragma unroll
        for (int v = 0; v < kBgPer; ++v)
            if ((unsigned int)il[v] < (unsigned int)nrows) atomicAdd(&L.cnt2[il[v] >> 1], (il[v] & 1) ? 0x10000u : 1u);
        __syncthreads();
        BG_STAMP(2);
        BG_STOP(2);
        // (2) exclusive scan over the rows (thread t: slots 4t .. 4t+3), long lists registered
        unsigned int cc[kBgPer], s4 = 0u;
#pragma unroll
        for (int q = 0; q < kBgPer / 2; ++q) {
            const unsigned int cw = L.cnt2[kBgPer / 2 * tid + q];
            cc[2 * q] = cw & 0xFFFFu; cc[2 * q + 1] = cw >> 16;
            s4 += cc[2 * q] + cc[2 * q + 1];
        }
        const unsigned int inc = wave_scan_incl(s4);
        if (lane == 63) L.wsum[wv] = inc;
        __syncthreads();
        {
            unsigned int base = 0u;
            {
#pragma unroll
                for (int q = 0; q < kBgThreads / 64; q += 4) {
                    const uint4 t = *reinterpret_cast<const uint4 *>(&L.wsum[q]);
                    base += (q < wv ? t.x : 0u) + (q + 1 < wv ? t.y : 0u) + (q + 2 < wv ? t.z : 0u) + (q + 3 < wv ? t.w : 0u);
                }
            }
            unsigned int ex = base + inc - s4;
#pragma unroll
            for (int q = 0; q < kBgPer / 2; ++q) {
                const unsigned int lo = ex, hi = ex + cc[2 * q];
                *reinterpret_cast<unsigned int *>(&L.start[kBgPer * tid + 2 * q]) = lo | (hi << 16);
                ex = hi + cc[2 * q + 1];
            }
            if (tid == kBgThreads - 1) L.start[kBgRows] = (unsigned short)ex;
        }
        __syncthreads();
        BG_STAMP(3);
        BG_STOP(3);
        // (3) placement, from the back of every list (arrival order inside a list: whatever the atomics give; put right below);
        //     the owners register their rows with long lists and hand the waves what they hold of them
#pragma unroll
        for (int v = 0; v < kBgPer; ++v)
            if ((unsigned int)il[v] < (unsigned int)nrows) {
                const int sh = (il[v] & 1) * 16;
                const unsigned int old = atomicSub(&L.cnt2[il[v] >> 1], 1u << sh);
                L.list[L.start[il[v]] + ((old >> sh) & 0xFFFFu) - 1u] = (unsigned short)(tid + v * kBgThreads);
            }
#pragma unroll
        for (int u = 0; u < kBgPer; ++u) {
            const int lr = tid + u * kBgThreads;
            if (u < rpt && lr < nrows && (int)L.start[lr + 1] - (int)L.start[lr] > kBgSmall) {
                const unsigned int k = atomicAdd(&L.nbig, 1u);
                L.big[k] = (unsigned short)lr;
                if (D3 && k < (unsigned int)kBgBigW) L.bigw[k] = float4{wx[u], wy[u], wz[u], __builtin_bit_cast(float, jown[u])};
            }
        }
        __syncthreads();
        BG_STAMP(4);
        BG_STOP(4);
        // (4) rows with long lists: one wave each -- bitmap sort in place, lane-parallel gathers, an ordered readlane chain
        const unsigned int nb = (unsigned int)__builtin_amdgcn_readfirstlane((int)L.nbig);
        for (unsigned int k = (unsigned int)wv; k < nb; k += kBgSortWaves) {
            const int lr = __builtin_amdgcn_readfirstlane((int)L.big[k]);
            const int start = __builtin_amdgcn_readfirstlane((int)L.start[lr]);
            const int c = __builtin_amdgcn_readfirstlane((int)L.start[lr + 1]) - start;
            unsigned int *bm = L.bitmap[wv];
            bm[lane] = 0u;
            bm[lane + 64] = 0u;
            wave_lds_sync();
            for (int e = lane; e < c; e += 64) {
                const unsigned int j = L.list[start + e];
                atomicOr(&bm[j >> 5], 1u << (j & 31u));
            }
            wave_lds_sync();
            unsigned int w0 = bm[2 * lane], w1 = bm[2 * lane + 1];
            const unsigned int p = __popc(w0) + __popc(w1);
            const unsigned int pin = wave_scan_incl(p);
            int pos = start + (int)(pin - p);
            while (w0) { L.list[pos++] = (unsigned short)(64 * lane + __ffs(w0) - 1); w0 &= w0 - 1u; }
            while (w1) { L.list[pos++] = (unsigned short)(64 * lane + 32 + __ffs(w1) - 1); w1 &= w1 - 1u; }
            wave_lds_sync();
            const size_t i = (size_t)(J.r0 + lr);
            if (D3) {
                P3 w;
                int jo;
                if (k < (unsigned int)kBgBigW) {
                    const float4 t = L.bigw[k];
                    w = P3{t.x, t.y, t.z}; jo = __builtin_bit_cast(int, t.w);
                } else {
                    w = *reinterpret_cast<const P3 *>(own + i * 3); jo = J.idx_own[i];
                }
                P3 ot{0.0f, 0.0f, 0.0f};  // the own term
                if ((first && own_first) || (last && !own_first)) {
                    const P3 o = *reinterpret_cast<const P3 *>(oth + (size_t)jo * 3);
                    ot = P3{c_own * (w.x - o.x), c_own * (w.y - o.y), c_own * (w.z - o.z)};
                }
                P3 a{0.0f, 0.0f, 0.0f};
                if (!first) a = MODE == 2 ? J.part[lr] : *reinterpret_cast<const P3 *>(J.g + i * 3);
                else if (own_first) a = P3{0.0f + ot.x, 0.0f + ot.y, 0.0f + ot.z};
                for (int e0 = 0; e0 < c; e0 += 64) {
                    float tx = 0.0f, ty = 0.0f, tz = 0.0f;
                    if (e0 + lane < c) {
                        const P3 o = *reinterpret_cast<const P3 *>(oth + (size_t)(c0 + L.list[start + e0 + lane]) * 3);
                        tx = c_oth * (o.x - w.x); ty = c_oth * (o.y - w.y); tz = c_oth * (o.z - w.z);
                    }
                    const int n = c - e0 < 64 ? c - e0 : 64;
                    for (int l = 0; l < n; ++l) {
                        a.x = a.x - lane_val(tx, l); a.y = a.y - lane_val(ty, l); a.z = a.z - lane_val(tz, l);
                    }
                }
                if (last && !own_first) a = P3{a.x + ot.x, a.y + ot.y, a.z + ot.z};
                if (lane == 0) {
                    if (MODE == 2) { if (last) bg_scatter_sample(J, (int)i, a); else J.part[lr] = a; }
                    else *reinterpret_cast<P3 *>(J.g + i * 3) = a;
                }
            } else {
                const int jo = J.idx_own[i];
                for (int d = 0; d < D; ++d) {
                    const float wd = own[i * D + d];
                    const float ot = c_own * (wd - oth[(size_t)jo * D + d]);
                    float a = first ? (own_first ? 0.0f + ot : 0.0f) : J.g[i * D + d];
                    for (int e0 = 0; e0 < c; e0 += 64) {
                        float t = 0.0f;
                        if (e0 + lane < c) t = c_oth * (oth[(size_t)(c0 + L.list[start + e0 + lane]) * D + d] - wd);
                        const int n = c - e0 < 64 ? c - e0 : 64;
                        for (int l = 0; l < n; ++l) a = a - lane_val(t, l);
                    }
                    if (last && !own_first) a = a + ot;
                    if (lane == 0) J.g[i * D + d] = a;
                }
            }
        }
        BG_STAMP(5);
        BG_STOP(5);
        // (5) the owners: short lists ordered in registers, subtracted in order (no barrier between (4) and (5): the waves
        //     touch only the lists / rows of the long rows, which the owners skip).  Row by row: two rows at a time with every
        //     stage's LDS reads issued together was measured slower at every list cap (4 / 6 / 8: the arrays of both rows spill)
#pragma unroll
        for (int u = 0; u < kBgPer; ++u) {
            const int lr = tid + u * kBgThreads;
            if (!(u < rpt && lr < nrows)) continue;
            const int start = L.start[lr], c = (int)L.start[lr + 1] - start;
            if (c > kBgSmall || (c == 0 && !first && !last)) continue;
            unsigned int e[kBgSmall];
#pragma unroll
            for (int k = 0; k < kBgSmall; ++k) e[k] = k < c ? (unsigned int)L.list[start + k] : 0xFFFFu;
            if (c > 4) {
                ce(e[0], e[1]); ce(e[2], e[3]); ce(e[4], e[5]); ce(e[6], e[7]);
                ce(e[0], e[2]); ce(e[1], e[3]); ce(e[4], e[6]); ce(e[5], e[7]);
                ce(e[1], e[2]); ce(e[5], e[6]); ce(e[0], e[4]); ce(e[3], e[7]);
                ce(e[1], e[5]); ce(e[2], e[6]);
                ce(e[1], e[4]); ce(e[3], e[6]);
                ce(e[2], e[4]); ce(e[3], e[5]);
                ce(e[3], e[4]);
            } else if (c > 2) {
                ce(e[0], e[1]); ce(e[2], e[3]); ce(e[0], e[2]); ce(e[1], e[3]); ce(e[1], e[2]);
            } else if (c == 2) {
                ce(e[0], e[1]);
            }
            const size_t i = (size_t)(J.r0 + lr);
            const int jo = jown[u];
            if (D3) {
                const P3 w{wx[u], wy[u], wz[u]};
                P3 ot{0.0f, 0.0f, 0.0f};
                if ((first && own_first) || (last && !own_first)) {
                    const P3 o{ojx[u], ojy[u], ojz[u]};
                    ot = P3{c_own * (w.x - o.x), c_own * (w.y - o.y), c_own * (w.z - o.z)};
                }
                P3 a{0.0f, 0.0f, 0.0f};
                if (!first) a = MODE == 2 ? J.part[lr] : *reinterpret_cast<const P3 *>(J.g + i * 3);
                else if (own_first) a = P3{0.0f + ot.x, 0.0f + ot.y, 0.0f + ot.z};
                {
#pragma unroll
                    for (int h = 0; h < kBgSmall; h += 4) {  // four gathers in flight
                        if (h >= c) break;
                        float ox[4], oy[4], oz[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {  // (unconditional: a slot beyond the list re-reads the round's first row)
                            const P3 t = *reinterpret_cast<const P3 *>(oth + (size_t)(c0 + (h + k < c ? e[h + k] : 0u)) * 3);
                            ox[k] = t.x; oy[k] = t.y; oz[k] = t.z;
                        }
                        asm volatile("" : "+v"(ox[0]), "+v"(oy[0]), "+v"(oz[0]), "+v"(ox[1]), "+v"(oy[1]), "+v"(oz[1]), "+v"(ox[2]), "+v"(oy[2]),
                                          "+v"(oz[2]), "+v"(ox[3]), "+v"(oy[3]), "+v"(oz[3]));  // (all four requested before the first is used)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (h + k < c) {
                                a.x = a.x - c_oth * (ox[k] - w.x);
                                a.y = a.y - c_oth * (oy[k] - w.y);
                                a.z = a.z - c_oth * (oz[k] - w.z);
                            }
                    }
                }
                if (last && !own_first) a = P3{a.x + ot.x, a.y + ot.y, a.z + ot.z};
                if (MODE == 2) { if (last) bg_scatter_sample(J, (int)i, a); else J.part[lr] = a; }
                else *reinterpret_cast<P3 *>(J.g + i * 3) = a;
            } else {
                for (int d = 0; d < D; ++d) {
                    const float wd = own[i * D + d];
                    const float ot = c_own * (wd - oth[(size_t)jo * D + d]);
                    float a = first ? (own_first ? 0.0f + ot : 0.0f) : J.g[i * D + d];
#pragma unroll
                    for (int k = 0; k < kBgSmall; ++k)
                        if (k < c) a = a - c_oth * (oth[(size_t)(c0 + e[k]) * D + d] - wd);
                    if (last && !own_first) a = a + ot;
                    J.g[i * D + d] = a;
                }
            }
        }
        BG_STAMP(6);
        if (!last) __syncthreads();  // the next round reuses the image / start / list (and reads the accumulators back)
    }
}

// blk: the block's number among the launch's row blocks; one_side >= 0: only that side's rows are formed (B * nsplit blocks)
__device__ __forceinline__ BgJob bg_job(const float *x, int N, const float *y, int M, int D, const int32_t *idx_x,
                                        const int32_t *idx_y, float ca, float cb, int nsplit, int blk, int one_side = -1) {
    const int part = blk % nsplit, bs = blk / nsplit;
    const int b = one_side >= 0 ? bs : bs >> 1, side = one_side >= 0 ? one_side : bs & 1;  // side 0: gx, 1: gy
    BgJob J{};
    J.side = side; J.D = D; J.b = b;
    J.R = side ? M : N; J.S = side ? N : M;
    J.own = (side ? y : x) + (size_t)b * J.R * D;
    J.oth = (side ? x : y) + (size_t)b * J.S * D;
    J.idx_own = (side ? idx_y : idx_x) + (size_t)b * J.R;
    J.idx_oth = (side ? idx_x : idx_y) + (size_t)b * J.S;
    J.c_own = side ? cb : ca; J.c_oth = side ? ca : cb;
    const int per = (J.R + nsplit - 1) / nsplit;
    J.r0 = part * per < J.R ? part * per : J.R;
    J.r1 = J.r0 + per < J.R ? J.r0 + per : J.R;
    return J;
}

// (round 6) spare blocks behind the row blocks of fx3d_chamfer_sampled_bwd's first launch: block e builds the ordered gather's tables of
// mesh e of a side -- the draws bucketed by face, sample_gather.h: they depend on the draws alone -- and leaves them in the scratch for the
// gather's blocks (7 us of a gather block's work done while the rows are formed)
struct SgTabJobs {
    const int32_t *face_idx[2];  // (n, B) draws of side x / y; nullptr: no tables for that side
    unsigned char *blob[2];
    int F[2], n[2], B;
};
static_assert(kBgThreads == sg::kSgThreads, "row blocks and table blocks share a launch");
#ifdef FX3D_RIDE_NOINLINE  // (A/B: the passengers as calls, the row blocks' register allocation as without them)
#define FX3D_RIDE_FN __attribute__((noinline))
#else
#define FX3D_RIDE_FN __forceinline__
#endif
__device__ FX3D_RIDE_FN void bg_passenger_block(int e, const SgTabJobs &tj, int ntab, const meshreg::Ride &ride, unsigned char *tab_lds) {
    if (e >= ntab) {  // the regularisers' adjoint (mesh_reg.h): reads the vertices and the forward's unit rows, writes gverts
        meshreg::adj_block(ride, e - ntab);
        return;
    }
    int side = 0;
    if (tj.face_idx[0]) { if (e >= tj.B) { e -= tj.B; side = 1; } } else side = 1;
    sg::SgMesh m{};
    m.face_idx = tj.face_idx[side] + (size_t)e * tj.n[side];
    m.F = tj.F[side]; m.n = tj.n[side];
    sg::sg_tables(tab_lds, m);
    sg::sg_tables_store(tab_lds, m.F, m.n, tj.blob[side] + (size_t)e * sg::sg_blob_bytes(m.F, m.n));
}
// RIDE: the launch carries passengers behind its nrow row blocks (fx3d_chamfer_sampled_bwd's first launch); fx3d_chamfer_bwd's own
// launches are compiled without them
template <bool D3, bool RIDE>
__global__ __launch_bounds__(kBgThreads, FX3D_BG_OCC) void chamfer_bwd_gather_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ y, int M, int D, const int32_t *__restrict__ idx_x,
    const int32_t *__restrict__ idx_y, float ca, float cb, float *__restrict__ gx, float *__restrict__ gy, int nsplit, int npass,
    SgTabJobs tj, int ntab, meshreg::Ride ride) {
    int blk = (int)blockIdx.x, one_side = -1;
    if constexpr (RIDE) {
        // the passengers take the FIRST npass block numbers: blocks are handed to the CUs in order, and behind a full round of row blocks
        // (eight meshes: 256) the tables started a row block's time late -- the launch took 21 us instead of 10
        if (blk < npass) {
            extern __shared__ __attribute__((aligned(16))) unsigned char tab_lds[];
            bg_passenger_block(blk, tj, ntab, ride, tab_lds);
            return;
        }
        blk -= npass;
        one_side = !gy ? 0 : (!gx ? 1 : -1);  // (a fitting loop differentiates w.r.t. one mesh: no blocks for the other side's rows)
    }
    __shared__ BgLds L;
    BgJob J = bg_job(x, N, y, M, D, idx_x, idx_y, ca, cb, nsplit, blk, one_side);
    if (!(J.side ? gy : gx)) return;  // (a side nobody asked for: fx3d_chamfer_sampled_bwd differentiates w.r.t. one mesh)
    J.g = (J.side ? gy : gx) + (size_t)J.b * J.R * D;
    bg_rows<D3 ? 0 : 1>(L, J);
}

// Adjoint of chamfer_distance(sample_points(m_x), sample_points(m_y)) w.r.t. the meshes' vertices, SCATTER form, in one launch: the
// gradient w.r.t. the sampled points is formed exactly as in chamfer_bwd_gather_kernel (D = 3), then every finished row goes onto
// the three vertices of its sampled face with global float atomics (sums in arrival order) instead of being written out.  A side
// without a mesh gradient (gverts == nullptr) is skipped.  (The ORDERED form -- fx3d_chamfer_sampled_bwd with the vertex -> face
// tables -- is two launches: chamfer_bwd_gather_kernel writes the rows, sample_bwd_gather_kernel gathers them, sample_gather.h.)
__global__ __launch_bounds__(kBgThreads, FX3D_BG_OCC) void chamfer_sampled_bwd_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ y, int M, const int32_t *__restrict__ idx_x,
    const int32_t *__restrict__ idx_y, float ca, float cb, SampledSide sx, SampledSide sy, int nsplit) {
    __shared__ BgLds L;
    extern __shared__ __attribute__((aligned(16))) unsigned char bg_dyn[];  // the accumulators between rounds (n > 4096 samples)
    BgJob J = bg_job(x, N, y, M, 3, idx_x, idx_y, ca, cb, nsplit, (int)blockIdx.x);
    J.smp = J.side ? sy : sx;
    if (!J.smp.gverts) return;
    J.part = reinterpret_cast<P3 *>(bg_dyn);
    bg_rows<2>(L, J);
}

// blocks per (cloud, side): ONE round of blocks on the chip (a block is a chain of dependent phases: a second round doubles the
// time), a block never owns fewer than 256 rows nor more than kBgRows.  Measured (kernel us, B x N): 32 x 4096: 15 / 11.7 / 18 / 28
// at 1 / 4 / 8 / 16 blocks per side; 8 x 5000: 17 / 13.5 / 11.6 at 1 / 4 / 16; 1 x 16384: 31 / 23 / 20 at 1 / 8 / 16.
int bg_nsplit(int B, int maxr, int sides = 2) {
    int nsplit = device_cus() / (sides * B);
    if (nsplit > maxr / 256) nsplit = maxr / 256;
    if (nsplit < 1) nsplit = 1;
    while ((maxr + nsplit - 1) / nsplit > kBgRows) ++nsplit;
    return nsplit;
}

}  // namespace

extern "C" {

fx3d_status fx3d_chamfer_bwd(const float *x, int32_t N, const float *y, int32_t M, int32_t B,
                             int32_t D, const int32_t *idx_x, const int32_t *idx_y, float w1,
                             float w2, float gout, int64_t B_global, float *gx, float *gy,
                             fx3d_stream_t s) {
    fx3d_status rc = chamfer_check_shapes("fx3d_chamfer_bwd", x, N, y, M, B, D);
    if (rc) return rc;
    FX3D_REQUIRE(idx_x && idx_y && gx && gy, "fx3d_chamfer_bwd: null pointer");
    FX3D_REQUIRE(B_global >= B, "fx3d_chamfer_bwd: B_global < B");
    hipStream_t st = as_stream(s);
    const float ca = gout * w1 * (float)(6.0 / ((double)D * N * (double)B_global));
    const float cb = gout * w2 * (float)(6.0 / ((double)D * M * (double)B_global));
    const int nsplit = bg_nsplit(B, N > M ? N : M);
    if ((long long)2 * B * nsplit < (1ll << 30) && !opt(OPT_BWD_GLOBAL_ATOMICS)) {
        ProfileScope prof("chamfer_bwd", st);
        if (D == 3)
            hipLaunchKernelGGL((chamfer_bwd_gather_kernel<true, false>), dim3(2 * B * nsplit), dim3(kBgThreads), 0, st, x, N, y, M, D, idx_x,
                               idx_y, ca, cb, gx, gy, nsplit, 0, SgTabJobs{}, 0, meshreg::Ride{});
        else
            hipLaunchKernelGGL((chamfer_bwd_gather_kernel<false, false>), dim3(2 * B * nsplit), dim3(kBgThreads), 0, st, x, N, y, M, D, idx_x,
                               idx_y, ca, cb, gx, gy, nsplit, 0, SgTabJobs{}, 0, meshreg::Ride{});
        FX3D_LAUNCH_CHECK();
        return FX3D_OK;
    }
    const long long total = (long long)B * (N + M);
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    {
        ProfileScope prof("chamfer_bwd", st);
        hipLaunchKernelGGL(chamfer_bwd_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, st, x, N, y, M,
                           B, D, idx_x, idx_y, ca, cb, gx, gy);
        hipLaunchKernelGGL(chamfer_bwd_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0, st, x, N, y, M,
                           B, D, idx_x, idx_y, ca, cb, gx, gy);
    }
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

#ifdef FX3D_BG_PROBE
__attribute__((visibility("default"))) int fx3d_debug_bg_probe(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bg_probe), sizeof(unsigned long long) * 1024 * 8);
}
#endif

}  // extern "C"

namespace {
struct SampledArgs {
    const int32_t *faces; int32_t Vmax, Fmax; const int32_t *face_idx; const float *r1, *r2; float *gverts;
    const int32_t *vf_rowptr, *vf_ent;
};
size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
// the ordered form's scratch: the adjoint's rows of both sides (published by a mesh's blocks, gathered by the last of them)
size_t sampled_ws_bytes(int N, int M, int B) {
    // (the tables' blobs are sized for the largest face count the ordered form takes at these draw counts: the caller need not name Fmax)
    return al256(sizeof(float) * 3 * (size_t)N * B) + al256(sizeof(float) * 3 * (size_t)M * B) +
           al256(sg::sg_blob_bytes(sg::kSgMaxF, N) * (size_t)B) + al256(sg::sg_blob_bytes(sg::kSgMaxF, M) * (size_t)B);
}
fx3d_status chamfer_sampled_bwd_impl(const char *fn, const float *x, int32_t N, const float *y, int32_t M, int32_t B, const int32_t *idx_x,
                                     const int32_t *idx_y, float w1, float w2, float gout, int64_t B_global, const SampledArgs &ax,
                                     const SampledArgs &ay, int32_t accumulate, const sg::SgStep &step_x, void *ws, size_t ws_bytes,
                                     fx3d_stream_t s, const fx3d_mesh_reg *reg = nullptr) {
    fx3d_status rc = chamfer_check_shapes(fn, x, N, y, M, B, 3);
    if (rc) return rc;
    FX3D_REQUIRE(idx_x && idx_y, "%s: null index array", fn);
    FX3D_REQUIRE(ax.gverts || ay.gverts, "%s: no gradient requested", fn);
    FX3D_REQUIRE(!ax.gverts || (ax.faces && ax.face_idx && ax.r1 && ax.r2 && ax.Vmax > 0 && ax.Fmax > 0), "%s: incomplete mesh of x", fn);
    FX3D_REQUIRE(!ay.gverts || (ay.faces && ay.face_idx && ay.r1 && ay.r2 && ay.Vmax > 0 && ay.Fmax > 0), "%s: incomplete mesh of y", fn);
    FX3D_REQUIRE((ax.vf_rowptr == nullptr) == (ax.vf_ent == nullptr) && (ay.vf_rowptr == nullptr) == (ay.vf_ent == nullptr),
                 "%s: vf_rowptr and vf_ent go together", fn);
    FX3D_REQUIRE(B_global >= B, "%s: B_global < B", fn);
    hipStream_t st = as_stream(s);
    const float ca = gout * w1 * (float)(6.0 / (3.0 * N * (double)B_global));
    const float cb = gout * w2 * (float)(6.0 / (3.0 * M * (double)B_global));
    const int nsplit = bg_nsplit(B, N > M ? N : M);
    FX3D_REQUIRE((long long)2 * B * nsplit < (1ll << 30), "%s: batch too large", fn);
    // Ordered form (no float atomics, bit-identical to the oracle): every requested side comes with its vertex -> face table and
    // fits the gather's LDS tables.  Otherwise: the float-atomic scatter.
    const bool ordered = (!ax.gverts || (ax.vf_rowptr && sg::sg_fits(ax.Fmax, N))) && (!ay.gverts || (ay.vf_rowptr && sg::sg_fits(ay.Fmax, M)));
    FX3D_REQUIRE(!step_x.vel || (ordered && ax.gverts), "%s: the optimiser step needs the ordered form (vertex -> face table, draws that fit "
                 "fx3d_sample_points_bwd_ordered)", fn);
    // the fit iteration's regularisers (fx3d_mesh_reg): their adjoint writes gverts_x, the sampling adjoint then accumulates on top
    meshreg::Ride ride_v{};
    const meshreg::Ride *ride = nullptr;
    if (reg) {
        FX3D_REQUIRE(ax.gverts && reg->V == (int64_t)ax.Vmax * B, "%s: fx3d_mesh_reg covers the source batch (V = %lld, B * Vmax = %lld)", fn,
                     (long long)reg->V, (long long)ax.Vmax * B);
        rc = mesh_reg_plan(reg, gout, ax.gverts, accumulate, st, fn, &ride_v);
        if (rc) return rc;
        if (ordered && ride_v.c_lap != 0.0f && ride_v.c_edge != 0.0f) ride = &ride_v;  // rides in the rows launch
        else {  // (a weight of zero drops its term, fx3d_mesh_losses_bwd knows how; the scatter form is one launch with no room)
            rc = mesh_reg_adjoint_standalone(reg, gout, ax.gverts, accumulate, st);
            if (rc) return rc;
        }
        accumulate = 1;
    }
    if (ordered) {
        // two launches: the chamfer adjoint's rows of the requested sides into the scratch (bit-identical to fx3d_chamfer_bwd), then the
        // ordered gather of every side (+ the optimiser step).  (One launch -- gather blocks behind the row blocks, waiting on a counter --
        // was built and measured equal inside the fit loop's graph, 104 against 105 us per iteration: the gather's tables, 7 us, do
        // overlap the rows there, but its tail -- acquire, staging, walk: ~9 us -- does not shrink; not kept: a bounded spin and a
        // dispatch-order argument for nothing.)
        const size_t need = sampled_ws_bytes(N, M, B);
        if (!ws || ws_bytes < need) {
            set_error("%s: workspace too small (%zu < %zu bytes)", fn, ws ? ws_bytes : (size_t)0, need);
            return FX3D_ERR_WORKSPACE;
        }
        char *w = static_cast<char *>(ws);
        float *gsx = reinterpret_cast<float *>(w); w += al256(sizeof(float) * 3 * (size_t)N * B);
        float *gsy = reinterpret_cast<float *>(w); w += al256(sizeof(float) * 3 * (size_t)M * B);
        unsigned char *tbx = reinterpret_cast<unsigned char *>(w); w += al256(ax.gverts ? sg::sg_blob_bytes(ax.Fmax, N) * (size_t)B : 0);
        unsigned char *tby = reinterpret_cast<unsigned char *>(w);
        {
            // the rows of the requested sides + (behind them, one block per mesh and side) the gather's tables
            SgTabJobs tj{};
            tj.B = B;
            size_t dyn = 0;
            int ntab = 0;
            if (ax.gverts) { tj.face_idx[0] = ax.face_idx; tj.blob[0] = tbx; tj.F[0] = ax.Fmax; tj.n[0] = N; dyn = std::max(dyn, sg::sg_tables_lds_bytes(ax.Fmax, N)); ntab += B; }
            if (ay.gverts) { tj.face_idx[1] = ay.face_idx; tj.blob[1] = tby; tj.F[1] = ay.Fmax; tj.n[1] = M; dyn = std::max(dyn, sg::sg_tables_lds_bytes(ay.Fmax, M)); ntab += B; }
            constexpr size_t kTabLds = 112 * 1024;  // beside the row blocks' 38 KB of static LDS
            if (dyn > kTabLds) {  // (meshes of > ~20 000 faces: every gather block builds its own tables, as fx3d_sample_points_bwd's do)
                tj = SgTabJobs{};
                dyn = 0; ntab = 0;
                tbx = tby = nullptr;
            } else {
                const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&chamfer_bwd_gather_kernel<true, true>), (int)kTabLds, "chamfer_bwd_gather_kernel");
                if (arc != FX3D_OK) return arc;
            }
            ProfileScope prof("chamfer_sampled_bwd", st);
            const int sides = (ax.gverts ? 1 : 0) + (ay.gverts ? 1 : 0), npass = ntab + (ride ? ride->nadj : 0);
            const int ns = bg_nsplit(B, sides == 2 ? (N > M ? N : M) : (ax.gverts ? N : M), sides);  // (row blocks of the requested sides only)
            hipLaunchKernelGGL((chamfer_bwd_gather_kernel<true, true>), dim3(npass + sides * B * ns), dim3(kBgThreads), dyn, st, x, N, y, M,
                               3, idx_x, idx_y, ca, cb, ax.gverts ? gsx : nullptr, ay.gverts ? gsy : nullptr, ns, npass, tj, ntab,
                               ride ? *ride : meshreg::Ride{});
            FX3D_LAUNCH_CHECK();
        }
        if (ax.gverts) {
            rc = sg::launch_sample_bwd_gather(ax.faces, ax.Vmax, ax.Fmax, B, N, ax.face_idx, ax.r1, ax.r2, gsx, ax.vf_rowptr, ax.vf_ent, ax.gverts,
                                              accumulate, step_x, st, tbx);
            if (rc) return rc;
        }
        if (ay.gverts) {
            rc = sg::launch_sample_bwd_gather(ay.faces, ay.Vmax, ay.Fmax, B, M, ay.face_idx, ay.r1, ay.r2, gsy, ay.vf_rowptr, ay.vf_ent, ay.gverts,
                                              accumulate, sg::SgStep{}, st, tby);
            if (rc) return rc;
        }
        return FX3D_OK;
    }
    if (!accumulate) {
        if (ax.gverts) FX3D_HIP(hipMemsetAsync(ax.gverts, 0, sizeof(float) * 3 * (size_t)ax.Vmax * B, st));
        if (ay.gverts) FX3D_HIP(hipMemsetAsync(ay.gverts, 0, sizeof(float) * 3 * (size_t)ay.Vmax * B, st));
    }
    const SampledSide sx{ax.faces, ax.face_idx, ax.r1, ax.r2, ax.gverts, ax.Vmax, ax.Fmax}, sy{ay.faces, ay.face_idx, ay.r1, ay.r2, ay.gverts, ay.Vmax, ay.Fmax};
    // sides beyond kBgChunk samples (the reference's default is 5000): the accumulators between the rounds live in LDS
    const int maxr = N > M ? N : M;
    const size_t dyn = maxr > kBgChunk ? sizeof(P3) * (size_t)((maxr + nsplit - 1) / nsplit) : 0;
    if (dyn > 0) {  // static BgLds (~38 KB) + up to 48 KB of accumulators: beyond the 64 KB a kernel gets without an opt-in
        const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&chamfer_sampled_bwd_kernel), (int)(sizeof(P3) * kBgRows),
                                                   "chamfer_sampled_bwd_kernel");
        if (arc != FX3D_OK) return arc;
    }
    ProfileScope prof("chamfer_sampled_bwd", st);
    hipLaunchKernelGGL(chamfer_sampled_bwd_kernel, dim3(2 * B * nsplit), dim3(kBgThreads), dyn, st, x, N, y, M, idx_x, idx_y, ca,
                       cb, sx, sy, nsplit);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}
}  // namespace

extern "C" {

fx3d_status fx3d_chamfer_sampled_bwd_workspace_bytes(int32_t N, int32_t M, int32_t B, size_t *bytes) {
    FX3D_REQUIRE(bytes, "fx3d_chamfer_sampled_bwd_workspace_bytes: null output");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0, "fx3d_chamfer_sampled_bwd_workspace_bytes: bad sizes");
    *bytes = sampled_ws_bytes(N, M, B);
    return FX3D_OK;
}

fx3d_status fx3d_chamfer_sampled_bwd(const float *x, int32_t N, const float *y, int32_t M, int32_t B, const int32_t *idx_x,
                                     const int32_t *idx_y, float w1, float w2, float gout, int64_t B_global,
                                     const int32_t *faces_x, int32_t Vmax_x, int32_t Fmax_x, const int32_t *face_idx_x,
                                     const float *r1_x, const float *r2_x, float *gverts_x, const int32_t *faces_y,
                                     int32_t Vmax_y, int32_t Fmax_y, const int32_t *face_idx_y, const float *r1_y,
                                     const float *r2_y, float *gverts_y, int32_t accumulate, const int32_t *vf_rowptr_x,
                                     const int32_t *vf_ent_x, const int32_t *vf_rowptr_y, const int32_t *vf_ent_y, void *ws,
                                     size_t ws_bytes, fx3d_stream_t s) {
    const SampledArgs ax{faces_x, Vmax_x, Fmax_x, face_idx_x, r1_x, r2_x, gverts_x, vf_rowptr_x, vf_ent_x};
    const SampledArgs ay{faces_y, Vmax_y, Fmax_y, face_idx_y, r1_y, r2_y, gverts_y, vf_rowptr_y, vf_ent_y};
    return chamfer_sampled_bwd_impl("fx3d_chamfer_sampled_bwd", x, N, y, M, B, idx_x, idx_y, w1, w2, gout, B_global, ax, ay, accumulate,
                                    sg::SgStep{}, ws, ws_bytes, s);
}

fx3d_status fx3d_chamfer_sampled_bwd_step(const float *x, int32_t N, const float *y, int32_t M, int32_t B, const int32_t *idx_x,
                                          const int32_t *idx_y, float w1, float w2, float gout, const int32_t *faces_x,
                                          int32_t V, int32_t F, const int32_t *face_idx_x, const float *r1_x, const float *r2_x,
                                          float *gverts_x, int32_t accumulate, const int32_t *vf_rowptr_x, const int32_t *vf_ent_x,
                                          float rho, float eta, float *vel, float *params, const float *base, float *out,
                                          uint64_t *ctr, uint64_t inc, void *ws, size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(vel && params && base && out && gverts_x, "fx3d_chamfer_sampled_bwd_step: null pointer");
    const SampledArgs ax{faces_x, V, F, face_idx_x, r1_x, r2_x, gverts_x, vf_rowptr_x, vf_ent_x};
    const SampledArgs ay{};
    const sg::SgStep st{rho, eta, vel, params, base, out, reinterpret_cast<unsigned long long *>(ctr), (unsigned long long)inc};
    return chamfer_sampled_bwd_impl("fx3d_chamfer_sampled_bwd_step", x, N, y, M, B, idx_x, idx_y, w1, w2, gout, B, ax, ay, accumulate, st,
                                    ws, ws_bytes, s);
}

fx3d_status fx3d_chamfer_sampled_bwd_step_reg(const float *x, int32_t N, const float *y, int32_t M, int32_t B, const int32_t *idx_x,
                                              const int32_t *idx_y, float w1, float w2, float gout, const int32_t *faces_x,
                                              int32_t V, int32_t F, const int32_t *face_idx_x, const float *r1_x, const float *r2_x,
                                              float *gverts_x, int32_t accumulate, const int32_t *vf_rowptr_x, const int32_t *vf_ent_x,
                                              float rho, float eta, float *vel, float *params, const float *base, float *out,
                                              uint64_t *ctr, uint64_t inc, void *ws, size_t ws_bytes, const fx3d_mesh_reg *reg,
                                              fx3d_stream_t s) {
    FX3D_REQUIRE(vel && params && base && out && gverts_x && reg, "fx3d_chamfer_sampled_bwd_step_reg: null pointer");
    const SampledArgs ax{faces_x, V, F, face_idx_x, r1_x, r2_x, gverts_x, vf_rowptr_x, vf_ent_x};
    const SampledArgs ay{};
    const sg::SgStep st{rho, eta, vel, params, base, out, reinterpret_cast<unsigned long long *>(ctr), (unsigned long long)inc};
    return chamfer_sampled_bwd_impl("fx3d_chamfer_sampled_bwd_step_reg", x, N, y, M, B, idx_x, idx_y, w1, w2, gout, B, ax, ay, accumulate, st,
                                    ws, ws_bytes, s, reg);
}

}  // extern "C"

extern "C" {

// Value AND gradient in one ABI call (the shape of `gradient(() -> chamfer_distance(A, B), ...)`, benchmarks/metrics.jl:24-38,
// examples/fit_mesh.jl:106-110): the forward with indices and the adjoint are queued back to back on the stream, the
// nearest-neighbour indices stay in the caller's scratch (or go to idx_x / idx_y when the caller wants them).  Two ABI calls
// leave the device idle between the launches for as long as the host needs for the second call.
fx3d_status fx3d_chamfer_fwd_bwd_workspace_bytes(int32_t N, int32_t M, int32_t B, int32_t D, size_t *bytes) {
    FX3D_REQUIRE(bytes, "fx3d_chamfer_fwd_bwd_workspace_bytes: null output");
    size_t fwd = 0;
    const fx3d_status rc = fx3d_chamfer_workspace_bytes(N, M, B, D, &fwd);
    if (rc) return rc;
    fwd = (fwd + 255) & ~(size_t)255;
    *bytes = fwd + ((sizeof(int32_t) * (size_t)B * ((size_t)N + M) + 255) & ~(size_t)255);
    return FX3D_OK;
}

fx3d_status fx3d_chamfer_fwd_bwd(const float *x, int32_t N, const float *y, int32_t M, int32_t B, int32_t D, float w1,
                                 float w2, float gout, int64_t B_global, float *loss_dev, float *loss_host, float *gx,
                                 float *gy, int32_t *idx_x, int32_t *idx_y, void *ws, size_t ws_bytes, fx3d_stream_t s) {
    fx3d_status rc = chamfer_check_shapes("fx3d_chamfer_fwd_bwd", x, N, y, M, B, D);
    if (rc) return rc;
    FX3D_REQUIRE(loss_dev && gx && gy, "fx3d_chamfer_fwd_bwd: null output pointer");
    FX3D_REQUIRE(B_global >= B, "fx3d_chamfer_fwd_bwd: B_global < B");
    size_t fwd = 0, need = 0;
    rc = fx3d_chamfer_workspace_bytes(N, M, B, D, &fwd);
    if (rc) return rc;
    fx3d_chamfer_fwd_bwd_workspace_bytes(N, M, B, D, &need);
    fwd = (fwd + 255) & ~(size_t)255;
    if (!ws || ws_bytes < need) {
        set_error("fx3d_chamfer_fwd_bwd: workspace too small (%zu < %zu bytes)", ws ? ws_bytes : (size_t)0, need);
        return FX3D_ERR_WORKSPACE;
    }
    int32_t *ix = reinterpret_cast<int32_t *>(static_cast<char *>(ws) + fwd);
    int32_t *iy = ix + (size_t)B * N;
    if (idx_x) ix = idx_x;
    if (idx_y) iy = idx_y;
    rc = chamfer_forward(x, N, y, M, B, D, loss_dev, (long long)B_global, w1, w2, ix, iy, ws, fwd, as_stream(s),
                         "fx3d_chamfer_fwd_bwd");
    if (rc) return rc;
    rc = fx3d_chamfer_bwd(x, N, y, M, B, D, ix, iy, w1, w2, gout, B_global, gx, gy, s);
    if (rc) return rc;
    if (loss_host) {
        FX3D_HIP(hipMemcpyAsync(loss_host, loss_dev, sizeof(float), hipMemcpyDeviceToHost, as_stream(s)));
        FX3D_HIP(hipStreamSynchronize(as_stream(s)));
    }
    return FX3D_OK;
}

}  // extern "C"
