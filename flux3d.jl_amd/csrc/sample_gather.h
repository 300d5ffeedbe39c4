// Ordered (atomic-free) adjoint of the barycentric sampling  samples = w1 v[f1] + w2 v[f2] + w3 v[f3]
// (src/transforms/mesh_func.jl:60-73; the reference's CPU adjoint is a gather + weighted sum, Zygote accumulates the three
// getindex pullbacks): the gradient w.r.t. a vertex v is
//     g[v] = base[v] + sum over the incident (face f, corner t) of v, ascending (f, t), of  inner(f, t)
//     inner(f, t) = 0 + sum over the samples k drawn on f, k ASCENDING, of  w_t(k) * gs[k]              (Float32, unfused)
// -- the order in which oracle/flux3d_oracle.c: fx3d_oracle_sample_points_bwd adds, so the result is the oracle's bit for bit and
// the same from run to run (round 5 scattered the 9 products of every sample with global float atomics: arrival order).
// sg_parts(V) 1024-thread blocks per mesh (each finishes a share of the vertices from tables of the whole mesh: built by the block itself
// -- fx3d_sample_points_bwd -- or, behind fx3d_chamfer_sampled_bwd's first launch, loaded as the blob a spare block of that launch built:
// sg_tables_store / _load), everything between the first and the last global access in LDS: the draws are bucketed by face with a counting sort (integer atomics, a block scan,
// placement from the back of every list), every list is put in ascending order in place (up to eight draws: a sorting network in
// the face's thread; longer: a wave through a bitmap over the sample ids), (gs, sqrt(r1), r2) of every draw is STAGED with coalesced
// loads -- one round trip to memory for the whole mesh --, faces of more than eight draws get their nine sums from nine lanes each,
// a thread per (vertex, corner) ENTRY of the vertex -> face table (fx3d_build_vertex_faces: a CSR over the vertices, entries
// face * 4 + corner ascending) forms inner(f, t) into LDS and a thread per vertex adds its entries in order.  The vertex's thread
// may go straight on to the optimiser step (SgStep: Flux.Optimise.Momentum + offset, examples/fit_mesh.jl:87-88,108-110) -- it owns
// the finished gradient row.  What fits: 4 F + 22.25 n + 18 K bytes of LDS <= kSgMaxLds (n <= ~5300 draws at the 5120 faces of the
// tutorial's sphere; the reference's default is 5000) -- other meshes keep the scatter.  (Measured on the way, one mesh of 5000 draws:
// gs / r1 / r2 gathered from memory per (entry, draw) 70 us; staged in list order 26; staged by draw id, one block 20; sixteen
// blocks 13 on an even mesh but 38 inside the fit loop, whose grown faces draw 20 - 50 each: table of the big faces 28, a thread
// per entry 19; the tables handed over from the first launch 12; draws and optimiser state requested up front 10.)
// Included by sampler.hip (fx3d_sample_points_bwd; the kernel) and chamfer_bwd.hip (fx3d_chamfer_sampled_bwd's second launch).
#pragma once
#include "fx3d_common.h"

namespace fx3d {
namespace sg {

#ifdef FX3D_SG_PROBE   // phase stamps of block 0 (wall clock, 100 MHz) for tools/sampled_bwd_time.py; never in the shipped build
__device__ unsigned long long g_sg_probe[32];
#define SG_STAMP(n) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_sg_probe[n] = wall_clock64(); } while (0)
#else
#define SG_STAMP(n) do { } while (0)
#endif

constexpr int kSgThreads = 1024;
constexpr int kSgWaves = kSgThreads / 64;
constexpr int kSgSmall = 8;         // lists up to this length are ordered in the consumer's registers
constexpr int kSgMaxN = 16384;      // samples per mesh: ids and list offsets are 16-bit in LDS
constexpr int kSgMaxF = 32768;      // faces per mesh (per-face counters: 16 bits each)
constexpr size_t kSgMaxLds = 156 * 1024;  // of the CU's 160 KB (the kernels that run this keep their static LDS under 4 KB)
constexpr int kSgRow = 5;           // staged floats per draw: gs (3), sqrt(r1), r2

struct SgMesh {                 // ONE mesh of the batch (pointers already offset to it)
    const int32_t *faces;       // (3, Fmax) mesh-local, 0-based -- unused by the gather itself (the table names the corners)
    const int32_t *face_idx;    // (n) the draws' faces
    const float *r1, *r2;       // (n) the draws' uniforms
    const float *gs;            // (3, n) gradient w.r.t. the samples
    const int32_t *vf_rowptr;   // (V + 1) vertex -> first entry
    const int32_t *vf_ent;      // (rowptr[V]) face * 4 + corner, ascending within a vertex
    float *gverts;              // (3, V) result; with `accumulate` also the base
    int V, F, n, accumulate;
};
struct SgStep {                 // optional optimiser step by the vertex's thread (vel == nullptr: none); meshes of equal vertex counts
    float rho, eta;
    float *vel, *x;             // (3, V) Momentum's velocity and the parameters (the offsets)
    const float *base;          // (3, V) the source mesh's vertices
    float *out;                 // (3, V) base + x: the next iteration's mesh
    unsigned long long *ctr;    // advanced by inc (the sampling seeds' device counter); may be nullptr
    unsigned long long inc;
};

constexpr int kSgInnerCap = 1536;   // (vertex, corner) entries whose sums a block parks in LDS at a time (a block's share of a mesh: ~1000)
struct SgLayout {
    size_t cnt, start, list, big, wsum, staged, inner, total;
    int bmwords;
};
__host__ __device__ inline SgLayout sg_layout(int F, int n) {
    SgLayout L{};
    size_t o = 0;
    L.cnt = o;    o += sizeof(unsigned int) * (size_t)((F + 2) / 2);             // two 16-bit counters per word
    L.start = o;  o += (sizeof(unsigned short) * (size_t)(F + 2) + 3) & ~(size_t)3;
    L.list = o;   o += (sizeof(unsigned short) * (size_t)(n + 1) + 3) & ~(size_t)3;
    L.big = o;    o += (sizeof(unsigned short) * (size_t)(n / (kSgSmall + 1) + 2) + 3) & ~(size_t)3;
    L.wsum = o;   o += sizeof(unsigned int) * (kSgWaves + 4);
    L.bmwords = (n + 31) / 32;
    L.staged = o;  // kSgRow floats per draw, in list order; the waves' bitmaps (long lists) live here before the staging
    const size_t st = sizeof(float) * kSgRow * (size_t)n, bm = sizeof(unsigned int) * (size_t)kSgWaves * (size_t)L.bmwords;
    o += st > bm ? st : bm;
    L.inner = o;  o += sizeof(float) * 3 * (size_t)kSgInnerCap;
    L.total = (o + 15) & ~(size_t)15;
    return L;
}
// meshes the ordered form takes (the others keep the float-atomic scatter)
inline bool sg_fits(int Fmax, int n) {
    return n > 0 && n <= kSgMaxN && Fmax > 0 && Fmax <= kSgMaxF && sg_layout(Fmax, n).total <= kSgMaxLds;
}

__device__ __forceinline__ void sg_ce(unsigned int &a, unsigned int &b) {
    const unsigned int lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}
__device__ __forceinline__ void sg_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned int sg_dpp_zero(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ unsigned int sg_wave_scan_incl(unsigned int v) {
    v += sg_dpp_zero<0x111, 0xF>(v);
    v += sg_dpp_zero<0x112, 0xF>(v);
    v += sg_dpp_zero<0x114, 0xF>(v);
    v += sg_dpp_zero<0x118, 0xF>(v);
    v += sg_dpp_zero<0x142, 0xA>(v);
    v += sg_dpp_zero<0x143, 0xC>(v);
    return v;
}
// All kSgThreads threads of a block call these together (they synchronise).  `lds`: sg_layout(m.F, m.n).total bytes.
// sg_tables: phases (0) - (4), the draws bucketed by face and every list in ascending order -- reads face_idx only.
__device__ __forceinline__ void sg_tables(unsigned char *lds, const SgMesh &m) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SgLayout L = sg_layout(m.F, m.n);
    unsigned int *cnt2 = reinterpret_cast<unsigned int *>(lds + L.cnt);
    unsigned short *start = reinterpret_cast<unsigned short *>(lds + L.start);
    unsigned short *list = reinterpret_cast<unsigned short *>(lds + L.list);
    unsigned short *big = reinterpret_cast<unsigned short *>(lds + L.big);
    unsigned int *bitmap = reinterpret_cast<unsigned int *>(lds + L.staged);  // (dead before the staging)
    unsigned int *wsum = reinterpret_cast<unsigned int *>(lds + L.wsum);  // [kSgWaves] + nbig at [kSgWaves]
    const int F = m.F, n = m.n;
    const int nw = (F + 2) / 2;
    SG_STAMP(0);
    // the thread's draws, requested before anything else and kept for the placement (phase 3): loaded inside the loops they were one
    // trip to the L2 per sweep, twice (`k < n ? load : skip` is a branch around the load with a full wait behind it).  kSgHold sweeps
    // live in registers (6144 draws; the ordered form takes ~5300 at 5120 faces), draws beyond them are read where they are used.
    constexpr int kSgHold = 6;
    unsigned int fk[kSgHold];
#pragma unroll
    for (int u = 0; u < kSgHold; ++u) {
        const int k = tid + u * kSgThreads;
        fk[u] = (unsigned int)m.face_idx[k < n ? k : 0];  // (n >= 1)
    }
    asm volatile("" : "+v"(fk[0]), "+v"(fk[1]), "+v"(fk[2]), "+v"(fk[3]), "+v"(fk[4]), "+v"(fk[5]));  // (all requested together: see chamfer_bwd.hip)
    static_assert(kSgHold == 6, "the pin above names six values");
    for (int i = tid; i < nw; i += kSgThreads) cnt2[i] = 0u;
    if (tid == 0) wsum[kSgWaves] = 0u;
    __syncthreads();
    // (1) draws per face
#pragma unroll
    for (int u = 0; u < kSgHold; ++u) {
        const unsigned int f = fk[u];
        if (tid + u * kSgThreads < n && f < (unsigned int)F) atomicAdd(&cnt2[f >> 1], (f & 1u) ? 0x10000u : 1u);
    }
    for (int k = tid + kSgHold * kSgThreads; k < n; k += kSgThreads) {
        const unsigned int f = (unsigned int)m.face_idx[k];
        if (f < (unsigned int)F) atomicAdd(&cnt2[f >> 1], (f & 1u) ? 0x10000u : 1u);
    }
    __syncthreads();
    SG_STAMP(1);
    // (2) exclusive scan over the faces: thread t takes the faces [t fpt, (t + 1) fpt), fpt even
    const int fpt = 2 * ((F + 2 + 2 * kSgThreads - 1) / (2 * kSgThreads));  // (fpt / 2) * kSgThreads words >= nw
    unsigned int s = 0u;
    for (int q = 0; q < fpt / 2; ++q) {
        const int w = (fpt / 2) * tid + q;
        if (w < nw) { const unsigned int cw = cnt2[w]; s += (cw & 0xFFFFu) + (cw >> 16); }
    }
    const unsigned int inc = sg_wave_scan_incl(s);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    {
        unsigned int ex = inc - s;
        for (int w = 0; w < kSgWaves; ++w) ex += w < wv ? wsum[w] : 0u;
        for (int q = 0; q < fpt / 2; ++q) {
            const int w = (fpt / 2) * tid + q;
            if (w < nw) {
                const unsigned int cw = cnt2[w];
                const int f0 = 2 * w;
                start[f0] = (unsigned short)ex;
                ex += cw & 0xFFFFu;
                if (f0 + 1 <= F) start[f0 + 1] = (unsigned short)ex;
                ex += cw >> 16;
            }
        }
        // (faces are 0 .. F - 1; entry F closes the last list: the word of face F holds a zero count, so it was written above)
    }
    __syncthreads();
    SG_STAMP(2);
    // (3) placement from the back of every list (arrival order; put right below), long lists registered
    auto place = [&](unsigned int f, int k) {
        if (f < (unsigned int)F) {
            const unsigned int sh = (f & 1u) * 16u;
            const unsigned int old = atomicSub(&cnt2[f >> 1], 1u << sh);
            list[(unsigned int)start[f] + ((old >> sh) & 0xFFFFu) - 1u] = (unsigned short)k;
        }
    };
#pragma unroll
    for (int u = 0; u < kSgHold; ++u)
        if (tid + u * kSgThreads < n) place(fk[u], tid + u * kSgThreads);
    for (int k = tid + kSgHold * kSgThreads; k < n; k += kSgThreads) place((unsigned int)m.face_idx[k], k);
    for (int f = tid; f < F; f += kSgThreads)
        if ((int)start[f + 1] - (int)start[f] > kSgSmall) big[atomicAdd(&wsum[kSgWaves], 1u)] = (unsigned short)f;
    __syncthreads();
    SG_STAMP(3);
    // (4) every list in ascending order, in place: short ones by their face's thread, long ones by a wave through a bitmap over
    //     the sample ids (lane l owns a contiguous run of words)
    for (int f = tid; f < F; f += kSgThreads) {  // (in registers: an insertion sort in LDS was a chain of dependent LDS round trips, 6 us)
        const int s0 = start[f], c = (int)start[f + 1] - s0;
        if (c < 2 || c > kSgSmall) continue;
        unsigned int id[kSgSmall];
#pragma unroll
        for (int k = 0; k < kSgSmall; ++k) id[k] = k < c ? (unsigned int)list[s0 + k] : 0xFFFFu;
        if (c > 4) {
            sg_ce(id[0], id[1]); sg_ce(id[2], id[3]); sg_ce(id[4], id[5]); sg_ce(id[6], id[7]);
            sg_ce(id[0], id[2]); sg_ce(id[1], id[3]); sg_ce(id[4], id[6]); sg_ce(id[5], id[7]);
            sg_ce(id[1], id[2]); sg_ce(id[5], id[6]); sg_ce(id[0], id[4]); sg_ce(id[3], id[7]);
            sg_ce(id[1], id[5]); sg_ce(id[2], id[6]);
            sg_ce(id[1], id[4]); sg_ce(id[3], id[6]);
            sg_ce(id[2], id[4]); sg_ce(id[3], id[5]);
            sg_ce(id[3], id[4]);
        } else if (c > 2) {
            sg_ce(id[0], id[1]); sg_ce(id[2], id[3]); sg_ce(id[0], id[2]); sg_ce(id[1], id[3]); sg_ce(id[1], id[2]);
        } else {
            sg_ce(id[0], id[1]);
        }
#pragma unroll
        for (int k = 0; k < kSgSmall; ++k)
            if (k < c) list[s0 + k] = (unsigned short)id[k];
    }
    {
        const unsigned int nb = (unsigned int)__builtin_amdgcn_readfirstlane((int)wsum[kSgWaves]);
        unsigned int *bm = bitmap + (size_t)wv * L.bmwords;
        const int wpl = (L.bmwords + 63) / 64;
        for (unsigned int b = (unsigned int)wv; b < nb; b += kSgWaves) {
            const int f = __builtin_amdgcn_readfirstlane((int)big[b]);
            const int s0 = __builtin_amdgcn_readfirstlane((int)start[f]);
            const int c = __builtin_amdgcn_readfirstlane((int)start[f + 1]) - s0;
            for (int w = lane; w < L.bmwords; w += 64) bm[w] = 0u;
            sg_wave_lds_sync();
            for (int e = lane; e < c; e += 64) {
                const unsigned int j = list[s0 + e];
                atomicOr(&bm[j >> 5], 1u << (j & 31u));
            }
            sg_wave_lds_sync();
            unsigned int p = 0u;
            for (int q = 0; q < wpl; ++q) {
                const int w = lane * wpl + q;
                if (w < L.bmwords) p += __popc(bm[w]);
            }
            const unsigned int pin = sg_wave_scan_incl(p);
            int pos = s0 + (int)(pin - p);
            for (int q = 0; q < wpl; ++q) {
                const int w = lane * wpl + q;
                unsigned int bits = w < L.bmwords ? bm[w] : 0u;
                while (bits) { list[pos++] = (unsigned short)(32 * w + __ffs(bits) - 1); bits &= bits - 1u; }
            }
            sg_wave_lds_sync();
        }
    }
    __syncthreads();
    SG_STAMP(4);
}

// The tables as a blob in memory (round 6, late): the bytes [start, staged) of the LDS layout -- list offsets, ordered lists, the faces of
// many draws and their count.  fx3d_chamfer_sampled_bwd builds them with spare blocks of the launch that forms the adjoint's rows (they need
// the draws only) and the gather's blocks load them: 7 us of every gather block's 19 leave the iteration's critical path.
__host__ __device__ inline size_t sg_blob_bytes(int F, int n) {
    const SgLayout L = sg_layout(F, n);
    return (L.staged - L.start + 15) & ~(size_t)15;
}
__device__ __forceinline__ void sg_tables_store(const unsigned char *lds, int F, int n, unsigned char *blob) {
    const SgLayout L = sg_layout(F, n);
    const unsigned int *src = reinterpret_cast<const unsigned int *>(lds + L.start);  // (both offsets are multiples of 4)
    unsigned int *dst = reinterpret_cast<unsigned int *>(blob);
    const int nw = (int)((L.staged - L.start) / 4);
    for (int i = threadIdx.x; i < nw; i += kSgThreads) dst[i] = src[i];
}
__device__ __forceinline__ void sg_tables_load(unsigned char *lds, int F, int n, const unsigned char *blob) {
    const SgLayout L = sg_layout(F, n);
    unsigned int *dst = reinterpret_cast<unsigned int *>(lds + L.start);
    const unsigned int *src = reinterpret_cast<const unsigned int *>(blob);
    const int nw = (int)((L.staged - L.start) / 4);
    for (int i = threadIdx.x; i < nw; i += kSgThreads) dst[i] = src[i];
    __syncthreads();
}
// LDS a block needs to BUILD the tables only (no staged rows: the long lists' bitmaps end the layout)
__host__ __device__ inline size_t sg_tables_lds_bytes(int F, int n) {
    const SgLayout L = sg_layout(F, n);
    return (L.staged + sizeof(unsigned int) * (size_t)kSgWaves * (size_t)L.bmwords + 15) & ~(size_t)15;
}

// sg_finish: phases (5) - (6) for the vertices [vb, ve) of the mesh -- a mesh's vertices may be shared out over several blocks
// (each with the mesh's tables in its own LDS -- built, ~7 us of one CU's time, or loaded, ~1 us; walking 2500 vertices would be 12 us).
__device__ __forceinline__ void sg_finish(unsigned char *lds, const SgMesh &m, const SgStep &st, int vb, int ve) {
    const int tid = threadIdx.x;
    const SgLayout L = sg_layout(m.F, m.n);
    const unsigned short *start = reinterpret_cast<const unsigned short *>(lds + L.start);
    const unsigned short *list = reinterpret_cast<const unsigned short *>(lds + L.list);
    float *staged = reinterpret_cast<float *>(lds + L.staged);
    const int n = m.n;
    // (5) stage (gs, sqrt(r1), r2) of every draw, row k = draw k: coalesced loads (consecutive lanes, consecutive draws), the only trip
    //     to memory for the draws' data, every load in flight at once.  (Staged in LIST order the loads were 25 k scattered dwords,
    //     one cache line each: 6 us; the walk below pays one more LDS read per draw for the indirection instead.)
    {
        constexpr int kU = 4;
        for (int k0 = tid; k0 < n; k0 += kU * kSgThreads) {
            float gx[kU], gy[kU], gz[kU], a1[kU], a2[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int k = k0 + u * kSgThreads;
                const size_t kk = (size_t)(k < n ? k : k0);
                const P3 t3 = *reinterpret_cast<const P3 *>(m.gs + 3 * kk);
                gx[u] = t3.x; gy[u] = t3.y; gz[u] = t3.z;
                a1[u] = m.r1[kk]; a2[u] = m.r2[kk];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int k = k0 + u * kSgThreads;
                if (k < n) {
                    float *row = staged + (size_t)kSgRow * k;
                    row[0] = gx[u]; row[1] = gy[u]; row[2] = gz[u]; row[3] = sqrtf(a1[u]); row[4] = a2[u];
                }
            }
        }
    }
    __syncthreads();
    SG_STAMP(5);
    // (5b) faces with more than kSgSmall draws (a handful once a fitted mesh has grown uneven: its large faces draw 20 - 50 of 5000):
    //      their nine sums -- (corner, coordinate) over the face's draws in order -- by nine lanes each, seven faces per wave at a time,
    //      parked in the bytes of the (dead) per-face counters; the face's first list slot then names its row of the table.  Walked
    //      as ordinary entries they made the waves of their three vertices run as many eight-entry steps as the face has draws
    //      (37 us in the fit loop's graph against 13 on an even mesh).
    {
        float *bigin = reinterpret_cast<float *>(lds + L.cnt);
        const int capb = (int)(sizeof(unsigned int) * (size_t)((m.F + 2) / 2) / (9 * sizeof(float)));
        const unsigned short *big = reinterpret_cast<const unsigned short *>(lds + L.big);
        const unsigned int *wsum = reinterpret_cast<const unsigned int *>(lds + L.wsum);
        int nb = (int)wsum[kSgWaves];
        nb = nb < capb ? nb : capb;  // (more long lists than the table holds: the rest are walked the slow way)
        unsigned short *listw = reinterpret_cast<unsigned short *>(lds + L.list);
        const int lane = tid & 63, wv = tid >> 6, grp = lane / 9, sub = lane % 9;
        for (int b0 = wv * 7; b0 < nb; b0 += kSgWaves * 7) {
            const int b = b0 + grp;
            const bool on = grp < 7 && b < nb;
            const int f = on ? (int)big[b] : 0;
            const int s0 = start[f], c = on ? (int)start[f + 1] - s0 : 0;
            const int t = sub / 3, d = sub % 3;
            float acc = 0.0f;
            for (int i = 0; i < c; ++i) {
                const float *row = staged + (size_t)kSgRow * listw[s0 + i];
                const float uu = row[3], vv = row[4];
                const float w = t == 0 ? 1.0f - uu : (t == 1 ? uu * (1.0f - vv) : uu * vv);
                acc = acc + w * row[d];
            }
            if (on) bigin[9 * b + sub] = acc;
            sg_wave_lds_sync();  // (every lane is done with the lists of this sweep's faces)
            if (on && sub == 0) listw[s0] = (unsigned short)(0x8000u | (unsigned int)b);
        }
    }
    __syncthreads();
    // (6) a thread per (vertex, corner) ENTRY forms inner(f, t) over the face's staged rows (ascending draws) and parks it in LDS; a
    //     thread per vertex then adds its entries in order, writes the row and, if asked, applies the optimiser step to it.  (A thread
    //     per vertex walking eight entries together cost a wave ~130 instructions per step for as many steps as its longest list:
    //     12 us on an even mesh, 28 us once the fitted mesh's faces had grown uneven; per entry a step is ~20 instructions and all
    //     sixteen waves share the entries.)  Vertices in batches of kSgThreads, a batch's entries in chunks of kSgInnerCap.
    {
        float *inner = reinterpret_cast<float *>(lds + L.inner);
        const float *bigin = reinterpret_cast<const float *>(lds + L.cnt);
        for (int vb0 = vb; vb0 < ve; vb0 += kSgThreads) {
            const int vend = vb0 + kSgThreads < ve ? vb0 + kSgThreads : ve;
            const int v = vb0 + tid;
            const bool ok = v < vend;
            const int e0 = ok ? m.vf_rowptr[v] : 0, e1 = ok ? m.vf_rowptr[v + 1] : 0;
            const int E_lo = m.vf_rowptr[vb0], E_hi = m.vf_rowptr[vend];
            P3 a{0.0f, 0.0f, 0.0f};
            if (ok && m.accumulate) a = *reinterpret_cast<const P3 *>(m.gverts + 3 * (size_t)v);
            // the optimiser's state of this vertex, requested here and used behind the sums: read where it is used it was nine
            // dependent trips to memory at the kernel's end (the stores between them may alias for all the compiler knows)
            P3 s_vel{0.0f, 0.0f, 0.0f}, s_x{0.0f, 0.0f, 0.0f}, s_base{0.0f, 0.0f, 0.0f};
            if (ok && st.vel) {
                s_vel = *reinterpret_cast<const P3 *>(st.vel + 3 * (size_t)v);
                s_x = *reinterpret_cast<const P3 *>(st.x + 3 * (size_t)v);
                s_base = *reinterpret_cast<const P3 *>(st.base + 3 * (size_t)v);
            }
            for (int ec = E_lo; ec < E_hi; ec += kSgInnerCap) {
                const int eend = ec + kSgInnerCap < E_hi ? ec + kSgInnerCap : E_hi;
                for (int e = ec + tid; e < eend; e += kSgThreads) {
                    const unsigned int en = (unsigned int)m.vf_ent[e];
                    const int f = (int)(en >> 2), t = (int)(en & 3u);
                    const int s0 = start[f];
                    int c = (int)start[f + 1] - s0;
                    float ax = 0.0f, ay = 0.0f, az = 0.0f;
                    if (c > kSgSmall) {  // a face of many draws: its sums are in the table (5b) unless the table was full
                        const unsigned int slot = list[s0];
                        if (slot & 0x8000u) {
                            const float *r9 = bigin + 9 * (size_t)(slot & 0x7fffu) + 3 * t;
                            ax = r9[0]; ay = r9[1]; az = r9[2];
                            c = 0;
                        }
                    }
                    for (int i = 0; i < c; ++i) {
                        const float *row = staged + (size_t)kSgRow * list[s0 + i];
                        const float uu = row[3], vv = row[4];
                        const float w = t == 0 ? 1.0f - uu : (t == 1 ? uu * (1.0f - vv) : uu * vv);
                        ax = ax + w * row[0]; ay = ay + w * row[1]; az = az + w * row[2];
                    }
                    float *o = inner + 3 * (size_t)(e - ec);
                    o[0] = ax; o[1] = ay; o[2] = az;
                }
                __syncthreads();
                {
                    const int lo = e0 > ec ? e0 : ec, hi = e1 < eend ? e1 : eend;
                    for (int e = lo; e < hi; ++e) {
                        const float *o = inner + 3 * (size_t)(e - ec);
                        a.x = a.x + o[0]; a.y = a.y + o[1]; a.z = a.z + o[2];
                    }
                }
                __syncthreads();  // (the next chunk, or the next batch of vertices, overwrites the sums)
            }
            if (ok) {
                *reinterpret_cast<P3 *>(m.gverts + 3 * (size_t)v) = a;
                if (st.vel) {  // fx3d_momentum_step_offset's arithmetic (mesh.hip: momentum_offset_kernel)
                    const float g3[3] = {a.x, a.y, a.z}, v3[3] = {s_vel.x, s_vel.y, s_vel.z}, x3[3] = {s_x.x, s_x.y, s_x.z},
                                b3[3] = {s_base.x, s_base.y, s_base.z};
                    float vn[3], xn[3], on[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        vn[d] = (st.rho * v3[d]) + (-st.eta * g3[d]);
                        xn[d] = (1.0f * x3[d]) + (1.0f * vn[d]);
                        on[d] = (1.0f * b3[d]) + (1.0f * xn[d]);
                    }
                    *reinterpret_cast<P3 *>(st.vel + 3 * (size_t)v) = P3{vn[0], vn[1], vn[2]};
                    *reinterpret_cast<P3 *>(st.x + 3 * (size_t)v) = P3{xn[0], xn[1], xn[2]};
                    *reinterpret_cast<P3 *>(st.out + 3 * (size_t)v) = P3{on[0], on[1], on[2]};
                }
            }
        }
    }
    if (st.vel && st.ctr && tid == 0 && vb == 0) *st.ctr += st.inc;
    SG_STAMP(6);
}

// blocks that share a mesh's vertices: ~160 vertices each, at most 16 (every block holds the mesh's tables)
__host__ __device__ inline int sg_parts(int V) {
    const int g = (V + 159) / 160;
    return g < 1 ? 1 : (g > 16 ? 16 : g);
}
__host__ __device__ inline void sg_part_range(int V, int parts, int j, int &vb, int &ve) {
    const int per = (V + parts - 1) / parts;
    vb = j * per < V ? j * per : V;
    ve = vb + per < V ? vb + per : V;
}

// the gather's launch (sampler.hip), also the second launch of fx3d_chamfer_sampled_bwd's ordered form (chamfer_bwd.hip)
fx3d_status launch_sample_bwd_gather(const int32_t *faces_padded, int Vmax, int Fmax, int B, int n, const int32_t *face_idx, const float *r1,
                                     const float *r2, const float *gs, const int32_t *vf_rowptr, const int32_t *vf_ent, float *gverts,
                                     int accumulate, const SgStep &step, hipStream_t st, const unsigned char *tables = nullptr);

}  // namespace sg
}  // namespace fx3d
