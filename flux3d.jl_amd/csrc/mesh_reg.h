// mesh_reg.h -- device bodies of the two fit_mesh regularisers (0.1 laplacian_loss + edge_loss, /root/reference/examples/fit_mesh.jl:80-83,
// src/metrics/mesh.jl:9-15,24-35) shared by their own launches (mesh.hip) and by the launches of the fit iteration they can ride in
// (round 6): the forward as extra blocks of the sampler's draw launch (sampler.hip), the adjoint as extra blocks of the launch that
// forms the chamfer adjoint's rows (chamfer_bwd.hip) -- two graph nodes (~ 4.4 us each) and their kernels' time off the iteration's
// critical path.  One source for the arithmetic: the riding blocks produce the bits of fx3d_mesh_losses / fx3d_mesh_losses_bwd.
#pragma once
#include "fx3d_common.h"

namespace fx3d {
namespace meshreg {

// XCD-aware work order of the grid-stride kernels (round 4): block L runs on XCD L % 8, each with its own L2.  With work item =
// blockIdx the eight XCDs each see every eighth 256-item chunk, so the vertices two neighbouring chunks share are fetched into two
// L2s; with the chunks of an XCD made CONTIGUOUS (XCD x takes logical blocks [start_x, start_x + n_x)) the neighbours share one.
// A bijection of [0, nb) for any nb.  (Partial sums stay in the slot of the PHYSICAL block.)
__device__ __forceinline__ long long xcd_logical_block(unsigned int L, unsigned int nb) {
    const unsigned int x = L & 7u, j = L >> 3, q = nb >> 3, r = nb & 7u;
    return (long long)x * q + (x < r ? x : r) + j;
}

// ---- laplacian_loss (src/metrics/mesh.jl:9-15): row i of L*verts' in ascending column order ----
__device__ __forceinline__ void lap_row(const float *__restrict__ verts,
                                        const int32_t *__restrict__ rowptr,
                                        const int32_t *__restrict__ colind,
                                        const float *__restrict__ vals, long long i, float &s0,
                                        float &s1, float &s2) {
    s0 = 0.0f; s1 = 0.0f; s2 = 0.0f;
    const int k1 = rowptr[i + 1];
    // eight entries per sweep (a mesh vertex has ~7: itself + 6 neighbours): every (weight, column) load first, then every
    // vertex gather, then the sums in ascending column order -- two dependent round trips per row instead of two per entry
    for (int k0 = rowptr[i]; k0 < k1; k0 += 8) {
        float w[8];
        int col[8];
        P3 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e < k1 ? k0 + e : k1 - 1;
            w[e] = vals[k];
            col[e] = colind[k];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const P3 *>(verts + 3ll * col[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k0 + e < k1) {
                s0 = s0 + w[e] * v[e].x;
                s1 = s1 + w[e] * v[e].y;
                s2 = s2 + w[e] * v[e].z;
            }
    }
}

// The same row with its (weight, column) entries taken from a WAVE-staged copy in LDS (round 4): the 64 rows of a wave are one
// contiguous run of the CSR arrays, which the wave copies with coalesced loads (every byte of the two streams crosses the L1
// once); a thread per row reading its own entries from memory has lanes ~28 bytes apart, eight partially used lines per load.
// Same operations in the same order as lap_row.
constexpr int kLapStage = 1024;  // entries a wave stages (64 rows of a closed mesh hold ~450); longer runs read memory directly
__device__ __forceinline__ void lap_row_lds(const float *__restrict__ verts, const float *w_s, const int *c_s, int k0, int k1,
                                            float &s0, float &s1, float &s2) {
    s0 = 0.0f; s1 = 0.0f; s2 = 0.0f;
    for (; k0 < k1; k0 += 8) {
        float w[8];
        int col[8];
        P3 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e < k1 ? k0 + e : k1 - 1;
            w[e] = w_s[k];
            col[e] = c_s[k];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const P3 *>(verts + 3ll * col[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k0 + e < k1) {
                s0 = s0 + w[e] * v[e].x;
                s1 = s1 + w[e] * v[e].y;
                s2 = s2 + w[e] * v[e].z;
            }
    }
}

// ---- forward: blocks [0, gV) take Laplacian rows, blocks [gV, gV + gE) edges; every block publishes one Float64 partial, the last
//      arriver adds each range in index order and writes both means (+ the weighted total).  By-product: u_r = (L v)_r / ||(L v)_r||
//      and 1/deg(r) per vertex (float4), all the adjoint needs.
struct FwdArgs {
    const float *verts;
    long long V;
    const int32_t *rowptr, *colind;
    const float *vals;
    const int32_t *e1, *e2;
    long long E;
    float target;
    int gV, gE;
    double *partials;
    unsigned int *ticket;
    float w_lap, w_edge;
    const float *base;
    float *loss_lap, *loss_edge, *total;
    float4 *u_out;
};
template <int NT>
struct FwdLds {
    double sm[NT / 64];
    int is_last;
    float w_stage[(NT / 64) * kLapStage];
    int c_stage[(NT / 64) * kLapStage];
};
// block `blk` of the gV + gE blocks (NT threads each) of one forward
template <int NT>
__device__ __forceinline__ void fwd_block(const FwdArgs &A, int blk, FwdLds<NT> &L) {
    const float *__restrict__ verts = A.verts;
    const int32_t *__restrict__ rowptr = A.rowptr, *__restrict__ colind = A.colind;
    const float *__restrict__ vals = A.vals;
    const long long V = A.V, E = A.E;
    const int gV = A.gV, gE = A.gE;
    double acc = 0.0;
    if (blk < gV) {
        // (round 4) the rows of a wave from its staged copy of the CSR run: laplacian_loss_kernel
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        float *w_s = L.w_stage + wv * kLapStage;
        int *c_s = L.c_stage + wv * kLapStage;
        for (long long ib = xcd_logical_block(blk, gV) * NT + wv * 64; ib < V; ib += (long long)gV * NT) {  // (wave-uniform; XCD-contiguous)
            const long long i = ib + lane;
            const bool okr = i < V;
            const int k0 = rowptr[okr ? i : V], k1 = okr ? rowptr[i + 1] : k0;
            const int k_lo = __builtin_amdgcn_readfirstlane(k0), k_hi = __builtin_amdgcn_readlane(k1, 63);
            const int nent = k_hi - k_lo;
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, invdeg = 0.0f;  // invdeg: any off-diagonal value of the row (all equal 1/deg(i)); 0 for an isolated vertex
            if (nent <= kLapStage) {
                for (int e = lane; e < nent; e += 64) {
                    w_s[e] = vals[k_lo + e];
                    c_s[e] = colind[k_lo + e];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                lap_row_lds(verts, w_s, c_s, k0 - k_lo, k1 - k_lo, s0, s1, s2);
                if (k1 - k0 >= 2) invdeg = c_s[k0 - k_lo] == (int)i ? w_s[k0 - k_lo + 1] : w_s[k0 - k_lo];
                __builtin_amdgcn_wave_barrier();
            } else if (okr) {
                lap_row(verts, rowptr, colind, vals, i, s0, s1, s2);
                if (k1 - k0 >= 2) invdeg = colind[k0] == (int)i ? vals[k0 + 1] : vals[k0];
            }
            if (okr) {
                const float nrm = sqrtf(((s0 * s0) + (s1 * s1)) + (s2 * s2));
                acc += (double)nrm;
                if (A.u_out) {
                    const bool ok = nrm > 0.0f;
                    A.u_out[i] = float4{ok ? s0 / nrm : 0.0f, ok ? s1 / nrm : 0.0f, ok ? s2 / nrm : 0.0f, invdeg};
                }
            }
        }
    } else {
        const int32_t *__restrict__ e1 = A.e1, *__restrict__ e2 = A.e2;
        for (long long e = xcd_logical_block(blk - gV, gE) * NT + threadIdx.x; e < E; e += (long long)gE * NT) {
            const float *a = verts + 3ll * e1[e], *b = verts + 3ll * e2[e];
            const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
            const float nrm = sqrtf(((d0 * d0) + (d1 * d1)) + (d2 * d2));
            const float t = nrm - A.target;
            acc += (double)(t * t);
        }
    }
    const double tot = block_sum<NT>(acc, L.sm);
    unsigned long long *pp = reinterpret_cast<unsigned long long *>(A.partials);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&pp[blk], __builtin_bit_cast(unsigned long long, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        L.is_last = ticket_arrive_last(A.ticket, (unsigned int)(gV + gE), (unsigned int)blk);
    }
    __syncthreads();
    if (!L.is_last) return;
    double a0 = 0.0, a1 = 0.0;
    for (int i = threadIdx.x; i < gV; i += NT)
        a0 += __builtin_bit_cast(double, __hip_atomic_load(&pp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int i = threadIdx.x; i < gE; i += NT)
        a1 += __builtin_bit_cast(double, __hip_atomic_load(&pp[gV + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const double t0 = block_sum<NT>(a0, L.sm);
    __syncthreads();
    const double t1 = block_sum<NT>(a1, L.sm);
    if (threadIdx.x == 0) {
        const float ll = (float)(t0 / (double)V), le = (float)(t1 / (double)E);
        if (A.loss_lap) *A.loss_lap = ll;
        if (A.loss_edge) *A.loss_edge = le;
        if (A.total) *A.total = ((A.base ? *A.base : 0.0f) + (A.w_lap * ll)) + (A.w_edge * le);  // the tutorial's sum, its order, unfused
        __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- adjoint, gather form: vertex i walks row i of the Laplacian's CSR (columns ascending):
//   Laplacian term  sum_{r in row i}  (c_l L[r,i]) u_r,  L[r,i] = -1 (r == i) or 1/deg(r);
//   edge term       sum_{j ~ i}  -+ g_ij d_ij           (j < i: edge (j,i), i is its second vertex; j > i: edge (i,j)).
// Both sums in the order in which the oracle's row-by-row / edge-by-edge scatter reaches vertex i, with its expressions.
template <bool LAP, bool EDGE>
__device__ __forceinline__ void adjoint_vertex(const float *__restrict__ verts, long long i, const int32_t *__restrict__ rowptr,
                                               const int32_t *__restrict__ colind, const float4 *__restrict__ u, float c_lap, float c_edge,
                                               float target, float *__restrict__ gverts, int accumulate) {
    const float v0 = verts[3 * i], v1 = verts[3 * i + 1], v2 = verts[3 * i + 2];
    float l0 = 0.0f, l1 = 0.0f, l2 = 0.0f, e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
    const int k1 = rowptr[i + 1];
    for (int k0 = rowptr[i]; k0 < k1; k0 += 8) {  // eight entries per sweep: columns, then all gathers, then the sums in order
        int col[8];
        float4 urr[8];
        P3 vrr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) col[e] = colind[k0 + e < k1 ? k0 + e : k1 - 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (LAP) urr[e] = u[col[e]];
            if (EDGE) vrr[e] = *reinterpret_cast<const P3 *>(verts + 3ll * col[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!(k0 + e < k1)) continue;
            const int r = col[e];
            if (LAP) {  // row r of the oracle's scatter: gverts[i] += (c * L[r,i]) * u_r
                const float4 ur = urr[e];
                const float w = c_lap * (r == (int)i ? -1.0f : ur.w);
                l0 = l0 + w * ur.x;
                l1 = l1 + w * ur.y;
                l2 = l2 + w * ur.z;
            }
            if (EDGE && r != (int)i) {
                const float vr[3] = {vrr[e].x, vrr[e].y, vrr[e].z};
                if (r < (int)i) {  // edge (r, i): d = v_r - v_i, vertex i receives -= g d
                    const float d0 = vr[0] - v0, d1 = vr[1] - v1, d2 = vr[2] - v2;
                    const float nrm = sqrtf(((d0 * d0) + (d1 * d1)) + (d2 * d2));
                    if (nrm > 0.0f) {
                        const float g = c_edge * 2.0f * (nrm - target) / nrm;
                        e0 = e0 - g * d0; e1 = e1 - g * d1; e2 = e2 - g * d2;
                    }
                } else {           // edge (i, r): d = v_i - v_r, vertex i receives += g d
                    const float d0 = v0 - vr[0], d1 = v1 - vr[1], d2 = v2 - vr[2];
                    const float nrm = sqrtf(((d0 * d0) + (d1 * d1)) + (d2 * d2));
                    if (nrm > 0.0f) {
                        const float g = c_edge * 2.0f * (nrm - target) / nrm;
                        e0 = e0 + g * d0; e1 = e1 + g * d1; e2 = e2 + g * d2;
                    }
                }
            }
        }
    }
    // (prev + laplacian term) + edge term: the order of the tutorial's chain (g_chamfer + g_lap) + g_edge
    float o0 = accumulate ? gverts[3 * i] : 0.0f, o1 = accumulate ? gverts[3 * i + 1] : 0.0f, o2 = accumulate ? gverts[3 * i + 2] : 0.0f;
    if (LAP) { o0 = accumulate ? o0 + l0 : l0; o1 = accumulate ? o1 + l1 : l1; o2 = accumulate ? o2 + l2 : l2; }
    if (EDGE) { o0 = (LAP || accumulate) ? o0 + e0 : e0; o1 = (LAP || accumulate) ? o1 + e1 : e1; o2 = (LAP || accumulate) ? o2 + e2 : e2; }
    gverts[3 * i] = o0; gverts[3 * i + 1] = o1; gverts[3 * i + 2] = o2;
}

// What rides (device view of include/flux3d_hip.h's fx3d_mesh_reg; filled by mesh.hip's mesh_reg_plan)
struct Ride {
    FwdArgs fwd;          // forward (draw launch); .total unused there
    float c_lap, c_edge;  // adjoint (rows launch): w_lap gout / V, w_edge gout / E
    float *gverts;
    int accumulate;
    int nadj;             // adjoint blocks: kAdjPerBlock vertices each
};
constexpr int kAdjPerBlock = 128;  // (a vertex is a chain of dependent gathers: few per CU, many CUs)
// block j of a Ride's adjoint blocks (any block size >= kAdjPerBlock); thread 0 of block 0 also writes the tutorial's sum
__device__ __forceinline__ void adj_block(const Ride &R, int j) {
    const FwdArgs &A = R.fwd;
    if (j == 0 && threadIdx.x == 0 && A.total)
        *A.total = ((A.base ? *A.base : 0.0f) + (A.w_lap * *A.loss_lap)) + (A.w_edge * *A.loss_edge);
    if ((int)threadIdx.x >= kAdjPerBlock) return;
    const long long i = (long long)j * kAdjPerBlock + threadIdx.x;
    if (i < A.V) adjoint_vertex<true, true>(A.verts, i, A.rowptr, A.colind, A.u_out, R.c_lap, R.c_edge, A.target, R.gverts, R.accumulate);
}

}  // namespace meshreg

// host side (mesh.hip): checks an fx3d_mesh_reg and lays out its launches; fn: the caller's name for messages
fx3d_status mesh_reg_plan(const fx3d_mesh_reg *reg, float gout, float *gverts, int accumulate, hipStream_t st, const char *fn, meshreg::Ride *out);
// the adjoint (+ the sum) of a planned Ride by launches of their own -- for callers whose launch cannot carry it
fx3d_status mesh_reg_adjoint_standalone(const fx3d_mesh_reg *reg, float gout, float *gverts, int accumulate, hipStream_t st);
}  // namespace fx3d
