// TriMesh kernels (gfx950): face areas, edge_loss, laplacian_loss and their adjoints.
//
// All four are gather + reduce over packed arrays: HBM/L2-bandwidth bound.  The
// reference does the gathers with `verts[:, faces]` temporaries (src/rep/mesh.jl:772,
// src/metrics/mesh.jl:27-28) and runs laplacian_loss's SpMM on the host even for CuArray meshes
// (`cpu(transpose(verts))`, src/metrics/mesh.jl:12); here each is a single fused pass with a
// deterministic two-stage reduction (per-block double partials -> one fixed-order block).
// Arithmetic order follows the reference expression by expression, unfused (-ffp-contract=off).
#include <cmath>

#include "fx3d_common.h"
#include "mesh_reg.h"

using namespace fx3d;

namespace {

#ifndef FX3D_MESH_THREADS
#define FX3D_MESH_THREADS 256
#endif
constexpr int kThreads = FX3D_MESH_THREADS;
constexpr int kMaxBlocks = 4096;  // partial-sum slots of the scratch; a launch takes at most grid_for()'s cap of them

// (round 4) Both gather kernels below ran one element per thread and iteration: index load -> vertex gathers -> arithmetic, two
// dependent round trips per element and 15-22 elements per thread at a 4 M-face mesh (3.2 / 2.1 TB/s of algorithmic bytes,
// profiles/r04_*_hbm_roofline.txt).  Four elements per iteration now: every index first, then every gather, then the
// arithmetic (the expressions are unchanged: same bits per element).
struct __attribute__((packed, aligned(4))) I3 { int32_t a, b, c; };  // one face of a (3, F) index array: a 12-byte load
__device__ __forceinline__ float tri_area_p(const P3 &v1, const P3 &v2, const P3 &v3) {
    const float t1[3] = {v1.x, v1.y, v1.z}, t2[3] = {v2.x, v2.y, v2.z}, t3[3] = {v3.x, v3.y, v3.z};
    return tri_area(t1, t2, t3);
}
using meshreg::xcd_logical_block;  // (mesh_reg.h)

__global__ __launch_bounds__(kThreads) void faces_areas_packed_kernel(
    const float *__restrict__ verts, const int32_t *__restrict__ faces, long long F,
    float *__restrict__ areas) {
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long f0 = xcd_logical_block(blockIdx.x, gridDim.x) * kThreads + threadIdx.x; f0 < F; f0 += 4 * stride) {
        I3 fc[4];
        P3 v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long f = f0 + u * stride;
            fc[u] = *reinterpret_cast<const I3 *>(faces + 3 * (f < F ? f : f0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u][0] = *reinterpret_cast<const P3 *>(verts + 3ll * fc[u].a);
            v[u][1] = *reinterpret_cast<const P3 *>(verts + 3ll * fc[u].b);
            v[u][2] = *reinterpret_cast<const P3 *>(verts + 3ll * fc[u].c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long f = f0 + u * stride;
            if (f < F) areas[f] = tri_area_p(v[u][0], v[u][1], v[u][2]);
        }
    }
}

__global__ __launch_bounds__(kThreads) void faces_areas_padded_kernel(
    const float *__restrict__ verts_padded, int Vmax, const int32_t *__restrict__ faces_padded,
    int Fmax, const int32_t *__restrict__ faces_len, int B, float *__restrict__ areas) {
    const long long total = (long long)B * Fmax;
    for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < total;
         k += (long long)gridDim.x * kThreads) {
        const int b = (int)(k / Fmax), f = (int)(k % Fmax);
        float a = 0.0f;  // _packed_to_padded pad value 0 (src/rep/mesh.jl:806)
        if (f < faces_len[b]) {
            const int32_t *fc = faces_padded + 3 * k;
            const float *vb = verts_padded + (size_t)b * Vmax * 3;
            a = tri_area(vb + 3ll * fc[0], vb + 3ll * fc[1], vb + 3ll * fc[2]);
        }
        areas[k] = a;
    }
}

// Fused finalisation of a mean over per-block partial sums: every block publishes its sum and takes a ticket; the
// last arriver adds the partials in index order (deterministic) and writes loss = Float32(sum / count), then
// returns the ticket to zero.  Hand-off through 8-byte agent-scope atomics as in nn1_f16_kernel (chamfer.hip).
__device__ __forceinline__ void mean_finalize_last_block(double tot, double *partials, unsigned int *ticket, double count,
                                                         float *loss, double *sm) {
    __shared__ int is_last;
    unsigned long long *pp = reinterpret_cast<unsigned long long *>(partials);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&pp[blockIdx.x], __builtin_bit_cast(unsigned long long, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = ticket_arrive_last(ticket, gridDim.x, blockIdx.x);
    }
    __syncthreads();
    if (!is_last) return;
    double acc = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kThreads)
        acc += __builtin_bit_cast(double, __hip_atomic_load(&pp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const double all = block_sum<kThreads>(acc, sm);
    if (threadIdx.x == 0) {
        *loss = (float)(all / count);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- edge_loss (src/metrics/mesh.jl:24-32) ---------------------------------------------------
__global__ __launch_bounds__(kThreads) void edge_loss_kernel(
    const float *__restrict__ verts, const int32_t *__restrict__ e1,
    const int32_t *__restrict__ e2, long long E, float target, double *__restrict__ partials,
    unsigned int *ticket, float *__restrict__ loss) {
    __shared__ double sm[kThreads / 64];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long e0 = xcd_logical_block(blockIdx.x, gridDim.x) * kThreads + threadIdx.x; e0 < E; e0 += 4 * stride) {  // four edges in flight
        int i1[4], i2[4];
        P3 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long e = e0 + u * stride < E ? e0 + u * stride : e0;
            i1[u] = e1[e];
            i2[u] = e2[e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const P3 *>(verts + 3ll * i1[u]);
            b[u] = *reinterpret_cast<const P3 *>(verts + 3ll * i2[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d0 = a[u].x - b[u].x, d1 = a[u].y - b[u].y, d2 = a[u].z - b[u].z;
            const float nrm = sqrtf(((d0 * d0) + (d1 * d1)) + (d2 * d2));
            const float t = nrm - target;
            if (e0 + u * stride < E) acc += (double)(t * t);
        }
    }
    const double tot = block_sum<kThreads>(acc, sm);
    mean_finalize_last_block(tot, partials, ticket, (double)E, loss, sm);
}

__global__ __launch_bounds__(kThreads) void edge_loss_bwd_kernel(
    const float *__restrict__ verts, const int32_t *__restrict__ e1,
    const int32_t *__restrict__ e2, long long E, float target, float c, float *gverts) {
    for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < E;
         e += (long long)gridDim.x * kThreads) {
        const int i = e1[e], j = e2[e];
        const float d0 = verts[3ll * i] - verts[3ll * j];
        const float d1 = verts[3ll * i + 1] - verts[3ll * j + 1];
        const float d2 = verts[3ll * i + 2] - verts[3ll * j + 2];
        const float nrm = sqrtf(((d0 * d0) + (d1 * d1)) + (d2 * d2));
        if (!(nrm > 0.0f)) continue;
        const float g = c * 2.0f * (nrm - target) / nrm;
        atomicAdd(&gverts[3ll * i], g * d0);
        atomicAdd(&gverts[3ll * i + 1], g * d1);
        atomicAdd(&gverts[3ll * i + 2], g * d2);
        atomicAdd(&gverts[3ll * j], -(g * d0));
        atomicAdd(&gverts[3ll * j + 1], -(g * d1));
        atomicAdd(&gverts[3ll * j + 2], -(g * d2));
    }
}

using meshreg::lap_row;
using meshreg::lap_row_lds;
using meshreg::kLapStage;  // (mesh_reg.h)

__global__ __launch_bounds__(kThreads) void laplacian_loss_kernel(
    const float *__restrict__ verts, long long V, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ colind, const float *__restrict__ vals,
    double *__restrict__ partials, unsigned int *ticket, float *__restrict__ loss) {
    __shared__ double sm[kThreads / 64];
    __shared__ float w_stage[(kThreads / 64) * kLapStage];
    __shared__ int c_stage[(kThreads / 64) * kLapStage];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *w_s = w_stage + wv * kLapStage;
    int *c_s = c_stage + wv * kLapStage;
    double acc = 0.0;
    for (long long ib = xcd_logical_block(blockIdx.x, gridDim.x) * kThreads + wv * 64; ib < V; ib += (long long)gridDim.x * kThreads) {  // (wave-uniform)
        const long long i = ib + lane;
        const bool ok = i < V;
        const int k0 = rowptr[ok ? i : V], k1 = ok ? rowptr[i + 1] : k0;
        const int k_lo = __builtin_amdgcn_readfirstlane(k0), k_hi = __builtin_amdgcn_readlane(k1, 63);
        const int nent = k_hi - k_lo;
        float s0, s1, s2;
        if (nent <= kLapStage) {
            for (int e = lane; e < nent; e += 64) {
                w_s[e] = vals[k_lo + e];
                c_s[e] = colind[k_lo + e];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            lap_row_lds(verts, w_s, c_s, k0 - k_lo, k1 - k_lo, s0, s1, s2);
            __builtin_amdgcn_wave_barrier();
        } else if (ok) {
            lap_row(verts, rowptr, colind, vals, i, s0, s1, s2);
        } else {
            s0 = s1 = s2 = 0.0f;
        }
        if (ok) acc += (double)sqrtf(((s0 * s0) + (s1 * s1)) + (s2 * s2));
    }
    const double tot = block_sum<kThreads>(acc, sm);
    mean_finalize_last_block(tot, partials, ticket, (double)V, loss, sm);
}

__global__ __launch_bounds__(kThreads) void laplacian_loss_bwd_kernel(
    const float *__restrict__ verts, long long V, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ colind, const float *__restrict__ vals, float c, float *gverts) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < V;
         i += (long long)gridDim.x * kThreads) {
        float s0, s1, s2;
        lap_row(verts, rowptr, colind, vals, i, s0, s1, s2);
        const float nrm = sqrtf(((s0 * s0) + (s1 * s1)) + (s2 * s2));
        if (!(nrm > 0.0f)) continue;
        const float u0 = s0 / nrm, u1 = s1 / nrm, u2 = s2 / nrm;
        const int k1 = rowptr[i + 1];
        for (int k0 = rowptr[i]; k0 < k1; k0 += 8) {  // (weights and columns of a sweep first: the atomics do not wait entry by entry)
            float w[8];
            int col[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e < k1 ? k0 + e : k1 - 1;
                w[e] = c * vals[k];
                col[e] = colind[k];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (k0 + e < k1) {
                    float *g = gverts + 3ll * col[e];
                    atomicAdd(&g[0], w[e] * u0);
                    atomicAdd(&g[1], w[e] * u1);
                    atomicAdd(&g[2], w[e] * u2);
                }
        }
    }
}

// The same adjoint WITHOUT float atomics and without scratch (round 3), bit-identical to the oracle's row-by-row scatter.  Vertex i
// receives (c L[r,i]) u_r from every row r that holds column i -- for a structurally symmetric L (the Laplacian of an undirected
// edge list) these are the columns of ITS OWN row, and ascending columns are the order in which the scatter reaches it.  A block takes
// kLapRows consecutive vertices; one THREAD PER ENTRY (i, r) of their rows recomputes u_r (the row sum of r: two dependent round
// trips, as in the forward) and picks L[r,i] up on the way, the contributions wait in LDS, and one thread per vertex adds its
// row's contributions in column order.  (A thread per vertex walking its neighbours' rows one after the other took 38 us on eight
// teapots -- a chain of ~20 round trips --, the scatter with its memset 14.7.)
constexpr int kLapRows = 32;
__global__ __launch_bounds__(kThreads) void laplacian_loss_bwd_gather_kernel(
    const float *__restrict__ verts, long long V, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
    const float *__restrict__ vals, float c, float *gverts, int accumulate, unsigned int *missing) {
    __shared__ float contrib[kThreads][3];
    __shared__ int erow[kThreads];      // row (relative to the block's first) of the entry a thread handles
    __shared__ float acc[kLapRows][3];
    const long long i0 = (long long)blockIdx.x * kLapRows;
    const int nr = V - i0 < kLapRows ? (int)(V - i0) : kLapRows;
    const int tid = threadIdx.x;
    const int e_lo = rowptr[i0], e_hi = rowptr[i0 + nr];
    if (tid < kLapRows) { acc[tid][0] = 0.0f; acc[tid][1] = 0.0f; acc[tid][2] = 0.0f; }
    for (int base = e_lo; base < e_hi; base += kThreads) {  // (one pass unless the rows are unusually long)
        __syncthreads();
        if (tid < nr) {  // the rows mark their entries of this pass
            const int k0 = rowptr[i0 + tid] > base ? rowptr[i0 + tid] : base;
            const int k1 = rowptr[i0 + tid + 1] < base + kThreads ? rowptr[i0 + tid + 1] : base + kThreads;
            for (int k = k0; k < k1; ++k) erow[k - base] = tid;
        }
        __syncthreads();
        const int k = base + tid;
        float x0 = 0.0f, x1 = 0.0f, x2 = 0.0f;
        if (k < e_hi) {
            const long long i = i0 + erow[tid], r = colind[k];
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, wri = 0.0f;
            bool has = false;
            const int q1 = rowptr[r + 1];
            for (int q0 = rowptr[r]; q0 < q1; q0 += 8) {  // row r as in lap_row: weights and columns, then the gathers, then the sums
                float w[8];
                int col[8];
                P3 v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int q = q0 + e < q1 ? q0 + e : q1 - 1;
                    w[e] = vals[q];
                    col[e] = colind[q];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const P3 *>(verts + 3ll * col[e]);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (q0 + e < q1) {
                        s0 = s0 + w[e] * v[e].x;
                        s1 = s1 + w[e] * v[e].y;
                        s2 = s2 + w[e] * v[e].z;
                        if (col[e] == (int)i) { wri = w[e]; has = true; }
                    }
            }
            const float nrm = sqrtf(((s0 * s0) + (s1 * s1)) + (s2 * s2));
            // L[i,r] is stored but L[r,i] is not: the precondition (structural symmetry) does not hold; the contribution this
            // form cannot see is counted (an integer atomic: the count is deterministic) so that the caller can tell
            if (!has && missing) atomicAdd(missing, 1u);
            if (has && nrm > 0.0f) {
                const float cw = c * wri;
                x0 = cw * (s0 / nrm); x1 = cw * (s1 / nrm); x2 = cw * (s2 / nrm);
            } else {
                has = false;
            }
            erow[tid] = has ? erow[tid] : -1 - erow[tid];  // (a row of zero norm / a missing transpose entry contributes nothing: skipped, not added as 0)
        }
        contrib[tid][0] = x0; contrib[tid][1] = x1; contrib[tid][2] = x2;
        __syncthreads();
        if (tid < nr) {  // in column order
            const int k0 = rowptr[i0 + tid] > base ? rowptr[i0 + tid] : base;
            const int k1 = rowptr[i0 + tid + 1] < base + kThreads ? rowptr[i0 + tid + 1] : base + kThreads;
            float a0 = acc[tid][0], a1 = acc[tid][1], a2 = acc[tid][2];
            for (int kk = k0; kk < k1; ++kk)
                if (erow[kk - base] >= 0) { a0 = a0 + contrib[kk - base][0]; a1 = a1 + contrib[kk - base][1]; a2 = a2 + contrib[kk - base][2]; }
            acc[tid][0] = a0; acc[tid][1] = a1; acc[tid][2] = a2;
        }
    }
    __syncthreads();
    if (tid < nr) {
        const long long i = i0 + tid;
        gverts[3 * i] = accumulate ? gverts[3 * i] + acc[tid][0] : acc[tid][0];
        gverts[3 * i + 1] = accumulate ? gverts[3 * i + 1] + acc[tid][1] : acc[tid][1];
        gverts[3 * i + 2] = accumulate ? gverts[3 * i + 2] + acc[tid][2] : acc[tid][2];
    }
}

// ---- both mesh losses in ONE launch, both adjoints in ONE gather launch (the fit_mesh regularisers,
//      examples/fit_mesh.jl:80-83: 0.1 laplacian_loss + edge_loss, launch bound at teapot scale) -----------------------
// Forward: blocks [0, gV) take Laplacian rows, blocks [gV, gV + gE) take edges; every block publishes one Float64
// partial, the last arriver adds each range in index order and writes both means (+ the weighted total).  As a
// by-product the row kernel stores u_r = (L v)_r / ||(L v)_r|| and 1/deg(r) per vertex (float4): all the adjoint needs.
// Adjoint, gather form, one thread per vertex i walking row i of the Laplacian's CSR (columns ascending):
//   Laplacian term  sum_{r in row i}  (c_l L[r,i]) u_r,  L[r,i] = -1 (r == i) or 1/deg(r);
//   edge term       sum_{j ~ i}  -+ g_ij d_ij           (j < i: edge (j,i), i is its second vertex; j > i: edge (i,j)).
// Both sums run in exactly the order in which the oracle's row-by-row / edge-by-edge scatter reaches vertex i (rows and
// sorted edges ascending), with its expressions, so the gradients are BIT-IDENTICAL to the oracle -- and run to run:
// no float atomics (the scatter versions above depend on the atomics' arrival order in the last bit).
// Requires the CSR to be the Laplacian of the SAME edge list (it is: both are cached per mesh, src/rep/mesh.jl:957-1002).
__global__ __launch_bounds__(kThreads) void mesh_losses_kernel(meshreg::FwdArgs A) {
    __shared__ meshreg::FwdLds<kThreads> L;
    meshreg::fwd_block<kThreads>(A, (int)blockIdx.x, L);  // (mesh_reg.h: the same body rides in the sampler's draw launch)
}

// u_r and 1/deg(r) alone (the adjoint without a preceding fused forward)
__global__ __launch_bounds__(kThreads) void lap_unit_rows_kernel(const float *__restrict__ verts, long long V,
                                                                const int32_t *__restrict__ rowptr,
                                                                const int32_t *__restrict__ colind,
                                                                const float *__restrict__ vals, float4 *__restrict__ u_out) {
    for (long long i = xcd_logical_block(blockIdx.x, gridDim.x) * kThreads + threadIdx.x; i < V; i += (long long)gridDim.x * kThreads) {
        float s0, s1, s2;
        lap_row(verts, rowptr, colind, vals, i, s0, s1, s2);
        const float nrm = sqrtf(((s0 * s0) + (s1 * s1)) + (s2 * s2));
        const int k0 = rowptr[i], k1 = rowptr[i + 1];
        float invdeg = 0.0f;
        if (k1 - k0 >= 2) invdeg = colind[k0] == (int)i ? vals[k0 + 1] : vals[k0];
        const bool ok = nrm > 0.0f;
        u_out[i] = float4{ok ? s0 / nrm : 0.0f, ok ? s1 / nrm : 0.0f, ok ? s2 / nrm : 0.0f, invdeg};
    }
}

template <bool LAP, bool EDGE>
__global__ __launch_bounds__(kThreads) void mesh_losses_bwd_gather_kernel(
    const float *__restrict__ verts, long long V, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
    const float4 *__restrict__ u, float c_lap, float c_edge, float target, float *__restrict__ gverts, int accumulate) {
    // (XCD-contiguous order for the Laplacian form: + 7 %; the edge form measured 8 % SLOWER with it and keeps the plain order)
    // (blockDim.x, not kThreads: small meshes are launched with one wave per block -- the fit loop's 9.6 k vertices are 38 blocks of 256
    //  but 151 of 64, and a vertex is a chain of dependent gathers: 10.0 -> ... us at eight teapots, round 6)
    const long long nt = (long long)blockDim.x;
    for (long long i = (LAP ? xcd_logical_block(blockIdx.x, gridDim.x) : (long long)blockIdx.x) * nt + threadIdx.x; i < V; i += (long long)gridDim.x * nt)
        meshreg::adjoint_vertex<LAP, EDGE>(verts, i, rowptr, colind, u, c_lap, c_edge, target, gverts, accumulate);  // (mesh_reg.h)
}

__global__ __launch_bounds__(kThreads) void lincomb_kernel(long long n, float a, const float *__restrict__ x, float b,
                                                          const float *__restrict__ y, float c,
                                                          const float *__restrict__ z, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        float v = (a * x[i]) + (b * y[i]);
        if (z) v = v + (c * z[i]);
        out[i] = v;
    }
}

// Flux.Optimise.Momentum in one pass: v <- rho v - eta g ; x <- x + v  (the arithmetic of two fx3d_lincomb calls)
__global__ __launch_bounds__(kThreads) void momentum_kernel(long long n, float rho, float eta, const float *__restrict__ g,
                                                           float *__restrict__ v, float *__restrict__ x) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const float vn = (rho * v[i]) + (-eta * g[i]);
        v[i] = vn;
        x[i] = (1.0f * x[i]) + (1.0f * vn);
    }
}

// ... and what the next iteration of the fit loop needs first, in the same pass: the offset mesh out = base + x
// (offset(src, x), src/transforms/mesh_func.jl:435-438) and the sampling seed's device counter (two launches less per iteration)
__global__ __launch_bounds__(kThreads) void momentum_offset_kernel(long long n, float rho, float eta, const float *__restrict__ g,
                                                                  float *__restrict__ v, float *__restrict__ x,
                                                                  const float *__restrict__ base, float *__restrict__ out,
                                                                  unsigned long long *ctr, unsigned long long inc) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const float vn = (rho * v[i]) + (-eta * g[i]);
        v[i] = vn;
        const float xn = (1.0f * x[i]) + (1.0f * vn);
        x[i] = xn;
        out[i] = (1.0f * base[i]) + (1.0f * xn);
    }
    if (ctr && blockIdx.x == 0 && threadIdx.x == 0) *ctr += inc;
}

// Grid of a grid-stride kernel over n elements, 256-thread blocks.  Plain gather / streaming kernels take up to 4096 blocks
// (every wave slot of the chip twice over); kernels that END IN A REDUCTION (one partial and one ticket arrival per block, the
// last block adds the partials) stop at 1024: edge_loss 20 / 22 / 26 us at 1024 / 2048 / 4096 blocks
// (profiles/r04_*_hbm_grid_sweep.txt).  Option mesh_max_blocks != 0 forces one cap for both.
// (round 5) `ew_cap`: the Laplacian adjoint's gather form wants NO cap -- one vertex per thread, 7 667 blocks at the 2 M-vertex sheet:
// 27.3 -> 25.2 us = 4.68 TB/s (caps 4 096 / 8 192 / 16 384 / none: 27.3 / 25.7 / 25.2 / 25.4); faces_areas loses half its speed
// beyond 4 096 (21.0 / 33.3 / 43.3 us), the edge forms do not care: same box, profiles/r05_v8_mesh_grid_caps.txt.  Both adjoints in one
// launch (<true, true>): 38.8 -> 36.9 us without the cap.
int grid_for(long long n, bool reduction = false, int ew_cap = kMaxBlocks) {
    long long g = (n + kThreads - 1) / kThreads;
    if (g < 1) g = 1;
    int cap = opt(OPT_MESH_MAX_BLOCKS);
    if (cap < 1 || cap > kMaxBlocks) cap = reduction ? 1024 : ew_cap;
    if (g > cap) g = cap;
    return (int)g;
}

fx3d_status copy_back(float *host, const float *dev, hipStream_t st) {
    if (host) {
        FX3D_HIP(hipMemcpyAsync(host, dev, sizeof(float), hipMemcpyDeviceToHost, st));
        FX3D_HIP(hipStreamSynchronize(st));
    }
    return FX3D_OK;
}

}  // namespace

extern "C" {

fx3d_status fx3d_faces_areas_packed(const float *verts, int64_t V, const int32_t *faces, int64_t F,
                                    float *areas, fx3d_stream_t s) {
    FX3D_REQUIRE(verts && faces && areas, "fx3d_faces_areas_packed: null pointer");
    FX3D_REQUIRE(V > 0 && F > 0 && V < (1ll << 31), "fx3d_faces_areas_packed: bad sizes V=%lld F=%lld",
                 (long long)V, (long long)F);
    ProfileScope prof("faces_areas", as_stream(s));
    hipLaunchKernelGGL(faces_areas_packed_kernel, dim3(grid_for(F)), dim3(kThreads), 0, as_stream(s),
                       verts, faces, (long long)F, areas);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_faces_areas_padded(const float *verts_padded, int32_t Vmax,
                                    const int32_t *faces_padded, int32_t Fmax,
                                    const int32_t *faces_len, int32_t B, float *areas,
                                    fx3d_stream_t s) {
    FX3D_REQUIRE(verts_padded && faces_padded && faces_len && areas, "fx3d_faces_areas_padded: null pointer");
    FX3D_REQUIRE(Vmax > 0 && Fmax > 0 && B > 0, "fx3d_faces_areas_padded: bad sizes");
    hipLaunchKernelGGL(faces_areas_padded_kernel, dim3(grid_for((long long)B * Fmax)), dim3(kThreads),
                       0, as_stream(s), verts_padded, Vmax, faces_padded, Fmax, faces_len, B, areas);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_lincomb(int64_t n, float a, const float *x, float b, const float *y, float c,
                         const float *z, float *out, fx3d_stream_t s) {
    FX3D_REQUIRE(x && y && out && n > 0, "fx3d_lincomb: bad argument");
    hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(s), (long long)n, a, x, b, y, c, z, out);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_momentum_step(int64_t n, float rho, float eta, const float *g, float *v, float *x, fx3d_stream_t s) {
    FX3D_REQUIRE(g && v && x && n > 0, "fx3d_momentum_step: bad argument");
    hipLaunchKernelGGL(momentum_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(s), (long long)n, rho, eta, g, v, x);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_momentum_step_offset(int64_t n, float rho, float eta, const float *g, float *v, float *x, const float *base,
                                      float *out, uint64_t *ctr, uint64_t inc, fx3d_stream_t s) {
    FX3D_REQUIRE(g && v && x && base && out && n > 0, "fx3d_momentum_step_offset: bad argument");
    hipLaunchKernelGGL(momentum_offset_kernel, dim3(grid_for(n)), dim3(kThreads), 0, as_stream(s), (long long)n, rho, eta, g, v, x,
                       base, out, reinterpret_cast<unsigned long long *>(ctr), (unsigned long long)inc);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_packed_to_padded(const float *packed, const int64_t *verts_len_host, int32_t B, int32_t Vmax,
                                  float *padded, fx3d_stream_t s) {
    FX3D_REQUIRE(packed && verts_len_host && padded && B > 0 && Vmax > 0, "fx3d_packed_to_padded: bad argument");
    hipStream_t st = as_stream(s);
    FX3D_HIP(hipMemsetAsync(padded, 0, sizeof(float) * 3 * (size_t)Vmax * B, st));  // pad value 0
    size_t cur = 0;
    for (int b = 0; b < B; ++b) {
        const size_t n = (size_t)verts_len_host[b];
        FX3D_REQUIRE(n <= (size_t)Vmax, "fx3d_packed_to_padded: mesh %d longer than Vmax", b);
        if (n) FX3D_HIP(hipMemcpyAsync(padded + (size_t)b * Vmax * 3, packed + cur * 3, n * 12, hipMemcpyDeviceToDevice, st));
        cur += n;
    }
    return FX3D_OK;
}

fx3d_status fx3d_padded_to_packed(const float *padded, const int64_t *verts_len_host, int32_t B, int32_t Vmax,
                                  float *packed, fx3d_stream_t s) {
    FX3D_REQUIRE(packed && verts_len_host && padded && B > 0 && Vmax > 0, "fx3d_padded_to_packed: bad argument");
    hipStream_t st = as_stream(s);
    size_t cur = 0;
    for (int b = 0; b < B; ++b) {
        const size_t n = (size_t)verts_len_host[b];
        FX3D_REQUIRE(n <= (size_t)Vmax, "fx3d_padded_to_packed: mesh %d longer than Vmax", b);
        if (n) FX3D_HIP(hipMemcpyAsync(packed + cur * 3, padded + (size_t)b * Vmax * 3, n * 12, hipMemcpyDeviceToDevice, st));
        cur += n;
    }
    return FX3D_OK;
}

fx3d_status fx3d_mesh_loss_workspace_bytes(int64_t count, size_t *bytes) {
    FX3D_REQUIRE(bytes, "fx3d_mesh_loss_workspace_bytes: null output");
    (void)count;
    *bytes = sizeof(double) * kMaxBlocks;
    return FX3D_OK;
}

fx3d_status fx3d_edge_loss(const float *verts, int64_t V, const int32_t *edges, int64_t E,
                           float target, float *loss_dev, float *loss_host, void *ws,
                           size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(verts && edges && loss_dev, "fx3d_edge_loss: null pointer");
    FX3D_REQUIRE(V > 0 && E > 0 && V < (1ll << 31), "fx3d_edge_loss: bad sizes V=%lld E=%lld",
                 (long long)V, (long long)E);
    if (!ws || ws_bytes < sizeof(double) * kMaxBlocks) {
        set_error("fx3d_edge_loss: workspace too small");
        return FX3D_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(s);
    double *partials = reinterpret_cast<double *>(ws);
    fx3d_status trc = FX3D_OK;
    unsigned int *ticket = ticket_slot(&trc, st);
    if (!ticket) return trc;
    const int g = grid_for(E, true);
    {
        ProfileScope prof("edge_loss", st);
        hipLaunchKernelGGL(edge_loss_kernel, dim3(g), dim3(kThreads), 0, st, verts, edges, edges + E,
                           (long long)E, target, partials, ticket, loss_dev);  // the last block writes the mean
    }
    FX3D_LAUNCH_CHECK();
    return copy_back(loss_host, loss_dev, st);
}

fx3d_status fx3d_edge_loss_bwd(const float *verts, int64_t V, const int32_t *edges, int64_t E,
                               float target, float gout, float *gverts, int32_t accumulate, fx3d_stream_t s) {
    // scatter form for a caller that holds nothing but an edge list (any list: duplicates, any order): float atomics, the last
    // bit depends on their arrival order.  The wrappers use fx3d_edge_loss_bwd_adj.
    FX3D_REQUIRE(verts && edges && gverts, "fx3d_edge_loss_bwd: null pointer");
    FX3D_REQUIRE(V > 0 && E > 0, "fx3d_edge_loss_bwd: bad sizes");
    hipStream_t st = as_stream(s);
    if (!accumulate) FX3D_HIP(hipMemsetAsync(gverts, 0, sizeof(float) * 3 * (size_t)V, st));
    hipLaunchKernelGGL(edge_loss_bwd_kernel, dim3(grid_for(E)), dim3(kThreads), 0, st, verts, edges,
                       edges + E, (long long)E, target, gout / (float)E, gverts);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_edge_loss_bwd_adj(const float *verts, int64_t V, const int32_t *rowptr, const int32_t *colind, int64_t E,
                                   float target, float gout, float *gverts, int32_t accumulate, fx3d_stream_t s) {
    // gather form: vertex i walks its neighbours (the Laplacian's CSR row, ascending; the diagonal is skipped) -- the order in
    // which the oracle's edge-by-edge scatter over the sorted edge list reaches it.  One launch, no atomics, no memset node.
    FX3D_REQUIRE(verts && rowptr && colind && gverts, "fx3d_edge_loss_bwd_adj: null pointer");
    FX3D_REQUIRE(V > 0 && E > 0 && V < (1ll << 31), "fx3d_edge_loss_bwd_adj: bad sizes");
    hipStream_t st = as_stream(s);
    ProfileScope prof("edge_loss_bwd", st);
    hipLaunchKernelGGL((mesh_losses_bwd_gather_kernel<false, true>), dim3(grid_for(V)), dim3(kThreads), 0, st, verts, (long long)V,
                       rowptr, colind, static_cast<const float4 *>(nullptr), 0.0f, gout / (float)E, target, gverts, accumulate);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_laplacian_loss(const float *verts, int64_t V, const int32_t *rowptr,
                                const int32_t *colind, const float *vals, float *loss_dev,
                                float *loss_host, void *ws, size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(verts && rowptr && colind && vals && loss_dev, "fx3d_laplacian_loss: null pointer");
    FX3D_REQUIRE(V > 0 && V < (1ll << 31), "fx3d_laplacian_loss: bad V=%lld", (long long)V);
    if (!ws || ws_bytes < sizeof(double) * kMaxBlocks) {
        set_error("fx3d_laplacian_loss: workspace too small");
        return FX3D_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(s);
    double *partials = reinterpret_cast<double *>(ws);
    fx3d_status trc = FX3D_OK;
    unsigned int *ticket = ticket_slot(&trc, st);
    if (!ticket) return trc;
    const int g = grid_for(V, true);
    {
        ProfileScope prof("laplacian_loss", st);
        hipLaunchKernelGGL(laplacian_loss_kernel, dim3(g), dim3(kThreads), 0, st, verts, (long long)V,
                           rowptr, colind, vals, partials, ticket, loss_dev);  // the last block writes the mean
    }
    FX3D_LAUNCH_CHECK();
    return copy_back(loss_host, loss_dev, st);
}

fx3d_status fx3d_mesh_losses_workspace_bytes(int64_t V, int64_t E, size_t *bytes) {
    FX3D_REQUIRE(bytes && V > 0 && E >= 0, "fx3d_mesh_losses_workspace_bytes: bad argument");
    // per-block partials of both ranges + the (4,V) unit rows of the Laplacian (16-byte aligned behind them)
    *bytes = sizeof(double) * 2 * kMaxBlocks + sizeof(float) * 4 * (size_t)V;
    return FX3D_OK;
}

fx3d_status fx3d_mesh_losses(const float *verts, int64_t V, const int32_t *rowptr, const int32_t *colind,
                             const float *vals, const int32_t *edges, int64_t E, float target, float w_lap, float w_edge,
                             const float *base_dev, float *loss_lap_dev, float *loss_edge_dev, float *total_dev,
                             void *ws, size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(verts && rowptr && colind && vals && edges, "fx3d_mesh_losses: null pointer");
    FX3D_REQUIRE(loss_lap_dev || loss_edge_dev || total_dev, "fx3d_mesh_losses: no output requested");
    FX3D_REQUIRE(V > 0 && E > 0 && V < (1ll << 31), "fx3d_mesh_losses: bad sizes V=%lld E=%lld", (long long)V, (long long)E);
    size_t need = 0;
    fx3d_mesh_losses_workspace_bytes(V, E, &need);
    if (!ws || ws_bytes < need) { set_error("fx3d_mesh_losses: workspace too small (%zu < %zu bytes)", ws ? ws_bytes : (size_t)0, need); return FX3D_ERR_WORKSPACE; }
    hipStream_t st = as_stream(s);
    double *partials = reinterpret_cast<double *>(ws);
    float4 *u = reinterpret_cast<float4 *>(partials + 2 * kMaxBlocks);
    fx3d_status trc = FX3D_OK;
    unsigned int *ticket = ticket_slot(&trc, st);
    if (!ticket) return trc;
    const int gV = grid_for(V, true), gE = grid_for(E, true);
    {
        ProfileScope prof("mesh_losses", st);
        const meshreg::FwdArgs A{verts, (long long)V, rowptr, colind, vals, edges, edges + E, (long long)E, target, gV, gE, partials, ticket,
                                 w_lap, w_edge, base_dev, loss_lap_dev, loss_edge_dev, total_dev, u};
        hipLaunchKernelGGL(mesh_losses_kernel, dim3(gV + gE), dim3(kThreads), 0, st, A);
    }
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_mesh_losses_bwd(const float *verts, int64_t V, const int32_t *rowptr, const int32_t *colind,
                                 const float *vals, int64_t E, float target, float g_lap, float g_edge, int32_t reuse_forward,
                                 float *gverts, int32_t accumulate, void *ws, size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(verts && rowptr && colind && vals && gverts, "fx3d_mesh_losses_bwd: null pointer");
    FX3D_REQUIRE(V > 0 && E > 0 && V < (1ll << 31), "fx3d_mesh_losses_bwd: bad sizes");
    size_t need = 0;
    fx3d_mesh_losses_workspace_bytes(V, E, &need);
    if (!ws || ws_bytes < need) { set_error("fx3d_mesh_losses_bwd: workspace too small (%zu < %zu bytes)", ws ? ws_bytes : (size_t)0, need); return FX3D_ERR_WORKSPACE; }
    hipStream_t st = as_stream(s);
    float4 *u = reinterpret_cast<float4 *>(reinterpret_cast<double *>(ws) + 2 * kMaxBlocks);
    if (g_lap == 0.0f && g_edge == 0.0f) {  // both terms dropped: the gradient is exactly zero; never read the (possibly unbuilt) unit rows
        if (!accumulate) FX3D_HIP(hipMemsetAsync(gverts, 0, sizeof(float) * 3 * (size_t)V, st));
        return FX3D_OK;
    }
    if (!reuse_forward && g_lap != 0.0f) {  // no fx3d_mesh_losses on the same vertices and workspace before this call: build the unit rows
        hipLaunchKernelGGL(lap_unit_rows_kernel, dim3(grid_for(V)), dim3(kThreads), 0, st, verts, (long long)V, rowptr, colind, vals, u);
        FX3D_LAUNCH_CHECK();
    }
    ProfileScope prof("mesh_losses_bwd", st);
    // (a weight of exactly zero drops its term: the Laplacian adjoint alone is how the wrappers differentiate laplacian_loss on
    // LARGE meshes -- the scratch-free fx3d_laplacian_loss_bwd_sym recomputes every neighbour's row, 7 x the traffic).
    // g_lap == 0 is tested FIRST: that instantiation never touches u, which is unbuilt in exactly that case.
    if (g_lap == 0.0f)
        hipLaunchKernelGGL((mesh_losses_bwd_gather_kernel<false, true>), dim3(grid_for(V)), dim3(kThreads), 0, st, verts, (long long)V,
                           rowptr, colind, u, 0.0f, g_edge / (float)E, target, gverts, accumulate);
    else if (g_edge == 0.0f)
        hipLaunchKernelGGL((mesh_losses_bwd_gather_kernel<true, false>), dim3(grid_for(V, false, 1 << 22)), dim3(kThreads), 0, st, verts, (long long)V,
                           rowptr, colind, u, g_lap / (float)V, 0.0f, target, gverts, accumulate);
    else if (V <= 65536)  // a small mesh: one wave per block, over four times as many CUs
        hipLaunchKernelGGL((mesh_losses_bwd_gather_kernel<true, true>), dim3((unsigned)((V + 63) / 64)), dim3(64), 0, st, verts, (long long)V,
                           rowptr, colind, u, g_lap / (float)V, g_edge / (float)E, target, gverts, accumulate);
    else
        hipLaunchKernelGGL((mesh_losses_bwd_gather_kernel<true, true>), dim3(grid_for(V, false, 1 << 22)), dim3(kThreads), 0, st, verts, (long long)V,
                           rowptr, colind, u, g_lap / (float)V, g_edge / (float)E, target, gverts, accumulate);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

}  // extern "C"

namespace {
__global__ void mesh_reg_total_kernel(const float *base, float w_lap, const float *ll, float w_edge, const float *le, float *total) {
    *total = ((base ? *base : 0.0f) + (w_lap * *ll)) + (w_edge * *le);
}
}  // namespace

namespace fx3d {
fx3d_status mesh_reg_plan(const fx3d_mesh_reg *reg, float gout, float *gverts, int accumulate, hipStream_t st, const char *fn, meshreg::Ride *out) {
    FX3D_REQUIRE(reg && reg->verts && reg->rowptr && reg->colind && reg->vals && reg->edges, "%s: fx3d_mesh_reg with a null pointer", fn);
    FX3D_REQUIRE(reg->loss_lap_dev && reg->loss_edge_dev, "%s: fx3d_mesh_reg needs loss_lap_dev and loss_edge_dev", fn);
    FX3D_REQUIRE(reg->V > 0 && reg->E > 0 && reg->V < (1ll << 31), "%s: fx3d_mesh_reg: bad sizes V=%lld E=%lld", fn, (long long)reg->V, (long long)reg->E);
    size_t need = 0;
    fx3d_mesh_losses_workspace_bytes(reg->V, reg->E, &need);
    if (!reg->ws || reg->ws_bytes < need) { set_error("%s: fx3d_mesh_reg workspace too small (%zu < %zu bytes)", fn, reg->ws ? reg->ws_bytes : (size_t)0, need); return FX3D_ERR_WORKSPACE; }
    fx3d_status trc = FX3D_OK;
    unsigned int *ticket = ticket_slot(&trc, st);
    if (!ticket) return trc;
    double *partials = reinterpret_cast<double *>(reg->ws);
    float4 *u = reinterpret_cast<float4 *>(partials + 2 * kMaxBlocks);
    meshreg::Ride R{};
    R.fwd = meshreg::FwdArgs{reg->verts, (long long)reg->V, reg->rowptr, reg->colind, reg->vals, reg->edges, reg->edges + reg->E, (long long)reg->E,
                             reg->target, grid_for(reg->V, true), grid_for(reg->E, true), partials, ticket, reg->w_lap, reg->w_edge, reg->base_dev,
                             reg->loss_lap_dev, reg->loss_edge_dev, reg->total_dev, u};
    R.c_lap = (reg->w_lap * gout) / (float)reg->V;
    R.c_edge = (reg->w_edge * gout) / (float)reg->E;
    R.gverts = gverts;
    R.accumulate = accumulate;
    R.nadj = (int)((reg->V + meshreg::kAdjPerBlock - 1) / meshreg::kAdjPerBlock);
    *out = R;
    return FX3D_OK;
}
fx3d_status mesh_reg_adjoint_standalone(const fx3d_mesh_reg *reg, float gout, float *gverts, int accumulate, hipStream_t st) {
    const fx3d_status rc = fx3d_mesh_losses_bwd(reg->verts, reg->V, reg->rowptr, reg->colind, reg->vals, reg->E, reg->target, reg->w_lap * gout,
                                                reg->w_edge * gout, 1, gverts, accumulate, reg->ws, reg->ws_bytes, reinterpret_cast<fx3d_stream_t>(st));
    if (rc) return rc;
    if (reg->total_dev) {
        hipLaunchKernelGGL(mesh_reg_total_kernel, dim3(1), dim3(1), 0, st, reg->base_dev, reg->w_lap, reg->loss_lap_dev, reg->w_edge, reg->loss_edge_dev,
                           reg->total_dev);
        FX3D_LAUNCH_CHECK();
    }
    return FX3D_OK;
}
}  // namespace fx3d

extern "C" {

fx3d_status fx3d_laplacian_loss_bwd(const float *verts, int64_t V, const int32_t *rowptr,
                                    const int32_t *colind, const float *vals, float gout,
                                    float *gverts, int32_t accumulate, fx3d_stream_t s) {
    // ANY CSR (asymmetric, pruned, directed, duplicate columns): the row-by-row scatter with float atomics -- always the
    // adjoint of the forward, the last bit depends on the atomics' arrival order (the raw entry point must not
    // assume a structure it cannot check).  Callers whose L is the Laplacian of an undirected edge list -- the Python and
    // Julia wrappers -- use fx3d_laplacian_loss_bwd_sym.
    FX3D_REQUIRE(verts && rowptr && colind && vals && gverts, "fx3d_laplacian_loss_bwd: null pointer");
    FX3D_REQUIRE(V > 0, "fx3d_laplacian_loss_bwd: bad V");
    hipStream_t st = as_stream(s);
    if (!accumulate) FX3D_HIP(hipMemsetAsync(gverts, 0, sizeof(float) * 3 * (size_t)V, st));
    hipLaunchKernelGGL(laplacian_loss_bwd_kernel, dim3(grid_for(V)), dim3(kThreads), 0, st, verts,
                       (long long)V, rowptr, colind, vals, gout / (float)V, gverts);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_laplacian_loss_bwd_sym(const float *verts, int64_t V, const int32_t *rowptr, const int32_t *colind,
                                        const float *vals, float gout, float *gverts, int32_t accumulate,
                                        uint32_t *missing_dev, fx3d_stream_t s) {
    // gather form: one launch, no atomics, bit-identical to the oracle.  PRECONDITION: L structurally symmetric (column i
    // in row r iff column r in row i), no duplicate columns in a row.  missing_dev (optional, caller-zeroed device counter)
    // receives the number of stored entries L[i,r] whose transpose L[r,i] is absent: non-zero means the precondition did
    // not hold and the gradient lacks those contributions -- use fx3d_laplacian_loss_bwd.
    FX3D_REQUIRE(verts && rowptr && colind && vals && gverts, "fx3d_laplacian_loss_bwd_sym: null pointer");
    FX3D_REQUIRE(V > 0, "fx3d_laplacian_loss_bwd_sym: bad V");
    if (opt(OPT_LAP_BWD_SCATTER)) return fx3d_laplacian_loss_bwd(verts, V, rowptr, colind, vals, gout, gverts, accumulate, s);
    hipStream_t st = as_stream(s);
    ProfileScope prof("laplacian_loss_bwd", st);
    hipLaunchKernelGGL(laplacian_loss_bwd_gather_kernel, dim3((unsigned int)((V + kLapRows - 1) / kLapRows)), dim3(kThreads), 0, st, verts,
                       (long long)V, rowptr, colind, vals, gout / (float)V, gverts, accumulate, missing_dev);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

}  // extern "C"

// ---- the reference's index arrays -> int32 0-based on the device (include/flux3d_hip.h: fx3d_index_convert) -----------------
namespace {
template <typename IT>
__global__ __launch_bounds__(kThreads) void index_convert_kernel(const IT *__restrict__ src, long long n, long long base, int clamp_pad,
                                                                 long long limit, int32_t *__restrict__ dst, unsigned int *bad) {
    unsigned int nbad = 0;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        long long v = (long long)src[i] - base;
        if (v < 0 && clamp_pad) v = 0;  // padding
        else if (limit > 0 ? (v < 0 || v >= limit) : (v < -2147483648ll || v > 2147483647ll)) { v = 0; ++nbad; }  // (no limit: still an int32)
        dst[i] = (int32_t)v;
    }
    if (bad && nbad) atomicAdd(bad, nbad);
}
size_t index_elem_bytes(int t) { return t == FX3D_IDX_I64 ? 8 : 4; }
}  // namespace

extern "C" {

fx3d_status fx3d_index_convert(const void *src_dev, int32_t index_type, int32_t index_base, int64_t count, int32_t clamp_pad,
                               int64_t limit, int32_t *dst_dev, uint32_t *bad_dev, fx3d_stream_t s) {
    FX3D_REQUIRE(index_type == FX3D_IDX_I32 || index_type == FX3D_IDX_U32 || index_type == FX3D_IDX_I64,
                 "fx3d_index_convert: index_type %d (FX3D_IDX_I32 / _U32 / _I64)", index_type);
    FX3D_REQUIRE(count >= 0 && limit >= 0 && limit <= (1ll << 31), "fx3d_index_convert: bad count / limit");
    if (count == 0) return FX3D_OK;
    FX3D_REQUIRE(src_dev && dst_dev, "fx3d_index_convert: null pointer");
    // a range check whose violations nobody can see would rewrite corrupt indices to vertex 0 in silence
    FX3D_REQUIRE(limit == 0 || bad_dev, "fx3d_index_convert: limit > 0 needs bad_dev (a zeroed device counter of the violations)");
    hipStream_t st = as_stream(s);
    const dim3 g(grid_for(count)), b(kThreads);
    if (index_type == FX3D_IDX_I64)
        hipLaunchKernelGGL(index_convert_kernel<long long>, g, b, 0, st, static_cast<const long long *>(src_dev), (long long)count,
                           (long long)index_base, clamp_pad, (long long)limit, dst_dev, bad_dev);
    else if (index_type == FX3D_IDX_U32)
        hipLaunchKernelGGL(index_convert_kernel<unsigned int>, g, b, 0, st, static_cast<const unsigned int *>(src_dev), (long long)count,
                           (long long)index_base, clamp_pad, (long long)limit, dst_dev, bad_dev);
    else
        hipLaunchKernelGGL(index_convert_kernel<int>, g, b, 0, st, static_cast<const int *>(src_dev), (long long)count,
                           (long long)index_base, clamp_pad, (long long)limit, dst_dev, bad_dev);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_index_upload_workspace_bytes(int32_t index_type, int64_t count, size_t *bytes) {
    FX3D_REQUIRE(bytes && count >= 0, "fx3d_index_upload_workspace_bytes: bad argument");
    FX3D_REQUIRE(index_type == FX3D_IDX_I32 || index_type == FX3D_IDX_U32 || index_type == FX3D_IDX_I64,
                 "fx3d_index_upload_workspace_bytes: index_type %d", index_type);
    *bytes = index_elem_bytes(index_type) * (size_t)count;
    return FX3D_OK;
}

fx3d_status fx3d_index_upload(const void *src_host, int32_t index_type, int32_t index_base, int64_t count, int32_t clamp_pad,
                              int64_t limit, int32_t *dst_dev, uint32_t *bad_dev, void *ws, size_t ws_bytes, fx3d_stream_t s) {
    size_t need = 0;
    const fx3d_status rc = fx3d_index_upload_workspace_bytes(index_type, count, &need);
    if (rc) return rc;
    if (count == 0) return FX3D_OK;
    FX3D_REQUIRE(src_host && dst_dev, "fx3d_index_upload: null pointer");
    if (!ws || ws_bytes < need) { set_error("fx3d_index_upload: workspace too small (%zu < %zu bytes)", ws ? ws_bytes : (size_t)0, need); return FX3D_ERR_WORKSPACE; }
    hipStream_t st = as_stream(s);
    FX3D_HIP(hipMemcpyAsync(ws, src_host, need, hipMemcpyHostToDevice, st));
    const fx3d_status crc = fx3d_index_convert(ws, index_type, index_base, count, clamp_pad, limit, dst_dev, bad_dev, s);
    if (crc) return crc;
    FX3D_HIP(hipStreamSynchronize(st));  // pageable host memory: blocking, like fx3d_memcpy_h2d
    return FX3D_OK;
}

}  // extern "C"
