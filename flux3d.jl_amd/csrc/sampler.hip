// sample_points kernels (gfx950): face-area-weighted triangle sampler.
//
// Replaces sample_points / _sample_points / _rand_barycentric_coords
// (src/transforms/mesh_func.jl:21-82).  The reference loops over meshes on the host, copies each
// probability vector D2H, builds a Distributions.Categorical alias table and draws with the
// host RNG (:43-55).  Here the whole batch is three launches, nothing leaves the device:
//   1. faces_areas_padded (mesh.hip)                         -- gather, HBM/L2 bound
//   2. face_cdf_kernel (one block per mesh up to 32 768 faces; five cdf_mb_* launches beyond): Float64 probabilities
//      (:32-39, incl. the last-padded-column fix-up) and their CDF, summed in the order specified in
//      oracle/flux3d_oracle.c (a radix-32 tree, every node left to right) so oracle and device agree bit-for-bit
//   3. sample_kernel: one thread per sample: Philox4x32-10 draw -> binary search in the CDF ->
//      barycentric point  (w1*v1 + w2*v2) + w3*v3, unfused Float32 (:67-71,:75-82)
#include <cmath>

#include "fx3d_common.h"
#include "sample_gather.h"
#include "mesh_reg.h"

using namespace fx3d;

// defined in mesh.hip
extern "C" fx3d_status fx3d_faces_areas_padded(const float *, int32_t, const int32_t *, int32_t,
                                               const int32_t *, int32_t, float *, fx3d_stream_t);

namespace {

constexpr int kThreads = 256;
constexpr int kCdfThreads = 1024;  // one block per mesh: the area / division passes are the parallel part
constexpr int kChunk = 32;  // FX_SCAN_CHUNK of the oracle
constexpr int kCdfFaces = 6;  // faces per thread and sweep of the area pass

// barycentric point, src/transforms/mesh_func.jl:67-71,:75-82
__device__ __forceinline__ void bary_point(const float *__restrict__ vb, const int32_t *__restrict__ fc,
                                           float r1, float r2, float *__restrict__ o) {
    const float *v1 = vb + 3ll * fc[0], *v2 = vb + 3ll * fc[1], *v3 = vb + 3ll * fc[2];
    const float u = sqrtf(r1);
    const float w1 = 1.0f - u, w2 = u * (1.0f - r2), w3 = u * r2;
#pragma unroll
    for (int d = 0; d < 3; ++d) o[d] = ((w1 * v1[d]) + (w2 * v2[d])) + (w3 * v3[d]);
}

__global__ __launch_bounds__(kThreads) void sample_explicit_kernel(
    const float *__restrict__ verts_padded, int Vmax, const int32_t *__restrict__ faces_padded,
    int Fmax, int B, int n, const int32_t *__restrict__ face_idx, const float *__restrict__ r1,
    const float *__restrict__ r2, float *__restrict__ out) {
    const long long total = (long long)B * n;
    for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < total;
         k += (long long)gridDim.x * kThreads) {
        const int b = (int)(k / n);
        const float *vb = verts_padded + (size_t)b * Vmax * 3;
        const int32_t *fc = faces_padded + ((size_t)b * Fmax + face_idx[k]) * 3;
        bary_point(vb, fc, r1[k], r2[k], out + 3 * k);
    }
}

// ---- the sampling CDF ------------------------------------------------------------------------------------------------
// SUMMATION ORDER = the specification shared with the oracle (oracle/flux3d_oracle.c): a radix-32 tree, every node summed
// left to right -- chunks of 32 faces, groups of 32 chunks, blocks of 32 groups (32 768 faces), groups of 32 blocks, the top.
// No chain is longer than 32 additions, so every level is one parallel pass; zero padding (+0.0 leaves a sum of non-negative
// terms unchanged) lets every chain run its full 32 steps, and a level with a single entry equals the level below it.
// Workspace per mesh (doubles): out[Fp] | E0[nchp] chunk offsets | E1[ngp] group offsets | T2[nbp] | BOFF[nbp] | E2[nbp] |
// misc[64] (den, fix, the fix-up chunk's probabilities), nch = Fp / 32, ng = ceil(nch / 32), nb = ceil(ng / 32), *p = rounded
// up to a multiple of 32.  The face areas are computed in place (compute_faces_areas_padded, src/rep/mesh.jl:799-808: pad -> 0).
constexpr int kCdfBlockFaces = kChunk * kChunk * kChunk;  // faces per level-2 block
struct CdfWs {
    int Fp, nch, nchp, ngp, nbp;
    size_t stride;  // doubles per mesh
    __host__ __device__ static CdfWs make(int Fmax) {
        CdfWs w{};
        w.Fp = (Fmax + kChunk - 1) / kChunk * kChunk;
        w.nch = w.Fp / kChunk;
        w.nchp = (w.nch + kChunk - 1) / kChunk * kChunk;
        const int ng = w.nchp / kChunk, nb = (ng + kChunk - 1) / kChunk;
        w.ngp = (ng + kChunk - 1) / kChunk * kChunk + kChunk;  // (+ the groups of a last, partly filled 8192-face block)
        w.nbp = (nb + kChunk - 1) / kChunk * kChunk;
        w.stride = (size_t)w.Fp + w.nchp + w.ngp + 3 * (size_t)w.nbp + 64;
        return w;
    }
    __host__ __device__ size_t e0() const { return (size_t)Fp; }
    __host__ __device__ size_t e1() const { return e0() + nchp; }
    __host__ __device__ size_t t2() const { return e1() + ngp; }
    __host__ __device__ size_t boff() const { return t2() + nbp; }
    __host__ __device__ size_t e2() const { return boff() + nbp; }
    __host__ __device__ size_t misc() const { return e2() + nbp; }
};

// left-to-right sum of 32 consecutive values (all loads first: the additions are the dependent chain)
__device__ __forceinline__ double chain32(const double *t) {
    double v[kChunk], s = 0.0;
#pragma unroll
    for (int i = 0; i < kChunk; ++i) v[i] = t[i];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) s += v[i];
    return s;
}
// ... and the exclusive prefixes in place; returns the total
__device__ __forceinline__ double chain32_excl(double *t) {
    double v[kChunk], s = 0.0;
#pragma unroll
    for (int i = 0; i < kChunk; ++i) v[i] = t[i];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
        t[i] = s;
        s += v[i];
    }
    return s;
}

#ifdef FX3D_CDF_PROBE  // wall-clock stamps (10 ns) of block 0's phases in the unused tail of the misc slots (tools/face_cdf_ab.py)
#define CDF_MARK(i) if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(out + W.misc() + 40)[i] = (long long)wall_clock64()
#else
#define CDF_MARK(i)
#endif
// One block per mesh of up to 32 768 faces (the fit loop's and C3's meshes), one launch.
// IN_LDS: the working copy (areas -> probabilities -> prefixes) and the chunk totals live in LDS (latency-bound chains).
// VLDS (with IN_LDS): the mesh's vertices are staged in LDS first (coalesced loads, in flight together with the face
// indices: ONE global round trip), the 3 gathers per face read LDS -- the block is alone on its mesh and one CU's
// texture path takes ~1.4 ns per face for scattered 12-byte loads (7.6 us of 16 for the fit loop's 5 k-face sphere).
// A second mesh batch in the same launch (chamfer_distance(m1, m2, n) samples both meshes: src/metrics/mesh.jl:41-42): blocks
// [0, B1) work on the first batch, blocks [B1, gridDim.x) on `o` -- two launches of a handful of blocks each become one.
struct CdfOther {
    const float *verts_padded;
    const int32_t *faces_padded, *faces_len;
    double *ws;
    int Vmax, Fmax;
};
template <bool IN_LDS, bool VLDS>
__global__ __launch_bounds__(kCdfThreads) void face_cdf_kernel(const float *__restrict__ verts_padded, int Vmax,
                                                            const int32_t *__restrict__ faces_padded,
                                                            const int32_t *__restrict__ faces_len, int Fmax, double eps,
                                                            double *__restrict__ ws, int B1, CdfOther o) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];  // IN_LDS: work[Fp + nch] + t0[nchp] (+ float4 verts[Vmax])
    int b = blockIdx.x;
    if (b >= B1) {  // (block-uniform)
        b -= B1;
        verts_padded = o.verts_padded; faces_padded = o.faces_padded; faces_len = o.faces_len; ws = o.ws; Vmax = o.Vmax; Fmax = o.Fmax;
    }
    const CdfWs W = CdfWs::make(Fmax);
    const int Fp = W.Fp, nch = W.nch, nchp = W.nchp, ng = nchp / kChunk;  // ng <= 32 (Fmax <= kCdfBlockFaces)
    const float *vb = verts_padded + (size_t)b * Vmax * 3;
    const int32_t *fb = faces_padded + (size_t)b * Fmax * 3;
    double *out = ws + (size_t)b * W.stride;
    double *cdf = IN_LDS ? dsm : out;
    // LDS copy: one pad double per chunk of 32, so that thread c walking chunk c (stride 33 doubles) and its neighbours
    // hit different banks -- with the plain stride of 256 B all 64 lanes of a wave shared one bank (face_cdf 22.8 -> 18.9 us)
    auto P = [](int k) { return IN_LDS ? k + (k >> 5) : k; };
    double *t0 = IN_LDS ? dsm + Fp + nch : out + W.e0();  // chunk totals, then chunk offsets (zero-padded to nchp)
    float4 *vl = reinterpret_cast<float4 *>(dsm + ((Fp + nch + nchp + 1) & ~1));
    __shared__ double t1[kChunk];  // group totals, then group offsets
    __shared__ double sh[2];
    __shared__ double pF[kChunk];  // probabilities of the chunk that holds the fix-up column, before the fix-up
    CDF_MARK(0);

    // areas: kCdfFaces faces per thread and sweep (one sweep for meshes of up to 6 k faces); the face-index loads do not
    // wait for faces_len (the padded array is addressable up to Fmax; rows beyond the mesh's own faces are replaced by a
    // valid face after the load)
    if (VLDS) {
        for (int v0 = 0; v0 < Vmax; v0 += 4 * kCdfThreads) {  // four loads in flight per thread, then the LDS writes
            P3 q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = v0 + e * kCdfThreads + threadIdx.x;
                q[e] = *reinterpret_cast<const P3 *>(vb + 3ll * (v < Vmax ? v : 0));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = v0 + e * kCdfThreads + threadIdx.x;
                if (v < Vmax) vl[v] = make_float4(q[e].x, q[e].y, q[e].z, 0.0f);
            }
        }
    }
    const int flen = faces_len[b];
    for (int k0 = 0; k0 < Fp; k0 += kCdfFaces * kCdfThreads) {
        struct __attribute__((packed, aligned(4))) I3 { int32_t a, b, c; };
        I3 fi[kCdfFaces];
#pragma unroll
        for (int e = 0; e < kCdfFaces; ++e) {
            const int k = k0 + e * kCdfThreads + threadIdx.x;
            fi[e] = I3{0, 0, 0};
            if (k0 + e * kCdfThreads < Fp) fi[e] = *reinterpret_cast<const I3 *>(fb + 3ll * (k < Fmax ? k : 0));  // (block-uniform)
        }
        if (VLDS && k0 == 0) __syncthreads();  // (the vertex loads above and these index loads were in flight together)
        CDF_MARK(8);
        P3 va[kCdfFaces], vb3[kCdfFaces], vc[kCdfFaces];
#pragma unroll
        for (int e = 0; e < kCdfFaces; ++e) {
            const int k = k0 + e * kCdfThreads + threadIdx.x;
            if (k >= flen) fi[e] = I3{0, 0, 0};  // padding rows: any in-range vertices (their area is set to 0 below)
            va[e] = vb3[e] = vc[e] = P3{0.0f, 0.0f, 0.0f};
            if (k0 + e * kCdfThreads < Fp) {
                if (VLDS) {
                    const float4 x = vl[fi[e].a], y = vl[fi[e].b], z = vl[fi[e].c];
                    va[e] = P3{x.x, x.y, x.z}; vb3[e] = P3{y.x, y.y, y.z}; vc[e] = P3{z.x, z.y, z.z};
                } else {
                    va[e] = *reinterpret_cast<const P3 *>(vb + 3ll * fi[e].a);
                    vb3[e] = *reinterpret_cast<const P3 *>(vb + 3ll * fi[e].b);
                    vc[e] = *reinterpret_cast<const P3 *>(vb + 3ll * fi[e].c);
                }
            }
        }
        CDF_MARK(9);
#pragma unroll
        for (int e = 0; e < kCdfFaces; ++e) {
            const int k = k0 + e * kCdfThreads + threadIdx.x;
            const float v1[3] = {va[e].x, va[e].y, va[e].z}, v2[3] = {vb3[e].x, vb3[e].y, vb3[e].z}, v3[3] = {vc[e].x, vc[e].y, vc[e].z};
            const float a = k < flen ? tri_area(v1, v2, v3) : 0.0f;
            if (k < Fp) cdf[P(k)] = (double)a;
        }
    }
    __syncthreads();
    CDF_MARK(1);
    for (int c = threadIdx.x; c < nchp; c += kCdfThreads) t0[c] = c < nch ? chain32(cdf + P(c * kChunk)) : 0.0;  // chunk totals of the areas
    __syncthreads();
    CDF_MARK(2);
    if (threadIdx.x < kChunk) t1[threadIdx.x] = threadIdx.x < ng ? chain32(t0 + threadIdx.x * kChunk) : 0.0;  // group totals
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = chain32(t1);
        sh[0] = s > eps ? s : eps;  // max(sum, eps), :35
    }
    __syncthreads();
    CDF_MARK(3);
    const double den = sh[0];
    const int cF = (Fmax - 1) / kChunk;  // chunk of the last PADDED column, where the fix-up lands (:36-37): the last chunk
    for (int k = threadIdx.x; k < Fmax; k += kCdfThreads) cdf[P(k)] = cdf[P(k)] / den;  // p, parallel (Float64 divisions)
    __syncthreads();
    CDF_MARK(4);
    for (int c = threadIdx.x; c < nch; c += kCdfThreads) {
        // local inclusive prefixes of p in place; the last prefix of a chunk is its total of p
        double l = 0.0, v[kChunk];
#pragma unroll
        for (int i = 0; i < kChunk; ++i) v[i] = cdf[P(c * kChunk) + i];
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            if (c == cF) pF[i] = v[i];
            l += v[i];
            cdf[P(c * kChunk) + i] = l;
        }
        t0[c] = l;
    }
    __syncthreads();
    CDF_MARK(5);
    // chunk offsets inside their group (exclusive, in place) + the group totals; then the same one level up.  The fix-up column
    // lies in the last chunk (Fp = roundup32(Fmax)): no offset depends on that chunk's own total -- only its local prefixes change.
    if (threadIdx.x < kChunk) t1[threadIdx.x] = threadIdx.x < ng ? chain32_excl(t0 + threadIdx.x * kChunk) : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sp = chain32_excl(t1);
        const double fix = 1.0 - sp;
        if (fix > 0.0) {  // p[Fmax-1] += fix
            double l = 0.0, v[kChunk];
#pragma unroll
            for (int i = 0; i < kChunk; ++i) v[i] = pF[i];
#pragma unroll
            for (int i = 0; i < kChunk; ++i) {
                l += v[i] + (cF * kChunk + i == Fmax - 1 ? fix : 0.0);
                cdf[P(cF * kChunk) + i] = l;
            }
        }
    }
    __syncthreads();
    CDF_MARK(6);
    for (int k = threadIdx.x; k < Fmax; k += kCdfThreads) {
        const int c = k / kChunk;
        out[k] = (t1[c / kChunk] + t0[c]) + cdf[P(k)];  // off0 = off1 + e0, cdf = off0 + l
    }
    CDF_MARK(7);
}

// ---- meshes beyond 32 768 faces: the same tree over many blocks and five launches (no chain longer than 32, every level a
//      parallel pass; the one-block kernel above took 1.9 ms for 500 k faces and 8.5 ms for 2 M).  A block of the three
//      face-parallel kernels takes 8192 faces = 256 chunks = 8 groups; the levels above the groups (<= 1024 level-2 nodes:
//      33 M faces) belong to one block per mesh. -----------------------------------------------------------------------------
constexpr int kCdfSub = 8192;                 // faces per block of K1 / K3 / K5
constexpr int kCdfSubChunks = kCdfSub / kChunk, kCdfSubGroups = kCdfSubChunks / kChunk;
// K1: areas of the block's faces -> out, chunk totals, group totals -> G (the E1 slots)
__global__ __launch_bounds__(kCdfThreads) void cdf_mb_areas_kernel(const float *__restrict__ verts_padded, int Vmax,
                                                                const int32_t *__restrict__ faces_padded,
                                                                const int32_t *__restrict__ faces_len, int Fmax, double *__restrict__ ws) {
    const CdfWs W = CdfWs::make(Fmax);
    const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *vb = verts_padded + (size_t)b * Vmax * 3;
    const int32_t *fb = faces_padded + (size_t)b * Fmax * 3;
    double *out = ws + (size_t)b * W.stride;
    const int flen = faces_len[b];
    __shared__ double t0[kCdfSubChunks];
    const int f0 = blk * kCdfSub;
    for (int k0 = 0; k0 < kCdfSub; k0 += 4 * kCdfThreads) {  // four faces per thread in flight
        struct __attribute__((packed, aligned(4))) I3 { int32_t a, b, c; };
        I3 fi[4];
        P3 va[4], vb3[4], vc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = f0 + k0 + e * kCdfThreads + tid;
            fi[e] = *reinterpret_cast<const I3 *>(fb + 3ll * (k < Fmax ? k : 0));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = f0 + k0 + e * kCdfThreads + tid;
            if (k >= flen) fi[e] = I3{0, 0, 0};
            va[e] = *reinterpret_cast<const P3 *>(vb + 3ll * fi[e].a);
            vb3[e] = *reinterpret_cast<const P3 *>(vb + 3ll * fi[e].b);
            vc[e] = *reinterpret_cast<const P3 *>(vb + 3ll * fi[e].c);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = f0 + k0 + e * kCdfThreads + tid;
            const float v1[3] = {va[e].x, va[e].y, va[e].z}, v2[3] = {vb3[e].x, vb3[e].y, vb3[e].z}, v3[3] = {vc[e].x, vc[e].y, vc[e].z};
            if (k < W.Fp) out[k] = (double)(k < flen ? tri_area(v1, v2, v3) : 0.0f);
        }
    }
    __syncthreads();
    if (tid < kCdfSubChunks) {
        const int c = blk * kCdfSubChunks + tid;  // this thread's chunk
        t0[tid] = c < W.nch ? chain32(out + (size_t)c * kChunk) : 0.0;
    }
    __syncthreads();
    if (tid < kCdfSubGroups) out[W.e1() + blk * kCdfSubGroups + tid] = chain32(t0 + tid * kChunk);
}

// K2 / K4: the levels above the groups, one block per mesh.  G = the group totals (ng of them in the E1 slots).
// SCAN = false: den = max(total, eps) -> misc[0].  SCAN = true: G becomes the group offsets inside their level-2 node (E1),
// BOFF[i] = off3[i / 32] + e2[i] the offsets of the level-2 nodes, fix = 1 - total -> misc[1].
template <bool SCAN>
__global__ __launch_bounds__(kCdfThreads) void cdf_mb_top_kernel(int Fmax, double eps, double *__restrict__ ws) {
    const CdfWs W = CdfWs::make(Fmax);
    const int b = blockIdx.x, tid = threadIdx.x;
    double *out = ws + (size_t)b * W.stride;
    const int ng = W.nchp / kChunk, nb = (ng + kChunk - 1) / kChunk;  // groups, level-2 nodes (<= 1024)
    __shared__ double t2[kChunk * kChunk], t3[kChunk];
    {
        double v[kChunk], sum = 0.0;
        double *G = out + W.e1() + (size_t)tid * kChunk;
#pragma unroll
        for (int i = 0; i < kChunk; ++i) v[i] = tid < nb && tid * kChunk + i < ng ? G[i] : 0.0;
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            if (SCAN && tid < nb && tid * kChunk + i < ng) G[i] = sum;
            sum += v[i];
        }
        t2[tid] = sum;
    }
    __syncthreads();
    if (tid < kChunk) t3[tid] = SCAN ? chain32_excl(t2 + tid * kChunk) : chain32(t2 + tid * kChunk);
    __syncthreads();
    if (tid == 0) {
        if (SCAN) {
            const double sp = chain32_excl(t3);
            out[W.misc() + 1] = 1.0 - sp;
        } else {
            const double sum = chain32(t3);
            out[W.misc()] = sum > eps ? sum : eps;  // max(sum, eps), :35
        }
    }
    __syncthreads();
    if (SCAN && tid < nb) out[W.boff() + tid] = t3[tid / kChunk] + t2[tid];
}

// K3: probabilities (all threads), chunk-local inclusive prefixes in place, chunk offsets inside their group (E0), group totals (G)
__global__ __launch_bounds__(kCdfThreads) void cdf_mb_prob_kernel(int Fmax, double *__restrict__ ws) {
    const CdfWs W = CdfWs::make(Fmax);
    const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    double *out = ws + (size_t)b * W.stride;
    const double den = out[W.misc()];
    const int cF = (Fmax - 1) / kChunk;
    __shared__ double t0[kCdfSubChunks];
    const int f0 = blk * kCdfSub;
#pragma unroll
    for (int e = 0; e < kCdfSub / kCdfThreads; ++e) {
        const int k = f0 + e * kCdfThreads + tid;
        if (k < Fmax) out[k] = out[k] / den;  // (columns past Fmax hold area 0)
    }
    __syncthreads();
    if (tid < kCdfSubChunks) {
        const int c = blk * kCdfSubChunks + tid;
        double l = 0.0;
        if (c < W.nch) {
            double v[kChunk];
            double *pc = out + (size_t)c * kChunk;
#pragma unroll
            for (int i = 0; i < kChunk; ++i) v[i] = pc[i];
            if (c == cF) {
#pragma unroll
                for (int i = 0; i < kChunk; ++i) out[W.misc() + 2 + i] = v[i];
            }
#pragma unroll
            for (int i = 0; i < kChunk; ++i) {
                l += v[i];
                pc[i] = l;
            }
        }
        t0[tid] = l;
    }
    __syncthreads();
    if (tid < kCdfSubGroups) out[W.e1() + blk * kCdfSubGroups + tid] = chain32_excl(t0 + tid * kChunk);
    __syncthreads();
    if (tid < kCdfSubChunks && blk * kCdfSubChunks + tid < W.nchp) out[W.e0() + blk * kCdfSubChunks + tid] = t0[tid];
}

// K5: the fix-up chunk, then cdf = ((BOFF[node] + E1[g]) + E0[c]) + l
__global__ __launch_bounds__(kCdfThreads) void cdf_mb_final_kernel(int Fmax, double *__restrict__ ws) {
    const CdfWs W = CdfWs::make(Fmax);
    const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    double *out = ws + (size_t)b * W.stride;
    const int cF = (Fmax - 1) / kChunk;
    if (cF / kCdfSubChunks == blk) {
        if (tid == 0) {
            const double fix = out[W.misc() + 1];
            if (fix > 0.0) {  // p[Fmax-1] += fix: only the last chunk's local prefixes change
                double l = 0.0, v[kChunk];
#pragma unroll
                for (int i = 0; i < kChunk; ++i) v[i] = out[W.misc() + 2 + i];
#pragma unroll
                for (int i = 0; i < kChunk; ++i) {
                    l += v[i] + (cF * kChunk + i == Fmax - 1 ? fix : 0.0);
                    out[(size_t)cF * kChunk + i] = l;
                }
            }
        }
        __syncthreads();
    }
    const int f0 = blk * kCdfSub;
#pragma unroll
    for (int e = 0; e < kCdfSub / kCdfThreads; ++e) {
        const int k = f0 + e * kCdfThreads + tid;
        if (k < Fmax) {
            const int c = k / kChunk;
            out[k] = ((out[W.boff() + c / (kChunk * kChunk)] + out[W.e1() + c / kChunk]) + out[W.e0() + c]) + out[k];
        }
    }
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

struct DrawSide {
    const float *verts_padded;
    const int32_t *faces_padded, *faces_len;
    const double *ws;
    float *out, *r1_out, *r2_out;
    int32_t *face_out;
    uint64_t seed_host;
    int Vmax, Fmax, B, n;
};
// (two sides: the draws of both meshes of chamfer_distance(m1, m2, n) in one launch; nb0 = blocks of side 0, each side's blocks
//  stride over their own samples)
// (round 6) blocks [nbd, gridDim.x): the forward of the fit iteration's regularisers (mesh_reg.h) -- they read the vertices only
__global__ __launch_bounds__(kThreads) void sample_seeded_kernel(DrawSide s0, DrawSide s1, int nb0, int nbd, const uint64_t *__restrict__ seed_dev,
                                                                 meshreg::FwdArgs reg_fwd) {
    if ((int)blockIdx.x >= nbd) {
        __shared__ meshreg::FwdLds<kThreads> L;
        meshreg::fwd_block<kThreads>(reg_fwd, (int)blockIdx.x - nbd, L);
        return;
    }
    const bool second = (int)blockIdx.x >= nb0;
    const DrawSide &S = second ? s1 : s0;
    const float *__restrict__ verts_padded = S.verts_padded;
    const int32_t *__restrict__ faces_padded = S.faces_padded, *__restrict__ faces_len = S.faces_len;
    const double *__restrict__ ws = S.ws;
    float *__restrict__ out = S.out, *__restrict__ r1_out = S.r1_out, *__restrict__ r2_out = S.r2_out;
    int32_t *__restrict__ face_out = S.face_out;
    const int Vmax = S.Vmax, Fmax = S.Fmax, n = S.n;
    const long long total = (long long)S.B * n;
    const size_t cstride = CdfWs::make(Fmax).stride;
    const uint64_t seed = S.seed_host + (seed_dev ? *seed_dev : 0);  // device part: advanced between replays of a graph
    const long long blk = second ? (long long)blockIdx.x - nb0 : (long long)blockIdx.x, nblk = second ? (long long)nbd - nb0 : (long long)nb0;
    for (long long k = blk * kThreads + threadIdx.x; k < total; k += nblk * kThreads) {
        const int b = (int)(k / n), sidx = (int)(k % n);
        const double *cdf = ws + (size_t)b * cstride;
        const int L = faces_len[b];
        uint32_t c[4] = {(uint32_t)sidx, (uint32_t)b, 0u, 0u};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint64_t bits = (((uint64_t)c[0] << 32) | c[1]) >> 11;
        const double uf = (double)bits * (1.0 / 9007199254740992.0) * cdf[L - 1];
        int lo = 0, hi = L - 1;  // first f with cdf[f] > uf
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] > uf) hi = mid; else lo = mid + 1;
        }
        const float r1 = (float)(c[2] >> 8) * (1.0f / 16777216.0f);
        const float r2 = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
        const float *vb = verts_padded + (size_t)b * Vmax * 3;
        const int32_t *fc = faces_padded + ((size_t)b * Fmax + lo) * 3;
        bary_point(vb, fc, r1, r2, out + 3 * k);
        if (face_out) face_out[k] = lo;
        if (r1_out) r1_out[k] = r1;
        if (r2_out) r2_out[k] = r2;
    }
}

// adjoint of samples = w1*v1 + w2*v2 + w3*v3 w.r.t. the padded verts (gather -> scatter-add)
__global__ __launch_bounds__(kThreads) void sample_bwd_kernel(
    const int32_t *__restrict__ faces_padded, int Vmax, int Fmax, int B, int n,
    const int32_t *__restrict__ face_idx, const float *__restrict__ r1,
    const float *__restrict__ r2, const float *__restrict__ gout, float *gverts) {
    const long long total = (long long)B * n;
    for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < total;
         k += (long long)gridDim.x * kThreads) {
        const int b = (int)(k / n);
        const int32_t *fc = faces_padded + ((size_t)b * Fmax + face_idx[k]) * 3;
        const float u = sqrtf(r1[k]), v = r2[k];
        const float w[3] = {1.0f - u, u * (1.0f - v), u * v};
        float *gb = gverts + (size_t)b * Vmax * 3;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int d = 0; d < 3; ++d) atomicAdd(&gb[3ll * fc[t] + d], w[t] * gout[3 * k + d]);
    }
}

// the ordered form (sample_gather.h): sg_parts(Vmax) blocks per mesh, no float atomics, bit-identical to the oracle's adjoint
__global__ __launch_bounds__(sg::kSgThreads) void sample_bwd_gather_kernel(
    const int32_t *__restrict__ faces_padded, int Vmax, int Fmax, int n, const int32_t *__restrict__ face_idx,
    const float *__restrict__ r1, const float *__restrict__ r2, const float *__restrict__ gout,
    const int32_t *__restrict__ vf_rowptr, const int32_t *__restrict__ vf_ent, float *gverts, int accumulate, int parts,
    sg::SgStep step, const unsigned char *__restrict__ tables) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_lds[];
    const size_t b = blockIdx.x / parts;
    const int j = blockIdx.x % parts;
    const sg::SgMesh m{faces_padded + b * Fmax * 3, face_idx + b * n, r1 + b * n, r2 + b * n, gout + b * n * 3,
                       vf_rowptr + b * (Vmax + 1), vf_ent + b * Fmax * 3, gverts + b * Vmax * 3, Vmax, Fmax, n, accumulate};
    int vb, ve;
    sg::sg_part_range(Vmax, parts, j, vb, ve);
    if (step.vel) {  // the optimiser's arrays are packed (3, sum V) = padded (3, Vmax, B): meshes of equal vertex counts
        const size_t off = b * (size_t)Vmax * 3;
        step.vel += off; step.x += off; step.base += off; step.out += off;
        if (b != 0) step.ctr = nullptr;  // (one block advances the seed counter)
    }
    if (tables) sg::sg_tables_load(sg_lds, Fmax, n, tables + b * sg::sg_blob_bytes(Fmax, n));  // built by an earlier launch (chamfer_bwd.hip)
    else sg::sg_tables(sg_lds, m);
    sg::sg_finish(sg_lds, m, step, vb, ve);
}

int grid_for(long long n) {
    long long g = (n + kThreads - 1) / kThreads;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

inline int roundup32(int v) { return (v + kChunk - 1) / kChunk * kChunk; }

size_t ws_bytes_needed(int Fmax, int B) {
    const size_t cdf = sizeof(double) * (size_t)B * CdfWs::make(Fmax).stride;
    const size_t areas = sizeof(float) * (size_t)B * Fmax;
    return cdf + ((areas + 15) / 16) * 16;
}

}  // namespace

namespace fx3d {
namespace sg {
fx3d_status launch_sample_bwd_gather(const int32_t *faces_padded, int Vmax, int Fmax, int B, int n, const int32_t *face_idx, const float *r1,
                                     const float *r2, const float *gs, const int32_t *vf_rowptr, const int32_t *vf_ent, float *gverts,
                                     int accumulate, const SgStep &step, hipStream_t st, const unsigned char *tables) {
    const size_t lds = sg_layout(Fmax, n).total;
    const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&sample_bwd_gather_kernel), (int)kSgMaxLds, "sample_bwd_gather_kernel");
    if (arc != FX3D_OK) return arc;
    const int parts = sg_parts(Vmax);
    ProfileScope prof("sample_bwd_gather", st);
    hipLaunchKernelGGL(sample_bwd_gather_kernel, dim3((unsigned)B * parts), dim3(kSgThreads), lds, st, faces_padded, Vmax, Fmax, n, face_idx,
                       r1, r2, gs, vf_rowptr, vf_ent, gverts, accumulate, parts, step, tables);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}
}  // namespace sg
}  // namespace fx3d

extern "C" {

fx3d_status fx3d_sample_points_explicit(const float *verts_padded, int32_t Vmax,
                                        const int32_t *faces_padded, int32_t Fmax, int32_t B,
                                        int32_t n, const int32_t *face_idx, const float *r1,
                                        const float *r2, float *out, fx3d_stream_t s) {
    FX3D_REQUIRE(verts_padded && faces_padded && face_idx && r1 && r2 && out,
                 "fx3d_sample_points_explicit: null pointer");
    FX3D_REQUIRE(Vmax > 0 && Fmax > 0 && B > 0 && n > 0, "fx3d_sample_points_explicit: bad sizes");
    hipLaunchKernelGGL(sample_explicit_kernel, dim3(grid_for((long long)B * n)), dim3(kThreads), 0,
                       as_stream(s), verts_padded, Vmax, faces_padded, Fmax, B, n, face_idx, r1, r2, out);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_sample_points_workspace_bytes(int32_t Fmax, int32_t B, size_t *bytes) {
    FX3D_REQUIRE(bytes, "fx3d_sample_points_workspace_bytes: null output");
    FX3D_REQUIRE(Fmax > 0 && B > 0, "fx3d_sample_points_workspace_bytes: bad sizes");
    *bytes = ws_bytes_needed(Fmax, B);
    return FX3D_OK;
}

}  // extern "C"

namespace {
struct CdfArgs {
    const float *verts_padded;
    const int32_t *faces_padded, *faces_len;
    void *ws;
    size_t ws_bytes;
    int Vmax, Fmax, B;
};
// which one-block variant a batch takes (0: multi-block path, 1: LDS copy + staged vertices, 2: LDS copy, 3: global working copy)
int cdf_variant(int Vmax, int Fmax, size_t *lds) {
    const int mb_opt = opt(OPT_CDF_MULTIBLOCK_FROM);  // (fx3d_set_option: the tests lower the switch point)
    const int mb_from = mb_opt > 0 ? mb_opt : kCdfBlockFaces;
    if (Fmax > mb_from || Fmax > kCdfBlockFaces) return 0;
    const CdfWs W = CdfWs::make(Fmax);
    const size_t cdf_lds = sizeof(double) * (size_t)(W.Fp + W.nch + W.nchp + 2);  // + one pad per chunk (bank spread), chunk totals padded to 32
    const size_t v_lds = sizeof(float) * 4 * (size_t)Vmax + 16;
    if (cdf_lds <= 60 * 1024 && cdf_lds + v_lds <= 150 * 1024) { *lds = cdf_lds + v_lds; return 1; }
    if (cdf_lds <= 60 * 1024) { *lds = cdf_lds; return 2; }
    *lds = 0;
    return 3;
}

fx3d_status cdf_check(const CdfArgs &a, const char *fn) {
    FX3D_REQUIRE(a.verts_padded && a.faces_padded && a.faces_len, "%s: null pointer", fn);
    FX3D_REQUIRE(a.Vmax > 0 && a.Fmax > 0 && a.B > 0, "%s: bad sizes", fn);
    if (!a.ws || a.ws_bytes < ws_bytes_needed(a.Fmax, a.B)) {
        set_error("%s: workspace too small (%zu < %zu)", fn, a.ws ? a.ws_bytes : (size_t)0, ws_bytes_needed(a.Fmax, a.B));
        return FX3D_ERR_WORKSPACE;
    }
    return FX3D_OK;
}

// one batch, or two in ONE launch when both take the same one-block variant (`b` may be null)
fx3d_status cdf_launch(const CdfArgs &a, const CdfArgs *b, double eps, hipStream_t st) {
    size_t la = 0, lb = 0;
    const int va = cdf_variant(a.Vmax, a.Fmax, &la), vb = b ? cdf_variant(b->Vmax, b->Fmax, &lb) : va;
    if (b && (va == 0 || va != vb)) {  // different kernels: one after the other
        const fx3d_status rc = cdf_launch(a, nullptr, eps, st);
        return rc ? rc : cdf_launch(*b, nullptr, eps, st);
    }
    double *cdf = reinterpret_cast<double *>(a.ws);
    ProfileScope prof("sample_cdf", st);
    if (va == 0) {
        const CdfWs W = CdfWs::make(a.Fmax);
        const int nb = (W.nchp / kChunk + kChunk - 1) / kChunk, nsub = (W.Fp + kCdfSub - 1) / kCdfSub;
        FX3D_REQUIRE(nb <= kChunk * kChunk, "fx3d_sample_points_cdf: more than 33 554 432 faces per mesh");
        FX3D_REQUIRE(a.B <= 65535, "fx3d_sample_points_cdf: more than 65535 meshes in one batch");
        hipLaunchKernelGGL(cdf_mb_areas_kernel, dim3(nsub, a.B), dim3(kCdfThreads), 0, st, a.verts_padded, a.Vmax, a.faces_padded, a.faces_len, a.Fmax, cdf);
        hipLaunchKernelGGL(cdf_mb_top_kernel<false>, dim3(a.B), dim3(kCdfThreads), 0, st, a.Fmax, eps, cdf);
        hipLaunchKernelGGL(cdf_mb_prob_kernel, dim3(nsub, a.B), dim3(kCdfThreads), 0, st, a.Fmax, cdf);
        hipLaunchKernelGGL(cdf_mb_top_kernel<true>, dim3(a.B), dim3(kCdfThreads), 0, st, a.Fmax, eps, cdf);
        hipLaunchKernelGGL(cdf_mb_final_kernel, dim3(nsub, a.B), dim3(kCdfThreads), 0, st, a.Fmax, cdf);
        FX3D_LAUNCH_CHECK();
        return FX3D_OK;
    }
    const CdfOther o = b ? CdfOther{b->verts_padded, b->faces_padded, b->faces_len, reinterpret_cast<double *>(b->ws), b->Vmax, b->Fmax}
                         : CdfOther{nullptr, nullptr, nullptr, nullptr, 0, 0};
    const int grid = a.B + (b ? b->B : 0);
    const size_t lds = la > lb ? la : lb;
    if (va == 1) {
        const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&face_cdf_kernel<true, true>), 150 * 1024,
                                                   "face_cdf_kernel");
        if (arc != FX3D_OK) return arc;
        hipLaunchKernelGGL((face_cdf_kernel<true, true>), dim3(grid), dim3(kCdfThreads), lds, st, a.verts_padded, a.Vmax,
                           a.faces_padded, a.faces_len, a.Fmax, eps, cdf, a.B, o);
    } else if (va == 2) {
        hipLaunchKernelGGL((face_cdf_kernel<true, false>), dim3(grid), dim3(kCdfThreads), lds, st, a.verts_padded, a.Vmax,
                           a.faces_padded, a.faces_len, a.Fmax, eps, cdf, a.B, o);
    } else {
        hipLaunchKernelGGL((face_cdf_kernel<false, false>), dim3(grid), dim3(kCdfThreads), 0, st, a.verts_padded, a.Vmax,
                           a.faces_padded, a.faces_len, a.Fmax, eps, cdf, a.B, o);
    }
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

struct DrawArgs {
    const float *verts_padded;
    const int32_t *faces_padded, *faces_len;
    const void *cdf_ws;
    size_t ws_bytes;
    float *out, *r1_out, *r2_out;
    int32_t *face_out;
    uint64_t seed;
    int Vmax, Fmax, B, n;
};
fx3d_status draw_check(const DrawArgs &a, const char *fn) {
    FX3D_REQUIRE(a.verts_padded && a.faces_padded && a.faces_len && a.out, "%s: null pointer", fn);
    FX3D_REQUIRE(a.Vmax > 0 && a.Fmax > 0 && a.B > 0 && a.n > 0, "%s: bad sizes", fn);
    if (!a.cdf_ws || a.ws_bytes < ws_bytes_needed(a.Fmax, a.B)) {
        set_error("%s: CDF workspace too small (%zu < %zu)", fn, a.cdf_ws ? a.ws_bytes : (size_t)0, ws_bytes_needed(a.Fmax, a.B));
        return FX3D_ERR_WORKSPACE;
    }
    return FX3D_OK;
}
DrawSide draw_side(const DrawArgs &a) {
    return DrawSide{a.verts_padded, a.faces_padded, a.faces_len, reinterpret_cast<const double *>(a.cdf_ws), a.out, a.r1_out, a.r2_out,
                    a.face_out, a.seed, a.Vmax, a.Fmax, a.B, a.n};
}
fx3d_status draw_launch(const DrawArgs &a, const DrawArgs *b, const uint64_t *seed_dev, hipStream_t st, const meshreg::FwdArgs *reg = nullptr) {
    ProfileScope prof("sample_draw", st);
    const int g0 = grid_for((long long)a.B * a.n), g1 = b ? grid_for((long long)b->B * b->n) : 0;
    const int gr = reg ? reg->gV + reg->gE : 0;
    hipLaunchKernelGGL(sample_seeded_kernel, dim3(g0 + g1 + gr), dim3(kThreads), 0, st, draw_side(a), draw_side(b ? *b : a), g0, g0 + g1, seed_dev,
                       reg ? *reg : meshreg::FwdArgs{});
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}
}  // namespace

extern "C" {

fx3d_status fx3d_sample_points_cdf(const float *verts_padded, int32_t Vmax, const int32_t *faces_padded,
                                   int32_t Fmax, const int32_t *faces_len, int32_t B, double eps, void *ws,
                                   size_t ws_bytes, fx3d_stream_t s) {
    const CdfArgs a{verts_padded, faces_padded, faces_len, ws, ws_bytes, Vmax, Fmax, B};
    const fx3d_status rc = cdf_check(a, "fx3d_sample_points_cdf");
    if (rc) return rc;
    return cdf_launch(a, nullptr, eps, as_stream(s));
}

// Both meshes of chamfer_distance(m1, m2, n) (src/metrics/mesh.jl:41-42) in ONE launch each: the two CDF builds, the two draws.
// Results are bit-identical to two separate fx3d_sample_points_cdf / _draw calls (every block works on its own batch).
fx3d_status fx3d_sample_points_cdf_pair(const float *verts0, int32_t Vmax0, const int32_t *faces0, int32_t Fmax0,
                                        const int32_t *faces_len0, int32_t B0, void *ws0, size_t ws_bytes0,
                                        const float *verts1, int32_t Vmax1, const int32_t *faces1, int32_t Fmax1,
                                        const int32_t *faces_len1, int32_t B1, void *ws1, size_t ws_bytes1, double eps,
                                        fx3d_stream_t s) {
    const CdfArgs a{verts0, faces0, faces_len0, ws0, ws_bytes0, Vmax0, Fmax0, B0}, b{verts1, faces1, faces_len1, ws1, ws_bytes1, Vmax1, Fmax1, B1};
    fx3d_status rc = cdf_check(a, "fx3d_sample_points_cdf_pair");
    if (rc) return rc;
    rc = cdf_check(b, "fx3d_sample_points_cdf_pair");
    if (rc) return rc;
    return cdf_launch(a, &b, eps, as_stream(s));
}

fx3d_status fx3d_sample_points_draw_pair(const float *verts0, int32_t Vmax0, const int32_t *faces0, int32_t Fmax0,
                                         const int32_t *faces_len0, int32_t B0, int32_t n0, uint64_t seed0, const void *cdf_ws0,
                                         size_t ws_bytes0, float *out0, int32_t *face_out0, float *r1_out0, float *r2_out0,
                                         const float *verts1, int32_t Vmax1, const int32_t *faces1, int32_t Fmax1,
                                         const int32_t *faces_len1, int32_t B1, int32_t n1, uint64_t seed1, const void *cdf_ws1,
                                         size_t ws_bytes1, float *out1, int32_t *face_out1, float *r1_out1, float *r2_out1,
                                         const uint64_t *seed_dev, fx3d_stream_t s) {
    const DrawArgs a{verts0, faces0, faces_len0, cdf_ws0, ws_bytes0, out0, r1_out0, r2_out0, face_out0, seed0, Vmax0, Fmax0, B0, n0};
    const DrawArgs b{verts1, faces1, faces_len1, cdf_ws1, ws_bytes1, out1, r1_out1, r2_out1, face_out1, seed1, Vmax1, Fmax1, B1, n1};
    fx3d_status rc = draw_check(a, "fx3d_sample_points_draw_pair");
    if (rc) return rc;
    rc = draw_check(b, "fx3d_sample_points_draw_pair");
    if (rc) return rc;
    return draw_launch(a, &b, seed_dev, as_stream(s));
}

fx3d_status fx3d_sample_points_draw_pair_reg(const float *verts0, int32_t Vmax0, const int32_t *faces0, int32_t Fmax0,
                                             const int32_t *faces_len0, int32_t B0, int32_t n0, uint64_t seed0, const void *cdf_ws0,
                                             size_t ws_bytes0, float *out0, int32_t *face_out0, float *r1_out0, float *r2_out0,
                                             const float *verts1, int32_t Vmax1, const int32_t *faces1, int32_t Fmax1,
                                             const int32_t *faces_len1, int32_t B1, int32_t n1, uint64_t seed1, const void *cdf_ws1,
                                             size_t ws_bytes1, float *out1, int32_t *face_out1, float *r1_out1, float *r2_out1,
                                             const uint64_t *seed_dev, const fx3d_mesh_reg *reg, fx3d_stream_t s) {
    const DrawArgs a{verts0, faces0, faces_len0, cdf_ws0, ws_bytes0, out0, r1_out0, r2_out0, face_out0, seed0, Vmax0, Fmax0, B0, n0};
    const DrawArgs b{verts1, faces1, faces_len1, cdf_ws1, ws_bytes1, out1, r1_out1, r2_out1, face_out1, seed1, Vmax1, Fmax1, B1, n1};
    fx3d_status rc = draw_check(a, "fx3d_sample_points_draw_pair_reg");
    if (rc) return rc;
    rc = draw_check(b, "fx3d_sample_points_draw_pair_reg");
    if (rc) return rc;
    meshreg::Ride R;
    rc = mesh_reg_plan(reg, 1.0f, nullptr, 0, as_stream(s), "fx3d_sample_points_draw_pair_reg", &R);
    if (rc) return rc;
    R.fwd.total = nullptr;  // (the sum needs the chamfer loss: fx3d_chamfer_sampled_bwd_step_reg writes it)
    R.fwd.base = nullptr;
    return draw_launch(a, &b, seed_dev, as_stream(s), &R.fwd);
}

fx3d_status fx3d_sample_points_draw(const float *verts_padded, int32_t Vmax, const int32_t *faces_padded,
                                    int32_t Fmax, const int32_t *faces_len, int32_t B, int32_t n, uint64_t seed,
                                    const uint64_t *seed_dev, const void *cdf_ws, size_t ws_bytes, float *out,
                                    int32_t *face_out, float *r1_out, float *r2_out, fx3d_stream_t s) {
    const DrawArgs a{verts_padded, faces_padded, faces_len, cdf_ws, ws_bytes, out, r1_out, r2_out, face_out, seed, Vmax, Fmax, B, n};
    const fx3d_status rc = draw_check(a, "fx3d_sample_points_draw");
    if (rc) return rc;
    return draw_launch(a, nullptr, seed_dev, as_stream(s));
}

fx3d_status fx3d_sample_points(const float *verts_padded, int32_t Vmax, const int32_t *faces_padded,
                               int32_t Fmax, const int32_t *faces_len, int32_t B, int32_t n,
                               double eps, uint64_t seed, float *out, int32_t *face_out,
                               float *r1_out, float *r2_out, void *ws, size_t ws_bytes,
                               fx3d_stream_t s) {
    FX3D_REQUIRE(out && n > 0, "fx3d_sample_points: bad argument");
    const fx3d_status rc = fx3d_sample_points_cdf(verts_padded, Vmax, faces_padded, Fmax, faces_len, B, eps, ws, ws_bytes, s);
    if (rc != FX3D_OK) return rc;
    return fx3d_sample_points_draw(verts_padded, Vmax, faces_padded, Fmax, faces_len, B, n, seed, nullptr, ws, ws_bytes, out,
                                   face_out, r1_out, r2_out, s);
}

#ifdef FX3D_SG_PROBE
__attribute__((visibility("default"))) int fx3d_debug_sg_probe(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sg::g_sg_probe), sizeof(unsigned long long) * 16);
}
#endif

fx3d_status fx3d_sample_points_bwd_ordered(int32_t Fmax, int32_t n, int32_t *ordered) {
    FX3D_REQUIRE(ordered, "fx3d_sample_points_bwd_ordered: null output");
    FX3D_REQUIRE(Fmax > 0 && n > 0, "fx3d_sample_points_bwd_ordered: bad sizes");
    *ordered = sg::sg_fits(Fmax, n) ? 1 : 0;
    return FX3D_OK;
}

fx3d_status fx3d_sample_points_bwd(const int32_t *faces_padded, int32_t Vmax, int32_t Fmax, int32_t B,
                                   int32_t n, const int32_t *face_idx, const float *r1,
                                   const float *r2, const float *gout, float *gverts, int32_t accumulate,
                                   const int32_t *vf_rowptr, const int32_t *vf_ent, fx3d_stream_t s) {
    FX3D_REQUIRE(faces_padded && face_idx && r1 && r2 && gout && gverts, "fx3d_sample_points_bwd: null pointer");
    FX3D_REQUIRE(Vmax > 0 && Fmax > 0 && B > 0 && n > 0 && (long long)B * 16 < (1ll << 30), "fx3d_sample_points_bwd: bad sizes");
    FX3D_REQUIRE((vf_rowptr == nullptr) == (vf_ent == nullptr), "fx3d_sample_points_bwd: vf_rowptr and vf_ent go together");
    hipStream_t st = as_stream(s);
    if (vf_rowptr && sg::sg_fits(Fmax, n))  // ordered: bit-reproducible, no memset, no float atomics
        return sg::launch_sample_bwd_gather(faces_padded, Vmax, Fmax, B, n, face_idx, r1, r2, gout, vf_rowptr, vf_ent, gverts, accumulate,
                                            sg::SgStep{}, st);
    if (!accumulate) FX3D_HIP(hipMemsetAsync(gverts, 0, sizeof(float) * 3 * (size_t)Vmax * B, st));
    hipLaunchKernelGGL(sample_bwd_kernel, dim3(grid_for((long long)B * n)), dim3(kThreads), 0, st,
                       faces_padded, Vmax, Fmax, B, n, face_idx, r1, r2, gout, gverts);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

}  // extern "C"
