// sample_points kernels (gfx950): face-area-weighted triangle sampler.
//
// Replaces sample_points / _sample_points / _rand_barycentric_coords
// (src/transforms/mesh_func.jl:21-82).  The reference loops over meshes on the host, copies each
// probability vector D2H, builds a Distributions.Categorical alias table and draws with the
// host RNG (:43-55).  Here the whole batch is three launches, nothing leaves the device:
//   1. faces_areas_padded (mesh.hip)                         -- gather, HBM/L2 bound
//   2. face_cdf_kernel: Float64 probabilities (:32-39, incl. the last-padded-column fix-up) and
//      their CDF, one block per mesh, summed in the order specified in oracle/flux3d_oracle.c
//      ("blocked" order, chunks of 32) so oracle and device agree bit-for-bit
//   3. sample_kernel: one thread per sample: Philox4x32-10 draw -> binary search in the CDF ->
//      barycentric point  (w1*v1 + w2*v2) + w3*v3, unfused Float32 (:67-71,:75-82)
#include <cmath>

#include "fx3d_common.h"

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

// One block per mesh.  ws layout per mesh: cdf[Fp] then tc[nchunks]  (doubles), Fp = roundup32.
// The SUMMATION ORDER is fixed by the specification shared with the oracle (chunks of 32 summed
// left to right, chunk totals summed left to right); everything order-free (the areas, the Float64 divisions, the
// final offset add) runs on all threads, the order-bound parts are 32- or nch-long add chains.
// The face areas are computed in place (compute_faces_areas_padded, src/rep/mesh.jl:799-808: pad faces -> 0).
//
// One thread's left-to-right sum of t[0..n) (the specified order); EXCL: t[c] is replaced by the sum of the values
// before it.  BATCH: t is zero-padded to a multiple of 32 and lives in LDS -- 32 values are loaded, then added: the
// additions are the dependent chain, the LDS latency is paid once per 32 instead of once per 8 (+0.0 does not change a
// sum of non-negative terms).
template <bool EXCL, bool BATCH>
__device__ __forceinline__ double chain_scan(double *t, int n) {
    double s = 0.0;
    if constexpr (BATCH) {
        for (int c = 0; c < n; c += kChunk) {
            double v[kChunk];
#pragma unroll
            for (int i = 0; i < kChunk; ++i) v[i] = t[c + i];
#pragma unroll
            for (int i = 0; i < kChunk; ++i) {
                if (EXCL) t[c + i] = s;
                s += v[i];
            }
        }
    } else {
#pragma unroll 8
        for (int c = 0; c < n; ++c) {
            const double v = t[c];
            if (EXCL) t[c] = s;
            s += v;
        }
    }
    return s;
}

#ifdef FX3D_CDF_PROBE  // wall-clock stamps (10 ns) of block 0's phases in the unused chunk-total slots of ws (tools/face_cdf_ab.py)
#define CDF_MARK(i) if (IN_LDS && threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(out + Fp)[i] = (long long)wall_clock64()
#else
#define CDF_MARK(i)
#endif
// IN_LDS: the working copy (areas -> probabilities -> prefixes) and the chunk totals live in LDS (latency-bound chains).
// VLDS (with IN_LDS): the mesh's vertices are staged in LDS first (coalesced loads, in flight together with the face
// indices: ONE global round trip), the 3 gathers per face read LDS -- the block is alone on its mesh and one CU's
// texture path takes ~1.4 ns per face for scattered 12-byte loads (7.6 us of 16 for the fit loop's 5 k-face sphere).
template <bool IN_LDS, bool VLDS>
__global__ __launch_bounds__(kCdfThreads) void face_cdf_kernel(const float *__restrict__ verts_padded, int Vmax,
                                                            const int32_t *__restrict__ faces_padded,
                                                            const int32_t *__restrict__ faces_len, int Fmax,
                                                            int Fp, double eps,
                                                            double *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];  // IN_LDS: work[Fp + nch] + tc[nchp] (+ float4 verts[Vmax])
    const int b = blockIdx.x;
    const int nch = Fp / kChunk, nchp = (nch + kChunk - 1) / kChunk * kChunk;
    const float *vb = verts_padded + (size_t)b * Vmax * 3;
    const int32_t *fb = faces_padded + (size_t)b * Fmax * 3;
    double *out = ws + (size_t)b * (Fp + nch);
    double *cdf = IN_LDS ? dsm : out;
    // LDS copy: one pad double per chunk of 32, so that thread c walking chunk c (stride 33 doubles) and its neighbours
    // hit different banks -- with the plain stride of 256 B all 64 lanes of a wave shared one bank (face_cdf 22.8 -> 18.9 us)
    auto P = [](int k) { return IN_LDS ? k + (k >> 5) : k; };
    double *tc = IN_LDS ? dsm + Fp + nch : out + Fp;
    const int nct = IN_LDS ? nchp : nch;  // chunk totals walked by the chains (LDS copy: zero-padded)
    float4 *vl = reinterpret_cast<float4 *>(dsm + ((Fp + nch + nchp + 1) & ~1));
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
    for (int c = threadIdx.x; c < nct; c += kCdfThreads) {  // chunk totals of the areas (all loads first: the additions are the chain)
        double t = 0.0;
        if (c < nch) {
            double v[kChunk];
#pragma unroll
            for (int i = 0; i < kChunk; ++i) v[i] = cdf[P(c * kChunk) + i];
#pragma unroll
            for (int i = 0; i < kChunk; ++i) t += v[i];
        }
        tc[c] = t;
    }
    __syncthreads();
    CDF_MARK(2);
    if (threadIdx.x == 0) {
        const double s = chain_scan<false, IN_LDS>(tc, nct);
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
        tc[c] = l;
    }
    __syncthreads();
    CDF_MARK(5);
    if (threadIdx.x == 0) {
        // sum of the chunk totals (left to right) and, in the same walk, their exclusive scan in place: the running value
        // before chunk c IS its offset.  The fix-up column lies in the last chunk (Fp = roundup32(Fmax)), whose offset
        // does not depend on its own total -- only its local prefixes change.
        const double sp = chain_scan<true, IN_LDS>(tc, nct);
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
    for (int k = threadIdx.x; k < Fmax; k += kCdfThreads) out[k] = tc[k / kChunk] + cdf[P(k)];
    CDF_MARK(7);
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

__global__ __launch_bounds__(kThreads) void sample_seeded_kernel(
    const float *__restrict__ verts_padded, int Vmax, const int32_t *__restrict__ faces_padded,
    int Fmax, int Fp, const int32_t *__restrict__ faces_len, int B, int n, uint64_t seed_host,
    const uint64_t *__restrict__ seed_dev, const double *__restrict__ ws, float *__restrict__ out,
    int32_t *__restrict__ face_out, float *__restrict__ r1_out, float *__restrict__ r2_out) {
    const long long total = (long long)B * n;
    const int nch = Fp / kChunk;
    const uint64_t seed = seed_host + (seed_dev ? *seed_dev : 0);  // device part: advanced between replays of a graph
    for (long long k = (long long)blockIdx.x * kThreads + threadIdx.x; k < total;
         k += (long long)gridDim.x * kThreads) {
        const int b = (int)(k / n), sidx = (int)(k % n);
        const double *cdf = ws + (size_t)b * (Fp + nch);
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

int grid_for(long long n) {
    long long g = (n + kThreads - 1) / kThreads;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

inline int roundup32(int v) { return (v + kChunk - 1) / kChunk * kChunk; }

size_t ws_bytes_needed(int Fmax, int B) {
    const int Fp = roundup32(Fmax);
    const size_t cdf = sizeof(double) * (size_t)B * (Fp + Fp / kChunk);
    const size_t areas = sizeof(float) * (size_t)B * Fmax;
    return cdf + ((areas + 15) / 16) * 16;
}

}  // namespace

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

fx3d_status fx3d_sample_points_cdf(const float *verts_padded, int32_t Vmax, const int32_t *faces_padded,
                                   int32_t Fmax, const int32_t *faces_len, int32_t B, double eps, void *ws,
                                   size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(verts_padded && faces_padded && faces_len, "fx3d_sample_points_cdf: null pointer");
    FX3D_REQUIRE(Vmax > 0 && Fmax > 0 && B > 0, "fx3d_sample_points_cdf: bad sizes");
    if (!ws || ws_bytes < ws_bytes_needed(Fmax, B)) {
        set_error("fx3d_sample_points_cdf: workspace too small (%zu < %zu)", ws ? ws_bytes : (size_t)0,
                  ws_bytes_needed(Fmax, B));
        return FX3D_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(s);
    const int Fp = roundup32(Fmax);
    double *cdf = reinterpret_cast<double *>(ws);
    ProfileScope prof("sample_cdf", st);
    const size_t cdf_lds = sizeof(double) * (size_t)(Fp + 2 * (Fp / kChunk) + kChunk + 2);  // + one pad per chunk (bank spread), chunk totals padded to 32
    const size_t v_lds = sizeof(float) * 4 * (size_t)Vmax + 16;
    if (cdf_lds <= 60 * 1024 && cdf_lds + v_lds <= 150 * 1024) {
        const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&face_cdf_kernel<true, true>), 150 * 1024,
                                                   "face_cdf_kernel");
        if (arc != FX3D_OK) return arc;
        hipLaunchKernelGGL((face_cdf_kernel<true, true>), dim3(B), dim3(kCdfThreads), cdf_lds + v_lds, st, verts_padded, Vmax,
                           faces_padded, faces_len, Fmax, Fp, eps, cdf);
    } else if (cdf_lds <= 60 * 1024) {
        hipLaunchKernelGGL((face_cdf_kernel<true, false>), dim3(B), dim3(kCdfThreads), cdf_lds, st, verts_padded, Vmax,
                           faces_padded, faces_len, Fmax, Fp, eps, cdf);
    } else {
        hipLaunchKernelGGL((face_cdf_kernel<false, false>), dim3(B), dim3(kCdfThreads), 0, st, verts_padded, Vmax,
                           faces_padded, faces_len, Fmax, Fp, eps, cdf);
    }
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_sample_points_draw(const float *verts_padded, int32_t Vmax, const int32_t *faces_padded,
                                    int32_t Fmax, const int32_t *faces_len, int32_t B, int32_t n, uint64_t seed,
                                    const uint64_t *seed_dev, const void *cdf_ws, size_t ws_bytes, float *out,
                                    int32_t *face_out, float *r1_out, float *r2_out, fx3d_stream_t s) {
    FX3D_REQUIRE(verts_padded && faces_padded && faces_len && out, "fx3d_sample_points_draw: null pointer");
    FX3D_REQUIRE(Vmax > 0 && Fmax > 0 && B > 0 && n > 0, "fx3d_sample_points_draw: bad sizes");
    if (!cdf_ws || ws_bytes < ws_bytes_needed(Fmax, B)) {
        set_error("fx3d_sample_points_draw: CDF workspace too small (%zu < %zu)", cdf_ws ? ws_bytes : (size_t)0,
                  ws_bytes_needed(Fmax, B));
        return FX3D_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(s);
    ProfileScope prof("sample_draw", st);
    hipLaunchKernelGGL(sample_seeded_kernel, dim3(grid_for((long long)B * n)), dim3(kThreads), 0, st,
                       verts_padded, Vmax, faces_padded, Fmax, roundup32(Fmax), faces_len, B, n, seed, seed_dev,
                       reinterpret_cast<const double *>(cdf_ws), out, face_out, r1_out, r2_out);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
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

fx3d_status fx3d_sample_points_bwd(const int32_t *faces_padded, int32_t Vmax, int32_t Fmax, int32_t B,
                                   int32_t n, const int32_t *face_idx, const float *r1,
                                   const float *r2, const float *gout, float *gverts, int32_t accumulate,
                                   fx3d_stream_t s) {
    FX3D_REQUIRE(faces_padded && face_idx && r1 && r2 && gout && gverts, "fx3d_sample_points_bwd: null pointer");
    FX3D_REQUIRE(Vmax > 0 && Fmax > 0 && B > 0 && n > 0, "fx3d_sample_points_bwd: bad sizes");
    hipStream_t st = as_stream(s);
    if (!accumulate) FX3D_HIP(hipMemsetAsync(gverts, 0, sizeof(float) * 3 * (size_t)Vmax * B, st));
    hipLaunchKernelGGL(sample_bwd_kernel, dim3(grid_for((long long)B * n)), dim3(kThreads), 0, st,
                       faces_padded, Vmax, Fmax, B, n, face_idx, r1, r2, gout, gverts);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

}  // extern "C"
