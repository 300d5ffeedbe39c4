// Feature-space k-NN on the matrix cores (4 <= D <= 128): the per-cloud pre-pass of fx3d_knn_ws and knn_mfma_kernel -- one
// translation unit of the k-NN family (knn_common.h).
#include "knn_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// knn_mfma_kernel<DK>: feature-space kNN (4 <= D <= 128, k+drop <= 32; the second EdgeConv runs at D = 64).
// Same selection scheme as knn_f16_d3_kernel (lane = query; 64 group minima -> tau in registers; per-lane mask
// lists; verified distance-only ranking), with the filter as a dense Float32 GEMM:
//   F[c][q] = fl(|c|^2) + sum_d c_d (-2 q_d)   on v_mfma_f32_32x32x2_f32 (rows = 32 candidates of a tile, columns
//   = the wave's 32 queries, the accumulator starts at |c|^2; two tiles on two accumulators).  With u = 2^-24 and
//   the usual gamma_n bounds, |F + |q|^2 - d_oracle| <= eps_q = 8 (D+4) u (|q|^2 + Cmax^2) for every candidate
//   (Cmax = largest candidate norm of the cloud; 2x head-room for the matrix core's internal rounding), so the
//   candidates with F <= tau + 2 eps_q are a superset of the k nearest.
// Block = 4 consumer waves (32 queries each) + 4 producer waves that stage the next candidate chunk into the other
// LDS buffer with global_load_lds_dwordx4 while the consumers work (one consumer wave per SIMD: nothing else
// would hide global latency; VALU work of any wave delays that SIMD's MFMAs, hence the direct loads).  The
// reduction dimension is permuted so that half-wave h owns d in [h*DP/2, (h+1)*DP/2): every operand fetch is a
// b128 (4 k-steps); the LDS image is lane-linear, the conflict-free rotation sits on the source addresses.
// In the exact phase all eight waves work: the four lanes of a query (two halves x consumer/producer) split its
// survivors; candidate and query rows are gathered from L2.
// Template modes: F16 = false: the Float32 GEMM above.  F16 = true (default for D % 4 == 0, M <= 4096): the cloud is
// centred per dimension and scaled by a power of two, operands are fp16; SPLIT = false (default) uses the rounded halves
// alone (one v_mfma_f32_32x32x16_f16 per K block, band 2^-10 (|q~|^2 + C~max^2)), SPLIT = true the 2-way split
// hi*hi + lo*hi + hi*lo (band 2^-18).  Producers convert while staging.
// Queries whose band holds more candidates than the key arrays (60) but whose lane lists are intact take the medium
// path (exact selection among their own survivors, up to kMMedCap); the rest of the leftovers the full exact merge.
// PRE (fx3d_knn_ws, the pre-pass has built the cloud's fp16 image): both waves of a pair run the filter (DUAL, 128 group minima per
// query).  Round 4, in this instantiation: every lane reads its pieces of its query row and of the centre straight from memory (no
// LDS staging, no barrier before the first chunk's); the image chunks come through registers (the direct-to-LDS form of round 4 measured 1.2 us slower
// and was removed with its switch in round 5); one instantiation of the four-tile loop per phase (phase A folds two tiles per v_min3 on the MFMA registers, phase B starts
// the accumulators at n_c - thr and shifts the signs in with v_alignbit); tau by knn_tau_8of16; survivors counted in phase B; the
// exact phase on 16-dimension COLUMN SLICES of the whole cloud (M <= 1024, D % 16 == 0, D <= 64; other shapes take the row
// stages of rounds 2-3: chosen by shape, the switch is gone), pairs in registers across the slices.  C4': kernel 60.8 -> 50.8 us (profiles/r04_v5_*).
constexpr int kMLCap = 40;        // rows of a lane's mask list (39 usable + the scratch head)
constexpr int kMKeyCap = 64;      // survivors per query handled by the fast path (three sentinels follow them inside the stride of 68)
constexpr int kMMedCap = 512;     // ... by the medium path: exact selection among the query's own survivors
constexpr int kMKeyStride = 68;   // row stride of the key arrays in words: 32 queries x b128 reads without bank conflicts

// fp16-split staging of a candidate chunk (producer side, F16 filter): unit = (row, group of 8 dimensions).
// A thread converts two float4 of a row (scaled by sc) into one hi piece and one lo piece of 8 halves each.
// LDS row: pieces [0, PPR/2) = hi of dimension groups, [PPR/2, PPR) = lo; piece c of row r sits at (c + r) mod PPR.
// (row, group) units per producer thread and chunk: CH * (DP/8) <= units * 256.  The single-piece image is half
// the size, so its chunks can be twice as long (fewer steps: a step cannot be shorter than the latency of the
// global loads issued one step ahead)
constexpr int kMUnitsSplit = 5, kMUnitsSingle = 6;
template <int DK, int kMUnits>
__device__ __forceinline__ void knn_f16_load_chunk(const float *__restrict__ yb, int D, int j0, int cn, int CH, int ptid,
                                                   float4 (&reg)[kMUnits][2]) {
    constexpr int DP = DK * 32, G = DP / 8;
    // unconditional loads from clamped (always valid) addresses, zeroed afterwards: predicated loads would be
    // issued one branch at a time, each waiting for its data
    const float4 zero4 = float4{0.f, 0.f, 0.f, 0.f};
    const int cnm1 = cn - 1;
#pragma unroll
    for (int u = 0; u < kMUnits; ++u) {
        const int un = ptid + u * kMProd;
        const int row = un / G, g = un - row * G;
        const int rowc = row < cnm1 ? row : cnm1;
        const int d0 = 8 * g < D ? 8 * g : 0, d1 = 8 * g + 4 < D ? 8 * g + 4 : 0;
        const float *src = yb + (size_t)(j0 + rowc) * D;
        reg[u][0] = *reinterpret_cast<const float4 *>(src + d0);
        reg[u][1] = *reinterpret_cast<const float4 *>(src + d1);
    }
#pragma unroll
    for (int u = 0; u < kMUnits; ++u) {
        const int un = ptid + u * kMProd;
        const int row = un / G, g = un - row * G;
        const bool ok = un < CH * G && row < cn;
        if (!(ok && 8 * g < D)) reg[u][0] = zero4;
        if (!(ok && 8 * g + 4 < D)) reg[u][1] = zero4;
    }
}
// fp16 single-piece image (SPLIT = false): rows of PPI = DP/8 pieces (16 bytes = 8 halves); piece c of row r sits
// at (c + r / RPB) mod PPI, RPB = rows per 256 bytes, so that 16 consecutive rows cover all LDS banks.
template <int PPI>
__device__ __forceinline__ int knn_hpiece_off(int row, int c) {
    constexpr int RPB = PPI >= 16 ? 1 : 16 / PPI;
    return (row * PPI + ((c + row / RPB) & (PPI - 1))) * 4;
}
// Converts and stores the units; the G = DP/8 consecutive lanes that hold one row also sum its scaled norm
// (3..4 butterfly steps).  norms != nullptr: phase A, norms[row] and the running maximum are recorded.
template <int CTRL>
__device__ __forceinline__ float knn_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int DK, bool SPLIT, int kMUnits>
__device__ __forceinline__ void knn_f16_store_chunk(float *img, int CH, int cn, float sc, const float *mu_lds, int ptid,
                                                    const float4 (&reg)[kMUnits][2], float *norms, float *norms_m,
                                                    const float *acoef_lds, float &tmax, bool &tnan) {
    constexpr int DP = DK * 32, G = DP / 8, PPR = DK * 8;
    static_assert(kMProd % G == 0, "a producer thread always converts the same group of eight dimensions");
    static_assert(G == 4 || G == 8 || G == 16, "the row sum below");
    // (centre and error coefficient are re-read from LDS per call: holding them in registers across the chunk loop spills)
    float mu8[8];
    {
        const float4 m0 = *reinterpret_cast<const float4 *>(mu_lds + 8 * (ptid % G)), m1 = *reinterpret_cast<const float4 *>(mu_lds + 8 * (ptid % G) + 4);
        mu8[0] = m0.x; mu8[1] = m0.y; mu8[2] = m0.z; mu8[3] = m0.w; mu8[4] = m1.x; mu8[5] = m1.y; mu8[6] = m1.z; mu8[7] = m1.w;
    }
    const float acoef = norms_m ? *acoef_lds : 0.0f;
#pragma unroll
    for (int u = 0; u < kMUnits; ++u) {
        const int un = ptid + u * kMProd;
        const int row = un / G, g = un - row * G;
        const float v[8] = {(reg[u][0].x - mu8[0]) * sc, (reg[u][0].y - mu8[1]) * sc, (reg[u][0].z - mu8[2]) * sc,
                            (reg[u][0].w - mu8[3]) * sc, (reg[u][1].x - mu8[4]) * sc, (reg[u][1].y - mu8[5]) * sc,
                            (reg[u][1].z - mu8[6]) * sc, (reg[u][1].w - mu8[7]) * sc};
        kh8 hi, lo;
        float part = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 hh_ = (_Float16)v[e];
            hi[e] = hh_;
            if (SPLIT) lo[e] = (_Float16)(v[e] - (float)hh_);
            part = __builtin_fmaf(v[e], v[e], part);
        }
        if (un < CH * G) {
            if (SPLIT) {
                *reinterpret_cast<kh8 *>(img + knn_piece_off<DK>(row, g)) = hi;
                *reinterpret_cast<kh8 *>(img + knn_piece_off<DK>(row, PPR / 2 + g)) = lo;
            } else {
                *reinterpret_cast<kh8 *>(img + knn_hpiece_off<G>(row, g)) = hi;
            }
        }
        if (norms) {  // wave-uniform
            // sum over the row's G lanes by DPP (VALU speed; the same tree as an xor butterfly in the row's first lane,
            // the only one that uses it)
            part = part + knn_dpp<0xB1>(part);                // quad_perm [1,0,3,2]
            part = part + knn_dpp<0x4E>(part);                // quad_perm [2,3,0,1]
            if (G >= 8) part = part + knn_dpp<0x141>(part);   // row_half_mirror
            if (G >= 16) part = part + knn_dpp<0x140>(part);  // row_mirror
            if (un < CH * G && g == 0) {
                const float t = row < cn ? part : INFINITY;  // rows beyond the cloud: F = +inf
                // norms_m: the candidate's own share of the filter error is folded into its norm, upwards for the
                // threshold search (phase A), downwards for the test (phase B)
                norms[row] = norms_m ? t + acoef * t : t;
                if (norms_m) norms_m[row] = row < cn ? t - acoef * t : INFINITY;
                if (row < cn) { tnan |= (t != t); tmax = fmaxf(tmax, t); }
            }
        }
    }
}

// staged exact phase: the oracle's squared distance of one query row (registers) to two staged candidate rows (LDS),
// dimension by dimension in order.  Differences and squares two dimensions per instruction (v_pk_add/mul_f32 on the
// natural register pairs), the sums one by one; the next four pieces of both rows are in flight while four are summed.
// FULL: D == DP (no guards).
template <int DP, bool FULL>
__device__ __forceinline__ void knn_pair_dist(const f32x4v (&q)[DP / 4], const float *cp0, const float *cp1, int D, float &s0, float &s1) {
    constexpr int NB = DP / 16;
    f32x4v c0[2][4], c1[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (FULL || 4 * t < D) {
            c0[0][t] = *reinterpret_cast<const f32x4v *>(cp0 + 4 * t);
            c1[0][t] = *reinterpret_cast<const f32x4v *>(cp1 + 4 * t);
        }
#pragma unroll
    for (int bk = 0; bk < NB; ++bk) {
        if (bk + 1 < NB) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (FULL || 16 * (bk + 1) + 4 * t < D) {
                    c0[(bk + 1) & 1][t] = *reinterpret_cast<const f32x4v *>(cp0 + 16 * (bk + 1) + 4 * t);
                    c1[(bk + 1) & 1][t] = *reinterpret_cast<const f32x4v *>(cp1 + 16 * (bk + 1) + 4 * t);
                }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (FULL || 16 * bk + 4 * t < D) {
                const f32x4v d0 = q[4 * bk + t] - c0[bk & 1][t], d1 = q[4 * bk + t] - c1[bk & 1][t];
                const f32x4v m0 = d0 * d0, m1 = d1 * d1;
                s0 = s0 + m0.x; s0 = s0 + m0.y; s0 = s0 + m0.z; s0 = s0 + m0.w;
                s1 = s1 + m1.x; s1 = s1 + m1.y; s1 = s1 + m1.z; s1 = s1 + m1.w;
            }
    }
}

// staged exact phase: a thread's share of one stage of candidate rows (8 pieces of 16 bytes, rows srow + i * RPI of
// the stage that starts at row g0; clamped addresses: always valid, unused rows are never stored)
__device__ __forceinline__ void knn_stage_fetch(const float *__restrict__ yb, int D, int M, int g0, int srow, int RPI, int scol,
                                                f32x4v (&reg)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int g = g0 + srow + i * RPI;
        g = g < M ? g : M - 1;
        reg[i] = *reinterpret_cast<const f32x4v *>(yb + (size_t)g * D + scol);
    }
}

// ------------------------------------------------------------------------------------------------
// Pre-pass of the feature-space kNN (fx3d_knn_ws): the per-cloud statistics and the fp16 image are built ONCE per cloud
// instead of by every block of the cloud (8 blocks per cloud at C4': the scale pass alone was 8 us of the 78, bound by the
// L2 -- every block read the whole cloud -- and the producers' conversion VALU delayed the consumers' MFMAs in both phases).
//   knn_pre_stats_kernel   grid (kPreParts, B): minima / maxima / sums per dimension of one eighth of the cloud's rows
//   knn_pre_image_kernel   grid (kPreParts, B): combines the eight parts (every block the same arithmetic: identical centre
//                          and scale), robust centre as in knn_mfma_kernel, converts its rows: fp16 image [Mpad][DP], scaled
//                          row norms (the candidate's error share folded in, upwards / downwards), largest norm of the part
// knn_mfma_kernel then reads the header, and its producer waves bring image chunks and norms in with direct-to-LDS loads (no
// VALU).  Workspace per cloud: KnnPre::cloud_bytes(M, DP).
constexpr int kPreParts = 8;
constexpr int kPreThreads = 256;
struct KnnPre {
    unsigned char *base;  // workspace
    size_t stride;        // bytes per cloud
    int Mpad;             // rows of the image (multiple of 256: chunks never need a clamp), DP halves per row
    // offsets inside a cloud's slab (bytes)
    size_t off_parts, off_hdr, off_cmax, off_nup, off_ndn, off_img;
    __host__ __device__ static KnnPre make(void *ws, int M, int DP) {
        KnnPre k{};
        k.base = static_cast<unsigned char *>(ws);
        k.Mpad = (M + 255) / 256 * 256;
        size_t o = 0;
        k.off_parts = o; o += (size_t)kPreParts * (3 * DP + 4) * 4;
        k.off_hdr = o; o += (size_t)(8 + DP) * 4;
        k.off_cmax = o; o += (size_t)kPreParts * 4;
        o = (o + 63) & ~(size_t)63;
        k.off_nup = o; o += (size_t)k.Mpad * 4;
        k.off_ndn = o; o += (size_t)k.Mpad * 4;
        k.off_img = o; o += (size_t)k.Mpad * DP * 2;
        k.stride = (o + 255) & ~(size_t)255;
        return k;
    }
    __host__ __device__ float *parts(int b) const { return reinterpret_cast<float *>(base + (size_t)b * stride + off_parts); }
    __host__ __device__ float *hdr(int b) const { return reinterpret_cast<float *>(base + (size_t)b * stride + off_hdr); }
    __host__ __device__ unsigned int *cmaxp(int b) const { return reinterpret_cast<unsigned int *>(base + (size_t)b * stride + off_cmax); }
    __host__ __device__ float *nup(int b) const { return reinterpret_cast<float *>(base + (size_t)b * stride + off_nup); }
    __host__ __device__ float *ndn(int b) const { return reinterpret_cast<float *>(base + (size_t)b * stride + off_ndn); }
    __host__ __device__ _Float16 *img(int b) const { return reinterpret_cast<_Float16 *>(base + (size_t)b * stride + off_img); }
};
// header floats: [0] sc  [1] funit  [2] acoef (candidate side)  [3] bits: 1 = non-finite / overflow-prone cloud  [8 ...] mu[DP]

// The parts' statistics travel between the blocks of a cloud INSIDE the fused pre-pass kernel: device-coherent accesses (relaxed
// atomics at agent scope: write-through stores, loads that do not hit a stale line -- the blocks may sit on different XCDs, each
// with its own L2) instead of agent-scope fences, which write back / invalidate a whole L2 per block (measured: + 0.1 us per block
// of the grid, serialised per XCD: C4' 76 -> 104 us).
__device__ __forceinline__ void knn_pre_put(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float knn_pre_get(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void knn_pre_stats_body(const float *__restrict__ y, int M, int D, int DP, const KnnPre &pre, int part, int b,
                                                   float *red) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *yb = y + (size_t)b * M * D;
    const int rq = D / 4;  // (kPreThreads % rq == 0: thread t always sees dimensions 4 (t % rq) ...)
    const int per = (M + kPreParts - 1) / kPreParts;
    const int r_lo = part * per < M ? part * per : M, r_hi = r_lo + per < M ? r_lo + per : M;
    const float4 *c4 = reinterpret_cast<const float4 *>(yb) + (size_t)r_lo * rq;
    const int total4 = (r_hi - r_lo) * rq;
    float4 lo4 = float4{INFINITY, INFINITY, INFINITY, INFINITY}, hi4 = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float4 sum4 = float4{0.f, 0.f, 0.f, 0.f};
    float poison = 0.0f;
    for (int e0 = tid; e0 < total4; e0 += 8 * kPreThreads) {
        float4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = c4[e0 + e * kPreThreads < total4 ? e0 + e * kPreThreads : e0];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            poison = __builtin_fmaf(v[e].x, 0.0f, poison); poison = __builtin_fmaf(v[e].y, 0.0f, poison);
            poison = __builtin_fmaf(v[e].z, 0.0f, poison); poison = __builtin_fmaf(v[e].w, 0.0f, poison);
            lo4.x = fminf(lo4.x, v[e].x); lo4.y = fminf(lo4.y, v[e].y); lo4.z = fminf(lo4.z, v[e].z); lo4.w = fminf(lo4.w, v[e].w);
            hi4.x = fmaxf(hi4.x, v[e].x); hi4.y = fmaxf(hi4.y, v[e].y); hi4.z = fmaxf(hi4.z, v[e].z); hi4.w = fmaxf(hi4.w, v[e].w);
            if (e0 + e * kPreThreads < total4) {
                sum4.x = sum4.x + v[e].x; sum4.y = sum4.y + v[e].y; sum4.z = sum4.z + v[e].z; sum4.w = sum4.w + v[e].w;
            }
        }
    }
    const bool anynan = __syncthreads_or(poison != poison) != 0;
    for (int m = rq; m < 64; m <<= 1) {  // lanes with equal lane % rq hold the same dimensions
        lo4.x = fminf(lo4.x, __shfl_xor(lo4.x, m, 64)); lo4.y = fminf(lo4.y, __shfl_xor(lo4.y, m, 64));
        lo4.z = fminf(lo4.z, __shfl_xor(lo4.z, m, 64)); lo4.w = fminf(lo4.w, __shfl_xor(lo4.w, m, 64));
        hi4.x = fmaxf(hi4.x, __shfl_xor(hi4.x, m, 64)); hi4.y = fmaxf(hi4.y, __shfl_xor(hi4.y, m, 64));
        hi4.z = fmaxf(hi4.z, __shfl_xor(hi4.z, m, 64)); hi4.w = fmaxf(hi4.w, __shfl_xor(hi4.w, m, 64));
        sum4.x = sum4.x + __shfl_xor(sum4.x, m, 64); sum4.y = sum4.y + __shfl_xor(sum4.y, m, 64);
        sum4.z = sum4.z + __shfl_xor(sum4.z, m, 64); sum4.w = sum4.w + __shfl_xor(sum4.w, m, 64);
    }
    if (lane < rq) {
        float *r8 = red + (size_t)(wv * 32 + lane) * 12;
        r8[0] = lo4.x; r8[1] = lo4.y; r8[2] = lo4.z; r8[3] = lo4.w;
        r8[4] = hi4.x; r8[5] = hi4.y; r8[6] = hi4.z; r8[7] = hi4.w;
        r8[8] = sum4.x; r8[9] = sum4.y; r8[10] = sum4.z; r8[11] = sum4.w;
    }
    __syncthreads();
    float *out = pre.parts(b) + (size_t)part * (3 * DP + 4);
    if (tid < rq) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float lo = INFINITY, hi = -INFINITY, sm = 0.0f;
            for (int w = 0; w < kPreThreads / 64; ++w) {
                const float *r8 = red + (size_t)(w * 32 + tid) * 12;
                lo = fminf(lo, r8[c]); hi = fmaxf(hi, r8[4 + c]); sm = sm + r8[8 + c];
            }
            knn_pre_put(out + 4 * tid + c, lo); knn_pre_put(out + DP + 4 * tid + c, hi); knn_pre_put(out + 2 * DP + 4 * tid + c, sm);
        }
    }
    if (tid == 0) knn_pre_put(out + 3 * DP, __builtin_bit_cast(float, anynan ? 1 : 0));
}
__global__ __launch_bounds__(kPreThreads) void knn_pre_stats_kernel(const float *__restrict__ y, int M, int D, int DP, KnnPre pre) {
    __shared__ float red[(kPreThreads / 64) * 32 * 12];
    knn_pre_stats_body(y, M, D, DP, pre, blockIdx.x, blockIdx.y, red);
}

// mu: [DP] floats, sh: 4 words of LDS ([0] bits of the extent  [1] skew flag  [2] bits of the bulk radius  [3] largest norm of the part)
template <int DK>
__device__ __forceinline__ void knn_pre_image_body(const float *__restrict__ y, int M, int D, int two_norms, const KnnPre &pre, int part, int b,
                                                   float *mu, unsigned int *sh) {
    constexpr int DP = DK * 32, G = DP / 8;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *yb = y + (size_t)b * M * D;
    const int rq = D / 4;
    const float *parts = pre.parts(b);
    // this part's rows: the first four sweeps (a whole part at C4': 128 rows) are requested BEFORE the parts' statistics are read --
    // the rows do not depend on them, and the kernel is two dependent global round trips otherwise (round 4)
    const int per = (M + kPreParts - 1) / kPreParts;
    const int r_lo = part * per < M ? part * per : M, r_hi = r_lo + per < M ? r_lo + per : M;
    const int g = tid % G;  // (kPreThreads % G == 0: a thread always converts the same eight dimensions)
    constexpr int RPS = kPreThreads / G;  // rows per sweep of the block
    const int d0 = 8 * g < D ? 8 * g : 0, d1 = 8 * g + 4 < D ? 8 * g + 4 : 0;
    float4 a0[4], a1[4];
    auto load_rows = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // every load of the four sweeps before the first use (a part is a few sweeps: latency, not bandwidth)
            const int row = r0 + u * RPS + tid / G;
            const float *src = yb + (size_t)(row < r_hi ? row : (r_lo < M ? r_lo : 0)) * D;
            a0[u] = *reinterpret_cast<const float4 *>(src + d0);
            a1[u] = *reinterpret_cast<const float4 *>(src + d1);
        }
    };
    load_rows(r_lo);
    if (tid < 4) sh[tid] = 0u;
    bool anynan = false;
    for (int p = 0; p < kPreParts; ++p) anynan |= __builtin_bit_cast(int, knn_pre_get(parts + (size_t)p * (3 * DP + 4) + 3 * DP)) != 0;
    __syncthreads();
    if (tid < rq) {
        float amax = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float lo = INFINITY, hi = -INFINITY, sm = 0.0f;
            for (int p = 0; p < kPreParts; ++p) {  // (fixed order: every block of the cloud gets the same centre)
                const float *q = parts + (size_t)p * (3 * DP + 4);
                lo = fminf(lo, knn_pre_get(q + 4 * tid + c)); hi = fmaxf(hi, knn_pre_get(q + DP + 4 * tid + c)); sm = sm + knn_pre_get(q + 2 * DP + 4 * tid + c);
            }
            float m0 = sm / (float)M;
            m0 = fminf(fmaxf(m0, lo), hi);
            mu[4 * tid + c] = m0;
            amax = fmaxf(amax, fmaxf(hi - m0, m0 - lo));
            if (fabsf(m0 - 0.5f * (lo + hi)) > 0.25f * (hi - lo)) sh[1] = 1u;
        }
        atomicMax(&sh[0], __builtin_bit_cast(unsigned int, amax));
    } else if (tid < DP / 4) {
        mu[4 * tid] = 0.0f; mu[4 * tid + 1] = 0.0f; mu[4 * tid + 2] = 0.0f; mu[4 * tid + 3] = 0.0f;
    }
    __syncthreads();
    if (sh[1] != 0u && !anynan) {  // robust centre: see knn_mfma_kernel (the same rule on the same 16 sampled rows)
        if (wv == 0) {
            float shiftmax = 0.0f, iqr2 = 0.0f;
            bool sw = false;
            float medv[DP / 64 > 0 ? DP / 64 : 1];
#pragma unroll
            for (int t = 0; t < (DP + 63) / 64; ++t) {
                const int d = lane + 64 * t;
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = yb[(size_t)((long long)i * M / 16) * D + (d < D ? d : 0)];
                float med = v[0], q1 = v[0], q3 = v[0];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    int rk = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) rk += (v[j] < v[i] || (v[j] == v[i] && j < i)) ? 1 : 0;
                    med = rk == 8 ? v[i] : med; q1 = rk == 4 ? v[i] : q1; q3 = rk == 12 ? v[i] : q3;
                }
                const float shift = d < D ? fabsf(mu[d < D ? d : 0] - med) : 0.0f;
                sw = sw || (shift > 8.0f * (q3 - q1));
                shiftmax = fmaxf(shiftmax, shift);
                if (d < D) iqr2 = __builtin_fmaf(q3 - q1, q3 - q1, iqr2);
                medv[t] = med;
            }
            if (__ballot(sw) != 0ull) {
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) {
                    shiftmax = fmaxf(shiftmax, __shfl_xor(shiftmax, m, 64));
                    iqr2 = iqr2 + __shfl_xor(iqr2, m, 64);
                }
#pragma unroll
                for (int t = 0; t < (DP + 63) / 64; ++t)
                    if (lane + 64 * t < D) mu[lane + 64 * t] = medv[t];
                if (lane == 0) {
                    sh[0] = __builtin_bit_cast(unsigned int, __builtin_bit_cast(float, sh[0]) + shiftmax);
                    sh[2] = __builtin_bit_cast(unsigned int, sqrtf(iqr2));
                }
            }
        }
        __syncthreads();
    }
    const float cinf = __builtin_bit_cast(float, sh[0]);
    float sc = 1.0f;
    if (cinf > 1.0e-30f && cinf < 1.0e30f) {
        int e;
        (void)frexpf(cinf * 1.000001f, &e);
        sc = ldexpf(1.0f, 10 - e);
    }
    const float rad = __builtin_bit_cast(float, sh[2]);
    const float funit = rad > 0.0f ? fminf(1.0f, fmaxf(sc * rad, 0x1p-12f)) : 1.0f;
    const float aq = 8.0f * (float)(4 * D + 8) * 0x1p-24f + 0x1.01p-10f;
    const float acoef = aq * 1.01f + 0x1p-26f * sqrtf((float)D) / funit + 0x1p-23f;
    const bool bad = anynan || !(cinf < 1.0e15f);
    if (part == 0) {
        float *h = pre.hdr(b);
        if (tid == 0) { h[0] = sc; h[1] = funit; h[2] = acoef; h[3] = bad ? 1.0f : 0.0f; }
        if (tid < DP) h[8 + tid] = mu[tid];
    }
    // ---- this part's rows -> image, norms
    _Float16 *img = pre.img(b);
    float *nup = pre.nup(b), *ndn = pre.ndn(b);
    float mu8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) mu8[e] = mu[8 * g + e];
    float tmax = 0.0f;
    bool tnan = false;
    for (int r0 = r_lo; r0 < r_hi; r0 += 4 * RPS) {
        if (r0 != r_lo) load_rows(r0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = r0 + u * RPS + tid / G;
            const bool ok = row < r_hi;
            if (!(8 * g < D)) a0[u] = float4{mu8[0], mu8[1], mu8[2], mu8[3]};       // padding dimensions: zero pieces
            if (!(8 * g + 4 < D)) a1[u] = float4{mu8[4], mu8[5], mu8[6], mu8[7]};
            const float v[8] = {(a0[u].x - mu8[0]) * sc, (a0[u].y - mu8[1]) * sc, (a0[u].z - mu8[2]) * sc, (a0[u].w - mu8[3]) * sc,
                                (a1[u].x - mu8[4]) * sc, (a1[u].y - mu8[5]) * sc, (a1[u].z - mu8[6]) * sc, (a1[u].w - mu8[7]) * sc};
            kh8 hi;
            float pt = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hi[e] = (_Float16)v[e];
                pt = __builtin_fmaf(v[e], v[e], pt);
            }
            pt = pt + knn_dpp<0xB1>(pt);
            pt = pt + knn_dpp<0x4E>(pt);
            if (G >= 8) pt = pt + knn_dpp<0x141>(pt);
            if (G >= 16) pt = pt + knn_dpp<0x140>(pt);
            if (ok) {
                *reinterpret_cast<kh8 *>(img + ((size_t)row * G + g) * 8) = hi;
                if (g == 0) {
                    nup[row] = two_norms ? pt + acoef * pt : pt;
                    ndn[row] = two_norms ? pt - acoef * pt : pt;
                    tnan |= (pt != pt);
                    tmax = fmaxf(tmax, pt);
                }
            }
        }
    }
    if (part == kPreParts - 1) {  // rows [M, Mpad): zero pieces, norm +inf (F = +inf: never selected)
        kh8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
        for (int un = tid; un < (pre.Mpad - M) * G; un += kPreThreads) *reinterpret_cast<kh8 *>(img + ((size_t)M * G + un) * 8) = z;
        for (int r = M + tid; r < pre.Mpad; r += kPreThreads) { nup[r] = INFINITY; ndn[r] = INFINITY; }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, m, 64));
    const bool anyn = __ballot(tnan) != 0;
    if (lane == 0) atomicMax(&sh[3], anyn ? 0x7fc00000u : __builtin_bit_cast(unsigned int, tmax));
    __syncthreads();
    if (tid == 0) pre.cmaxp(b)[part] = bad ? 0x7fc00000u : sh[3];
}
template <int DK>
__global__ __launch_bounds__(kPreThreads) void knn_pre_image_kernel(const float *__restrict__ y, int M, int D, int two_norms, KnnPre pre) {
    __shared__ float mu[DK * 32];
    __shared__ unsigned int sh[4];
    knn_pre_image_body<DK>(y, M, D, two_norms, pre, blockIdx.x, blockIdx.y, mu, sh);
}
// (Round 4 tried the whole pre-pass as ONE launch of one 1024-thread block per cloud -- the cloud in registers between the statistics
//  and the conversion, no meeting: 12.7 us under rocprofv3 against 4.8 + 6.4 for the two launches, calls 1-2 us slower at four of five
//  shapes (profiles/r04_v4_knn_prepass_ab.txt): 32 CUs stream 256 KB each at ~50 GB/s.  Removed; the two launches stay.)

// producer wave pw brings the norms of chunk [j0, j0 + CH) into the block's norm arrays (direct-to-LDS; not waited for here)
__device__ __forceinline__ void knn_pre_stage_norms(const float *__restrict__ gnup, const float *__restrict__ gndn, int j0, int CH, float *nup,
                                                    float *ndn, int pw, int lane) {
    const int nin = CH / 4 / 64;  // wave-instructions per norm array (CH / 4 pieces of four floats); CH = 64 -> a quarter wave
    for (int i = pw; i < (nin > 0 ? nin : 1); i += kMWaves)
        if (i * 64 + lane < CH / 4) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gnup + j0 + (size_t)(i * 64 + lane) * 4),
                                             (__attribute__((address_space(3))) void *)(nup + (size_t)i * 256), 16, 0, 0);
            if (ndn)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gndn + j0 + (size_t)(i * 64 + lane) * 4),
                                                 (__attribute__((address_space(3))) void *)(ndn + (size_t)i * 256), 16, 0, 0);
        }
}
// producer wave pw brings chunk [j0, j0 + CH) of the pre-pass image into `img` (single-piece layout of knn_hpiece_off: the
// rotation sits on the source address) and, in phase A, its norms into the block's norm arrays -- direct-to-LDS loads only
template <int DK, bool WAIT = true>
__device__ __forceinline__ void knn_pre_stage_chunk(const _Float16 *__restrict__ gimg, const float *__restrict__ gnup,
                                                    const float *__restrict__ gndn, int j0, int CH, float *img, float *nup,
                                                    float *ndn, bool norms, int pw, int lane) {
    constexpr int PPI = DK * 4;                       // 16-byte pieces per image row (DP halves)
    constexpr int RPB = PPI >= 16 ? 1 : 16 / PPI;
    const int ninstr = CH * PPI / 64 / kMWaves;       // wave-instructions of this producer wave (CH is a multiple of 64)
    for (int i = 0; i < ninstr; ++i) {
        const int S0 = (pw * ninstr + i) * 64;        // first 16-byte slot of this wave-instruction
        const int S = S0 + lane;
        const int row = S / PPI, pos = S & (PPI - 1);
        const int c = (pos - row / RPB) & (PPI - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gimg + ((size_t)(j0 + row) * PPI + c) * 8),
                                         (__attribute__((address_space(3))) void *)(img + (size_t)S0 * 4), 16, 0, 0);
    }
    if (norms) {
        const int nin = CH / 4 / 64;  // wave-instructions per norm array (CH / 4 pieces of four floats); CH = 64 -> a quarter wave
        for (int i = pw; i < (nin > 0 ? nin : 1); i += kMWaves)
            if (i * 64 + lane < CH / 4) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gnup + j0 + (size_t)(i * 64 + lane) * 4),
                                                 (__attribute__((address_space(3))) void *)(nup + (size_t)i * 256), 16, 0, 0);
                if (ndn)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gndn + j0 + (size_t)(i * 64 + lane) * 4),
                                                     (__attribute__((address_space(3))) void *)(ndn + (size_t)i * 256), 16, 0, 0);
            }
    }
    if (WAIT) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces have landed
        __builtin_amdgcn_wave_barrier();
    }
}

template <int DK, bool F16, bool SPLIT, bool PRE = false>
__global__ __launch_bounds__(kMThreads) void knn_mfma_kernel(const float *__restrict__ x, int N,
                                                             const float *__restrict__ y, int M, int B, int D,
                                                             int k, int drop, int32_t *__restrict__ idx,
                                                             float *__restrict__ dist, int CH, int img_floats,
                                                             int keep_norms, int two_norms, int srl, void *pre_ws, int xdiv, int csl, int regstage) {
    constexpr int DP = DK * 32;      // padded feature dimension
    constexpr int RS = DP + 4;       // row stride of the query rows staged in the prologue (floats)
    constexpr int PPR = DK * 8;      // 16-byte pieces per candidate row
    constexpr int NT = DP / 8;       // b128 operand fetches per tile and half
    constexpr int NB16 = DP / 16;    // K blocks of the fp16 filter
    constexpr int RSI = (F16 && !SPLIT) ? DP / 2 : DP;  // image row stride in floats (single-piece fp16: hi halves only)
    constexpr int PPI = RSI / 4;     // 16-byte pieces per image row
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int buf_floats = CH * RSI + CH;                                  // image [CH][RSI] + norms [CH]
    // fixed-size bookkeeping first, then lists | med | nall: from the lists on everything is dead once the survivors are
    // decoded, so the staged exact phase uses that whole tail of the allocation for candidate rows
    int *lcnt = reinterpret_cast<int *>(sm + img_floats);                  // [kMWaves][64]  list lengths
    int *qn_n = lcnt + kMWaves * 64;                                       // [kMWaves][32]  survivors per query
    int *qflag = qn_n + kMWaves * 32;                                      // [kMWaves][32]  1 = fast path
    int *qbelow = qflag + kMWaves * 32;                                    // [kMWaves][32]  entries with rank < kk
    unsigned int *cmax = reinterpret_cast<unsigned int *>(qbelow + kMWaves * 32);  // bits of max |c|^2 (>= 0)
    float *mu = reinterpret_cast<float *>(cmax + 4);                       // [DP] F16: per-dimension centre of the cloud
    unsigned long long *qstpk = reinterpret_cast<unsigned long long *>(mu + DP);  // [kMWaves][32] survivors per row stage (packed prefix)
    unsigned short *lcnt2 = reinterpret_cast<unsigned short *>(lcnt);      // DUAL (below): [2 kMWaves][64] list lengths, in lcnt's space
    int *lists = reinterpret_cast<int *>(qstpk + kMWaves * 32);            // [kMWaves][kMLCap][64] mask words (DUAL: [2 kMWaves][kMLCap / 2][64])
    int *med = lists + kMWaves * kMLCap * 64;                              // [2 kMWaves][kMMedCap + 128] medium path: ids + merge lists
    float *nall = reinterpret_cast<float *>(med + 2 * kMWaves * (kMMedCap + 128));  // [nchunk*CH] all candidate norms (keep_norms)
    // DUAL: the four parts' packed stage counts per query [kMWaves][32][4] (4 KiB) live in the norm arrays, which are dead after
    // the chunk loop and at least that large (one array of >= 2304 floats, or two of >= 512)
    unsigned long long *qpk = reinterpret_cast<unsigned long long *>(nall);
    // block L runs on XCD L % 8: give every cloud's blocks ids with equal L % 8 so that its candidates stay in
    // one L2 (8 or more clouds; fewer: plain order, a cloud's blocks spread over all XCDs)
    const int nbx = (N + kMWaves * 32 - 1) / (kMWaves * 32);
    const int L = blockIdx.x;
    const bool by_xcd = B >= 8;
    const int b = by_xcd ? ((L >> 3) / nbx) * 8 + (L & 7) : L / nbx;
    const int bxq = by_xcd ? (L >> 3) % nbx : L % nbx;
    if (b >= B) return;
    // pre-pass workspace (fx3d_knn_ws): this cloud's image and norms; the layout is a function of (M, DP)
    constexpr bool use_pre = PRE;  // (a separate instantiation: the kernel without a pre-pass keeps its registers)
    const _Float16 *pre_img = nullptr;
    const float *pre_nup = nullptr, *pre_ndn = nullptr;
    if (use_pre) {
        const KnnPre pre = KnnPre::make(pre_ws, M, DP);
        pre_img = pre.img(b); pre_nup = pre.nup(b); pre_ndn = pre.ndn(b);
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool consumer = wv < kMWaves;
    const int cw = consumer ? wv : wv - kMWaves;   // the consumer wave this wave is paired with
    const int ptid = tid - kMProd;                 // producer thread id (negative for consumers)
    // PRE: the image arrives by direct-to-LDS loads, the "producer" waves are free -- BOTH waves of a pair (they share a SIMD)
    // run the filter, on alternate double pairs of tiles: two waves per SIMD hide each other's LDS latencies and MFMA -> VALU
    // dependencies (a lone consumer wave stalled for more than half of its cycles).  Per query 128 group minima instead of 64
    // (a tighter tau), four lane lists instead of two (the decode is shared by four lanes).
    constexpr bool DUAL = PRE;
    constexpr int LCAP = DUAL ? kMLCap / 2 : kMLCap;  // rows of a lane's mask list
    const int half = consumer ? 0 : 1;
    const int h = lane >> 5, jl = lane & 31;
    const int kk = k + drop;
    const float *xb = x + (size_t)(b / xdiv) * N * D, *yb = y + (size_t)b * M * D;  // (xdiv > 1: candidate slices as virtual clouds share their queries)
    const int q0 = (bxq * kMWaves + cw) * 32;
    const bool wave_active = q0 < N;
    const int qi = q0 + jl;
    const bool vec4y = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(yb) & 15) == 0);
    const bool vec4x = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(xb) & 15) == 0);
    const int nchunk = (M + CH - 1) / CH;
    // filter error per unit of (candidate norm + query norm), scaled units: fp32 accumulation + centring + the oracle's own
    // rounding 8 (4D + 8) u (4x head-room), operand representation 2^-10 (rounded halves) or 2^-18 (2-way split)
    float *nallm = two_norms ? nall + (size_t)nchunk * CH : nullptr;  // [nchunk*CH] norms for the phase-B test
    KNN_PROBE_MARK(0);

    if (tid == 0) {
        *cmax = 0u;
        cmax[2] = 0u;  // F16: a mean far from the middle of its range was seen (scale pass)
        cmax[3] = 0u;  // F16: bulk radius of a cloud centred on its medians (0: not in use)
        const float aq = 8.0f * (float)(4 * D + 8) * 0x1p-24f + (SPLIT ? 0x1p-18f : 0x1.01p-10f);
        // candidate side: + its share of the subnormal floor, + the rounding of n (1 +- A); parked in LDS (cmax[1])
        reinterpret_cast<float *>(cmax)[1] = aq * 1.01f + 0x1p-26f * sqrtf((float)D) + 0x1p-23f;
    }
    float sc = 1.0f;  // F16: power-of-two scale with |sc * c| < 1 for every candidate
    float funit = 1.0f;  // F16: unit of the absolute error terms (see the scale pass)
    // (round 4) behind the pre-pass the kernel's start is ONE global round trip: the first chunk of the image (and the norms) is requested
    // right here, and every lane reads its pieces of its query row and of the centre straight from memory (below) -- no staging of the
    // query rows through LDS, no barrier before the first chunk's.  (It was three dependent round trips -- header, query rows in a loop
    // of load -> LDS store, first chunk -- and two block barriers: 10.4 k cycles.)
    const bool early = F16 && use_pre && vec4x;
    if (F16 && use_pre) {
        // ---- the pre-pass (knn_pre_*_kernel) has the centre, the scale and the largest scaled norm of this cloud
        if (!early) __syncthreads();
        const KnnPre pre = KnnPre::make(pre_ws, M, DP);
        const float *h = pre.hdr(b);
        sc = h[0];
        funit = h[1];
        if (!early && tid < DP) mu[tid] = h[8 + tid];
        if (tid == 0) {  // (the same thread zeroed these words above; they are read after the chunk loop's barriers)
            reinterpret_cast<float *>(cmax)[1] = h[2];
            unsigned int m = h[3] != 0.0f ? 0x7fc00000u : 0u;
            const unsigned int *cp = pre.cmaxp(b);
            for (int p = 0; p < kPreParts; ++p) m = cp[p] > m ? cp[p] : m;  // (NaN pattern > every finite norm)
            *cmax = m;
        }
        if (!early) __syncthreads();
    } else if (F16) {
        // ---- centre and scale: per-dimension MEAN mu (robust against a few far points, unlike the mid-range) and the
        //      largest |c - mu| of the cloud, one coalesced pass (F16 => 16-byte loads are legal).  Distances do not
        //      depend on the origin, the fp16 band does: it grows with |q~|^2 + |c~|^2, so a common offset of a few
        //      standard deviations would flood the lists.  Thread t always sees the same four dimensions when the
        //      block size is a multiple of D/4.  (Any mu is correct; it only has to be the same for all points.)
        __syncthreads();
        const int rq = D / 4;
        const bool centre = (kMThreads % rq) == 0 && rq <= 32;
        float4 lo4 = float4{INFINITY, INFINITY, INFINITY, INFINITY}, hi4 = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        float4 sum4 = float4{0.f, 0.f, 0.f, 0.f};
        bool tnan = false;
        float poison = 0.0f;
        const float4 *c4 = reinterpret_cast<const float4 *>(yb);
        const int total4 = M * (D / 4);
        constexpr int kInFlight = 8;  // 16-byte loads in flight per thread (16 did not help: the pass is bound by the L2, every block reads its whole cloud)
        for (int e0 = tid; e0 < total4; e0 += kInFlight * kMThreads) {
            float4 v[kInFlight];
#pragma unroll
            for (int e = 0; e < kInFlight; ++e) v[e] = c4[e0 + e * kMThreads < total4 ? e0 + e * kMThreads : e0];  // (clamped: same dimensions)
#pragma unroll
            for (int e = 0; e < kInFlight; ++e) {
                // NaN or +-inf coordinates (x * 0 is NaN for them): no scale exists, every query takes the exact path
                poison = __builtin_fmaf(v[e].x, 0.0f, poison); poison = __builtin_fmaf(v[e].y, 0.0f, poison);
                poison = __builtin_fmaf(v[e].z, 0.0f, poison); poison = __builtin_fmaf(v[e].w, 0.0f, poison);
                lo4.x = vmin_f32(lo4.x, v[e].x); lo4.y = vmin_f32(lo4.y, v[e].y); lo4.z = vmin_f32(lo4.z, v[e].z); lo4.w = vmin_f32(lo4.w, v[e].w);
                hi4.x = vmax_f32(hi4.x, v[e].x); hi4.y = vmax_f32(hi4.y, v[e].y); hi4.z = vmax_f32(hi4.z, v[e].z); hi4.w = vmax_f32(hi4.w, v[e].w);
                if (e0 + e * kMThreads < total4) {  // (the clamped duplicates must not enter the mean)
                    sum4.x = sum4.x + v[e].x; sum4.y = sum4.y + v[e].y; sum4.z = sum4.z + v[e].z; sum4.w = sum4.w + v[e].w;
                }
            }
        }
        tnan = poison != poison;
        const bool anynan = __syncthreads_or(tnan) != 0;
        float *red = sm;  // [kMThreads / 64][32][12] scratch in the (still unused) chunk buffers
        if (centre) {
            for (int m = rq; m < 64; m <<= 1) {  // lanes with equal lane % rq hold the same dimensions
                lo4.x = fminf(lo4.x, __shfl_xor(lo4.x, m, 64)); lo4.y = fminf(lo4.y, __shfl_xor(lo4.y, m, 64));
                lo4.z = fminf(lo4.z, __shfl_xor(lo4.z, m, 64)); lo4.w = fminf(lo4.w, __shfl_xor(lo4.w, m, 64));
                hi4.x = fmaxf(hi4.x, __shfl_xor(hi4.x, m, 64)); hi4.y = fmaxf(hi4.y, __shfl_xor(hi4.y, m, 64));
                hi4.z = fmaxf(hi4.z, __shfl_xor(hi4.z, m, 64)); hi4.w = fmaxf(hi4.w, __shfl_xor(hi4.w, m, 64));
                sum4.x = sum4.x + __shfl_xor(sum4.x, m, 64); sum4.y = sum4.y + __shfl_xor(sum4.y, m, 64);
                sum4.z = sum4.z + __shfl_xor(sum4.z, m, 64); sum4.w = sum4.w + __shfl_xor(sum4.w, m, 64);
            }
            if (lane < rq) {
                float *r8 = red + (size_t)(wv * 32 + lane) * 12;
                r8[0] = lo4.x; r8[1] = lo4.y; r8[2] = lo4.z; r8[3] = lo4.w;
                r8[4] = hi4.x; r8[5] = hi4.y; r8[6] = hi4.z; r8[7] = hi4.w;
                r8[8] = sum4.x; r8[9] = sum4.y; r8[10] = sum4.z; r8[11] = sum4.w;
            }
        }
        __syncthreads();
        float amax = 0.0f;
        if (centre) {
            if (tid < rq) {
                float lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                float sm4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int w = 0; w < kMThreads / 64; ++w) {
                    const float *r8 = red + (size_t)(w * 32 + tid) * 12;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { lo[c] = fminf(lo[c], r8[c]); hi[c] = fmaxf(hi[c], r8[4 + c]); sm4[c] = sm4[c] + r8[8 + c]; }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float m0 = sm4[c] / (float)M;
                    m0 = fminf(fmaxf(m0, lo[c]), hi[c]);  // (rounding of the sum cannot leave the range)
                    mu[4 * tid + c] = m0;
                    amax = fmaxf(amax, fmaxf(hi[c] - m0, m0 - lo[c]));
                    // a mean far from the middle of its range: skewed data or a few far points (checked below on a sample)
                    if (fabsf(m0 - 0.5f * (lo[c] + hi[c])) > 0.25f * (hi[c] - lo[c])) cmax[2] = 1u;
                }
            } else if (tid < DP / 4) {
                mu[4 * tid] = 0.0f; mu[4 * tid + 1] = 0.0f; mu[4 * tid + 2] = 0.0f; mu[4 * tid + 3] = 0.0f;
            }
        } else {
            if (tid < DP / 4) { mu[4 * tid] = 0.0f; mu[4 * tid + 1] = 0.0f; mu[4 * tid + 2] = 0.0f; mu[4 * tid + 3] = 0.0f; }
            if (tid < total4)  // (threads without an element hold +-inf)
                amax = fmaxf(fmaxf(fmaxf(fabsf(lo4.x), fabsf(hi4.x)), fmaxf(fabsf(lo4.y), fabsf(hi4.y))),
                             fmaxf(fmaxf(fabsf(lo4.z), fabsf(hi4.z)), fmaxf(fabsf(lo4.w), fabsf(hi4.w))));
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m, 64));
        if (lane == 0) atomicMax(cmax, anynan ? 0x7fc00000u : __builtin_bit_cast(unsigned int, amax));
        __syncthreads();
        if (centre && cmax[2] != 0u && !anynan) {  // (block-uniform)
            // ---- robust centre.  A few points far from the bulk pull the mean towards them (one point 10^6 x the extent away
            //      among 1024: by 10^3 extents), every query then sits |q~| >> extent from the centre and its band ~ 2^-10 |q~|^2
            //      swallows the whole cloud (1.7 ms instead of 80 us).  Per-dimension MEDIAN and quartiles of 16 rows spread
            //      over the cloud; when a mean lies more than 8 interquartile ranges from the median, every dimension is
            //      centred on its median instead (any centre is correct) and the extent grows by the largest shift (an upper
            //      bound, no second pass).  Skewed but clean data (one-sided features) keep their means.
            if (wv == 0) {
                float shiftmax = 0.0f, iqr2 = 0.0f;
                bool sw = false;
                float medv[DP / 64 > 0 ? DP / 64 : 1];
#pragma unroll
                for (int t = 0; t < (DP + 63) / 64; ++t) {
                    const int d = lane + 64 * t;
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = yb[(size_t)((long long)i * M / 16) * D + (d < D ? d : 0)];
                    float med = v[0], q1 = v[0], q3 = v[0];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        int rk = 0;
#pragma unroll
                        for (int j = 0; j < 16; ++j) rk += (v[j] < v[i] || (v[j] == v[i] && j < i)) ? 1 : 0;
                        med = rk == 8 ? v[i] : med; q1 = rk == 4 ? v[i] : q1; q3 = rk == 12 ? v[i] : q3;
                    }
                    const float shift = d < D ? fabsf(mu[d < D ? d : 0] - med) : 0.0f;
                    sw = sw || (shift > 8.0f * (q3 - q1));
                    shiftmax = fmaxf(shiftmax, shift);
                    if (d < D) iqr2 = __builtin_fmaf(q3 - q1, q3 - q1, iqr2);
                    medv[t] = med;
                }
                if (__ballot(sw) != 0ull) {
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) {
                        shiftmax = fmaxf(shiftmax, __shfl_xor(shiftmax, m, 64));
                        iqr2 = iqr2 + __shfl_xor(iqr2, m, 64);
                    }
#pragma unroll
                    for (int t = 0; t < (DP + 63) / 64; ++t)
                        if (lane + 64 * t < D) mu[lane + 64 * t] = medv[t];
                    if (lane == 0) {
                        *cmax = __builtin_bit_cast(unsigned int, __builtin_bit_cast(float, *cmax) + shiftmax);
                        reinterpret_cast<float *>(cmax)[3] = sqrtf(iqr2);  // the bulk's radius (unscaled): the unit of the absolute error terms below
                    }
                }
            }
            __syncthreads();
        }
        const float cinf = __builtin_bit_cast(float, *cmax);
        if (cinf > 1.0e-30f && cinf < 1.0e30f) {
            int e;
            (void)frexpf(cinf * 1.000001f, &e);  // = m 2^e, m in [0.5, 1)
            // |sc (c - mu)| < 2^10: ten binades above 1 so that a bulk far smaller than the largest |c - mu| (a few far
            // points) still sits in fp16's normal range; queries up to 30 x the cloud's extent stay below 6e4
            sc = ldexpf(1.0f, 10 - e);
        }
        {
            // unit s of the absolute (fp16 subnormal) error terms: |x| <= (x^2 / s + s) / 2 for any s > 0 turns the linear bound
            // 2^-24 sqrt(D) (|q~| + |c~| / 2) into shares of the squared norms.  s = 1 unless the cloud was centred on its medians
            // because of far points: then the bulk may sit far below 1 in scaled units, and with s = 1 the constant term
            // 2^-23 sqrt(D) would dwarf its squared distances (the whole cloud inside every band).
            const float rad = reinterpret_cast<const float *>(cmax)[3];
            funit = rad > 0.0f ? fminf(1.0f, fmaxf(sc * rad, 0x1p-12f)) : 1.0f;
        }
        __syncthreads();
        if (tid == 0 && funit < 1.0f) {
            const float aq = 8.0f * (float)(4 * D + 8) * 0x1p-24f + (SPLIT ? 0x1p-18f : 0x1.01p-10f);
            reinterpret_cast<float *>(cmax)[1] = aq * 1.01f + 0x1p-26f * sqrtf((float)D) / funit + 0x1p-23f;
        }
        // from here on: bits of the largest SCALED squared norm.  A cloud whose extent lets exact Float32 distances overflow
        // (D (61 cinf)^2 >= 3.4e38 for usable queries) is handled like a non-finite one: its +Inf ties are ordered by index in
        // the oracle, which only the brute-force merge reproduces.
        if (tid == 0) *cmax = (anynan || !(cinf < 1.0e15f)) ? 0x7fc00000u : 0u;
    }

    // ---- B operand: the wave's 32 query rows, staged through LDS (coalesced), then -2 q in registers ------------
    float4 a[NT];            // f32 filter: -2 q, this lane's half of the permuted reduction dimension
    kh8 ah[NB16], al[NB16];  // fp16 filter: hi / lo halves of -2 sc q, 8 dimensions per K block and half-wave
    float qn = 0.0f;
    bool qok = true;
    if (early) {
        // K block bb covers dimensions 16 bb + 8 h + [0, 8) in half-wave h: hi halves of -2 sc (q - mu), every piece read from memory
        // (16 bytes of the lane's own row -- rows beyond N read row N - 1 and are never used -- and of the pre-pass header's centre,
        // zero beyond D); all loads in flight together
        const float *qg = xb + (size_t)(qi < N ? qi : N - 1) * D;
        const float *mg = KnnPre::make(pre_ws, M, DP).hdr(b) + 8;
        float4 qv[NB16][2], mv[NB16][2];
#pragma unroll
        for (int bb = 0; bb < NB16; ++bb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int d0 = 16 * bb + 8 * h + 4 * u;
                qv[bb][u] = *reinterpret_cast<const float4 *>(qg + (d0 < D ? d0 : 0));
                mv[bb][u] = *reinterpret_cast<const float4 *>(mg + d0);
            }
        // the first chunk's direct-to-LDS loads (and the norms) go out BEHIND this wave's own loads: the memory counter retires in
        // order, so requested first they made the producer waves wait for the whole chunk before they could touch their header
        // values (their operands were ready 3 k cycles after the consumers', and the block's first barrier with them)
        if (!consumer) {
            if (DK <= 2 && regstage)
                for (int c = 1; c < nchunk; ++c)
                    knn_pre_stage_norms(pre_nup, pre_ndn, c * CH, CH, nall + (size_t)c * CH, nallm ? nallm + (size_t)c * CH : nullptr, wv - kMWaves, lane);
            knn_pre_stage_chunk<DK, false>(pre_img, pre_nup, pre_ndn, 0, CH, sm, nall, nallm, true, wv - kMWaves, lane);
        }
        float amax = 0.0f;
#pragma unroll
        for (int bb = 0; bb < NB16; ++bb) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool in = 16 * bb + 8 * h + 4 * u < D;
                const float v[4] = {qv[bb][u].x, qv[bb][u].y, qv[bb][u].z, qv[bb][u].w}, m4[4] = {mv[bb][u].x, mv[bb][u].y, mv[bb][u].z, mv[bb][u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qs_ = in ? (v[e] - m4[e]) * sc : 0.0f;  // centred like the candidates
                    qn = qn + qs_ * qs_;
                    const float av = -2.0f * qs_;
                    amax = fmaxf(amax, fabsf(av));
                    ah[bb][4 * u + e] = (_Float16)av;
                }
            }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        qok = amax < 6.0e4f;  // inside the fp16 range (false for NaN too)
        qn = qn + __shfl_xor(qn, 32, 64);
    } else if (consumer || DUAL) {  // (DUAL: both waves of a pair stage the same rows -- identical values -- and derive the same operands)
        float *qs = sm + (size_t)cw * 32 * RS;
        const int nrow = wave_active ? ((N - q0) < 32 ? (N - q0) : 32) : 0;
        const float *src = xb + (size_t)q0 * D;
        if (vec4x) {
            const int rq = D / 4;
            for (int e = lane; e < nrow * rq; e += 64) {
                const int row = e / rq, c4 = e - row * rq;
                *reinterpret_cast<float4 *>(qs + (size_t)row * RS + 4 * c4) = reinterpret_cast<const float4 *>(src)[e];
            }
        } else {
            for (int e = lane; e < nrow * D; e += 64) {
                const int row = e / D, d = e - row * D;
                qs[(size_t)row * RS + d] = src[e];
            }
        }
        for (int e = lane; e < 32 * DP; e += 64) {  // zero padding: columns >= D, rows >= nrow
            const int row = e / DP, d = e - row * DP;
            if (row >= nrow || d >= D) qs[(size_t)row * RS + d] = 0.0f;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (F16) {
            // K block bb covers dimensions 16 bb + 8 h + [0, 8) in half-wave h: hi and lo halves of -2 sc q
            float amax = 0.0f;
#pragma unroll
            for (int bb = 0; bb < NB16; ++bb) {
                const float *qr = qs + (size_t)jl * RS + 16 * bb + 8 * h;
                const float4 v0 = *reinterpret_cast<const float4 *>(qr), v1 = *reinterpret_cast<const float4 *>(qr + 4);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float qs_ = (v[e] - mu[16 * bb + 8 * h + e]) * sc;  // centred like the candidates
                    qn = qn + qs_ * qs_;
                    const float av = -2.0f * qs_;
                    amax = fmaxf(amax, fabsf(av));
                    const _Float16 hh_ = (_Float16)av;
                    ah[bb][e] = hh_;
                    if (SPLIT) al[bb][e] = (_Float16)(av - (float)hh_);
                }
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            qok = amax < 6.0e4f;  // inside the fp16 range (false for NaN too)
        } else {
            const float *qr = qs + (size_t)jl * RS + h * (DP / 2);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 v = *reinterpret_cast<const float4 *>(qr + 4 * t);
                qn = qn + v.x * v.x;
                qn = qn + v.y * v.y;
                qn = qn + v.z * v.z;
                qn = qn + v.w * v.w;
                a[t] = float4{-2.0f * v.x, -2.0f * v.y, -2.0f * v.z, -2.0f * v.w};
            }
        }
        qn = qn + __shfl_xor(qn, 32, 64);
    }
    if (!early) __syncthreads();
    KNN_PROBE_MARK(1);

    // ---- chunk schedule: phase A walks the chunks forwards, phase B backwards (its first chunk is resident) ----
    const int nstep = 2 * nchunk;
    constexpr int kMUnits = SPLIT ? kMUnitsSplit : kMUnitsSingle;
    float4 preg[kMUnits][2];  // F16 producers: the chunk after next, loaded one step ahead
    float pmax = 0.0f;        // F16 producers: largest scaled norm seen
    bool pnan = false;
    int stage_ev = 0;                    // F16 producers: staging events done (chunks 0..n-1, n-2..0)
    const int nevents = 2 * nchunk - 1;
    if (F16 && use_pre) {
        if (!consumer && early) {
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of the first chunk (requested at the kernel's start) have landed
            __builtin_amdgcn_wave_barrier();
        } else if (!consumer) {
            // (regstage: the image chunks of the loop below come through registers; the norms of ALL chunks arrive here, once)
            if (DK <= 2 && regstage)
                for (int c = 1; c < nchunk; ++c)
                    knn_pre_stage_norms(pre_nup, pre_ndn, c * CH, CH, nall + (size_t)c * CH, nallm ? nallm + (size_t)c * CH : nullptr, wv - kMWaves, lane);
            knn_pre_stage_chunk<DK>(pre_img, pre_nup, pre_ndn, 0, CH, sm, nall, nallm, true, wv - kMWaves, lane);
        }
    } else if (F16) {
        if (!consumer) {
            knn_f16_load_chunk<DK, kMUnits>(yb, D, 0, M < CH ? M : CH, CH, ptid, preg);
            knn_f16_store_chunk<DK, SPLIT, kMUnits>(sm, CH, M < CH ? M : CH, sc, mu, ptid, preg, nall, nallm, reinterpret_cast<const float *>(cmax + 1), pmax, pnan);
            if (nchunk == 1) {
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, m, 64));
                const bool anyn = __ballot(pnan) != 0;
                if (lane == 0) atomicMax(cmax, anyn ? 0x7fc00000u : __builtin_bit_cast(unsigned int, pmax));
            }
            if (nevents > 1) {
                const int c1 = 1 < nchunk ? 1 : 2 * nchunk - 3;
                knn_f16_load_chunk<DK, kMUnits>(yb, D, c1 * CH, (M - c1 * CH) < CH ? (M - c1 * CH) : CH, CH, ptid, preg);
            }
            stage_ev = 1;
        }
    } else {
        if (D < DP || !vec4y) {  // padding columns must read as zeros; the direct loads never touch them
            for (int e = tid; e < 2 * buf_floats / 4; e += kMThreads)
                reinterpret_cast<float4 *>(sm)[e] = float4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
        }
        if (!consumer) {
            const int cn = M < CH ? M : CH;
            knn_stage_chunk<DK>(yb, D, 0, cn, CH, sm, keep_norms ? nall : sm + (size_t)CH * DP, cmax, true, true, vec4y,
                                wv - kMWaves, lane);
        }
    }
    __syncthreads();
    KNN_PROBE_MARK(2);

    float mn[32];  // group minima: [r] even tiles, [16 + r] odd tiles -> 64 groups per query
#pragma unroll
    for (int r = 0; r < 32; ++r) mn[r] = INFINITY;
    float thr = 0.0f;
    int cnt = 0, totb = 0;  // list words with a survivor; survivors seen by phase B (all of them: the overflow flag tells when words were lost)
    int *mylist = lists + (DUAL ? wv * LCAP : cw * kMLCap) * 64 + lane;  // entry e at mylist[e * 64]

    int cur = 0;  // buffer holding the chunk of this step
    for (int step = 0; step < nstep; ++step) {
        const int phase = step >= nchunk ? 1 : 0;
        const int ci = phase ? nstep - 1 - step : step;
        const int j0 = ci * CH;
        const int cn = (M - j0) < CH ? (M - j0) : CH;
        const int cn_pad = (cn + 63) & ~63;
        const int nstep1 = step + 1;
        const int ci_next = nstep1 >= nchunk ? nstep - 1 - nstep1 : nstep1;
        const bool stage_next = nstep1 < nstep && ci_next != ci;
        // regstage (round 4): the next chunk of the image through REGISTERS -- every thread requests its 16-byte pieces now and writes
        // them to the other buffer after its share of the filter (no VALU either way).  The direct-to-LDS loads moved ~16 bytes per
        // cycle and CU and did not overlap the compute (a 256-row step = its compute, 3.0 k cycles, + its staging, 2.4 k); loads to
        // registers run at the L1's 64 bytes per cycle.
        constexpr bool REGST = PRE && DK <= 2;  // (D > 64: eight pieces per thread -- 32 registers the kernel does not have)
        constexpr int NCR = REGST ? (PPI / 2 > 0 ? PPI / 2 : 1) : 1;  // 16-byte pieces per thread and 256-row chunk
        f32x4v creg[NCR];
        if (REGST && stage_next && regstage) {
            const int j0n = ci_next * CH;
            constexpr int RPBc = PPI >= 16 ? 1 : 16 / PPI;
#pragma unroll
            for (int i = 0; i < NCR; ++i) {
                const int S = tid + i * kMThreads;
                const int row = S / PPI, pos = S & (PPI - 1);
                const int c = (pos - row / RPBc) & (PPI - 1);
                if (S < CH * PPI) creg[i] = *reinterpret_cast<const f32x4v *>(pre_img + ((size_t)(j0n + row) * PPI + c) * 8);
            }
        } else if (DUAL && !consumer && stage_next) {  // the next chunk's direct loads first: they land while this wave computes
            const int j0n = ci_next * CH;
            knn_pre_stage_chunk<DK, false>(pre_img, pre_nup, pre_ndn, j0n, CH, sm + (size_t)(1 - cur) * buf_floats, nall + (size_t)ci_next * CH,
                                           nallm ? nallm + (size_t)ci_next * CH : nullptr, nstep1 < nchunk, wv - kMWaves, lane);
        }
        if (consumer || DUAL) {
            if (wave_active) {
                const float *cand = sm + (size_t)cur * buf_floats;
                const float *cnorm = keep_norms ? (phase && nallm ? nallm : nall) + (size_t)ci * CH : cand + (size_t)CH * DP;
                const int npair = cn_pad / 64;
                const int tile0 = j0 / 32;
                int pr_first = 0;
                // phase B of the fp16 filters accumulates on n_c - thr: the sign of the result is the test.  (The Float32 GEMM keeps the
                // compare: its staging leaves the rows beyond the cloud's end unwritten -- norm +inf, stale pieces -- and inf + NaN has
                // no usable sign; the fp16 images are zero there.)
                const float tsub = (F16 && phase) ? thr : 0.0f;
                if (F16 && !SPLIT && PRE) {  // (without the pre-pass the producers' staging registers leave no room: 68 spills)
                    // single-piece fp16 filter: TWO pairs of tiles per iteration -- the second pair's operand fetches and MFMAs are
                    // issued before the first pair's results are folded, so the fold (VALU) of one overlaps the matrix work of the
                    // other and one round of LDS latency serves four tiles (a lone consumer wave per SIMD hides nothing otherwise).
                    // One instantiation per phase (round 4): phase A's accumulators ARE the norm loads' destinations (no VALU) and two
                    // tiles fold per v_min3 straight from the MFMA registers; phase B starts them at n_c - thr (one v_sub each).
                    constexpr int RPB2 = PPI >= 16 ? 1 : 16 / PPI;
                    auto run4 = [&](auto phc) {
                        constexpr bool PHB = decltype(phc)::value;
                        for (; pr_first + 1 < npair; pr_first += 2) {
                            if (DUAL && ((pr_first >> 1) & 1) != half) continue;  // the pair's waves take alternate double pairs
                            f32x16v accs[4];
                            kh8 ops[4][NB16];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {  // q = 2 * (pair) + (tile of the pair)
                                const int rbase = (pr_first + (q >> 1)) * 64 + 32 * (q & 1);
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const float4 n0 = *reinterpret_cast<const float4 *>(cnorm + rbase + 8 * g + 4 * h);
                                    if (PHB) {
                                        accs[q][4 * g] = n0.x - thr; accs[q][4 * g + 1] = n0.y - thr; accs[q][4 * g + 2] = n0.z - thr; accs[q][4 * g + 3] = n0.w - thr;
                                    } else {
                                        accs[q][4 * g] = n0.x; accs[q][4 * g + 1] = n0.y; accs[q][4 * g + 2] = n0.z; accs[q][4 * g + 3] = n0.w;
                                    }
                                }
                                const float *cq = cand + (size_t)(rbase + jl) * RSI;
#pragma unroll
                                for (int bb = 0; bb < NB16; ++bb)
                                    ops[q][bb] = *reinterpret_cast<const kh8 *>(cq + ((2 * bb + h + jl / RPB2) & (PPI - 1)) * 4);
                            }
#pragma unroll
                            for (int bb = 0; bb < NB16; ++bb) {  // four independent accumulators in turn: no MFMA waits for its predecessor
#pragma unroll
                                for (int q = 0; q < 4; ++q) accs[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ops[q][bb], ah[bb], accs[q], 0, 0, 0);
                            }
                            // (round 4) the fold / mask work runs at raised wave priority: the pair's other wave is then the one whose MFMAs are
                            // in the pipe while this one issues VALU -- same-box A/B 51.95 -> 50.7 us (the reverse, priority on the MFMA block, costs
                            // 12 us: the issuing wave hogs the slots its partner's fold needs; static priorities by wave role: no effect)
                            __builtin_amdgcn_s_setprio(1);
                            if (!PHB) {
                                // (any partition of the tiles into the 32 groups of a lane will do; the two accumulators issued last are
                                //  read 20+ issue slots after their MFMAs: knn_f16_d3_kernel's order)
                                KNN_MFMA_SETTLE4(accs[0], accs[1], accs[2], accs[3]);
#pragma unroll
                                for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[r]) : "v"(accs[0][r]), "v"(accs[1][r]));
#pragma unroll
                                for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[16 + r]) : "v"(accs[2][r]), "v"(accs[3][r]));
                            } else {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    unsigned int m = 0;
#pragma unroll
                                    for (int i = 0; i < 16; ++i) {  // the sign of F - thr is the test: one v_alignbit per row shifts it in (row r at bit r)
                                        const float av = accs[q][15 - i];
                                        m = __builtin_amdgcn_alignbit(m, __builtin_bit_cast(unsigned int, av), 31);
                                    }
                                    const int pp = cnt < LCAP - 1 ? cnt : LCAP - 1;
                                    mylist[pp * 64] = (int)((unsigned int)(tile0 + pr_first * 2 + q) << 16 | m);
                                    cnt += m != 0 ? 1 : 0;
                                    totb += __builtin_popcount(m);
                                }
                            }
                            __builtin_amdgcn_s_setprio(0);
                        }
                    };
                    if (phase == 0) run4(std::false_type{});
                    else run4(std::true_type{});
                }
                for (int pr = (DUAL && half) ? npair : pr_first; pr < npair; ++pr) {  // (DUAL: a last lone pair goes to the first wave)
                    // rows pr*64 + jl and + 32 share (row mod PPR) = jl mod PPR: one rotated offset per fetch
                    const float *c0 = cand + (size_t)(pr * 64 + jl) * RSI, *c1 = c0 + (size_t)32 * RSI;
                    // accumulators start at the candidate norms: register r of half h is row (r&3) + 8(r>>2) + 4h
                    f32x16v acc0, acc1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 n0 = *reinterpret_cast<const float4 *>(cnorm + pr * 64 + 8 * g + 4 * h);
                        const float4 n1 = *reinterpret_cast<const float4 *>(cnorm + pr * 64 + 32 + 8 * g + 4 * h);
                        acc0[4 * g] = n0.x - tsub; acc0[4 * g + 1] = n0.y - tsub; acc0[4 * g + 2] = n0.z - tsub; acc0[4 * g + 3] = n0.w - tsub;
                        acc1[4 * g] = n1.x - tsub; acc1[4 * g + 1] = n1.y - tsub; acc1[4 * g + 2] = n1.z - tsub; acc1[4 * g + 3] = n1.w - tsub;
                    }
                    if (F16 && SPLIT) {
                        // A = candidate pieces (rows), B = query pieces (columns); hi*hi + lo*hi + hi*lo
                        kh8 h0[NB16], l0[NB16], h1[NB16], l1[NB16];
#pragma unroll
                        for (int bb = 0; bb < NB16; ++bb) {
                            const int ph = ((2 * bb + h + jl) & (PPR - 1)) * 4, pl = ((PPR / 2 + 2 * bb + h + jl) & (PPR - 1)) * 4;
                            h0[bb] = *reinterpret_cast<const kh8 *>(c0 + ph);
                            l0[bb] = *reinterpret_cast<const kh8 *>(c0 + pl);
                            h1[bb] = *reinterpret_cast<const kh8 *>(c1 + ph);
                            l1[bb] = *reinterpret_cast<const kh8 *>(c1 + pl);
                        }
#pragma unroll
                        for (int bb = 0; bb < NB16; ++bb) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0[bb], ah[bb], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1[bb], ah[bb], acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l0[bb], ah[bb], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(l1[bb], ah[bb], acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0[bb], al[bb], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1[bb], al[bb], acc1, 0, 0, 0);
                        }
                    } else if (F16) {
                        // single-piece filter: one MFMA per K block and tile on the rounded (hi) halves
                        constexpr int RPB = PPI >= 16 ? 1 : 16 / PPI;
                        kh8 h0[NB16], h1[NB16];
#pragma unroll
                        for (int bb = 0; bb < NB16; ++bb) {
                            const int ph = ((2 * bb + h + jl / RPB) & (PPI - 1)) * 4;
                            h0[bb] = *reinterpret_cast<const kh8 *>(c0 + ph);
                            h1[bb] = *reinterpret_cast<const kh8 *>(c1 + ph);
                        }
#pragma unroll
                        for (int bb = 0; bb < NB16; ++bb) {
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0[bb], ah[bb], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1[bb], ah[bb], acc1, 0, 0, 0);
                        }
                    } else {
                        float4 b0[NT], b1[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const int po = ((h * NT + t + jl) & (PPR - 1)) * 4;
                            b0[t] = *reinterpret_cast<const float4 *>(c0 + po);
                            b1[t] = *reinterpret_cast<const float4 *>(c1 + po);
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) {  // A = candidates (rows), B = queries (columns)
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t].x, a[t].x, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t].x, a[t].x, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t].y, a[t].y, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t].y, a[t].y, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t].z, a[t].z, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t].z, a[t].z, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[t].w, a[t].w, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[t].w, a[t].w, acc1, 0, 0, 0);
                        }
                    }
                    if (phase == 0) {
                        KNN_MFMA_SETTLE2(acc0, acc1);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mn[r] = vmin_acc(mn[r], acc0[r]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mn[16 + r] = vmin_acc(mn[16 + r], acc1[r]);
                    } else {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            // one word per tile: (tile index << 16) | mask of the rows with F <= thr; stored at the
                            // list head unconditionally, the head advances when the mask is not empty
                            unsigned int m = 0;
#pragma unroll
                            for (int i = 0; i < 16; ++i) {  // (ascending i: a descending unrolled loop over the vector's elements read element 0 every time)
                                const float av = tt ? acc1[15 - i] : acc0[15 - i];
                                if (F16) m = __builtin_amdgcn_alignbit(m, __builtin_bit_cast(unsigned int, av), 31);
                                else m |= (av <= thr) ? (1u << (15 - i)) : 0u;
                            }
                            const int pp = cnt < LCAP - 1 ? cnt : LCAP - 1;
                            mylist[pp * 64] = (int)((unsigned int)(tile0 + pr * 2 + tt) << 16 | m);
                            cnt += m != 0 ? 1 : 0;
                            totb += __builtin_popcount(m);
                        }
                    }
                }
            }
        } else if (stage_next) {
            const int j0n = ci_next * CH;
            const int cnn = (M - j0n) < CH ? (M - j0n) : CH;
            float *img = sm + (size_t)(1 - cur) * buf_floats;
            if (F16 && use_pre) {  // (not reached when DUAL: kept for a PRE build without it)
                const bool phase_a = nstep1 < nchunk;  // (the norms of all chunks stay in LDS: phase B brings the image only)
                knn_pre_stage_chunk<DK>(pre_img, pre_nup, pre_ndn, j0n, CH, img, nall + (size_t)ci_next * CH,
                                        nallm ? nallm + (size_t)ci_next * CH : nullptr, phase_a, wv - kMWaves, lane);
            } else if (F16) {
                // the registers hold chunk ci_next (loaded one step ago); then fetch the chunk after it
                knn_f16_store_chunk<DK, SPLIT, kMUnits>(img, CH, cnn, sc, mu, ptid, preg, stage_ev < nchunk ? nall + (size_t)stage_ev * CH : nullptr,
                                        stage_ev < nchunk && nallm ? nallm + (size_t)stage_ev * CH : nullptr,
                                        reinterpret_cast<const float *>(cmax + 1), pmax, pnan);
                if (stage_ev == nchunk - 1) {  // last phase-A chunk: publish this wave's maximum norm
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, m, 64));
                    const bool anyn = __ballot(pnan) != 0;
                    if (lane == 0) atomicMax(cmax, anyn ? 0x7fc00000u : __builtin_bit_cast(unsigned int, pmax));
                }
                ++stage_ev;
                if (stage_ev < nevents) {
                    const int cnx = stage_ev < nchunk ? stage_ev : 2 * nchunk - 2 - stage_ev;
                    knn_f16_load_chunk<DK, kMUnits>(yb, D, cnx * CH, (M - cnx * CH) < CH ? (M - cnx * CH) : CH, CH, ptid, preg);
                }
            } else {
                const bool phase_a = nstep1 < nchunk;
                knn_stage_chunk<DK>(yb, D, j0n, cnn, CH, img, keep_norms ? nall + (size_t)ci_next * CH : img + (size_t)CH * DP,
                                    cmax, phase_a, phase_a || !keep_norms, vec4y, wv - kMWaves, lane);
            }
        }
        if (REGST && stage_next && regstage) {
            float *nimg = sm + (size_t)(1 - cur) * buf_floats;
#pragma unroll
            for (int i = 0; i < NCR; ++i) {
                const int S = tid + i * kMThreads;
                if (S < CH * PPI) *reinterpret_cast<f32x4v *>(nimg + (size_t)S * 4) = creg[i];
            }
        } else if (DUAL && !consumer && stage_next) {
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of the next chunk have landed
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        KNN_PROBE_MARK(3 + step);
        if (stage_next) cur = 1 - cur;
        if (step == nchunk - 1 && (consumer || DUAL)) {
            // ---- tau: kk-th smallest of the 64 group minima of every query (32 in this lane, 32 in its partner) ----
            const float c2 = __builtin_bit_cast(float, *cmax);
            float tau;
            if (DUAL && kk <= 24 && M >= 128) {
                // 128 group minima per query in the layout of the D = 3 kernel (32 per lane x two half-lanes x the pair's two waves):
                // its reduced selection (round 4; the sort of all 32 + two 32-value merges below were 5 us of this kernel)
                tau = knn_tau_8of16<kMWaves>(mn, reinterpret_cast<float *>(lists), wv, jl, h, kk);
                __syncthreads();  // the exchange space becomes the lane lists
            } else {
            k3_sort_regs<32>(mn);
            float oth[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) oth[r] = __shfl_xor(mn[31 - r], 32, 64);
#pragma unroll
            for (int r = 0; r < 32; ++r)  // half 0 keeps the 32 smallest of the 64 (a bitonic sequence)
                mn[r] = h ? vmax_f32(mn[r], oth[r]) : vmin_f32(mn[r], oth[r]);
#pragma unroll
            for (int j = 16; j > 0; j >>= 1) {  // one bitonic merge sorts it ascending
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int l = i ^ j;
                    if (l > i) {
                        const float lo = vmin_f32(mn[i], mn[l]), hi = vmax_f32(mn[i], mn[l]);
                        mn[i] = lo;
                        mn[l] = hi;
                    }
                }
            }
            if (DUAL) {
                // the other wave of the pair holds the minima of the other tiles: the 32 smallest of the 128 through LDS (the
                // lane lists are not in use yet), one more bitonic merge
                float *xch = reinterpret_cast<float *>(lists);  // [2 kMWaves][32][33]
                if (h == 0) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) xch[(wv * 32 + jl) * 33 + r] = mn[r];
                }
                __syncthreads();
                {
                    const float *po = xch + (((wv + kMWaves) % (2 * kMWaves)) * 32 + jl) * 33;
#pragma unroll
                    for (int r = 0; r < 32; ++r) mn[r] = vmin_f32(mn[r], po[31 - r]);
#pragma unroll
                    for (int j = 16; j > 0; j >>= 1) {
#pragma unroll
                        for (int i = 0; i < 32; ++i) {
                            const int l = i ^ j;
                            if (l > i) {
                                const float lo = vmin_f32(mn[i], mn[l]), hi = vmax_f32(mn[i], mn[l]);
                                mn[i] = lo;
                                mn[l] = hi;
                            }
                        }
                    }
                }
                __syncthreads();  // the exchange space becomes the lane lists
            }
            float val = mn[0];
#pragma unroll
            for (int r = 1; r < 32; ++r) val = (kk - 1) == r ? mn[r] : val;
            tau = __shfl(val, jl, 64);  // kk <= 32: always among the 32 smallest (half 0)
            }
            float eps;
            if (F16) {
                // scaled units (c~ = sc c, |c~| < 1; qn = |sc q|^2): split representation 3 2^-22 |a~||c~|, fp32
                // accumulation of the 3D exact products (3D+1) u, the oracle's own (D+2) u, fp16 underflow floor
                // single piece: the rounded operands differ by 2^-11 relative each, sum |c~_d a_d| <= 2 |c~||q~| <= qn + c2
                // |F^ - (sc^2 d_oracle - qn)| <= A (n_c + qn) + floor for candidate c with scaled norm n_c (A = acoef_q).
                // two_norms: the candidate's share A n_c is already inside the norms (upwards in phase A, downwards
                // in phase B), the query keeps B_q = A qn + floor_q: a far candidate no longer widens everybody's
                // band.  Otherwise n_c <= c2 for all of them.
                const float acoef_q = 8.0f * (float)(4 * D + 8) * 0x1p-24f + (SPLIT ? 0x1p-18f : 0x1.01p-10f);
                const float floor_q = 0x1p-24f * sqrtf((float)D) * (qn / funit + 2.0f * funit);
                eps = two_norms ? acoef_q * qn + floor_q : acoef_q * (qn + c2) + floor_q + 0x1p-26f * sqrtf((float)D) * c2 / funit;
                eps = (qok && c2 == c2) ? eps : INFINITY;  // c2 is NaN for a non-finite / overflow-prone cloud (scale pass)
            } else {
                eps = (8.0f * (float)(D + 4) * 0x1p-24f) * (qn + c2);
                eps = qn + c2 < 1.0e38f ? eps : INFINITY;  // (|q| + |c|)^2 <= 2 (qn + c2): no exact distance overflows
            }
            thr = tau + 2.0f * eps;  // NaN / inf => slow path below
            // (round 4) phase B starts the accumulators at n_c - thr instead of n_c and keeps the SIGN of the result (one v_alignbit per
            // row where the compare cost v_cmp + v_cndmask + v_or and two wait states).  The accumulation now carries thr through its
            // D + 1 roundings: against the compare form the result moves by at most (D + 1) u (2 n_c + |thr| + 2 sum |products|)
            // (65 u (4 n_c + 2 qn + |thr|) at D = 64), u = 2^-24.  The candidate's and the query's own shares sit inside the budget the
            // filter already grants them (8 (4 D + 8) u = 2112 u each at D = 64, of which the compare form uses ~130 u); the threshold's share,
            // (K + 1) u |thr| with K = D products (3 D for the split filter), is added here four times over -- at D = 64 2^-16 |thr|,
            // 1/64 of the band of a candidate at the boundary (its norm is of the threshold's size) -- and it makes the test strict
            // (a candidate at the threshold gives a negative result, never +-0).
            thr = thr + (4.0f * (float)((SPLIT ? 3 : 1) * D + 1) * 0x1p-24f) * (fabsf(thr) + qn);
        }
    }
    KNN_PROBE_MARK(20);

    // ---- exact phase -----------------------------------------------------------------------------------------------
    // (1) consumers: list lengths, survivors per query, fast-path flag
    const int need = kk < M ? kk : M;
    const int part = (consumer ? 0 : 2) + h;  // the query's four lanes: two half-waves x the pair's two waves
    if (DUAL) {
        // every lane publishes the per-stage counts of its own list (bytes of a 64-bit word; srl == 0: the total in byte 0)
        // and whether the list overflowed (top bit: a stage holds < 128 survivors of the <= 64 that matter)
        const int nv = cnt < LCAP - 1 ? cnt : LCAP - 1;
        const int tsh = srl > 0 ? srl - 5 : 31;
        unsigned long long pk = 0;
        int totx = 0;
        if (srl > 0) {
            for (int e = 0; e < nv; ++e) {
                const unsigned int w = (unsigned int)mylist[e * 64];
                const int pc = __builtin_popcount(w & 0xffffu);
                totx += pc;
                pk += (unsigned long long)pc << (((w >> 16) >> tsh) * 8);
            }
        } else {  // no row stages (column slices, gather): the total phase B counted -- no walk over the list (a chain of LDS round trips)
            totx = totb;
            pk = (unsigned long long)(unsigned int)totb;
        }
        // (the bytes are only meaningful while none can carry: a part with more than 63 survivors -- the query is not a fast one
        //  then -- publishes its plain total behind a marker bit instead)
        qpk[(cw * 32 + jl) * 4 + part] = (totx <= 63 ? pk : (1ull << 62) | (unsigned long long)totx) | (cnt > LCAP - 1 ? 1ull << 63 : 0ull);
        lcnt2[wv * 64 + lane] = (unsigned short)nv;
    } else if (consumer) {
        const int nv = cnt < kMLCap - 1 ? cnt : kMLCap - 1;
        int tot = 0;
        for (int e = 0; e < nv; ++e) tot += __builtin_popcount((unsigned int)mylist[e * 64] & 0xffffu);
        const int totp = __shfl_xor(tot, 32, 64);
        const int cntp = __shfl_xor(cnt, 32, 64);
        const int n = tot + totp;
        const bool lists_ok = wave_active && qi < N && thr < INFINITY && cnt <= kMLCap - 1 && cntp <= kMLCap - 1 && n >= need;
        const bool fast = lists_ok && n <= kMKeyCap;
        const bool medium = lists_ok && n > kMKeyCap && n <= kMMedCap;  // too many for the key arrays, lists intact
        lcnt[cw * 64 + lane] = (h ? totp : 0) | (nv << 16);  // start offset of this lane's ids | entries
        if (h == 0) { qn_n[cw * 32 + jl] = n; qflag[cw * 32 + jl] = fast ? 1 : (medium ? 2 : 0); qbelow[cw * 32 + jl] = 0; }
    }
    __syncthreads();  // the chunk buffers are free from here on: they hold the keys
    KNN_PROBE_MARK(21);
    unsigned int *qd = reinterpret_cast<unsigned int *>(sm) + (size_t)(cw * 32 + jl) * kMKeyStride;                       // distance bits
    int *qj = reinterpret_cast<int *>(sm) + (size_t)kMWaves * 32 * kMKeyStride + (size_t)(cw * 32 + jl) * kMKeyStride;     // indices
    int n = qn_n[cw * 32 + jl];
    bool fast = qflag[cw * 32 + jl] == 1;
    bool handled = qflag[cw * 32 + jl] == 2;  // answered by the medium path
    unsigned long long dpk[4] = {0ull, 0ull, 0ull, 0ull};  // DUAL: the four parts' packed counts
    if (DUAL) {
        bool ovf = false, big = false;
        int tot4 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long v = qpk[(cw * 32 + jl) * 4 + q];
            ovf |= (v >> 63) != 0;
            const bool bigp = ((v >> 62) & 1ull) != 0;  // more than 63 survivors in this part alone: its plain total
            big |= bigp;
            dpk[q] = bigp ? 0ull : v & ~(3ull << 62);
            tot4 += bigp ? (int)(unsigned int)v : (int)((dpk[q] * 0x0101010101010101ull) >> 56);  // sum of the bytes
        }
        n = tot4;
        const bool lists_ok = wave_active && qi < N && thr < INFINITY && !ovf && n >= need;
        fast = lists_ok && !big && n <= kMKeyCap;
        handled = lists_ok && !fast && n <= kMMedCap;
        if (part == 0) { qn_n[cw * 32 + jl] = n; qflag[cw * 32 + jl] = fast ? 1 : (handled ? 2 : 0); qbelow[cw * 32 + jl] = 0; }
    }
    // staged exact phase (srl > 0): the thread's share of the first stage of candidate rows is requested here, so that
    // it arrives while the lists are decoded
    f32x4v sreg[8];
    // (D > 64: the rows are staged and evaluated in two column halves of <= 64 dimensions -- see the exact phase below)
    const int DS = DP > 64 && D > 64 ? 64 : D;                     // staged width of a row (half), floats
    const int PR = DS >> 2, RPI = srl > 0 ? kMThreads / PR : 0;  // rows per sweep of the block (PR divides the block size)
    const int srow = srl > 0 ? tid / PR : 0, scol = (tid - srow * PR) * 4;
    if (srl > 0) knn_stage_fetch(yb, D, M, 0, srow, RPI, scol, sreg);
    if (csl > 0) knn_stage_fetch(yb, D, M, 0, tid >> 2, kMThreads / 4, 4 * (tid & 3), sreg);  // column slices: rows (tid >> 2) + 128 i, piece tid & 3
    if (wave_active) {
        // ---- medium path (tight clusters, many duplicates: more candidates inside the band than the key arrays hold):
        //      the wave decodes the query's two lane lists into an id list and selects exactly among those ids,
        //      instead of scanning all M candidates in the fallback.  The lists are intact until the barrier after (2).
        const unsigned long long mmask = __ballot(handled);
        int *ids = med + wv * (kMMedCap + 128);
        for (unsigned int bm = (unsigned int)mmask | (unsigned int)(mmask >> 32); bm; bm &= bm - 1) {
            const int j = __builtin_ctz(bm);
            if ((j & 1) != (consumer ? 0 : 1)) continue;  // the pair's two waves share the queries
            int total = 0;
            for (int h2 = 0; h2 < (DUAL ? 4 : 2); ++h2) {  // (DUAL: h2 = 2 * (wave of the pair) + half-wave)
                const int src = (h2 & 1) * 32 + j;
                const int lw = DUAL ? cw + kMWaves * (h2 >> 1) : cw;  // the wave that holds the list
                const int nv2 = DUAL ? lcnt2[lw * 64 + src] : lcnt[cw * 64 + src] >> 16;
                const unsigned int w = lane < nv2 ? (unsigned int)lists[((DUAL ? lw * LCAP : cw * kMLCap) + lane) * 64 + src] : 0u;  // nv2 < 64
                const int pc = __builtin_popcount(w & 0xffffu);
                int incl = pc;
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) {
                    const int t = __shfl_up(incl, m, 64);
                    if (lane >= m) incl += t;
                }
                int pos = total + incl - pc;
                unsigned int m16 = w & 0xffffu;
                const int rowbase = (int)(w >> 16) * 32 + 4 * (h2 & 1);
                while (m16) {
                    const int r = __builtin_ctz(m16);
                    m16 &= m16 - 1;
                    ids[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                }
                total += __shfl(incl, 63, 64);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float bd;
            int bj;
            knn_exact_bruteforce(xb + (size_t)(q0 + j) * D, yb, total, D, kk, lane, reinterpret_cast<float *>(ids + kMMedCap),
                                 ids + kMMedCap + 64, bd, bj, ids);
            const int r = lane - drop;
            if (r >= 0 && r < k) {
                idx[((size_t)b * N + q0 + j) * k + r] = bj;
                if (dist) dist[((size_t)b * N + q0 + j) * k + r] = bd;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // (2) consumers decode their mask words into candidate ids (integer work only).  srl > 0 (staged exact phase): the
    //     ids of a query are grouped by row stage (2^srl candidate rows, at most 8 stages): per-stage counts of the two
    //     half-wave lists in the bytes of a 64-bit word (n <= 60 < 256), prefix sums by one multiplication
    if (DUAL) {
        if (fast) {  // every one of the query's four lanes decodes its own list behind the lists of the parts before it
            const int nv = lcnt2[wv * 64 + lane];
            const unsigned long long incl = (dpk[0] + dpk[1] + dpk[2] + dpk[3]) * 0x0101010101010101ull;  // byte s: survivors in stages 0..s
            unsigned long long before = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) before += q < part ? dpk[q] : 0ull;
            unsigned long long startpk = (srl > 0 ? incl << 8 : 0ull) + before;  // byte s: where this lane's ids of stage s go
            const int tsh = srl > 0 ? srl - 5 : 31;
            for (int e0 = 0; e0 < nv; e0 += 4) {  // four list words in flight
                unsigned int w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = (unsigned int)mylist[(e0 + u < LCAP ? e0 + u : LCAP - 1) * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned int m = e0 + u < nv ? (w[u] & 0xffffu) : 0u;
                    const int rowbase = (int)(w[u] >> 16) * 32 + 4 * h;
                    const int sh = srl > 0 ? (int)((w[u] >> 16) >> tsh) * 8 : 0;
                    int pos = (int)(startpk >> sh) & 0xff;
                    startpk += (unsigned long long)__builtin_popcount(m) << sh;
                    while (m) {
                        const int r = __builtin_ctz(m);
                        m &= m - 1;
                        qj[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                    }
                }
            }
            if (part == 3) { qd[n] = 0xffffffffu; qd[n + 1] = 0xffffffffu; qd[n + 2] = 0xffffffffu;
                             qj[n] = 0x7fffffff; qj[n + 1] = 0x7fffffff; qj[n + 2] = 0x7fffffff; }  // sentinels for the b128 sweeps
        }
    } else if (consumer && fast) {
        const int meta = lcnt[cw * 64 + lane];
        const int nv = meta >> 16;
        if (srl > 0) {
            const int tsh = srl - 5;  // tile -> stage
            unsigned long long pk = 0;
            for (int e = 0; e < nv; ++e) {
                const unsigned int w = (unsigned int)mylist[e * 64];
                pk += (unsigned long long)__builtin_popcount(w & 0xffffu) << (((w >> 16) >> tsh) * 8);
            }
            const unsigned long long pko = ((unsigned long long)(unsigned int)__shfl_xor((int)(pk >> 32), 32, 64) << 32) |
                                           (unsigned int)__shfl_xor((int)pk, 32, 64);
            const unsigned long long incl = (pk + pko) * 0x0101010101010101ull;  // byte s: survivors in stages 0..s
            unsigned long long startpk = (incl << 8) + (h ? pko : 0ull);         // byte s: where this lane's ids of stage s go
            if (h == 0) qstpk[cw * 32 + jl] = incl;
            for (int e0 = 0; e0 < nv; e0 += 4) {  // four list words in flight
                unsigned int w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = (unsigned int)mylist[(e0 + u < kMLCap ? e0 + u : kMLCap - 1) * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned int m = e0 + u < nv ? (w[u] & 0xffffu) : 0u;
                    const int rowbase = (int)(w[u] >> 16) * 32 + 4 * h;
                    const int sh = (int)((w[u] >> 16) >> tsh) * 8;
                    int pos = (int)(startpk >> sh) & 0xff;
                    startpk += (unsigned long long)__builtin_popcount(m) << sh;
                    while (m) {
                        const int r = __builtin_ctz(m);
                        m &= m - 1;
                        qj[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                    }
                }
            }
        } else {
            int pos = meta & 0xffff;
            for (int e0 = 0; e0 < nv; e0 += 4) {  // four list words in flight
                unsigned int w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = (unsigned int)mylist[(e0 + u < kMLCap ? e0 + u : kMLCap - 1) * 64];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    unsigned int m = e0 + u < nv ? (w[u] & 0xffffu) : 0u;
                    const int rowbase = (int)(w[u] >> 16) * 32 + 4 * h;
                    while (m) {
                        const int r = __builtin_ctz(m);
                        m &= m - 1;
                        qj[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                    }
                }
            }
        }
        if (h) { qd[n] = 0xffffffffu; qd[n + 1] = 0xffffffffu; qd[n + 2] = 0xffffffffu;
                 qj[n] = 0x7fffffff; qj[n + 1] = 0x7fffffff; qj[n + 2] = 0x7fffffff; }  // sentinels for the b128 sweeps
    }
    __syncthreads();  // ids visible to the producer partners; the lane lists are dead: their space holds the slots
    KNN_PROBE_MARK(22);
    // (3) the query's survivors are split over its four lanes (two halves x consumer / producer wave)
    const int per = (n + 3) >> 2;
    const int mystart = part * per < n ? part * per : n;
    const int mycount = (mystart + per <= n ? per : n - mystart);
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(lists) + (size_t)(cw * 32 + jl) * 33;  // [..][32 + 1 pad]
    if (csl > 0) {
        // ---- column slices (round 4): ALL candidate rows pass through LDS, 16 dimensions at a time (csl = D / 16 slices), as four
        //      planes of 16-byte pieces ([piece c][row], plane stride 16 Mp + 32 bytes: the coalesced staging writes and the reads of
        //      consecutive rows are conflict-free).  In every slice the four lanes of a query share ALL its survivors evenly -- the
        //      row stages below share the survivors of one stage at a time: ~1.7 per lane against a fullest lane of 3-4 (42 % of the
        //      lane slots held a pair); whole queries differ far less (27 +- 5 survivors) -- and a pair's running sum waits in its
        //      distance slot between slices: the oracle's order of additions.  The query's slice is 16 floats in registers (the row
        //      stages held the whole row: 64), the next slice's pieces of rows and query are in flight while this one is summed.
        const int MPc = (M + kMThreads / 4 - 1) / (kMThreads / 4) * (kMThreads / 4);
        const int PS = MPc * 4 + 8;  // plane stride, floats
        float *stg = reinterpret_cast<float *>(lists);
        const bool act = wave_active && fast;
        const int crow = tid >> 2, cc = tid & 3;
        const float *qrow = xb + (size_t)(act ? qi : 0) * D;
        f32x4v qs[4], qnx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) qs[t] = *reinterpret_cast<const f32x4v *>(qrow + 4 * t);
        // this lane's pairs stay in registers across the slices: the candidate's plane offset and the running sum (n <= 64 survivors per
        // query: at most 16 per lane); four pairs are in flight -- their 16 pieces are requested together, their four chains of additions
        // interleave (a slice of ONE pair is 16 dependent additions: two pairs in flight left the phase latency bound, 2.5 us per slice)
        const int cnt_l = act ? mycount : 0;
        int jo[16];
        float acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            jo[u] = u < cnt_l ? qj[mystart + u] * 4 : 0;
            acc[u] = 0.0f;
        }
        for (int s = 0; s < csl; ++s) {
            if (s) __syncthreads();  // every lane is done with the previous slice
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = crow + i * (kMThreads / 4);
                if (row < MPc) *reinterpret_cast<f32x4v *>(stg + (size_t)cc * PS + (size_t)row * 4) = sreg[i];
            }
            __syncthreads();
            if (s < 3) KNN_PROBE_MARK(26 + 2 * s);
            if (s + 1 < csl) {
                knn_stage_fetch(yb, D, M, 0, crow, kMThreads / 4, 16 * (s + 1) + 4 * cc, sreg);
#pragma unroll
                for (int t = 0; t < 4; ++t) qnx[t] = *reinterpret_cast<const f32x4v *>(qrow + 16 * (s + 1) + 4 * t);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (__ballot(4 * g < cnt_l) != 0ull) {  // (wave-uniform: some lane still has a pair in this group)
                    f32x4v c[4][4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < 4; ++t) c[e][t] = *reinterpret_cast<const f32x4v *>(stg + jo[4 * g + e] + t * PS);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        f32x4v m[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x4v d = qs[t] - c[e][t];
                            m[e] = d * d;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = acc[4 * g + e] + m[e].x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = acc[4 * g + e] + m[e].y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = acc[4 * g + e] + m[e].z;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * g + e] = acc[4 * g + e] + m[e].w;
                    }
                }
            }
            if (s < 3) KNN_PROBE_MARK(27 + 2 * s);
#pragma unroll
            for (int t = 0; t < 4; ++t) qs[t] = qnx[t];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < cnt_l) qd[mystart + u] = __builtin_bit_cast(unsigned int, acc[u]);
    } else if (srl > 0) {
        // staged: the candidate rows come through LDS one stage (2^srl rows) at a time, loaded coalesced once per block
        // (every row exactly once: M * 4D bytes from L2 instead of 4D per survivor), rows 16 bytes apart in the banks.
        // Per stage the four lanes of a query split its survivors of that stage; the query row sits in registers.
        // The oracle's distance of every id: same operations in the same order as the gather below.
        const int SRW = 1 << srl, RSX = DS + 4;
        float *stg = reinterpret_cast<float *>(lists);
        const int nstage = (M + SRW - 1) >> srl;
        const bool act = wave_active && fast;
        const unsigned long long incl = !act ? 0ull : (DUAL ? (dpk[0] + dpk[1] + dpk[2] + dpk[3]) * 0x0101010101010101ull : qstpk[cw * 32 + jl]);
        if constexpr (DP <= 64) {
            f32x4v qreg[DP / 4];
            {
                const float *qrow = xb + (size_t)(act ? qi : 0) * D;
#pragma unroll
                for (int t = 0; t < DP / 4; ++t)
                    qreg[t] = 4 * t < D ? *reinterpret_cast<const f32x4v *>(qrow + 4 * t) : f32x4v{0.f, 0.f, 0.f, 0.f};
            }
            for (int s = 0; s < nstage; ++s) {
                if (s) __syncthreads();  // every lane is done with the previous stage
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = srow + i * RPI;
                    if (row < SRW) *reinterpret_cast<f32x4v *>(stg + (size_t)row * RSX + scol) = sreg[i];
                }
                __syncthreads();
                if (s < 3) KNN_PROBE_MARK(26 + 2 * s);
                if (s + 1 < nstage) knn_stage_fetch(yb, D, M, (s + 1) << srl, srow, RPI, scol, sreg);  // in flight while this stage is evaluated
                if (act) {
                    const int start = s ? (int)(incl >> (8 * (s - 1))) & 0xff : 0, end = (int)(incl >> (8 * s)) & 0xff;
                    const int per = (end - start + 3) >> 2;
                    const int a0 = start + part * per < end ? start + part * per : end;
                    const int a1 = a0 + per < end ? a0 + per : end;
                    for (int p0 = a0; p0 < a1; p0 += 2) {
                        const bool two = p0 + 1 < a1;
                        const float *cp0 = stg + (size_t)(qj[p0] - (s << srl)) * RSX;
                        const float *cp1 = stg + (size_t)(qj[two ? p0 + 1 : p0] - (s << srl)) * RSX;
                        float s0 = 0.0f, s1 = 0.0f;
                        if (D == DP) knn_pair_dist<DP, true>(qreg, cp0, cp1, D, s0, s1);
                        else knn_pair_dist<DP, false>(qreg, cp0, cp1, D, s0, s1);
                        qd[p0] = __builtin_bit_cast(unsigned int, s0);
                        if (two) qd[p0 + 1] = __builtin_bit_cast(unsigned int, s1);
                    }
                }
                if (s < 3) KNN_PROBE_MARK(27 + 2 * s);
            }
        } else {
            // D > 64 (the fourth EdgeConv's 128 features): a 128-dimension query row is 128 registers -- with the stage registers
            // and the pair buffers the kernel spilled 330-380 of them.  The exact phase runs once per column half instead:
            // dimensions 0..63 of every row are staged and summed first (the partial sum waits in the pair's distance slot),
            // then 64..D-1 continue it -- the oracle's order; each half is the D = 64 phase (same stage size, same traffic).
            const int nhalf = D > 64 ? 2 : 1;
            for (int hf = 0; hf < nhalf; ++hf) {
                const int hoff = 64 * hf, wdt = hf ? D - 64 : DS;  // this half's first column and width
                f32x4v qreg[16];
                {
                    const float *qrow = xb + (size_t)(act ? qi : 0) * D + hoff;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        qreg[t] = 4 * t < wdt ? *reinterpret_cast<const f32x4v *>(qrow + 4 * t) : f32x4v{0.f, 0.f, 0.f, 0.f};
                }
                for (int s = 0; s < nstage; ++s) {
                    if (s || hf) __syncthreads();  // every lane is done with the previous stage
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int row = srow + i * RPI;
                        if (row < SRW) *reinterpret_cast<f32x4v *>(stg + (size_t)row * RSX + scol) = sreg[i];
                    }
                    __syncthreads();
                    if (s < 3 && hf == 0) KNN_PROBE_MARK(26 + 2 * s);
                    // the next stage of this half, or the first stage of the second half, in flight while this one is evaluated
                    // (a half narrower than 64 columns: the pieces beyond it re-read its last piece and are never used)
                    if (s + 1 < nstage) {
                        knn_stage_fetch(yb + hoff, D, M, (s + 1) << srl, srow, RPI, scol < wdt ? scol : wdt - 4, sreg);
                    } else if (hf + 1 < nhalf) {
                        knn_stage_fetch(yb + 64, D, M, 0, srow, RPI, scol < D - 64 ? scol : D - 68, sreg);
                    }
                    if (act) {
                        const int start = s ? (int)(incl >> (8 * (s - 1))) & 0xff : 0, end = (int)(incl >> (8 * s)) & 0xff;
                        const int per = (end - start + 3) >> 2;
                        const int a0 = start + part * per < end ? start + part * per : end;
                        const int a1 = a0 + per < end ? a0 + per : end;
                        for (int p0 = a0; p0 < a1; p0 += 2) {
                            const bool two = p0 + 1 < a1;
                            const float *cp0 = stg + (size_t)(qj[p0] - (s << srl)) * RSX;
                            const float *cp1 = stg + (size_t)(qj[two ? p0 + 1 : p0] - (s << srl)) * RSX;
                            float s0 = hf ? __builtin_bit_cast(float, qd[p0]) : 0.0f;
                            float s1 = hf && two ? __builtin_bit_cast(float, qd[p0 + 1]) : 0.0f;
                            if (wdt == 64) knn_pair_dist<64, true>(qreg, cp0, cp1, wdt, s0, s1);
                            else knn_pair_dist<64, false>(qreg, cp0, cp1, wdt, s0, s1);
                            qd[p0] = __builtin_bit_cast(unsigned int, s0);
                            if (two) qd[p0 + 1] = __builtin_bit_cast(unsigned int, s1);
                        }
                    }
                    if (s < 3 && hf == 0) KNN_PROBE_MARK(27 + 2 * s);
                }
            }
        }
    } else if (wave_active && fast) {
        // the oracle's distance of every id.  The query row sits in registers; candidate rows are gathered from L2
        // one full 128-byte line per request (32 dimensions), two candidates in flight
        const float *qrow = xb + (size_t)qi * D;
        if (vec4y && vec4x) {
            constexpr int QR = DP > 64 ? 1 : DP / 4;  // D > 64: the query pieces are re-read (L1) with every 32-dimension block
            float4 qreg[QR];
            if (DP <= 64) {
#pragma unroll
                for (int t = 0; t < QR; ++t)
                    qreg[t] = 4 * t < D ? *reinterpret_cast<const float4 *>(qrow + 4 * t) : float4{0.f, 0.f, 0.f, 0.f};
            }
            for (int p0 = mystart; p0 < mystart + mycount; p0 += 2) {
                const bool two = p0 + 1 < mystart + mycount;
                const float *cp0 = yb + (size_t)qj[p0] * D;
                const float *cp1 = yb + (size_t)qj[two ? p0 + 1 : p0] * D;
                float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                for (int d0 = 0; d0 < DP; d0 += 32) {
                    if (d0 < D) {
                        float4 c0[8], c1[8], q8[DP > 64 ? 8 : 1];
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (d0 + 4 * t < D) {
                                c0[t] = *reinterpret_cast<const float4 *>(cp0 + d0 + 4 * t);
                                c1[t] = *reinterpret_cast<const float4 *>(cp1 + d0 + 4 * t);
                                if (DP > 64) q8[t] = *reinterpret_cast<const float4 *>(qrow + d0 + 4 * t);
                            }
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (d0 + 4 * t < D) {
                                const float4 qv = DP > 64 ? q8[DP > 64 ? t : 0] : qreg[DP > 64 ? 0 : d0 / 4 + t];
                                float t0 = qv.x - c0[t].x, t1 = qv.y - c0[t].y, t2 = qv.z - c0[t].z, t3 = qv.w - c0[t].w;
                                s0 = s0 + t0 * t0;
                                s0 = s0 + t1 * t1;
                                s0 = s0 + t2 * t2;
                                s0 = s0 + t3 * t3;
                                t0 = qv.x - c1[t].x; t1 = qv.y - c1[t].y; t2 = qv.z - c1[t].z; t3 = qv.w - c1[t].w;
                                s1 = s1 + t0 * t0;
                                s1 = s1 + t1 * t1;
                                s1 = s1 + t2 * t2;
                                s1 = s1 + t3 * t3;
                            }
                    }
                }
                qd[p0] = __builtin_bit_cast(unsigned int, s0);
                if (two) qd[p0 + 1] = __builtin_bit_cast(unsigned int, s1);
            }
        } else {
            for (int p0 = mystart; p0 < mystart + mycount; ++p0) {
                const float *cp = yb + (size_t)qj[p0] * D;
                float sd = 0.0f;
                for (int d = 0; d < D; ++d) {
                    const float t = qrow[d] - cp[d];
                    sd = sd + t * t;
                }
                qd[p0] = __builtin_bit_cast(unsigned int, sd);
            }
        }
    }
    __syncthreads();
    KNN_PROBE_MARK(23);
    // (4) rank on the distance bits (squared distances are >= +0: unsigned order), verified as in knn_f16_d3_kernel.  Passes of eight of
    //     the lane's entries against all n keys; a remainder of at most four / two entries in every lane of the wave takes a narrower
    //     pass (round 4: n = 33 ... 36 survivors -- nine entries per lane -- cost a second full pass, 1152 instead of 720 operations).
    {
        const int myc = (wave_active && fast) ? mycount : 0;
        const int nloop = (wave_active && fast) ? n : 0;
        int below = 0;
        auto rank_pass = [&](auto wc, int e0) {
            constexpr int W = decltype(wc)::value;
            unsigned int md[W];
            int rank[W];
#pragma unroll
            for (int u = 0; u < W; ++u) {
                md[u] = e0 + u < myc ? qd[mystart + e0 + u] : 0xffffffffu;
                rank[u] = 0;
            }
            for (int i = 0; i < nloop; i += 4) {  // (nloop: n for the lanes of a fast query, 0 for the others)
                const uint4 o = *reinterpret_cast<const uint4 *>(qd + i);
#pragma unroll
                for (int u = 0; u < W; ++u) {  // compare + add-with-carry: two VALU ops per pair
                    unsigned long long cc;
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.x), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.y), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.z), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.w), "v"(md[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < W; ++u)
                if (e0 + u < myc && rank[u] < kk) {
                    slots[rank[u]] = ((unsigned long long)md[u] << 32) | (unsigned int)qj[mystart + e0 + u];
                    below += 1 + (rank[u] << 8);
                }
        };
        for (int e0 = 0;;) {
            const int rem = myc - e0;
            if (__ballot(rem > 0) == 0ull) break;
            if (__ballot(rem > 2) == 0ull) { rank_pass(std::integral_constant<int, 2>{}, e0); e0 += 2; }
            else if (__ballot(rem > 4) == 0ull) { rank_pass(std::integral_constant<int, 4>{}, e0); e0 += 4; }
            else { rank_pass(std::integral_constant<int, 8>{}, e0); e0 += 8; }
        }
        if (below) atomicAdd(&qbelow[cw * 32 + jl], below);
    }
    __syncthreads();
    KNN_PROBE_MARK(24);
    // (5) verify (count and rank sum, see knn_f16_d3_kernel); slots [drop, kk) are the answer, in order: the query's
    //     four lanes share the writes, 16 bytes at a time
    const bool bad = wave_active && qi < N && fast && qbelow[cw * 32 + jl] != kk + ((kk * (kk - 1) / 2) << 8);  // (n >= kk here)
    if (wave_active && qi < N && fast && !bad) {
        const size_t obase = ((size_t)b * N + qi) * k;
        if ((k & 3) == 0 && ((reinterpret_cast<uintptr_t>(idx) | (dist ? reinterpret_cast<uintptr_t>(dist) : 0)) & 15) == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int v = part + 4 * u;
                if (4 * v < k) {
                    unsigned long long key[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) key[e] = slots[drop + 4 * v + e];
                    *reinterpret_cast<int4 *>(idx + obase + 4 * v) =
                        int4{(int)(unsigned int)key[0], (int)(unsigned int)key[1], (int)(unsigned int)key[2], (int)(unsigned int)key[3]};
                    if (dist)
                        *reinterpret_cast<float4 *>(dist + obase + 4 * v) =
                            float4{__builtin_bit_cast(float, (unsigned int)(key[0] >> 32)), __builtin_bit_cast(float, (unsigned int)(key[1] >> 32)),
                                   __builtin_bit_cast(float, (unsigned int)(key[2] >> 32)), __builtin_bit_cast(float, (unsigned int)(key[3] >> 32))};
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = drop + part + 4 * u;
                if (r < kk) {
                    const unsigned long long key = slots[r];
                    idx[obase + r - drop] = (int)(unsigned int)key;
                    if (dist) dist[obase + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
                }
            }
        }
    }
    if (!wave_active) return;
    // tied queries are ranked again on the full keys, and the leftovers (exact merge) answered, by the pair's two waves on
    // alternate queries
    const bool slowq = qi < N && !fast && !handled;
    const unsigned long long badmask = __ballot(bad);
    const unsigned int bad32 = ((unsigned int)badmask | (unsigned int)(badmask >> 32)) & (consumer ? 0x55555555u : 0xaaaaaaaau);  // (the pair's two waves take alternate queries)
    if (__builtin_popcount(bad32) <= 3) {
        // a few tied queries (ordinary data): the whole wave per query -- the quick form for a single one (knn_common.h)
        for (unsigned int bm = bad32; bm; bm &= bm - 1) {
            const int j = __builtin_ctz(bm);
            const int qs = cw * 32 + j;
            unsigned long long *sj = reinterpret_cast<unsigned long long *>(lists) + (size_t)qs * 33;
            knn_rank_ties(reinterpret_cast<const unsigned int *>(sm) + (size_t)qs * kMKeyStride,
                          reinterpret_cast<const int *>(sm) + (size_t)kMWaves * 32 * kMKeyStride + (size_t)qs * kMKeyStride, qn_n[qs], kk,
                          sj, lane);
            for (int r = drop + lane; r < kk; r += 64) {
                const unsigned long long key = sj[r];
                idx[((size_t)b * N + q0 + j) * k + r - drop] = (int)(unsigned int)key;
                if (dist) dist[((size_t)b * N + q0 + j) * k + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
            }
        }
    } else {
        // many tied queries: the wave's tied queries at once, four lanes per query (knn_common.h: knn_rank_ties4)
        const int j = 2 * (lane >> 2) + (consumer ? 0 : 1), pl = lane & 3;
        const bool mine = ((bad32 >> j) & 1u) != 0;
        const int qs = cw * 32 + j;
        unsigned long long *sj = reinterpret_cast<unsigned long long *>(lists) + (size_t)qs * 33;
        knn_rank_ties4(reinterpret_cast<const unsigned int *>(sm) + (size_t)qs * kMKeyStride,
                       reinterpret_cast<const int *>(sm) + (size_t)kMWaves * 32 * kMKeyStride + (size_t)qs * kMKeyStride, mine ? qn_n[qs] : 0, kk,
                       sj, pl);
        if (mine)
            for (int r = drop + pl; r < kk; r += 4) {
                const unsigned long long key = sj[r];
                idx[((size_t)b * N + q0 + j) * k + r - drop] = (int)(unsigned int)key;
                if (dist) dist[((size_t)b * N + q0 + j) * k + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
            }
    }
    // leftovers (overflowing lists, non-finite bands), wave-cooperative (scratch: behind all the slots)
    int *wscratch = lists + kMWaves * 32 * 33 * 2 + wv * 128;
    const unsigned long long slowmask = __ballot(slowq);
    const unsigned int slow32 = (unsigned int)slowmask | (unsigned int)(slowmask >> 32);
    for (int j = consumer ? 0 : 1; j < 32; j += 2) {
        if (!((slow32 >> j) & 1u) || q0 + j >= N) continue;
        float bd;
        int bj;
        __builtin_amdgcn_wave_barrier();
        knn_exact_bruteforce(xb + (size_t)(q0 + j) * D, yb, M, D, kk, lane, reinterpret_cast<float *>(wscratch), wscratch + 64, bd, bj);
        const int r = lane - drop;
        if (r >= 0 && r < k) {
            idx[((size_t)b * N + q0 + j) * k + r] = bj;
            if (dist) dist[((size_t)b * N + q0 + j) * k + r] = bd;
        }
    }
    KNN_PROBE_MARK(25);
}




size_t knn_pre_bytes(int M, int B, int D) {
    const int DP = (D + 31) / 32 * 32 == 96 ? 128 : (D + 31) / 32 * 32;
    return KnnPre::make(nullptr, M, DP).stride * (size_t)B;
}
bool knn_pre_shape_ok(int M, int D, int kk) {
    return D >= 4 && D <= 128 && kk <= 32 && M >= 64 && M <= 4096 && D % 4 == 0 && kPreThreads % (D / 4) == 0 && D / 4 <= 32;
}
// the shapes fx3d_knn_ws serves with the pre-pass: the single-piece fp16 filter (the default of knn_mfma_kernel)
bool knn_pre_eligible(const float *x, const float *y, int M, int D, int kk) {
    if (!knn_pre_shape_ok(M, D, kk)) return false;
    if (((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) != 0) return false;
    return !(opt(OPT_KNN_NO_MFMA) || opt(OPT_KNN_NO_PREPASS));
}

template <int DK, bool F16, bool SPLIT>
fx3d_status launch_knn_mfma_dk(const float *x, int N, const float *y, int M, int B, int D, int k, int drop,
                               int32_t *idx, float *dist, hipStream_t st, void *pre_ws = nullptr, int xdiv = 1) {
    constexpr int DP = DK * 32, RS = DP + 4, RSI = (F16 && !SPLIT) ? DP / 2 : DP;
    // list lengths + per-query counters + cmax + per-dimension centre + per-stage survivor counts ...
    const bool use_pre = pre_ws != nullptr && F16 && !SPLIT;
    const size_t small = (size_t)kMWaves * 64 * 4 + (size_t)3 * kMWaves * 32 * 4 + 64 + (size_t)DP * 4 + (size_t)kMWaves * 32 * 8;
    static_assert(kMWaves * 32 * 33 * 8 + 2 * kMWaves * 128 * 4 <= kMWaves * kMLCap * 64 * 4, "rank slots + fallback scratch alias the mask lists");
    const int keep_norms = M <= 4096;  // all candidate norms stay in LDS: phase B does not recompute them
    // ... then the tail that is dead after the decode: lists (later the slots), medium path (id lists + merge scratch, one
    // per wave), candidate norms
    size_t tail = (size_t)kMWaves * kMLCap * 64 * 4 + (size_t)2 * kMWaves * (kMMedCap + 128) * 4;
    if (keep_norms) tail += (size_t)((M + 255) / 256 * 256 + 256) * 4;
    // fp16 filter, room permitting: a second norms array (the candidate's error share folded in, upwards / downwards)
    const int two_norms = F16 && keep_norms && M <= 2048;
    if (two_norms) tail += (size_t)((M + 255) / 256 * 256 + 256) * 4;
    const size_t fixed = small + tail;
    const size_t budget = 150 * 1024 - fixed;                                  // floats*4 for the two chunk buffers
    int CH = (int)(budget / 2 / ((size_t)RSI * 4 + 4)) / 64 * 64;
    if (CH > 256) CH = 256;
    constexpr int kMUnits = SPLIT ? kMUnitsSplit : kMUnitsSingle;
    if (F16 && !use_pre && CH > kMUnits * kMProd * 8 / DP / 64 * 64) CH = kMUnits * kMProd * 8 / DP / 64 * 64;  // producer register budget
    const int mpad = (M + 63) / 64 * 64;
    if (CH > mpad) CH = mpad;
    if (use_pre) CH = CH >= 256 ? 256 : (CH >= 128 ? 128 : 64);  // chunks tile the image's 256-row padding exactly
    size_t img = 2 * ((size_t)CH * RSI + CH);                                  // floats
    const size_t qstage = (size_t)kMWaves * 32 * RS;                           // prologue: query rows
    const size_t exact = (size_t)2 * kMWaves * 32 * kMKeyStride;               // exact phase: distance bits + indices
    if (img < qstage) img = qstage;
    if (img < exact) img = exact;
    img = (img + 3) & ~(size_t)3;
    size_t lds = img * 4 + fixed;
    // staged exact phase: candidate rows pass through the tail in stages of 2^srl rows of 4D + 16 bytes (at most 8 stages,
    // at most 8 sweeps of the block per stage; the allocation may grow up to the limit for it).  0 = gather from L2.
    int srl = 0;
    const int DS = DP > 64 && D > 64 ? 64 : D;  // staged width of a row: D > 64 goes through in two column halves
    const int PR = DS / 4;
    const bool stageable = D % 4 == 0 && (kMThreads % PR) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (stageable) {
        const size_t room = 152 * 1024 - (img * 4 + small);
        for (int l = 8; l >= 5; --l) {
            const size_t need = ((size_t)1 << l) * ((size_t)DS * 4 + 16);
            if (need <= room && ((size_t)1 << l) * PR <= 8 * (size_t)kMThreads && ((M + (1 << l) - 1) >> l) <= 8) {
                srl = l;
                if (img * 4 + small + need > lds) lds = img * 4 + small + need;
                break;
            }
        }
    }
    // column slices (round 4) instead of row stages: every row of the cloud, 16 dimensions at a time, as four planes of 16-byte
    // pieces -- when the whole cloud's slice fits the same tail (M <= 1024 at the kernel's 512 threads x 8 pieces)
    int csl = 0;
    if (stageable && D % 16 == 0 && D <= 64 && M <= 8 * (kMThreads / 4)) {
        const size_t room = 152 * 1024 - (img * 4 + small);
        const size_t mpc = (size_t)(M + kMThreads / 4 - 1) / (kMThreads / 4) * (kMThreads / 4);
        const size_t need = 4 * (mpc * 16 + 32);
        if (need <= room) {
            csl = D / 16;
            srl = 0;  // (the decode does not group the ids by row stage)
            if (img * 4 + small + need > lds) lds = img * 4 + small + need;
        }
    }
    const fx3d_status arc = ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_mfma_kernel<DK, F16, SPLIT>), 152 * 1024, "knn_mfma_kernel");
    if (arc != FX3D_OK) return arc;
    FX3D_REQUIRE(lds <= 152 * 1024, "fx3d_knn: internal LDS plan exceeds the CU (D=%d)", D);
    const int qpb = kMWaves * 32;
    const int nbx = (N + qpb - 1) / qpb;
    const int bpad = B >= 8 ? (B + 7) / 8 * 8 : B;
    KnnPre pre{};
    if (use_pre) {
        pre = KnnPre::make(pre_ws, M, DP);
        hipLaunchKernelGGL(knn_pre_stats_kernel, dim3(kPreParts, B), dim3(kPreThreads), 0, st, y, M, D, DP, pre);
        hipLaunchKernelGGL((knn_pre_image_kernel<DK>), dim3(kPreParts, B), dim3(kPreThreads), 0, st, y, M, D, two_norms, pre);
    }
    if (use_pre) {
        const fx3d_status arc2 = ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_mfma_kernel<DK, F16, SPLIT, F16 && !SPLIT>), 152 * 1024,
                                                    "knn_mfma_kernel<pre>");
        if (arc2 != FX3D_OK) return arc2;
        hipLaunchKernelGGL((knn_mfma_kernel<DK, F16, SPLIT, F16 && !SPLIT>), dim3(nbx * bpad), dim3(kMThreads), lds, st, x, N, y, M, B, D,
                           k, drop, idx, dist, CH, (int)img, keep_norms, two_norms, srl, pre_ws, xdiv, csl, 1);
    } else
        hipLaunchKernelGGL((knn_mfma_kernel<DK, F16, SPLIT>), dim3(nbx * bpad), dim3(kMThreads), lds, st, x, N, y, M, B, D,
                           k, drop, idx, dist, CH, (int)img, keep_norms, two_norms, srl, nullptr, xdiv, csl, 0);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status launch_knn_mfma(const float *x, int N, const float *y, int M, int B, int D, int k, int drop, int32_t *idx,
                            float *dist, hipStream_t st, void *pre_ws = nullptr, int xdiv = 1) {
    const int dk = (D + 31) / 32;
    // fp16-split filter: needs 16-byte loads (D % 4 == 0, aligned clouds) and all norms in LDS up front
    const bool f16 = D % 4 == 0 && M <= 4096 && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                     ((size_t)M * D * 4) % 16 == 0;
    if (f16) {
        switch (dk) {
            case 1: return launch_knn_mfma_dk<1, true, false>(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
            case 2: return launch_knn_mfma_dk<2, true, false>(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
            default: return launch_knn_mfma_dk<4, true, false>(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
        }
    }
    switch (dk) {
        case 1: return launch_knn_mfma_dk<1, false, true>(x, N, y, M, B, D, k, drop, idx, dist, st, nullptr, xdiv);
        case 2: return launch_knn_mfma_dk<2, false, true>(x, N, y, M, B, D, k, drop, idx, dist, st, nullptr, xdiv);
        default: return launch_knn_mfma_dk<4, false, true>(x, N, y, M, B, D, k, drop, idx, dist, st, nullptr, xdiv);
    }
}

}  // namespace

namespace fx3d {
fx3d_status knn_mfma_launch(const float *x, int N, const float *y, int M, int B, int D, int k, int drop, int32_t *idx, float *dist,
                            hipStream_t st, void *pre_ws, int xdiv) {
    return launch_knn_mfma(x, N, y, M, B, D, k, drop, idx, dist, st, pre_ws, xdiv);
}
bool knn_mfma_pre_shape_ok(int M, int D, int kk) { return knn_pre_shape_ok(M, D, kk); }
bool knn_mfma_pre_eligible(const float *x, const float *y, int M, int D, int kk) { return knn_pre_eligible(x, y, M, D, kk); }
size_t knn_mfma_pre_bytes(int M, int B, int D) { return knn_pre_bytes(M, B, D); }
}  // namespace fx3d
