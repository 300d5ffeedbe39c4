// D = 3 k-NN on the matrix cores (knn_f16_d3_kernel): one translation unit of the k-NN family (knn_common.h).
#include "knn_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// knn_f16_d3_kernel: kNN for D = 3 (DGCNN's first EdgeConv, BASELINE config 4) with the chamfer kernel's
// fp16-split filter (chamfer.hip, nn1_f16_kernel: t = |c~|^2 + qm~ . c~ on ONE v_mfma_f32_32x32x16_f16 per
// 32 x 32 tile, |t - s^2 (d_oracle - |q'|^2)| <= 2^-20 (4 + 2S)).
//   A block = 4 groups of 32 queries x 2 waves per group; the two waves of a group take alternate pairs of
//   32-candidate tiles (two waves per SIMD hide each other's LDS latencies).  Lane l = (hh, jq) of a wave holds
//   query jq of its group and the 16 candidate rows (r&3)+8(r>>2)+4hh of its tiles: four lanes per query.
//   Phase A: per lane and register the minimum over its tiles (one v_min3 folds two tiles): 32 group minima per
//            lane, 128 per query, each over M/128 candidates.  The kk-th smallest of them bounds the kk-th
//            smallest filter value: tau.  Selection = 32-element sorting network in registers, the partner
//            lane's values by v_permlane32_swap, the other wave's 32 smallest through LDS, two bitonic merges.
//   Phase B: the filter again with the threshold folded into the MFMA (K slot 15: 1 x -thr16, thr16 the
//            smallest fp16 above the threshold): the sign of the result is the test, one v_alignbit per row
//            shifts it into the tile's 16-bit row mask; (tile, mask) words go to the LANE's private LDS list
//            (unconditional store at the list head, the head advances when the mask is not empty).
//            threshold = tau (1 + 4 beta + ...) + (10.1 beta + ...) |q~|^2 + floor on the upper-bound values of the image
//            (mean centring, 2^7 scale, folded norms: chamfer.hip make_pieces; band_b1 / band_a below).  A superset
//            of the k nearest, ~1.1 kk entries per query at config 4.
//   Exact:   every lane decodes its list and evaluates the oracle's distance of its entries (the query is in its
//            registers); the ranking of a query's keys is shared by its four lanes: rank = number of keys with
//            a smaller distance = output slot, verified by count and rank sum, ties re-ranked on (distance,
//            index).  Bit-identical to fx3d_oracle_knn.
//   Queries outside the fp16 range, with overflowing lists or non-finite thresholds take the brute-force merge.
// EdgeConv's features of ONE (point i, neighbour rank r) pair for F = 3 (src/models/dgcnn.jl:36-51): cat(x_i, x_j - x_i),
// layout 0 = (2F,K,N,B), 1 = (K*N,2F,B).  Used by the rare paths of the fused kernel (ties, exact fallback).
__device__ __forceinline__ void knn_d3_feature_entry(float *__restrict__ feat, int layout, int b, int N, int k, int i, int r,
                                                     const float *a, const float *c) {
    if (layout == 0) {
        float *o = feat + (((size_t)b * N + i) * k + r) * 6;
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
        o[3] = c[0] - a[0]; o[4] = c[1] - a[1]; o[5] = c[2] - a[2];
    } else {
        const size_t KN = (size_t)k * N;
        float *o = feat + (size_t)b * 6 * KN + (size_t)i * k + r;
        o[0] = a[0]; o[KN] = a[1]; o[2 * KN] = a[2];
        o[3 * KN] = c[0] - a[0]; o[4 * KN] = c[1] - a[1]; o[5 * KN] = c[2] - a[2];
    }
}

// FEAT: EdgeConv's graph build in one kernel (self-kNN, x == y): the epilogue also writes cat(x_i, x_j - x_i).
template <bool FEAT, class C>
__global__ __launch_bounds__(C::T) void knn_f16_d3_kernel(const float *__restrict__ x, int N,
                                                               const float *__restrict__ y, int M, int B, int k,
                                                               int drop, int32_t *__restrict__ idx,
                                                               float *__restrict__ dist, int CH, int img_bytes,
                                                               int raw_ok, float *__restrict__ feat, int layout, int med_cap, int med_off, int xdiv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char k3sm[];
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * C::W];  // per wave: min, max, sum, sampled second moment (padded to 4 dims)
    __shared__ int nfar;                    // candidates of the cloud beyond the robust range ...
    __shared__ int farlist[kK3FarCap];      // ... their indices: outside the filter, every query evaluates them exactly
    kh8 *imgp = reinterpret_cast<kh8 *>(k3sm);  // piece (blk, half, row) at (blk*2 + half)*32 + row
    constexpr int kListBytes = C::W * C::CAP * 64 * 4;      // lane lists; also the tau exchange and, later, the slots
    constexpr int kCtrInts = C::G * 32 * 8;               // per query: 4 part counts, overflow, n, fast, below
    int *lists_all = reinterpret_cast<int *>(k3sm + img_bytes);                                  // [C::W][C::CAP][64]
    int *ctr = reinterpret_cast<int *>(k3sm + img_bytes + kListBytes);                           // [C::G*32][8]
    const float4 *rawc = reinterpret_cast<const float4 *>(k3sm + img_bytes + kListBytes + kCtrInts * 4);  // [M] when raw_ok

    const int nbx = (N + C::G * 32 - 1) / (C::G * 32);
    const int L = blockIdx.x;
    const bool by_xcd = B >= 8;
    const int b = by_xcd ? ((L >> 3) / nbx) * 8 + (L & 7) : L / nbx;
    const int bxq = by_xcd ? (L >> 3) % nbx : L % nbx;
    if (b >= B) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int jq = lane & 31, hh = lane >> 5;
    const int kk = k + drop;
    const float *xb = x + (size_t)(b / xdiv) * N * 3, *yb = y + (size_t)b * M * 3;  // (xdiv > 1: candidate slices as virtual clouds share their queries)
    const int grp = wv % C::G, half = wv / C::G;  // query group; which pairs of tiles this wave takes
    const int q0 = (bxq * C::G + grp) * 32;
    const bool wave_active = q0 < N;
    for (int e = tid; e < kCtrInts; e += C::T) ctr[e] = 0;
    if (tid == 0) nfar = 0;  // (ordered before its first use by the barrier of the cloud pass)
    const int qi = q0 + jq;  // this lane's query (loaded here: the latency hides behind the pass over the cloud)
    const int qc = qi < N ? qi : N - 1;
    const float qr[3] = {xb[(size_t)qc * 3], xb[(size_t)qc * 3 + 1], xb[(size_t)qc * 3 + 2]};
    KNN_PROBE_MARK(0);

    // ---- one pass over the cloud: bounding box (-> centre mu, power-of-two scale sc with |c~| <= 1) and, for
    //      clouds up to kTRawMax points, the raw coordinates parked in LDS for the staging and the exact phase ----
    float mu[3], cinf = 0.0f, varmax = 0.0f;
    bool allfin = true;
    {
        float4 *raww = reinterpret_cast<float4 *>(k3sm + img_bytes + kListBytes + kCtrInts * 4);
        float mn3[3] = {INFINITY, INFINITY, INFINITY}, mx3[3] = {-INFINITY, -INFINITY, -INFINITY}, sm3[3] = {0.f, 0.f, 0.f};
        float sqt = 0.0f;  // second moment (all three coordinates) of a sample (~M/4 points) about the cloud's first point: the spread
        const float pil[3] = {yb[0], yb[1], yb[2]};
        // thread t takes points t, t + C::T, ...: 12-byte loads, consecutive lanes on consecutive points (coalesced,
        // and the 16-byte LDS slots of a wave's points are consecutive: no bank conflicts)
        const int nsweep = (M + C::T - 1) / C::T;
        for (int i0 = 0; i0 < nsweep; i0 += 4) {
            P3 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = (i0 + e) * C::T + tid;
                v[e] = *reinterpret_cast<const P3 *>(yb + (size_t)(pt < M ? pt : M - 1) * 3);  // (clamped: always valid)
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pt = (i0 + e) * C::T + tid;
                mn3[0] = fminf(mn3[0], v[e].x); mx3[0] = fmaxf(mx3[0], v[e].x);
                mn3[1] = fminf(mn3[1], v[e].y); mx3[1] = fmaxf(mx3[1], v[e].y);
                mn3[2] = fminf(mn3[2], v[e].z); mx3[2] = fmaxf(mx3[2], v[e].z);
                if (pt < M) {
                    sm3[0] = sm3[0] + v[e].x; sm3[1] = sm3[1] + v[e].y; sm3[2] = sm3[2] + v[e].z;
                    if (raw_ok) raww[pt] = float4{v[e].x, v[e].y, v[e].z, 0.0f};
                }
                if (((i0 + e) & 3) == (wv & 3) && pt < M) {  // (wave-uniform first condition: 64-point runs all over the cloud)
                    sqt = __builtin_fmaf(v[e].x - pil[0], v[e].x - pil[0], sqt); sqt = __builtin_fmaf(v[e].y - pil[1], v[e].y - pil[1], sqt);
                    sqt = __builtin_fmaf(v[e].z - pil[2], v[e].z - pil[2], sqt);
                }
            }
        }
        {   // wave level by DPP (the result is in lane 63), one 16-byte row per (wave, statistic)
            float4 lo4, hi4, st4;
            lo4.x = wave_min_l63(mn3[0]); lo4.y = wave_min_l63(mn3[1]); lo4.z = wave_min_l63(mn3[2]); lo4.w = 0.0f;
            hi4.x = wave_max_l63(mx3[0]); hi4.y = wave_max_l63(mx3[1]); hi4.z = wave_max_l63(mx3[2]); hi4.w = 0.0f;
            st4.x = wave_sum_l63(sm3[0]); st4.y = wave_sum_l63(sm3[1]); st4.z = wave_sum_l63(sm3[2]); st4.w = wave_sum_l63(sqt);
            if (lane == 63) {
                float4 *r4 = reinterpret_cast<float4 *>(red);
                r4[wv * 4] = lo4; r4[wv * 4 + 1] = hi4; r4[wv * 4 + 2] = st4;
            }
        }
        __syncthreads();
        {
            const float4 *r4 = reinterpret_cast<const float4 *>(red);
            float4 lo4 = r4[0], hi4 = r4[1], st4 = r4[2];
#pragma unroll
            for (int w = 1; w < C::W; ++w) {
                const float4 a0 = r4[w * 4], a1 = r4[w * 4 + 1], a2 = r4[w * 4 + 2];
                lo4.x = fminf(lo4.x, a0.x); lo4.y = fminf(lo4.y, a0.y); lo4.z = fminf(lo4.z, a0.z);
                hi4.x = fmaxf(hi4.x, a1.x); hi4.y = fmaxf(hi4.y, a1.y); hi4.z = fmaxf(hi4.z, a1.z);
                st4.x = st4.x + a2.x; st4.y = st4.y + a2.y; st4.z = st4.z + a2.z; st4.w = st4.w + a2.w;
            }
            const float lo3[3] = {lo4.x, lo4.y, lo4.z}, hi3[3] = {hi4.x, hi4.y, hi4.z}, st3[3] = {st4.x, st4.y, st4.z};
            varmax = st4.w / fmaxf(0.25f * (float)M, 1.0f);  // TOTAL variance of the three coordinates, from ~M/4 sampled points (a heuristic's input)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                // centre = the MEAN (a stray far point moves the box centre, hardly the mean); any centre is correct
                mu[d] = fminf(fmaxf(st3[d] / (float)M, lo3[d]), hi3[d]);
                const float off = st3[d] / (float)M - pil[d];
                varmax = varmax - off * off;  // (variance about the mean from the moment about a data point)
                cinf = fmaxf(cinf, fmaxf(hi3[d] - mu[d], mu[d] - lo3[d]));
                allfin = allfin && fabsf(st3[d]) < INFINITY;  // a NaN / +-Inf coordinate makes the sum non-finite (fminf / fmaxf skip NaNs)
            }
        }
        cinf = cinf * 1.000001f;
    }
    // not sane (non-finite or huge coordinates): every query takes the brute-force merge, which orders distances as the
    // oracle does (dist_key); a finite cloud with cinf < 1e16 has no infinite or NaN distance to a usable query
    const bool sane = allfin && cinf < 1.0e16f;  // usable queries lie within 234 cinf of the centre: 3 (235 cinf)^2 stays finite
    // robust range, as in nn1_f16_kernel (chamfer.hip): a few points far from the bulk must not set the scale (one point 10^5 x
    // the extent away sent every query to the exact merge: 143 vs 22 us).  rng = min(cinf, 16 x the mean max-norm deviation
    // about the re-centred mean); gates: kRobustGate / kRobustHarm (fx3d_common.h).  Candidates beyond the range
    // get norm +inf in the image (never below a threshold) and go on a side list that every query appends to its survivors.
    float rng = cinf;
    if (sane && 3.0f * cinf * cinf > kRobustGate * varmax) {
        const float4 r = robust_range3<C::T, true>(yb, M, raw_ok != 0, rawc, red, mu[0], mu[1], mu[2], cinf);
        mu[0] = r.x; mu[1] = r.y; mu[2] = r.z; rng = r.w;
    }
    const bool has_far = sane && rng < cinf;
    float sc = 1.0f;
    if (sane && rng > 1.0e-30f) {
        int e;
        (void)frexpf(rng, &e);
        sc = ldexpf(1.0f, 7 - e);  // |c~| < 2^7: a bulk far smaller than the farthest point stays out of fp16's subnormals
    }
    KNN_PROBE_MARK(1);

    // ---- this lane's query: B operand, band --------------------------------------------------------------------
    const float m0 = -2.0f * ((qr[0] - mu[0]) * sc), m1 = -2.0f * ((qr[1] - mu[1]) * sc), m2 = -2.0f * ((qr[2] - mu[2]) * sc);
    const float S = (fabsf(m0) + fabsf(m1)) + fabsf(m2);
    const bool qok = S < 3.0e4f;  // inside the fp16 range (also false for NaN)
    // band (chamfer.hip, make_pieces: the image holds upper bounds U_c = t^ + beta n_c): a candidate among the kk nearest
    // has U_c <= tau_U (1 + 18 beta) + 22.3 beta |q~|^2 + floor, + the oracle's rounding, + the threshold as a 17th
    // MFMA term in phase B (2^-21 of its magnitude)
    const float qn = 0.25f * ((m0 * m0 + m1 * m1) + m2 * m2);
    const float band_b1 = 1.0f + 18.0f * kK3BetaC + 0x1p-20f + 0x1p-21f;
    const float band_a = (22.3f * kK3BetaC + 0x1p-19f + 0x1p-21f) * qn + 0x1p-24f * (S + 4.0f);
    kh8 bq;
    {
        _Float16 hx, lx, hy, ly, hz, lz;
        k3_split2h(qok ? m0 : 0.f, hx, lx); k3_split2h(qok ? m1 : 0.f, hy, ly); k3_split2h(qok ? m2 : 0.f, hz, lz);
        const _Float16 one = (_Float16)1.0f, z = (_Float16)0.0f;
        bq = hh == 0 ? kh8{hx, lx, hx, hy, ly, hy, hz, lz} : kh8{hz, one, one, one, lx, ly, lz, z};
    }

    float mn[32];  // group minima: [r] first / [16 + r] second tile of this wave's pairs -> 128 groups per query
#pragma unroll
    for (int r = 0; r < 32; ++r) mn[r] = INFINITY;
    float thr = 0.0f;
    int cnt = 0, tot = 0;
    int *mylist = lists_all + wv * C::CAP * 64 + lane;  // entry e at mylist[e * 64]
    f32x16v zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
    const int nchunk = (M + CH - 1) / CH;

    for (int phase = 0; phase < 2; ++phase) {
        for (int ci = 0; ci < nchunk; ++ci) {
            const int j0 = (phase == 0 ? ci : nchunk - 1 - ci) * CH;  // phase B backwards: its first chunk is staged
            const int cn = (M - j0) < CH ? (M - j0) : CH;
            const int cn_pad = (cn + 63) & ~63;
            if (!(phase == 1 && ci == 0)) {
                __syncthreads();
                // (two separate loops: a select between an LDS and a global pointer trips the compiler)
                if (raw_ok) {
                    for (int pt = tid; pt < cn_pad; pt += C::T) {
                        const int i0 = ((pt >> 5) * 2) * 32 + (pt & 31);
                        kh8 p0, p1;
                        if (pt < cn) {
                            const float4 rc = rawc[j0 + pt];
                            k3_pieces_far((rc.x - mu[0]) * sc, (rc.y - mu[1]) * sc, (rc.z - mu[2]) * sc, has_far, phase == 0, j0 + pt, &nfar, farlist, p0, p1);
                        } else {  // padding: n1 = +inf => t = +inf, never below a finite threshold
                            k3_make_pieces(0.f, 0.f, 0.f, p0, p1);
                            p1[1] = (_Float16)INFINITY;
                        }
                        imgp[i0] = p0;
                        imgp[i0 + 32] = p1;
                    }
                } else {
                    for (int pt = tid; pt < cn_pad; pt += C::T) {
                        const int i0 = ((pt >> 5) * 2) * 32 + (pt & 31);
                        kh8 p0, p1;
                        if (pt < cn) {
                            const float *src = yb + (size_t)(j0 + pt) * 3;
                            k3_pieces_far((src[0] - mu[0]) * sc, (src[1] - mu[1]) * sc, (src[2] - mu[2]) * sc, has_far, phase == 0, j0 + pt, &nfar, farlist, p0, p1);
                        } else {
                            k3_make_pieces(0.f, 0.f, 0.f, p0, p1);
                            p1[1] = (_Float16)INFINITY;
                        }
                        imgp[i0] = p0;
                        imgp[i0 + 32] = p1;
                    }
                }
                __syncthreads();
                KNN_PROBE_MARK(2);
            }
            if (wave_active) {
                const kh8 *pa = imgp + hh * 32 + jq;
                const int npair = cn_pad / 64;
                const int tile0 = j0 / 32;
                int pr = half;  // this wave's pairs of 32-candidate tiles: half, half + 2, ...
                if (phase == 0) {
                    for (; pr + 2 < npair; pr += 4) {  // two pairs per step: one v_min3 folds two tiles' rows
                        f32x16v a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2) * 64], bq, zero, 0, 0, 0);
                        f32x16v b0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2 + 4) * 64], bq, zero, 0, 0, 0);
                        f32x16v a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2 + 1) * 64], bq, zero, 0, 0, 0);
                        f32x16v b1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2 + 5) * 64], bq, zero, 0, 0, 0);
                        KNN_MFMA_SETTLE4(a0, b0, a1, b1);
#pragma unroll
                        for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[r]) : "v"(a0[r]), "v"(b0[r]));
#pragma unroll
                        for (int r = 0; r < 16; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mn[16 + r]) : "v"(a1[r]), "v"(b1[r]));
                    }
                }
                for (; pr < npair; pr += 2) {
                    f32x16v acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2) * 64], bq, zero, 0, 0, 0);
                    f32x16v acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[(pr * 2 + 1) * 64], bq, zero, 0, 0, 0);
                    if (phase == 0) {
                        KNN_MFMA_SETTLE2(acc0, acc1);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mn[r] = vmin_acc(mn[r], acc0[r]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) mn[16 + r] = vmin_acc(mn[16 + r], acc1[r]);
                    } else {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            // one word per tile: (tile index << 16) | mask of the rows with t - thr16 < 0 (row r at
                            // bit 15 - r: one v_alignbit shifts the sign in); stored at the list head
                            // unconditionally, the head advances when the mask is not empty
                            unsigned int m = 0;
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                m = __builtin_amdgcn_alignbit(m, __builtin_bit_cast(unsigned int, tt ? acc1[r] : acc0[r]), 31);
                            const int pp = cnt < C::CAP - 1 ? cnt : C::CAP - 1;
                            mylist[pp * 64] = (int)((unsigned int)(tile0 + pr * 2 + tt) << 16 | m);
                            cnt += m != 0 ? 1 : 0;
                            tot += __builtin_popcount(m);
                        }
                    }
                }
            }
        }
        KNN_PROBE_MARK(phase ? 5 : 3);
        if (phase == 0) {
            // ---- tau: kk-th smallest of the 128 group minima of every query: 32 in this lane, 32 in its partner
            //      lane, 64 in the other wave of the group.  Sorting network in registers, one exchange with the
            //      partner lane, one exchange with the other wave through LDS, bitonic merges in between.
            float tau = INFINITY;
            float *xch = reinterpret_cast<float *>(lists_all);  // the lists are not in use yet
            if (C::KKMAX <= 32 && kk <= 24 && M >= 128) {
                // (round 4) the eight smallest of each half-lane set of 16 instead of a sort of all 32: knn_tau_8of16
                tau = knn_tau_8of16<C::G>(mn, xch, wv, jq, hh, kk);
            } else {
                k3_sort_regs<32>(mn);
                {
                    float oth[32];  // the partner lane's values (v_permlane32_swap: no LDS round trip)
    #pragma unroll
                    for (int r = 0; r < 32; ++r)  // mn[r] <- lanes 0-31's value, oth[r] <- lanes 32-63's, in every lane
                        // (inline asm: this compiler's __builtin_amdgcn_permlane32_swap returns its first result twice)
                        asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(mn[r]), "=&v"(oth[r]));
    #pragma unroll
                    for (int r = 0; r < 32; ++r)  // the 32 smallest of the wave's 64 (a bitonic sequence), in both half-lanes
                        mn[r] = vmin_f32(mn[r], oth[31 - r]);
                }
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
                if (hh == 0) {
    #pragma unroll
                    for (int r = 0; r < 32; ++r) xch[(wv * 32 + jq) * 33 + r] = mn[r];
                }
                __syncthreads();
                // the kk-th smallest of the union of this wave's 32 smallest X and the other wave's Y (both ascending) without merging
                // them: min over the splits (i values from X, kk - i from Y) of max(X[i-1], Y[kk-i-1]).  For kk <= 32 that is the kk-th
                // smallest of all 128 group minima; for 32 < kk <= 64 the kk-th smallest of these 64 -- an upper bound of it (equal unless
                // one wave holds more than 32 of the kk smallest).  ~3 VALU + one LDS read per split instead of 32 reads + a 32-value
                // bitonic merge (192 VALU).
                if (C::KKMAX <= 32) {
                    const float *po = xch + (((wv + C::G) % C::W) * 32 + jq) * 33;  // the group's other wave
    #pragma unroll
                    for (int r = 0; r < 32; ++r) mn[r] = vmin_f32(mn[r], po[31 - r]);     // the 32 smallest of the 128
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
                    tau = mn[0];  // kk <= 32: among the 32 smallest
    #pragma unroll
                    for (int r = 1; r < 32; ++r) tau = (kk - 1) == r ? mn[r] : tau;
                } else
                {
                    const float *po = xch + (((wv + C::G) % C::W) * 32 + jq) * 33;  // the group's other wave
                    float yv[33];  // (all reads first, clamped addresses: a branch per split made them a chain of LDS round trips)
    #pragma unroll
                    for (int i = 0; i <= 32; ++i) {
                        const int yi = kk - i - 1;
                        yv[i] = po[yi < 0 ? 0 : (yi > 31 ? 31 : yi)];
                    }
    #pragma unroll
                    for (int i = 0; i <= 32; ++i) {
                        const int ny = kk - i;  // (uniform: the selects below take scalar conditions)
                        const float a = i >= 1 ? mn[i >= 1 ? i - 1 : 0] : -INFINITY;
                        const float t = vmax_f32(a, ny >= 1 ? yv[i] : -INFINITY);
                        tau = vmin_f32(tau, (ny >= 0 && ny <= 32) ? t : INFINITY);
                    }
                }
            }
            thr = __builtin_fmaf(tau, band_b1, band_a);
            {
                // phase B subtracts the threshold inside the MFMA (K slot 15: candidate side 1, query side -thr16)
                // and keeps the sign: thr16 = the smallest fp16 value strictly above thr, so that t <= thr gives a
                // negative difference (no -0) -- at most a few more survivors than the Float32 threshold
                _Float16 h = (_Float16)thr;
                unsigned short hb = __builtin_bit_cast(unsigned short, h);
                if ((float)h <= thr) hb = (hb & 0x7fffu) == 0 ? 0x0001u : ((hb & 0x8000u) ? hb - 1 : hb + 1);
                h = __builtin_bit_cast(_Float16, hb);
                if (!((float)h < INFINITY)) thr = INFINITY;  // (also NaN) -> the query is not usable
                if (hh == 1) bq[7] = -h;
            }
            __syncthreads();  // the exchange space becomes the lane lists
            KNN_PROBE_MARK(4);
        }
    }

    // ---- exact phase: the four lanes of a query (two half-lanes x two waves) share its survivors ------------------
    const int part = half * 2 + hh;
    const int qslot = grp * 32 + jq;
    int *qctr = ctr + qslot * 8;  // [0..3] entries decoded by part, [4] overflow, [5] below
    const int need = kk < M ? kk : M;
    const int nv = cnt < C::CAP - 1 ? cnt : C::CAP - 1;
    const int nf = has_far ? nfar : 0;              // (complete: every chunk was staged before the last barrier)
    const bool far_ok = nf <= kK3FarCap;             // more far candidates than the side list holds: no query is usable
    qctr[part] = tot + (part == 3 && far_ok ? nf : 0);  // the query's last lane appends the far candidates to its own entries
    if (cnt > C::CAP - 1) qctr[4] = 1;
    if (med_cap > 0) atomicOr(&qctr[6], nv << (8 * part));  // list lengths of the four parts (< 24 each): the medium path's decode
    __syncthreads();  // every wave is done with the image: its space now holds the keys
    KNN_PROBE_MARK(6);
    unsigned int *qd = reinterpret_cast<unsigned int *>(k3sm) + (size_t)qslot * C::KS;                          // distance bits
    int *qj = reinterpret_cast<int *>(k3sm) + (size_t)C::G * 32 * C::KS + (size_t)qslot * C::KS;     // indices
    const int c0 = qctr[0], c1 = qctr[1], c2 = qctr[2], c3 = qctr[3];
    const int n = c0 + c1 + c2 + c3;
    const int off = part == 0 ? 0 : (part == 1 ? c0 : (part == 2 ? c0 + c1 : c0 + c1 + c2));
    const bool usable = sane && far_ok && qok && thr < INFINITY;
    const bool fast = wave_active && qi < N && usable && qctr[4] == 0 && n <= C::KCAP && n >= need;  // (+ 3 sentinels: inside the stride)
    // ---- medium path (tight clusters, duplicated points, lattices: more candidates inside the band than the key arrays hold):
    //      a wave decodes the query's four lane lists into an id list (+ the far candidates) and selects exactly among those,
    //      instead of scanning all M candidates in the fallback.  The lists are intact until the barrier after the decode.
    const bool medium = med_cap > 0 && wave_active && qi < N && usable && qctr[4] == 0 && n > C::KCAP && n <= med_cap && n >= need;
    if (med_cap > 0 && wave_active) {
        const unsigned long long mm = __ballot(medium);
        int *ids = reinterpret_cast<int *>(k3sm + med_off) + wv * (med_cap + 128);
        for (unsigned int bm = (unsigned int)mm | (unsigned int)(mm >> 32); bm; bm &= bm - 1) {
            const int j = __builtin_ctz(bm);
            if ((j & 1) != half) continue;  // the group's two waves share the queries
            const int *cj = ctr + (grp * 32 + j) * 8;
            int total = 0;
            for (int p2 = 0; p2 < 4; ++p2) {  // p2 = 2 * (wave of the group) + half-wave
                const int src = (p2 & 1) * 32 + j;
                const int nv2 = (cj[6] >> (8 * p2)) & 0xff;
                const unsigned int w = lane < nv2 ? (unsigned int)lists_all[((grp + C::G * (p2 >> 1)) * C::CAP + lane) * 64 + src] : 0u;
                const int pc = __builtin_popcount(w & 0xffffu);
                int incl = pc;
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) {
                    const int t = __shfl_up(incl, m, 64);
                    if (lane >= m) incl += t;
                }
                int pos = total + incl - pc;
                unsigned int m16 = w & 0xffffu;
                const int rowbase = (int)(w >> 16) * 32 + 4 * (p2 & 1);
                while (m16) {
                    const int r = 15 - __builtin_ctz(m16);  // (row r at bit 15 - r)
                    m16 &= m16 - 1;
                    ids[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                }
                total += __shfl(incl, 63, 64);
            }
            for (int f = lane; f < nf; f += 64) ids[total + f] = farlist[f];  // (nf = 0 unless far_ok and has_far: usable)
            total += nf;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float bd;
            int bj;
            knn_exact_bruteforce<(C::KKMAX > 32)>(xb + (size_t)(q0 + j) * 3, yb, total, 3, kk, lane, reinterpret_cast<float *>(ids + med_cap), ids + med_cap + 64,
                                 bd, bj, ids);
            const int r = lane - drop;
            if (r >= 0 && r < k) {
                idx[((size_t)b * N + q0 + j) * k + r] = bj;
                if (dist) dist[((size_t)b * N + q0 + j) * k + r] = bd;
                if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, q0 + j, r, xb + (size_t)(q0 + j) * 3, yb + (size_t)bj * 3);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (fast) {
        // (1) decode the (tile, mask) words into candidate ids: integer work only, no memory latency in the chain
        int pos = off;
        for (int e0 = 0; e0 < nv; e0 += 4) {  // four list words in flight
            unsigned int w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = (unsigned int)mylist[(e0 + u < C::CAP ? e0 + u : C::CAP - 1) * 64];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned int m = e0 + u < nv ? (w[u] & 0xffffu) : 0u;
                const int rowbase = (int)(w[u] >> 16) * 32 + 4 * hh;
                while (m) {
                    const int r = 15 - __builtin_ctz(m);
                    m &= m - 1;
                    qj[pos++] = rowbase + (r & 3) + 8 * (r >> 2);
                }
            }
        }
        if (part == 3)
            for (int f = 0; f < nf; ++f) qj[pos++] = farlist[f];  // (never inside a mask: their filter value is +inf)
        // (2) the oracle's distance of the ids this lane just wrote (its own LDS writes: no barrier), eight in flight
        for (int p0 = off; p0 < pos; p0 += 8) {
            int id[8];
            float c0f[8], c1f[8], c2f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) id[u] = qj[p0 + u < pos ? p0 + u : off];
            if (raw_ok) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 rc = rawc[id[u]];
                    c0f[u] = rc.x; c1f[u] = rc.y; c2f[u] = rc.z;
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float *c = yb + (size_t)id[u] * 3;
                    c0f[u] = c[0]; c1f[u] = c[1]; c2f[u] = c[2];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float t0 = qr[0] - c0f[u], t1 = qr[1] - c1f[u], t2 = qr[2] - c2f[u];
                float sd = t0 * t0;
                sd = sd + t1 * t1;
                sd = sd + t2 * t2;
                if (p0 + u < pos) qd[p0 + u] = __builtin_bit_cast(unsigned int, sd);
            }
        }
        if (part == 3) {  // sentinels: the rank loops read four keys at a time
            qd[n] = 0xffffffffu; qd[n + 1] = 0xffffffffu; qd[n + 2] = 0xffffffffu;
            qj[n] = 0x7fffffff; qj[n + 1] = 0x7fffffff; qj[n + 2] = 0x7fffffff;
        }
    }
    __syncthreads();  // keys visible to the query's four lanes; the lane lists are dead: their space holds the slots
    KNN_PROBE_MARK(8);
    const int per = (n + 3) >> 2;  // the ranking is shared evenly
    const int mystart = part * per < n ? part * per : n;
    const int mycount = mystart + per <= n ? per : n - mystart;
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(lists_all) + (size_t)qslot * C::SS;  // [..][KKMAX + 1 pad]
    // ---- order: rank of a survivor = number of survivors of its query with a smaller distance (squared distances
    //      are >= +0: unsigned order of the bits).  Ties in the distance are resolved by the index in the
    //      reference; they are rare, so the ranks are computed on the distances alone and VERIFIED: the ranks below
    //      kk are a permutation of 0..kk-1 iff exactly kk entries have rank < kk and their ranks sum to
    //      kk (kk - 1) / 2: an entry's rank is at most its position in the sorted order, strictly less for every
    //      entry tied with an earlier one; a tie straddling the kk boundary makes the count kk + 1.  A query that
    //      fails is ranked again on the full keys.
    if (fast) {
        int below = 0;  // own entries with rank < kk: count | sum of ranks << 8
        for (int e0 = 0; e0 < mycount; e0 += 8) {
            unsigned int md[8];
            int rank[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                md[u] = e0 + u < mycount ? qd[mystart + e0 + u] : 0xffffffffu;
                rank[u] = 0;
            }
            for (int i = 0; i < n; i += 4) {
                const uint4 o = *reinterpret_cast<const uint4 *>(qd + i);
#pragma unroll
                for (int u = 0; u < 8; ++u) {  // compare + add-with-carry: two VALU ops per pair
                    unsigned long long cc;
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.x), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.y), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.z), "v"(md[u]));
                    asm("v_cmp_lt_u32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank[u]), "=&s"(cc) : "v"(o.w), "v"(md[u]));
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u < mycount && rank[u] < kk) {
                    slots[rank[u]] = ((unsigned long long)md[u] << 32) | (unsigned int)qj[mystart + e0 + u];
                    below += 1 + (rank[u] << 8);
                }
        }
        if (below) atomicAdd(&qctr[5], below);
    }
    __syncthreads();
    KNN_PROBE_MARK(9);
    const bool bad = fast && qctr[5] != kk + ((kk * (kk - 1) / 2) << 8);  // (n >= kk here)
    if (part == 0 && wave_active && qi < N) {
        KNN_PROBE_STAT(0, 1);
        KNN_PROBE_STAT(1, !fast);
        KNN_PROBE_STAT(2, !usable);
        KNN_PROBE_STAT(3, qctr[4] != 0);
        KNN_PROBE_STAT(4, n > C::KCAP);
        KNN_PROBE_STAT(5, n < need);
        KNN_PROBE_STAT(6, n);
        KNN_PROBE_STAT(8, bad);
    }
    if (fast && !bad) {
        // slots [drop, kk) are the answer, in order; the query's four lanes share the writes, 16 bytes at a time
        const size_t obase = ((size_t)b * N + qi) * k;
        if ((k & 3) == 0 && ((reinterpret_cast<uintptr_t>(idx) | (dist ? reinterpret_cast<uintptr_t>(dist) : 0)) & 15) == 0) {
#pragma unroll
            for (int u = 0; u < C::KKMAX / 16; ++u) {
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
                    if (FEAT) {
                        // four neighbours of point qi: 16-byte stores along the rank dimension (mlp layout: one per
                        // channel row; cat layout: 24 contiguous floats)
                        float cx[4], cy[4], cz[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = (int)(unsigned int)key[e];
                            if (raw_ok) {
                                const float4 rc = rawc[j];
                                cx[e] = rc.x - qr[0]; cy[e] = rc.y - qr[1]; cz[e] = rc.z - qr[2];
                            } else {
                                const float *c = yb + (size_t)j * 3;
                                cx[e] = c[0] - qr[0]; cy[e] = c[1] - qr[1]; cz[e] = c[2] - qr[2];
                            }
                        }
                        if (layout == 1 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0) {
                            const size_t KN = (size_t)k * N;
                            float *o = feat + (size_t)b * 6 * KN + (size_t)qi * k + 4 * v;
                            *reinterpret_cast<float4 *>(o) = float4{qr[0], qr[0], qr[0], qr[0]};
                            *reinterpret_cast<float4 *>(o + KN) = float4{qr[1], qr[1], qr[1], qr[1]};
                            *reinterpret_cast<float4 *>(o + 2 * KN) = float4{qr[2], qr[2], qr[2], qr[2]};
                            *reinterpret_cast<float4 *>(o + 3 * KN) = float4{cx[0], cx[1], cx[2], cx[3]};
                            *reinterpret_cast<float4 *>(o + 4 * KN) = float4{cy[0], cy[1], cy[2], cy[3]};
                            *reinterpret_cast<float4 *>(o + 5 * KN) = float4{cz[0], cz[1], cz[2], cz[3]};
                        } else if (layout == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0) {
                            float4 *o = reinterpret_cast<float4 *>(feat + (obase + 4 * v) * 6);
                            o[0] = float4{qr[0], qr[1], qr[2], cx[0]};
                            o[1] = float4{cy[0], cz[0], qr[0], qr[1]};
                            o[2] = float4{qr[2], cx[1], cy[1], cz[1]};
                            o[3] = float4{qr[0], qr[1], qr[2], cx[2]};
                            o[4] = float4{cy[2], cz[2], qr[0], qr[1]};
                            o[5] = float4{qr[2], cx[3], cy[3], cz[3]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float cc[3] = {cx[e], cy[e], cz[e]};
                                float *o0 = layout == 0 ? feat + (obase + 4 * v + e) * 6 : feat + (size_t)b * 6 * k * N + (size_t)qi * k + 4 * v + e;
                                const size_t st = layout == 0 ? 1 : (size_t)k * N;
                                o0[0] = qr[0]; o0[st] = qr[1]; o0[2 * st] = qr[2];
                                o0[3 * st] = cc[0]; o0[4 * st] = cc[1]; o0[5 * st] = cc[2];
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < C::KKMAX / 4; ++u) {
                const int r = drop + part + 4 * u;
                if (r < kk) {
                    const unsigned long long key = slots[r];
                    idx[obase + r - drop] = (int)(unsigned int)key;
                    if (dist) dist[obase + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
                    if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, qi, r - drop, qr, yb + (size_t)(unsigned int)key * 3);
                }
            }
        }
    }
    KNN_PROBE_MARK(10);
    if (!wave_active) return;
    // ties are ranked again, and the leftovers (exact merge) answered, by the group's two waves on alternate queries
    const bool slowq = qi < N && !fast && !medium;
    const unsigned long long badmask = __ballot(bad), slowmask = __ballot(slowq);
    if ((badmask | slowmask) == 0) return;
    const unsigned int bad32 = ((unsigned int)badmask | (unsigned int)(badmask >> 32)) & (half ? 0xaaaaaaaau : 0x55555555u);  // (the group's two waves take alternate queries)
    if (__builtin_popcount(bad32) <= 3) {
        // ties in the distance among the first kk of a FEW queries (ordinary data: a handful per launch): the whole wave ranks one
        // query's keys again on (distance, index) -- the launch waits for this wave, and this form is the quick one for a single query
        for (unsigned int bm = bad32; bm; bm &= bm - 1) {
            const int j = __builtin_ctz(bm);
            const int qs = grp * 32 + j;
            const int *cj = ctr + qs * 8;
            unsigned long long *sj = reinterpret_cast<unsigned long long *>(lists_all) + (size_t)qs * C::SS;
            knn_rank_ties(reinterpret_cast<const unsigned int *>(k3sm) + (size_t)qs * C::KS,
                          reinterpret_cast<const int *>(k3sm) + (size_t)C::G * 32 * C::KS + (size_t)qs * C::KS,
                          cj[0] + cj[1] + cj[2] + cj[3], kk, sj, lane);
            for (int r = drop + lane; r < kk; r += 64) {
                const unsigned long long key = sj[r];
                idx[((size_t)b * N + q0 + j) * k + r - drop] = (int)(unsigned int)key;
                if (dist) dist[((size_t)b * N + q0 + j) * k + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
                if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, q0 + j, r - drop, xb + (size_t)(q0 + j) * 3, yb + (size_t)(unsigned int)key * 3);
            }
        }
    } else {
        // many tied queries (lattices and duplicated points tie in EVERY query): the wave's tied queries AT ONCE, four lanes per query
        const int j = 2 * (lane >> 2) + half, pl = lane & 3;
        const bool mine = ((bad32 >> j) & 1u) != 0;
        const int qs = grp * 32 + j;
        const int *cj = ctr + qs * 8;
        unsigned long long *sj = reinterpret_cast<unsigned long long *>(lists_all) + (size_t)qs * C::SS;
        knn_rank_ties4(reinterpret_cast<const unsigned int *>(k3sm) + (size_t)qs * C::KS,
                       reinterpret_cast<const int *>(k3sm) + (size_t)C::G * 32 * C::KS + (size_t)qs * C::KS,
                       mine ? cj[0] + cj[1] + cj[2] + cj[3] : 0, kk, sj, pl);
        if (mine)
            for (int r = drop + pl; r < kk; r += 4) {
                const unsigned long long key = sj[r];
                idx[((size_t)b * N + q0 + j) * k + r - drop] = (int)(unsigned int)key;
                if (dist) dist[((size_t)b * N + q0 + j) * k + r - drop] = __builtin_bit_cast(float, (unsigned int)(key >> 32));
                if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, q0 + j, r - drop, xb + (size_t)(q0 + j) * 3, yb + (size_t)(unsigned int)key * 3);
            }
    }
    // leftovers, wave-cooperative (scratch: behind the slots)
    int *wscratch = lists_all + C::G * 32 * C::SS * 2 + wv * 128;
    const unsigned int slow32 = (unsigned int)slowmask | (unsigned int)(slowmask >> 32);
    for (int j = half; j < 32; j += 2) {
        if (!((slow32 >> j) & 1u) || q0 + j >= N) continue;
        float bd;
        int bj;
        __builtin_amdgcn_wave_barrier();
        knn_exact_bruteforce<(C::KKMAX > 32)>(xb + (size_t)(q0 + j) * 3, yb, M, 3, kk, lane, reinterpret_cast<float *>(wscratch), wscratch + 64, bd, bj);
        const int r = lane - drop;
        if (r >= 0 && r < k) {
            idx[((size_t)b * N + q0 + j) * k + r] = bj;
            if (dist) dist[((size_t)b * N + q0 + j) * k + r] = bd;
            if (FEAT) knn_d3_feature_entry(feat, layout, b, N, k, q0 + j, r, xb + (size_t)(q0 + j) * 3, yb + (size_t)bj * 3);
        }
    }
    KNN_PROBE_MARK(11);
}

// dynamic LDS of geometry C for a cloud of M candidates without the optional parts (raw coordinates, medium-path scratch)
template <class C>
size_t knn_f16_d3_core_lds(int M) {
    int CH = (M + 63) / 64 * 64;
    if (CH > kTChunk) CH = kTChunk;
    size_t img = (size_t)CH * 32;
    const size_t keys = (size_t)C::G * 32 * C::KS * 8;
    if (img < keys) img = keys;
    return img + (size_t)C::W * C::CAP * 64 * 4 + (size_t)C::G * 32 * 8 * 4;
}
template <class C, int COMPACT = 0>  // COMPACT = blocks per CU the allocation is held to (0: one block, the whole CU)
fx3d_status launch_knn_f16_d3_geom(const float *x, int N, const float *y, int M, int B, int k, int drop, int32_t *idx,
                                   float *dist, hipStream_t st, float *feat, int layout, int xdiv) {
    int CH = (M + 63) / 64 * 64;
    if (CH > kTChunk) CH = kTChunk;
    const size_t keys = (size_t)C::G * 32 * C::KS * 8;  // distance bits + indices
    // (COMPACT: the LDS image no larger than the key arrays -- larger clouds go through it in several chunks)
    if (COMPACT && (size_t)CH * 32 > keys) CH = (int)(keys / 32 / 64 * 64);
    size_t img = (size_t)CH * 32;
    if (img < keys) img = keys;
    const size_t fixed = (size_t)C::W * C::CAP * 64 * 4 + (size_t)C::G * 32 * 8 * 4;  // lists (exchange, slots) + counters
    // COMPACT: the whole allocation stays below half a CU's LDS (two blocks per CU); the optional parts only if they fit under that
    const size_t raw_limit = COMPACT ? k3_compact_lds(COMPACT ? COMPACT : 1) : 152 * 1024, all_limit = COMPACT ? k3_compact_lds(COMPACT ? COMPACT : 1) : 156 * 1024;
    const int raw_ok = M <= kTRawMax && img + fixed + (size_t)M * 16 <= raw_limit;
    size_t lds = img + fixed + (raw_ok ? (size_t)M * 16 : 0);
    // medium path scratch (id list + merge lists per wave), when it fits next to everything else
    int med_cap = 0, med_off = 0;
    for (int cap = 512; cap >= 128; cap >>= 1)
        if (lds + (size_t)C::W * (cap + 128) * 4 <= all_limit) {
            med_cap = cap; med_off = (int)lds; lds += (size_t)C::W * (cap + 128) * 4;
            break;
        }
    const fx3d_status arc = feat ? ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_f16_d3_kernel<true, C>), 156 * 1024, "knn_f16_d3_kernel<feat>")
                                 : ensure_dynamic_lds(reinterpret_cast<const void *>(&knn_f16_d3_kernel<false, C>), 156 * 1024, "knn_f16_d3_kernel");
    if (arc != FX3D_OK) return arc;
    const int nbx = (N + C::G * 32 - 1) / (C::G * 32);
    const int bpad = B >= 8 ? (B + 7) / 8 * 8 : B;
    if (feat)
        hipLaunchKernelGGL((knn_f16_d3_kernel<true, C>), dim3(nbx * bpad), dim3(C::T), lds, st, x, N, y, M, B, k, drop, idx, dist,
                           CH, (int)img, raw_ok, feat, layout, med_cap, med_off, xdiv);
    else
        hipLaunchKernelGGL((knn_f16_d3_kernel<false, C>), dim3(nbx * bpad), dim3(C::T), lds, st, x, N, y, M, B, k, drop, idx, dist,
                           CH, (int)img, raw_ok, feat, layout, med_cap, med_off, xdiv);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}
fx3d_status launch_knn_f16_d3(const float *x, int N, const float *y, int M, int B, int k, int drop, int32_t *idx,
                              float *dist, hipStream_t st, float *feat = nullptr, int layout = 0, int xdiv = 1) {
    const bool compact = true;
    if (k + drop <= 32) return launch_knn_f16_d3_geom<K3Base>(x, N, y, M, B, k, drop, idx, dist, st, feat, layout, xdiv);
    // (measured, tools/knn_compact_ab.py: 1.3-1.6 x over the wide geometry at M = 1600 ... 8192 too -- several image chunks --, except
    //  where many queries overflow the 88 keys and fall to the exact merge over a large cloud: k + drop > 44 with M > 4096)
    if (k + drop <= K3Mid::KKMAX && compact && (k + drop <= 44 || M <= 4096))
        return launch_knn_f16_d3_geom<K3Mid, 2>(x, N, y, M, B, k, drop, idx, dist, st, feat, layout, xdiv);
    return launch_knn_f16_d3_geom<K3Wide>(x, N, y, M, B, k, drop, idx, dist, st, feat, layout, xdiv);
}


}  // namespace

namespace fx3d {
fx3d_status knn_d3_launch(const float *x, int N, const float *y, int M, int B, int k, int drop, int32_t *idx, float *dist, hipStream_t st,
                          float *feat, int layout, int xdiv) {
    return launch_knn_f16_d3(x, N, y, M, B, k, drop, idx, dist, st, feat, layout, xdiv);
}
}  // namespace fx3d
