// Internal helpers shared by the translation units of libflux3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "flux3d_hip.h"

namespace fx3d {

void set_error(const char *fmt, ...);

inline fx3d_status hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    if (e == hipErrorOutOfMemory) return FX3D_ERR_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return FX3D_ERR_NO_DEVICE;
    return FX3D_ERR_HIP;
}

#define FX3D_HIP(call)                                                          \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) return fx3d::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define FX3D_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            fx3d::set_error(__VA_ARGS__);        \
            return FX3D_ERR_INVALID_ARG;         \
        }                                        \
    } while (0)

// checks the launch itself (configuration errors); execution errors surface at the next sync
#define FX3D_LAUNCH_CHECK() FX3D_HIP(hipGetLastError())

inline hipStream_t as_stream(fx3d_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Opt a kernel in to more than 64 KiB of dynamic LDS, once per (kernel, device): the attribute is per device, and
// a process may drive several (runtime.hip).
// Library-owned arrival counter for a fused "last block finalises" reduction: zero at hand-out, the kernel returns
// it to zero (chamfer.hip).  A launch that `st` is capturing into a graph gets a slot of its own, never reused.
// nullptr + *rc on allocation failure.
unsigned int *ticket_slot(fx3d_status *rc, hipStream_t st);
// Words per ticket slot: the arrival counter at [0] + 16 first-level counters, one per 64-byte line (ticket_arrive_last).
constexpr int kTicketGroups = 16;
constexpr int kTicketStride = 16 * (kTicketGroups + 1);
fx3d_status ensure_dynamic_lds(const void *kernel, int bytes, const char *name);
// chamfer.hip's argument check and forward driver (loss with batch size Bg, optional indices), for chamfer_bwd.hip
fx3d_status chamfer_check_shapes(const char *fn, const void *x, int N, const void *y, int M, int B, int D);
fx3d_status chamfer_forward(const float *x, int N, const float *y, int M, int B, int D, float *loss_dev, long long Bg, float w1,
                            float w2, int32_t *idx_x, int32_t *idx_y, void *ws, size_t ws_bytes, hipStream_t st, const char *fn);
// compute units of the calling thread's current device (cached per device; 256 on an MI355X in SPX mode, fewer in the
// partitioned modes): the launch plans size their rounds of blocks with it instead of a constant
int device_cus();

// ---- optional per-kernel event timing (runtime.hip) -------------------------------------------
bool profile_on();
void profile_mark(const char *name, hipStream_t st, bool begin);
struct ProfileScope {  // brackets one dominant-kernel launch when profiling is enabled
    const char *name;
    hipStream_t st;
    bool on;
    ProfileScope(const char *n, hipStream_t s) : name(n), st(s), on(profile_on()) {
        if (on) profile_mark(name, st, true);
    }
    ~ProfileScope() {
        if (on) profile_mark(name, st, false);
    }
};

// ---- variant switches (runtime.hip): explicit options instead of environment reads on the launch path.  The FX3D_*
//      environment variables of the same names only seed the defaults, once, at the first use of the library;
//      fx3d_set_option / fx3d_get_option (flux3d_hip.h) change / read them afterwards.  Process-wide, atomic.
enum Opt {
    OPT_NN1_VARIANT,          // 3 = fp16-split MFMA filter + exact re-scan (default), 0 = the exact VALU loop (A/B)
    OPT_NN1_NOSPLIT,          // 1: never split the candidates of a cloud over blocks
    OPT_BWD_GLOBAL_ATOMICS,   // 1: chamfer adjoint with global float atomics instead of the LDS accumulator
    OPT_KNN_NO_MFMA,          // 1: wave-per-query kernels only
    OPT_KNN_NO_PREPASS,       // 1: fx3d_knn_ws ignores its scratch (no per-cloud pre-pass)
    OPT_KNN_SLICES,           // fx3d_knn_ws: candidate slices per cloud: 0 = automatic, 1 = never, 2 / 4 / 8 = forced
    OPT_EDGECONV_UNFUSED,     // 1: EdgeConv graph build as search + feature kernels
    OPT_LAP_BWD_SCATTER,      // 1: fx3d_laplacian_loss_bwd as the scatter with float atomics (any CSR) instead of the gather (symmetric structure)
    OPT_CDF_MULTIBLOCK_FROM,  // faces per mesh from which the sampling CDF takes the multi-block path (0 = the built-in limit)
    OPT_NN1_TINY_MPAIRS,      // D = 3 nn1 / chamfer: problems of at most this many MILLION ordered pair evaluations (2 B N M) run on the
                              // exact small-problem kernel (nn1_tiny_kernel) instead of the fp16-filter kernel; 0 = never
    OPT_MESH_MAX_BLOCKS,      // grid cap of the grid-stride mesh kernels (areas, losses, adjoints), 256-thread blocks; 0 = automatic
    OPT_NN1_PRUNE,            // 1 (default): chamfer / nn1 with a workspace run the fp16 kernel with spatial pruning (one-chunk plans); 0 = never
    OPT_COUNT
};
int opt(Opt o);

constexpr int kWave = 64;  // gfx950 wavefront

#ifdef __HIPCC__
// Arrival of one block (ONE thread calls this, after its partial result is stored and drained) at a launch's ticket: true for
// the last of `nblocks` arrivals.  Same-address device atomics serialise at ~9 ns each: 1024 blocks that finish together -- the
// grid-stride mesh kernels -- spent 9 us of a 29 us launch queueing at one counter (round 4: edge_loss 24 / 29 / 39 / 61 us at
// 512 / 1024 / 2048 / 4096 blocks).  Beyond 64 blocks the arrival is two-level: block b counts at first-level counter b % 16
// (its own cache line), the last arriver of a group resets it and counts at the ticket itself, the last of the 16 groups is
// the launch's last block.  The caller resets ticket[0] when it is done (as before).
__device__ __forceinline__ bool ticket_arrive_last(unsigned int *ticket, unsigned int nblocks, unsigned int blk) {
    if (nblocks <= 64u) return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1u;
    const unsigned int g = blk % (unsigned int)kTicketGroups;
    const unsigned int gsize = (nblocks - g + (unsigned int)kTicketGroups - 1u) / (unsigned int)kTicketGroups;
    unsigned int *c1 = ticket + 16 * (g + 1u);
    if (__hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gsize - 1u) return false;
    __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)kTicketGroups - 1u;
}
#endif

// ||(v2-v1) x (v3-v1)|| / 2 : _lg_cross (src/rep/utils.jl:4-21), _norm (:29),
// compute_faces_areas_packed (src/rep/mesh.jl:772-779).
__device__ __forceinline__ float tri_area(const float *__restrict__ v1, const float *__restrict__ v2,
                                          const float *__restrict__ v3) {
    const float p0 = v1[0], p1 = v1[1], p2 = v1[2];
    const float a1 = v2[0] - p0, a2 = v2[1] - p1, a3 = v2[2] - p2;
    const float b1 = v3[0] - p0, b2 = v3[1] - p1, b3 = v3[2] - p2;
    const float c1 = (a2 * b3) - (a3 * b2);
    const float c2 = (a3 * b1) - (a1 * b3);
    const float c3 = (a1 * b2) - (a2 * b1);
    const float s = ((c1 * c1) + (c2 * c2)) + (c3 * c3);
    return sqrtf(s) / 2.0f;
}


// ---- device-side reductions (wave64) -------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over a block of NT threads (NT multiple of 64, <= 1024); result valid in thread 0.
// Fixed order: lanes by shuffle tree, then waves 0..n-1 sequentially => deterministic.
template <int NT>
__device__ inline double block_sum(double v, double *smem /* >= NT/64 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) smem[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += smem[i];
    }
    return r;
}

struct __attribute__((packed, aligned(4))) P3 { float x, y, z; };  // one point of a (3, N) cloud: a 12-byte load

// wave64 min / max by DPP (VALU speed; __shfl_xor would go through the LDS crossbar)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_min_f(float v) {
    v = fminf(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fminf(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fminf(v, dpp_mov<0x141>(v));  // row_half_mirror
    v = fminf(v, dpp_mov<0x140>(v));  // row_mirror  -> every lane of a 16-lane row holds the row minimum
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}
__device__ __forceinline__ float wave_max_f(float v) { return -wave_min_f(-v); }
// the same reductions finished by DPP too (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): the result is
// in lane 63 only -- six VALU operations per value, no readlanes
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_keep(float v) {  // lanes outside ROWMASK (or without a source) keep v
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
// (minima and maxima on order-preserving integer keys: integer min / max take the DPP operand directly, float min / max
//  would canonicalise both operands first -- three instructions per step instead of one)
__device__ __forceinline__ int fkey(float v) {  // signed-integer order == float order (-0 < +0; NaNs beyond the infinities)
    const int b = __builtin_bit_cast(int, v);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float fkey_inv(int k) { return __builtin_bit_cast(float, k ^ ((k >> 31) & 0x7fffffff)); }
template <int CTRL>
__device__ __forceinline__ int dpp_movi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_keepi(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xF, false); }
template <bool MAX>
__device__ __forceinline__ int imm(int a, int b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); }
template <bool MAX>
__device__ __forceinline__ int row_mm_key(int k) {  // every lane of a 16-lane row ends with the row's min / max
    k = imm<MAX>(k, dpp_movi<0xB1>(k));
    k = imm<MAX>(k, dpp_movi<0x4E>(k));
    k = imm<MAX>(k, dpp_movi<0x141>(k));
    k = imm<MAX>(k, dpp_movi<0x140>(k));
    return k;
}
template <bool MAX>
__device__ __forceinline__ float wave_mm_l63(float v) {
    int k = row_mm_key<MAX>(fkey(v));
    k = imm<MAX>(k, dpp_keepi<0x142, 0xA>(k));  // rows 1, 3 += rows 0, 2
    k = imm<MAX>(k, dpp_keepi<0x143, 0xC>(k));  // rows 2, 3 += row 1 (which holds rows 0..1)
    return fkey_inv(k);
}
__device__ __forceinline__ float wave_min_l63(float v) { return wave_mm_l63<false>(v); }
__device__ __forceinline__ float wave_max_l63(float v) { return wave_mm_l63<true>(v); }
__device__ __forceinline__ float wave_sum_l63(float v) {  // fixed order: deterministic
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    v = v + dpp_mov<0x141>(v);
    v = v + dpp_mov<0x140>(v);
    const float a = dpp_keep<0x142, 0xA>(v);
    v = ((threadIdx.x >> 4) & 1) ? v + a : v;  // (rows 0 and 2 keep their own value: v + v must not happen)
    const float b = dpp_keep<0x143, 0xC>(v);
    v = ((threadIdx.x >> 5) & 1) ? v + b : v;
    return v;
}
// the same for a double (two 32-bit DPP moves per step; __shfl_down would be two ds_bpermute round trips per step): lane 63
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, CTRL, 0xF, 0xF, true);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_keep64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int l0 = (int)(unsigned int)b, h0 = (int)(unsigned int)(b >> 32);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(l0, l0, CTRL, ROWMASK, 0xF, false);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(h0, h0, CTRL, ROWMASK, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_l63_f64(double v) {  // fixed order: deterministic
    v = v + dpp_mov64<0xB1>(v);
    v = v + dpp_mov64<0x4E>(v);
    v = v + dpp_mov64<0x141>(v);
    v = v + dpp_mov64<0x140>(v);
    const double a = dpp_keep64<0x142, 0xA>(v);
    v = ((threadIdx.x >> 4) & 1) ? v + a : v;
    const double b = dpp_keep64<0x143, 0xC>(v);
    v = ((threadIdx.x >> 5) & 1) ? v + b : v;
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {  // the same butterfly with adds (fixed order: deterministic)
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    v = v + dpp_mov<0x141>(v);
    v = v + dpp_mov<0x140>(v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (a + b) + (c + d);
}


// The robust-range rounds run when cinf^2 > kRobustGate * variance (the farthest point more than 8 sigma out; the sampled
// variance contains the outliers themselves, so one far point among N makes the ratio ~N at most: the gate cannot be much higher),
// and they stop after the first one unless the range it finds is kRobustHarm times below cinf: the folded norms keep the
// filter sharp with a point up to ~10^3 x the extent away, only beyond that does the bulk sink below fp16's resolution.
constexpr float kRobustGate = 64.0f;
constexpr float kRobustHarm = 8.0f;  // (round 0 measures about a mean the outliers pulled: one extreme point among N gives cinf / rng ~ N / 32)

// The robust range of a D = 3 cloud with outliers (nn1_f16_kernel, knn_f16_d3_kernel): clean clouds never call it.
// NT threads per block; `parked`: the raw points sit in LDS as float4 (LINEAR: park[pt]; otherwise in the first piece
// of their fp16-image slot, park[((pt >> 5) * 2) * 32 + (pt & 31)]), else they are read from `cb`; red: >= 8 NT/64 floats.  Returns (mu, rng): the re-centred mean and 16 x the mean max-norm
// deviation of the points within the previous range.  All threads of the block call it together (it synchronises).
template <int NT, bool LINEAR>
__device__ __forceinline__ float4 robust_range3(const float *__restrict__ cb, int NC, bool parked, const float4 *park,
                                               float *red, float mu0, float mu1, float mu2, float cinf) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float mu[3] = {mu0, mu1, mu2};
    float rng = cinf;
    {
        // round 0: deviations from the mean of all points.  If that finds outliers (rng < cinf) the mean itself was pulled
        // by them: round 1 re-centres on the points inside the range, round 2 measures the deviations about that centre.
        for (int round = 0; round < 3; ++round) {
            const bool trim = round > 0;
            float dev = 0.0f, cntf = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
            auto take = [&](float x, float y, float z) {
                const float dv = fmaxf(fmaxf(fabsf(x - mu[0]), fabsf(y - mu[1])), fabsf(z - mu[2]));
                if (!trim || dv <= rng) { dev += dv; cntf += 1.0f; sx += x; sy += y; sz += z; }
            };
            for (int pt = tid; pt < NC; pt += NT) {
                if (parked) {
                    const float4 r = park[LINEAR ? pt : ((pt >> 5) * 2) * 32 + (pt & 31)];  // parked by this thread in the cloud pass
                    take(r.x, r.y, r.z);
                } else {
                    const P3 r = *reinterpret_cast<const P3 *>(cb + (size_t)pt * 3);
                    take(r.x, r.y, r.z);
                }
            }
            // (round 0 needs the deviation only: the count is NC; round 1 the count and the sums; round 2 deviation + count)
            float v5[5] = {dev, cntf, sx, sy, sz};
            const int k0 = round == 1 ? 1 : 0, k1 = round == 0 ? 1 : (round == 1 ? 5 : 2);
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (k >= k0 && k < k1) v5[k] = wave_sum_l63(v5[k]);  // DPP: the total is in lane 63
            __syncthreads();  // (red was read above)
            if (lane == 63) {
                float4 *r4 = reinterpret_cast<float4 *>(red);
                r4[wv * 2] = float4{v5[0], v5[1], v5[2], v5[3]};
                r4[wv * 2 + 1] = float4{v5[4], 0.0f, 0.0f, 0.0f};
            }
            __syncthreads();
            float t5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            {
                const float4 *r4 = reinterpret_cast<const float4 *>(red);
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) {
                    const float4 a = r4[w * 2];
                    t5[0] = t5[0] + a.x; t5[1] = t5[1] + a.y; t5[2] = t5[2] + a.z; t5[3] = t5[3] + a.w;
                    if (round == 1) t5[4] = t5[4] + r4[w * 2 + 1].x;
                }
            }
            if (round == 0) t5[1] = (float)NC;
            if (!(t5[1] > 0.0f)) break;                      // (uniform)
            if (round == 1) {  // re-centre on the points inside the range; measured again about the new centre next round
                mu[0] = t5[2] / t5[1]; mu[1] = t5[3] / t5[1]; mu[2] = t5[4] / t5[1];
                continue;
            }
            const float r16 = 16.0f * (t5[0] / t5[1]);
            if (r16 > 0.0f && r16 < rng) rng = r16;
            if (round == 0 && !(rng * kRobustHarm < cinf)) { rng = cinf; break; }  // no (harmful) outliers: the plain mean and cinf stand
        }
    }
    return float4{mu[0], mu[1], mu[2], rng};
}

// Sum `n` doubles written by a previous kernel, single block of 256, fixed order.
// (file-local copy per translation unit: no relocatable device code needed)
__global__ static void reduce_partials_kernel(const double *__restrict__ partials, int64_t n,
                                              double *__restrict__ out) {
    __shared__ double sm[4];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += partials[i];
    double r = block_sum<256>(acc, sm);
    if (threadIdx.x == 0) out[0] = r;
}

}  // namespace fx3d
