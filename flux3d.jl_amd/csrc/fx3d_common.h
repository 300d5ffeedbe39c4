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
fx3d_status ensure_dynamic_lds(const void *kernel, int bytes, const char *name);

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

constexpr int kWave = 64;  // gfx950 wavefront

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
