// pointcloud_to_voxel (src/conversions.jl:91-131) for device-resident clouds.
//
// The reference normalises every cloud by ONE scalar min and max (maximum over coordinates and points,
// :94-96), lays a res^3 lattice of centres (i + 0.5)/res, i = 1..res (Float64, note the 1-based i: the
// lattice is shifted one cell up, :102-113), finds the nearest cloud point of every centre with a KD-tree,
// recomputes that distance in Float64 and marks the voxel when it is <= 0.6/res^2 (:125-129).
//
// "nearest point within radius" == "any point within radius", so instead of res^3 x N distance
// evaluations the kernel walks the cloud once: a point can only mark centres closer than sqrt(0.6)/res
// (0.775 cell widths) per axis, i.e. at most 2 lattice indices per axis (4 are tested).  The candidate
// centres are tested with the reference's own Float64 arithmetic (unfused, dimension order), so occupancy
// is bit-identical; the work is O(N), the traffic is the cloud read + the res^3*B grid write.
#include "fx3d_common.h"

namespace fx3d {
namespace {

constexpr int kVoxThreads = 256;

// one block per cloud: scalar min / max over all 3N coordinates (src/conversions.jl:94-95)
__global__ __launch_bounds__(kVoxThreads) void cloud_range_kernel(const float *__restrict__ p, int N,
                                                                  float *__restrict__ range) {
    const int b = blockIdx.x;
    const float *pb = p + (size_t)b * N * 3;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    bool nan = false;
    for (int e = threadIdx.x; e < 3 * N; e += kVoxThreads) {
        const float v = pb[e];
        nan |= (v != v);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    __shared__ float slo[kVoxThreads], shi[kVoxThreads];
    __shared__ int snan;
    if (threadIdx.x == 0) snan = 0;
    __syncthreads();
    slo[threadIdx.x] = lo;
    shi[threadIdx.x] = hi;
    if (nan) snan = 1;
    __syncthreads();
    for (int s = kVoxThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            slo[threadIdx.x] = fminf(slo[threadIdx.x], slo[threadIdx.x + s]);
            shi[threadIdx.x] = fmaxf(shi[threadIdx.x], shi[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // Julia's maximum/minimum propagate NaN; a NaN range yields an all-NaN cloud and an empty grid
        const float q = __builtin_nanf("");
        range[2 * b] = snan ? q : slo[0];
        range[2 * b + 1] = snan ? q : shi[0];
    }
}

__global__ __launch_bounds__(kVoxThreads) void voxel_scatter_kernel(const float *__restrict__ p, int N, int B,
                                                                    int res, const float *__restrict__ range,
                                                                    float *__restrict__ vox) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * kVoxThreads + threadIdx.x;
    if (i >= N) return;
    const float lo = range[2 * b], hi = range[2 * b + 1];
    const float span = hi - lo;
    const float *q = p + ((size_t)b * N + i) * 3;
    // cloud = (p .- verts_min) ./ (verts_max - verts_min), Float32 (:96)
    const double c0 = (double)((q[0] - lo) / span);
    const double c1 = (double)((q[1] - lo) / span);
    const double c2 = (double)((q[2] - lo) / span);
    if (!(c0 == c0) || !(c1 == c1) || !(c2 == c2)) return;
    const double rres = (double)res;
    const double thr = 0.6 / (double)((long long)res * res);  // 0.6 / (resolution * resolution) (:127)
    // lattice indices whose centre can be within the radius: |(k + 0.5)/res - c| <= 0.775/res
    const int k0 = (int)floor(c0 * rres - 1.3), k1 = (int)floor(c1 * rres - 1.3), k2 = (int)floor(c2 * rres - 1.3);
    float *vb = vox + (size_t)b * res * res * res;
    for (int a = 0; a < 4; ++a) {
        const int x = k0 + a;
        if (x < 1 || x > res) continue;
        const double dx = ((double)x + 0.5) / rres - c0;
        const double dx2 = dx * dx;
        for (int bb = 0; bb < 4; ++bb) {
            const int y = k1 + bb;
            if (y < 1 || y > res) continue;
            const double dy = ((double)y + 0.5) / rres - c1;
            const double dxy = dx2 + dy * dy;
            for (int cc = 0; cc < 4; ++cc) {
                const int z = k2 + cc;
                if (z < 1 || z > res) continue;
                const double dz = ((double)z + 0.5) / rres - c2;
                const double d = dxy + dz * dz;  // sum(dists_vec .^ 2, dims = 1): ((dx^2 + dy^2) + dz^2)
                // reshape(dists, res, res, res, B): the innermost loop variable z is the first dimension
                if (d <= thr) vb[((size_t)(x - 1) * res + (y - 1)) * res + (z - 1)] = 1.0f;
            }
        }
    }
}

}  // namespace
}  // namespace fx3d

using namespace fx3d;

extern "C" {

fx3d_status fx3d_voxel_workspace_bytes(int32_t B, size_t *bytes) {
    FX3D_REQUIRE(bytes && B > 0, "fx3d_voxel_workspace_bytes: bad arguments");
    *bytes = sizeof(float) * 2 * (size_t)B;
    return FX3D_OK;
}

fx3d_status fx3d_pointcloud_to_voxel(const float *points, int32_t N, int32_t B, int32_t res, float *voxels,
                                     void *ws, size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(points && voxels && ws, "fx3d_pointcloud_to_voxel: null pointer");
    FX3D_REQUIRE(N > 0 && B > 0 && res > 0 && res <= 1024, "fx3d_pointcloud_to_voxel: bad sizes");
    FX3D_REQUIRE(ws_bytes >= sizeof(float) * 2 * (size_t)B, "fx3d_pointcloud_to_voxel: workspace too small");
    hipStream_t st = as_stream(s);
    float *range = static_cast<float *>(ws);
    FX3D_HIP(hipMemsetAsync(voxels, 0, sizeof(float) * (size_t)res * res * res * B, st));
    ProfileScope prof("pointcloud_to_voxel", st);
    hipLaunchKernelGGL(cloud_range_kernel, dim3(B), dim3(kVoxThreads), 0, st, points, N, range);
    hipLaunchKernelGGL(voxel_scatter_kernel, dim3((N + kVoxThreads - 1) / kVoxThreads, B), dim3(kVoxThreads), 0, st,
                       points, N, B, res, range, voxels);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

}  // extern "C"
