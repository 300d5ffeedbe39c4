// The chamfer loss in the reference's own Float32 arithmetic (fx3d_chamfer_loss_pairwise_f32): one translation unit of the chamfer
// family (chamfer.hip: forward, chamfer_bwd.hip: adjoint).
#include "fx3d_common.h"

using namespace fx3d;

// ------------------------------------------------------------------------------------------------
// The loss in the reference's own arithmetic: Float32 pairwise `mean` of the materialised squared
// differences, src/metrics/pcloud.jl:47-48 with Base's mapreduce_impl (pairwise_blocksize 1024) -- bit for bit
// oracle/flux3d_oracle.c: fx3d_oracle_chamfer_loss_pairwise, from the forward's NN indices.  Three small launches:
// (1) one thread per direction walks the recursion and lists the leaves (ranges of < = 1024 elements, in order);
// (2) one wave per leaf: its lanes evaluate the leaf's elements T[e] = (a_e - c_gathered_e)^2 in parallel into LDS, lane 0
//     adds them left to right, exactly the reference's order; (3) one thread per direction adds the leaf sums up the same
//     recursion, divides by Float32(length), x 3.0f0, w1 dA + w2 dB.  Not the hot path (the default stays the Float64 sum
//     finalised inside the nn1 launch): the mode for a host that wants the reference's last bit.
namespace {
constexpr int kPwBlock = 1024;  // Base.pairwise_blocksize

struct PwDir {
    const float *a, *c;         // own cloud, the other cloud
    const int32_t *idx;         // own row -> row of the other cloud
    int R, S;                   // rows of a / of c per batch element
    unsigned long long len;     // D * R * B
    unsigned long long *first;  // [cap + 1] leaf starts (+ sentinel = len)
    float *sums;                // [cap]
    unsigned int *nleaf;        // [1]
};
struct PwParams { PwDir d[2]; int D, B; float w1, w2; float *loss; };

__global__ void pairwise_leaves_kernel(PwParams p) {
    const int dir = threadIdx.x;
    if (dir >= 2) return;
    const PwDir &q = p.d[dir];
    unsigned long long st_f[64], st_l[64];
    int sp = 0;
    unsigned int n = 0;
    st_f[0] = 0; st_l[0] = q.len - 1; sp = 1;
    while (sp) {
        --sp;
        const unsigned long long f = st_f[sp], l = st_l[sp];
        if (l - f < (unsigned long long)kPwBlock) { q.first[n++] = f; continue; }
        const unsigned long long mid = f + ((l - f) >> 1);
        st_f[sp] = mid + 1; st_l[sp] = l; ++sp;   // right half later ...
        st_f[sp] = f; st_l[sp] = mid; ++sp;       // ... the left one first: leaves come out in ascending order
    }
    q.first[n] = q.len;
    *q.nleaf = n;
}

__global__ __launch_bounds__(64) void pairwise_leaf_sums_kernel(PwParams p, unsigned int cap0) {
    __shared__ float t[kPwBlock];
    const int dir = blockIdx.x >= cap0 ? 1 : 0;
    const unsigned int k = dir ? blockIdx.x - cap0 : blockIdx.x;
    const PwDir &q = p.d[dir];
    if (k >= *q.nleaf) return;
    const unsigned long long f = q.first[k];
    const int cnt = (int)(q.first[k + 1] - f);
    const int D = p.D;
    for (int j = threadIdx.x; j < cnt; j += 64) {
        const unsigned long long e = f + j;          // column-major (D, R, B): e = (b R + i) D + d
        const unsigned long long row = e / D;        // = b R + i
        const int d = (int)(e - row * D);
        const unsigned long long b = row / q.R;
        const float v = q.a[e] - q.c[(b * q.S + q.idx[row]) * D + d];
        t[j] = v * v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = t[0];
        if (cnt > 1) {
            v = t[0] + t[1];
            for (int j = 2; j < cnt; ++j) v = v + t[j];
        }
        q.sums[k] = v;
    }
}

__global__ void pairwise_combine_kernel(PwParams p) {
    __shared__ float res[2];
    const int dir = threadIdx.x;
    if (dir < 2) {
        const PwDir &q = p.d[dir];
        // post-order walk of the same recursion; leaves are consumed in the order they were listed
        unsigned long long st_f[64], st_l[64];
        float st_v[64];
        unsigned char st_s[64];
        int sp = 0;
        unsigned int k = 0;
        float ret = 0.0f;
        st_f[0] = 0; st_l[0] = q.len - 1; st_s[0] = 0; sp = 1;
        while (sp) {
            const int top = sp - 1;
            const unsigned long long f = st_f[top], l = st_l[top];
            if (l - f < (unsigned long long)kPwBlock) { ret = q.sums[k++]; --sp; continue; }
            const unsigned long long mid = f + ((l - f) >> 1);
            if (st_s[top] == 0) { st_s[top] = 1; st_f[sp] = f; st_l[sp] = mid; st_s[sp] = 0; ++sp; }
            else if (st_s[top] == 1) { st_v[top] = ret; st_s[top] = 2; st_f[sp] = mid + 1; st_l[sp] = l; st_s[sp] = 0; ++sp; }
            else { ret = st_v[top] + ret; --sp; }
        }
        res[dir] = (ret / (float)q.len) * 3.0f;   // mean(...) * 3.0f0, src/metrics/pcloud.jl:47-48
    }
    __syncthreads();
    if (threadIdx.x == 0) *p.loss = (p.w1 * res[0]) + (p.w2 * res[1]);
}

inline size_t pw_cap(unsigned long long len) { return (size_t)(len / 512 + 2); }
}  // namespace

extern "C" {

fx3d_status fx3d_chamfer_pairwise_workspace_bytes(int32_t N, int32_t M, int32_t B, int32_t D, size_t *bytes) {
    FX3D_REQUIRE(bytes, "fx3d_chamfer_pairwise_workspace_bytes: null output");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0, "fx3d_chamfer_pairwise_workspace_bytes: empty input");
    const size_t c0 = pw_cap((unsigned long long)D * N * B), c1 = pw_cap((unsigned long long)D * M * B);
    *bytes = (c0 + 1 + c1 + 1) * sizeof(unsigned long long) + (c0 + c1) * sizeof(float) + 4 * sizeof(unsigned int);
    return FX3D_OK;
}

fx3d_status fx3d_chamfer_loss_pairwise_f32(const float *x, int32_t N, const float *y, int32_t M, int32_t B, int32_t D,
                                           const int32_t *idx_x, const int32_t *idx_y, float w1, float w2, float *loss_dev,
                                           float *loss_host, void *ws, size_t ws_bytes, fx3d_stream_t s) {
    fx3d_status rc = chamfer_check_shapes("fx3d_chamfer_loss_pairwise_f32", x, N, y, M, B, D);
    if (rc) return rc;
    FX3D_REQUIRE(idx_x && idx_y && loss_dev, "fx3d_chamfer_loss_pairwise_f32: null pointer");
    size_t need = 0;
    (void)fx3d_chamfer_pairwise_workspace_bytes(N, M, B, D, &need);
    if (!ws || ws_bytes < need) {
        set_error("fx3d_chamfer_loss_pairwise_f32: workspace too small (%zu < %zu bytes)", ws ? ws_bytes : (size_t)0, need);
        return FX3D_ERR_WORKSPACE;
    }
    const unsigned long long l0 = (unsigned long long)D * N * B, l1 = (unsigned long long)D * M * B;
    const size_t c0 = pw_cap(l0), c1 = pw_cap(l1);
    PwParams p{};
    unsigned long long *u = reinterpret_cast<unsigned long long *>(ws);
    float *f = reinterpret_cast<float *>(u + c0 + 1 + c1 + 1);
    unsigned int *cnt = reinterpret_cast<unsigned int *>(f + c0 + c1);
    p.d[0] = PwDir{x, y, idx_x, N, M, l0, u, f, cnt};
    p.d[1] = PwDir{y, x, idx_y, M, N, l1, u + c0 + 1, f + c0, cnt + 1};
    p.D = D; p.B = B; p.w1 = w1; p.w2 = w2; p.loss = loss_dev;
    hipStream_t st = as_stream(s);
    FX3D_REQUIRE(c0 + c1 < (1ull << 31), "fx3d_chamfer_loss_pairwise_f32: problem too large");
    hipLaunchKernelGGL(pairwise_leaves_kernel, dim3(1), dim3(64), 0, st, p);
    hipLaunchKernelGGL(pairwise_leaf_sums_kernel, dim3((unsigned)(c0 + c1)), dim3(64), 0, st, p, (unsigned int)c0);
    hipLaunchKernelGGL(pairwise_combine_kernel, dim3(1), dim3(64), 0, st, p);
    FX3D_LAUNCH_CHECK();
    if (loss_host) {
        FX3D_HIP(hipMemcpyAsync(loss_host, loss_dev, sizeof(float), hipMemcpyDeviceToHost, st));
        FX3D_HIP(hipStreamSynchronize(st));
    }
    return FX3D_OK;
}

}  // extern "C"

