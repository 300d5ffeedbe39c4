// Host-pointer convenience variants of the ops ("device-pointer + host-pointer variants").
//
// The reference's CPU methods take plain `Array`s (src/metrics/pcloud.jl:54-70, src/models/dgcnn.jl:3-7,
// src/transforms/mesh_func.jl:21-58, src/metrics/mesh.jl:9-32): a host that has not adopted the device array type yet
// calls these with its own host buffers -- column-major, exactly Julia's memory -- and gets host results back.  Each call
// stages its inputs into device scratch owned by the calling thread (grow-only, reused across calls, freed at thread
// exit), runs the SAME device entry points as everything else (no CPU code path), copies the outputs back and
// synchronises: synchronous like the reference's own functions.  The PCIe copies are inside the call, so these are
// the convenient form, not the fast one (C2: ~156 us per call against ~52 us with resident clouds).
#include <cstring>
#include <vector>

#include "fx3d_common.h"

using namespace fx3d;

namespace {

struct Scratch {  // one per host thread and device: a few grow-only device buffers
    struct Buf { void *p = nullptr; size_t n = 0; };
    Buf b[8];
    int dev = -1;
    hipStream_t st = nullptr;
    ~Scratch() { release(); }
    void release() {
        for (Buf &x : b) { if (x.p) (void)hipFree(x.p); x = Buf{}; }
        if (st) { (void)hipStreamDestroy(st); st = nullptr; }
    }
    fx3d_status bind() {  // (the calling thread's current device; scratch follows it)
        int d = 0;
        FX3D_HIP(hipGetDevice(&d));
        if (d != dev) { release(); dev = d; }
        if (!st) FX3D_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        return FX3D_OK;
    }
    fx3d_status get(int i, size_t bytes, void **out) {
        Buf &x = b[i];
        if (bytes > x.n) {
            if (x.p) FX3D_HIP(hipFree(x.p));
            x = Buf{};
            const size_t want = bytes + bytes / 4 + 256;
            FX3D_HIP(hipMalloc(&x.p, want));
            x.n = want;
        }
        *out = x.p;
        return FX3D_OK;
    }
};
thread_local Scratch g_scr;

#define FX3D_TRY(call) do { const fx3d_status rc__ = (call); if (rc__) return rc__; } while (0)

template <typename T>
fx3d_status up(int slot, const T *host, size_t count, T **dev) {
    void *p = nullptr;
    FX3D_TRY(g_scr.get(slot, count * sizeof(T), &p));
    FX3D_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, g_scr.st));
    *dev = static_cast<T *>(p);
    return FX3D_OK;
}
template <typename T>
fx3d_status room(int slot, size_t count, T **dev) {
    void *p = nullptr;
    FX3D_TRY(g_scr.get(slot, count * sizeof(T), &p));
    *dev = static_cast<T *>(p);
    return FX3D_OK;
}
template <typename T>
fx3d_status down(T *host, const T *dev, size_t count) {
    if (host) FX3D_HIP(hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, g_scr.st));
    return FX3D_OK;
}
fx3d_status finish() {
    FX3D_HIP(hipStreamSynchronize(g_scr.st));
    return FX3D_OK;
}

}  // namespace

extern "C" {

// chamfer_distance(A, B; w1, w2) / _nearest_neighbors on host arrays: x (D,N,B), y (D,M,B) column-major Float32.
// loss (1 float), idx_x (N*B), idx_y (M*B) int32 0-based: any of the three outputs may be NULL (not all).
fx3d_status fx3d_chamfer_distance_host(const float *x, int32_t N, const float *y, int32_t M, int32_t B, int32_t D, float w1,
                                       float w2, float *loss, int32_t *idx_x, int32_t *idx_y) {
    FX3D_REQUIRE(x && y && (loss || idx_x || idx_y), "fx3d_chamfer_distance_host: null pointer");
    FX3D_REQUIRE(N > 0 && M > 0 && B > 0 && D > 0, "fx3d_chamfer_distance_host: empty input (N=%d M=%d B=%d D=%d)", N, M, B, D);
    FX3D_TRY(g_scr.bind());
    float *dx, *dy, *dl;
    int32_t *dix = nullptr, *diy = nullptr;
    FX3D_TRY(up(0, x, (size_t)D * N * B, &dx));
    FX3D_TRY(up(1, y, (size_t)D * M * B, &dy));
    FX3D_TRY(room(2, 1, &dl));
    if (idx_x || idx_y) {
        FX3D_TRY(room(3, (size_t)N * B, &dix));
        FX3D_TRY(room(4, (size_t)M * B, &diy));
    }
    size_t wsb = 0;
    FX3D_TRY(fx3d_chamfer_workspace_bytes(N, M, B, D, &wsb));
    void *ws = nullptr;
    FX3D_TRY(g_scr.get(5, wsb, &ws));
    FX3D_TRY(fx3d_chamfer_fwd(dx, N, dy, M, B, D, w1, w2, dl, nullptr, dix, diy, ws, wsb, reinterpret_cast<fx3d_stream_t>(g_scr.st)));
    FX3D_TRY(down(loss, dl, 1));
    FX3D_TRY(down(idx_x, dix, (size_t)N * B));
    FX3D_TRY(down(idx_y, diy, (size_t)M * B));
    return finish();
}

// knn(KDTree(y), x, k, true) per batch element (src/models/dgcnn.jl:3-7): y == NULL -> self search.  idx (k,N,B) int32
// 0-based, dist (k,N,B) squared Float32 distances (optional).
fx3d_status fx3d_knn_host(const float *x, int32_t N, const float *y, int32_t M, int32_t B, int32_t D, int32_t k,
                          int32_t drop_first, int32_t *idx, float *dist) {
    FX3D_REQUIRE(x && idx, "fx3d_knn_host: null pointer");
    FX3D_REQUIRE(N > 0 && B > 0 && D > 0 && k > 0, "fx3d_knn_host: empty input");
    if (!y) M = N;
    FX3D_REQUIRE(M > 0, "fx3d_knn_host: empty candidate cloud");
    FX3D_TRY(g_scr.bind());
    float *dx, *dy = nullptr, *dd = nullptr;
    int32_t *di;
    FX3D_TRY(up(0, x, (size_t)D * N * B, &dx));
    if (y) FX3D_TRY(up(1, y, (size_t)D * M * B, &dy));
    FX3D_TRY(room(3, (size_t)k * N * B, &di));
    if (dist) FX3D_TRY(room(2, (size_t)k * N * B, &dd));
    size_t wsb = 0;
    FX3D_TRY(fx3d_knn_workspace_bytes(N, M, B, D, k, drop_first, &wsb));
    void *ws = nullptr;
    if (wsb) FX3D_TRY(g_scr.get(5, wsb + 256, &ws));
    if (ws) ws = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    FX3D_TRY(fx3d_knn_ws(dx, N, dy ? dy : dx, M, B, D, k, drop_first, di, dd, ws, wsb, reinterpret_cast<fx3d_stream_t>(g_scr.st)));
    FX3D_TRY(down(idx, di, (size_t)k * N * B));
    FX3D_TRY(down(dist, dd, (size_t)k * N * B));
    return finish();
}

// sample_points(m, n; eps) (src/transforms/mesh_func.jl:21-58) on host arrays: verts_padded (3,Vmax,B) Float32,
// faces_padded (3,Fmax,B) int32 0-based mesh-local, faces_len (B) int32; out (3,n,B).  Device Philox stream keyed by seed.
fx3d_status fx3d_sample_points_host(const float *verts_padded, int32_t Vmax, const int32_t *faces_padded, int32_t Fmax,
                                    const int32_t *faces_len, int32_t B, int32_t n, double eps, uint64_t seed, float *out) {
    FX3D_REQUIRE(verts_padded && faces_padded && faces_len && out, "fx3d_sample_points_host: null pointer");
    FX3D_REQUIRE(Vmax > 0 && Fmax > 0 && B > 0 && n > 0, "fx3d_sample_points_host: empty input");
    FX3D_TRY(g_scr.bind());
    float *dv, *dout;
    int32_t *df, *dl;
    FX3D_TRY(up(0, verts_padded, (size_t)3 * Vmax * B, &dv));
    FX3D_TRY(up(1, faces_padded, (size_t)3 * Fmax * B, &df));
    FX3D_TRY(up(2, faces_len, (size_t)B, &dl));
    FX3D_TRY(room(3, (size_t)3 * n * B, &dout));
    size_t wsb = 0;
    FX3D_TRY(fx3d_sample_points_workspace_bytes(Fmax, B, &wsb));
    void *ws = nullptr;
    FX3D_TRY(g_scr.get(5, wsb, &ws));
    FX3D_TRY(fx3d_sample_points(dv, Vmax, df, Fmax, dl, B, n, eps, seed, dout, nullptr, nullptr, nullptr, ws, wsb,
                                reinterpret_cast<fx3d_stream_t>(g_scr.st)));
    FX3D_TRY(down(out, dout, (size_t)3 * n * B));
    return finish();
}

// edge_loss(m, target) / laplacian_loss(m) (src/metrics/mesh.jl:9-32) on host arrays: verts (3,V) packed, edges (E,2)
// int32 0-based column-major; the Laplacian as 0-based CSR (fx3d_build_laplacian_csr).
fx3d_status fx3d_edge_loss_host(const float *verts, int64_t V, const int32_t *edges, int64_t E, float target, float *loss) {
    FX3D_REQUIRE(verts && edges && loss, "fx3d_edge_loss_host: null pointer");
    FX3D_REQUIRE(V > 0 && E > 0, "fx3d_edge_loss_host: empty input");
    FX3D_TRY(g_scr.bind());
    float *dv, *dl;
    int32_t *de;
    FX3D_TRY(up(0, verts, (size_t)3 * V, &dv));
    FX3D_TRY(up(1, edges, (size_t)2 * E, &de));
    FX3D_TRY(room(2, 1, &dl));
    size_t wsb = 0;
    FX3D_TRY(fx3d_mesh_loss_workspace_bytes(E, &wsb));
    void *ws = nullptr;
    FX3D_TRY(g_scr.get(5, wsb, &ws));
    FX3D_TRY(fx3d_edge_loss(dv, V, de, E, target, dl, nullptr, ws, wsb, reinterpret_cast<fx3d_stream_t>(g_scr.st)));
    FX3D_TRY(down(loss, dl, 1));
    return finish();
}

fx3d_status fx3d_laplacian_loss_host(const float *verts, int64_t V, const int32_t *rowptr, const int32_t *colind,
                                     const float *vals, float *loss) {
    FX3D_REQUIRE(verts && rowptr && colind && vals && loss, "fx3d_laplacian_loss_host: null pointer");
    FX3D_REQUIRE(V > 0, "fx3d_laplacian_loss_host: empty input");
    FX3D_TRY(g_scr.bind());
    const int64_t nnz = rowptr[V];
    FX3D_REQUIRE(nnz > 0, "fx3d_laplacian_loss_host: empty Laplacian");
    float *dv, *dvals, *dl;
    int32_t *dr, *dc;
    FX3D_TRY(up(0, verts, (size_t)3 * V, &dv));
    FX3D_TRY(up(1, rowptr, (size_t)V + 1, &dr));
    FX3D_TRY(up(2, colind, (size_t)nnz, &dc));
    FX3D_TRY(up(3, vals, (size_t)nnz, &dvals));
    FX3D_TRY(room(4, 1, &dl));
    size_t wsb = 0;
    FX3D_TRY(fx3d_mesh_loss_workspace_bytes(V, &wsb));
    void *ws = nullptr;
    FX3D_TRY(g_scr.get(5, wsb, &ws));
    FX3D_TRY(fx3d_laplacian_loss(dv, V, dr, dc, dvals, dl, nullptr, ws, wsb, reinterpret_cast<fx3d_stream_t>(g_scr.st)));
    FX3D_TRY(down(loss, dl, 1));
    return finish();
}

}  // extern "C"
