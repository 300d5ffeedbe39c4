"""ctypes front-end of the CPU ORACLE (oracle/flux3d_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py -- never from the product package ``flux3d.jl_amd``.

Array conventions mirror the reference (Julia, column-major): a point cloud is a numpy array of
shape ``(D, N, B)`` in Fortran order (or anything ``np.asfortranarray`` can turn into one), which
is byte-identical to Julia's ``Array{Float32,3}`` of ``src/rep/pcloud.jl:25-28``.  Indices are
0-based here; the Julia shim adds 1.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libflux3d_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "flux3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.fx3d_oracle_edges_packed.restype = C.c_int64
        _lib.fx3d_oracle_laplacian_csr.restype = C.c_int64
    return _lib


def _f32(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float32))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _dims(x):
    x = _f32(x)
    if x.ndim == 2:
        x = x.reshape(x.shape[0], x.shape[1], 1, order="F")
    assert x.ndim == 3
    return x, x.shape[0], x.shape[1], x.shape[2]


# ---------------------------------------------------------------------------- point clouds
def nn1(x, y, want_dist=False, kdtree=False):
    """_nearest_neighbors (src/metrics/pcloud.jl:54-70). Returns idx_x (N,B), idx_y (M,B)."""
    x, D, N, B = _dims(x)
    y, D2, M, B2 = _dims(y)
    assert D == D2 and B == B2
    ix = np.zeros((N, B), np.int32, order="F")
    iy = np.zeros((M, B), np.int32, order="F")
    if kdtree:
        rc = lib().fx3d_oracle_nn1_kdtree(_p(x), N, _p(y), M, B, D, _p(ix), _p(iy))
        assert rc == 0
        return ix, iy
    dx = np.zeros((N, B), np.float32, order="F") if want_dist else None
    dy = np.zeros((M, B), np.float32, order="F") if want_dist else None
    rc = lib().fx3d_oracle_nn1(_p(x), N, _p(y), M, B, D, _p(ix), _p(iy), _p(dx), _p(dy))
    assert rc == 0
    return (ix, iy, dx, dy) if want_dist else (ix, iy)


def usable_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota."""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def nn1_allcores(x, y, kdtree=False, threads=None):
    """cpu_baseline leg only: the same searches spread over the usable host cores (OpenMP).
    Returns (idx_x, idx_y, threads_used)."""
    x, D, N, B = _dims(x)
    y, D2, M, B2 = _dims(y)
    assert D == D2 and B == B2
    ix = np.zeros((N, B), np.int32, order="F")
    iy = np.zeros((M, B), np.int32, order="F")
    nt = lib().fx3d_oracle_nn1_allcores(_p(x), N, _p(y), M, B, D, _p(ix), _p(iy), int(bool(kdtree)),
                                        int(threads or usable_cores()))
    assert nt >= 1
    return ix, iy, nt


def chamfer_distance(x, y, w1=1.0, w2=1.0, return_all=False, kdtree=False):
    """_chamfer_distance (src/metrics/pcloud.jl:39-52)."""
    x, D, N, B = _dims(x)
    y, _, M, _ = _dims(y)
    loss = C.c_float(0)
    if kdtree:
        rc = lib().fx3d_oracle_chamfer_fwd_kdtree(_p(x), N, _p(y), M, B, D, C.c_float(w1),
                                                  C.c_float(w2), C.byref(loss))
        assert rc == 0
        return np.float32(loss.value)
    ix = np.zeros((N, B), np.int32, order="F")
    iy = np.zeros((M, B), np.int32, order="F")
    sums = np.zeros(2, np.float64)
    rc = lib().fx3d_oracle_chamfer_fwd(_p(x), N, _p(y), M, B, D, C.c_float(w1), C.c_float(w2),
                                       C.byref(loss), _p(ix), _p(iy), _p(sums))
    assert rc == 0
    if return_all:
        return np.float32(loss.value), ix, iy, sums
    return np.float32(loss.value)


def chamfer_loss_pairwise(x, y, ix, iy, w1=1.0, w2=1.0):
    """The loss in the reference's own arithmetic: Float32 pairwise `mean` (block 1024) of the materialised squared
    differences, from given NN indices (src/metrics/pcloud.jl:47-50)."""
    x, D, N, B = _dims(x)
    y, _, M, _ = _dims(y)
    ix = np.asfortranarray(ix, dtype=np.int32)
    iy = np.asfortranarray(iy, dtype=np.int32)
    out = C.c_float(0)
    rc = lib().fx3d_oracle_chamfer_loss_pairwise(_p(x), N, _p(y), M, B, D, _p(ix), _p(iy), C.c_float(w1), C.c_float(w2),
                                                 C.byref(out))
    assert rc == 0
    return np.float32(out.value)


def chamfer_bwd(x, y, ix, iy, w1=1.0, w2=1.0, gout=1.0):
    x, D, N, B = _dims(x)
    y, _, M, _ = _dims(y)
    ix = np.asfortranarray(ix, dtype=np.int32)
    iy = np.asfortranarray(iy, dtype=np.int32)
    gx = np.zeros_like(x)
    gy = np.zeros_like(y)
    rc = lib().fx3d_oracle_chamfer_bwd(_p(x), N, _p(y), M, B, D, _p(ix), _p(iy), C.c_float(w1),
                                       C.c_float(w2), C.c_float(gout), _p(gx), _p(gy))
    assert rc == 0
    return gx, gy


def knn(x, k, y=None, drop_first=False, want_dist=True):
    """knn(KDTree(y), x, k, true) (src/models/dgcnn.jl:5-6). Returns idx (k,N,B), dist (k,N,B)."""
    x, D, N, B = _dims(x)
    yy = x if y is None else _dims(y)[0]
    M = yy.shape[1]
    idx = np.zeros((k, N, B), np.int32, order="F")
    dist = np.zeros((k, N, B), np.float32, order="F") if want_dist else None
    rc = lib().fx3d_oracle_knn(_p(x), N, _p(yy), M, B, D, k, int(drop_first), _p(idx), _p(dist))
    assert rc == 0, rc
    return (idx, dist) if want_dist else idx


def knn_gather(x, idx):
    """X[:, idxs] -> (F,K,N,B) (src/models/dgcnn.jl:6,36)."""
    x, F, N, B = _dims(x)
    idx = np.asfortranarray(idx, dtype=np.int32)
    k = idx.shape[0]
    out = np.zeros((F, k, N, B), np.float32, order="F")
    lib().fx3d_oracle_knn_gather(_p(x), N, B, F, k, _p(idx), _p(out))
    return out


# ---------------------------------------------------------------------------------- meshes
def _i64(a):
    return np.asfortranarray(np.asarray(a, dtype=np.int64))


def faces_areas_packed(verts, faces0):
    """compute_faces_areas_packed (src/rep/mesh.jl:765-780); faces0 0-based (3,F)."""
    v = _f32(verts)
    f = _i64(faces0)
    out = np.zeros(f.shape[1], np.float32)
    rc = lib().fx3d_oracle_faces_areas_packed(_p(v), C.c_int64(v.shape[1]), _p(f),
                                              C.c_int64(f.shape[1]), _p(out))
    assert rc == 0, rc
    return out


def faces_areas_padded(verts_padded, faces_padded0, faces_len):
    v = _f32(verts_padded)
    f = _i64(faces_padded0)
    fl = np.asarray(faces_len, np.int64)
    B = v.shape[2]
    out = np.zeros((1, f.shape[1], B), np.float32, order="F")
    lib().fx3d_oracle_faces_areas_padded(_p(v), v.shape[1], _p(f), f.shape[1], _p(fl), B, _p(out))
    return out


def face_probs(areas_padded, eps=1e-6):
    a = _f32(areas_padded)
    Fmax, B = a.shape[1], a.shape[2]
    out = np.zeros((1, Fmax, B), np.float64, order="F")
    lib().fx3d_oracle_face_probs(_p(a), Fmax, B, C.c_double(eps), _p(out))
    return out


def sample_points_explicit(verts_padded, faces_padded0, face_idx, r1, r2):
    v = _f32(verts_padded)
    f = _i64(faces_padded0)
    fi = np.asfortranarray(face_idx, dtype=np.int32)
    r1 = _f32(r1)
    r2 = _f32(r2)
    n, B = fi.shape
    out = np.zeros((3, n, B), np.float32, order="F")
    lib().fx3d_oracle_sample_points_explicit(_p(v), v.shape[1], _p(f), f.shape[1], B, n, _p(fi),
                                             _p(r1), _p(r2), _p(out))
    return out


def sample_points_bwd(faces_padded0, faces_len, Vmax, face_idx, r1, r2, gout, base=None):
    """Adjoint of sample_points w.r.t. verts_padded for fixed draws, in the device's (fixed) summation order."""
    f = _i64(faces_padded0)
    fl = np.asarray(faces_len, np.int64)
    fi = np.asfortranarray(face_idx, dtype=np.int32)
    r1, r2, g = _f32(r1), _f32(r2), _f32(gout)
    n, B = fi.shape
    out = np.zeros((3, Vmax, B), np.float32, order="F") if base is None else np.array(base, np.float32, order="F", copy=True)
    rc = lib().fx3d_oracle_sample_points_bwd(_p(f), _p(fl), int(Vmax), f.shape[1], B, n, _p(fi), _p(r1), _p(r2), _p(g),
                                             _p(out), int(base is not None))
    assert rc == 0
    return out


def sample_points_seeded(verts_padded, faces_padded0, faces_len, n, seed, eps=1e-6,
                         return_draws=False):
    v = _f32(verts_padded)
    f = _i64(faces_padded0)
    fl = np.asarray(faces_len, np.int64)
    B = v.shape[2]
    out = np.zeros((3, n, B), np.float32, order="F")
    fi = np.zeros((n, B), np.int32, order="F")
    r1 = np.zeros((n, B), np.float32, order="F")
    r2 = np.zeros((n, B), np.float32, order="F")
    lib().fx3d_oracle_sample_points_seeded(_p(v), v.shape[1], _p(f), f.shape[1], _p(fl), B, n,
                                           C.c_double(eps), C.c_uint64(seed), _p(out), _p(fi),
                                           _p(r1), _p(r2))
    return (out, fi, r1, r2) if return_draws else out


def philox(c0, c1, c2, c3, k0, k1):
    out = np.zeros(4, np.uint32)
    lib().fx3d_oracle_philox(C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3),
                             C.c_uint32(k0), C.c_uint32(k1), _p(out))
    return out


def edges_packed(faces0, V, want_f2e=False):
    """_compute_edges_packed (src/rep/mesh.jl:907-955); returns edges (E,2) 0-based [+ (F,3)]."""
    f = _i64(faces0)
    F = f.shape[1]
    e = np.zeros((3 * F, 2), np.int64)  # C order rows (v0,v1)
    f2e = np.zeros((F, 3), np.int64) if want_f2e else None
    E = lib().fx3d_oracle_edges_packed(_p(f), C.c_int64(F), C.c_int64(V), _p(e), _p(f2e))
    assert E >= 0, E
    e = e[:E].copy()
    return (e, f2e) if want_f2e else e


def laplacian_csr(edges0, V):
    """_compute_laplacian_packed (src/rep/mesh.jl:957-1002) as CSR."""
    e = np.ascontiguousarray(edges0, dtype=np.int64)
    E = e.shape[0]
    rowptr = np.zeros(V + 1, np.int64)
    colind = np.zeros(2 * E + V, np.int64)
    vals = np.zeros(2 * E + V, np.float32)
    nnz = lib().fx3d_oracle_laplacian_csr(_p(e), C.c_int64(E), C.c_int64(V), _p(rowptr),
                                          _p(colind), _p(vals))
    return rowptr, colind[:nnz].copy(), vals[:nnz].copy()


def laplacian_loss(verts_packed, rowptr, colind, vals):
    v = _f32(verts_packed)
    loss = C.c_float(0)
    lib().fx3d_oracle_laplacian_loss(_p(v), C.c_int64(v.shape[1]), _p(rowptr), _p(colind),
                                     _p(vals), C.byref(loss))
    return np.float32(loss.value)


def edge_loss(verts_packed, edges0, target=0.0):
    v = _f32(verts_packed)
    e = np.ascontiguousarray(edges0, dtype=np.int64)
    loss = C.c_float(0)
    lib().fx3d_oracle_edge_loss(_p(v), C.c_int64(v.shape[1]), _p(e), C.c_int64(e.shape[0]),
                                C.c_float(target), C.byref(loss))
    return np.float32(loss.value)


def laplacian_loss_bwd(verts_packed, rowptr, colind, vals, gout=1.0):
    v = _f32(verts_packed)
    g = np.zeros_like(v)
    lib().fx3d_oracle_laplacian_loss_bwd(_p(v), C.c_int64(v.shape[1]), _p(rowptr), _p(colind),
                                         _p(vals), C.c_float(gout), _p(g))
    return g


def edge_loss_bwd(verts_packed, edges0, target=0.0, gout=1.0):
    v = _f32(verts_packed)
    e = np.ascontiguousarray(edges0, dtype=np.int64)
    g = np.zeros_like(v)
    lib().fx3d_oracle_edge_loss_bwd(_p(v), C.c_int64(v.shape[1]), _p(e), C.c_int64(e.shape[0]),
                                    C.c_float(target), C.c_float(gout), _p(g))
    return g


# ------------------------------------------------------------------------- EdgeConv graph features
def edge_features(x, idx, layout=1):
    """EdgeConv's input features (src/models/dgcnn.jl:36-51), array op by array op:
    KNNGraph (F,K,N,B) gathered with idx (k,N,B, 0-based); X repeated K times along a new dim 2 (:39-43);
    cat(X, KNNGraph - X, dims=1) (:45) -> layout 0;  PermutedDimsArray (2,3,1,4) + reshape (N*K, 2F, B)
    (:48-51) -> layout 1."""
    x, F, N, B = _dims(x)
    idx = np.asarray(idx)
    k = idx.shape[0]
    graph = knn_gather(x, idx)                                  # (F,K,N,B)
    xr = np.broadcast_to(x.reshape(F, 1, N, B, order="F"), (F, k, N, B))
    out = np.concatenate([xr, graph - xr], axis=0)              # (2F,K,N,B)
    if layout == 0:
        return np.asfortranarray(out)
    out = np.transpose(out, (1, 2, 0, 3))                       # (K,N,2F,B)
    return np.asfortranarray(out).reshape(N * k, 2 * F, B, order="F")


def edge_features_bwd(g, F, N, B, k, layout=1):
    """Adjoint of edge_features w.r.t. X with the graph held constant (@nograd, src/models/dgcnn.jl:9):
    reverse of the reshape/permute/cat/repeat chain; the K copies' gradients are added in rank order."""
    g = np.asarray(g, dtype=np.float32)
    if layout == 1:
        g = g.reshape(k, N, 2 * F, B, order="F").transpose(2, 0, 1, 3)   # (2F,K,N,B)
    else:
        g = g.reshape(2 * F, k, N, B, order="F")
    d = g[:F] - g[F:]
    acc = np.zeros((F, N, B), np.float32)
    for r in range(k):
        acc = acc + d[:, r]
    return np.asfortranarray(acc)


# ------------------------------------------------------------------------------- pointcloud_to_voxel
def pointcloud_to_voxel(points, res=32):
    """pointcloud_to_voxel (src/conversions.jl:91-131) -> (res,res,res,B) Float32 0/1."""
    x, D, N, B = _dims(points)
    assert D == 3
    out = np.zeros((res, res, res, B), np.float32, order="F")
    rc = lib().fx3d_oracle_pointcloud_to_voxel(_p(x), N, B, int(res), _p(out))
    assert rc == 0
    return out
