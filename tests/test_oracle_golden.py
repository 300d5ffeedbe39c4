"""Pin the CPU oracle against the reference's own known answers -- CPU only.

If these fail, nothing downstream (GPU parity) means anything.  Sources of every number are in
tests/golden/ref_known_answers.json.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _mesh_arrays(verts_rows, faces_rows):
    vl = [np.asfortranarray(np.array(v, np.float32).T) for v in verts_rows]
    fl = [np.asfortranarray(np.array(f, np.int64).T) for f in faces_rows]
    return vl, fl


def _packed(vl, fl):
    offs = np.concatenate([[0], np.cumsum([v.shape[1] for v in vl])[:-1]])
    vp = np.asfortranarray(np.concatenate(vl, axis=1))
    fp0 = np.asfortranarray(np.concatenate([f - 1 + o for f, o in zip(fl, offs)], axis=1))
    return vp, fp0


def _load_obj(name):
    from flux3d_jl_amd import load_obj
    return load_obj(os.path.join(GOLDEN, name))


def test_face_areas_known_answer(oracle, known):
    """test/rep.jl:259-260,351-387: areas [0.125,0.1,0.02,0.0] and [0.0415,0.1] (tol 1e-4)."""
    k = known["areas_batch"]
    vl, fl = _mesh_arrays(k["verts"], k["faces"])
    vp, fp0 = _packed(vl, fl)
    got = oracle.faces_areas_packed(vp, fp0)
    exp = np.concatenate([np.array(a, np.float32) for a in k["areas"]])
    assert np.allclose(got, exp, rtol=k["tol"], atol=k["tol"])
    # padded form: tail zero (test/rep.jl:386-387)
    Fmax = max(f.shape[1] for f in fl)
    Vmax = max(v.shape[1] for v in vl)
    vpad = np.zeros((3, Vmax, 2), np.float32, order="F")
    fpad = np.zeros((3, Fmax, 2), np.int64, order="F")
    for i, (v, f) in enumerate(zip(vl, fl)):
        vpad[:, : v.shape[1], i] = v
        fpad[:, : f.shape[1], i] = f - 1
    ap = oracle.faces_areas_padded(vpad, fpad, [f.shape[1] for f in fl])
    assert np.allclose(ap[0, :4, 0], k["areas"][0], atol=1e-4)
    assert np.allclose(ap[0, :2, 1], k["areas"][1], atol=1e-4)
    assert np.all(ap[0, 2:, 1] == 0)


def test_teapot_laplacian_known_answer(oracle, known):
    """README.md:111-112: laplacian_loss(teapot) == 0.05888283f0."""
    v, f = _load_obj("teapot.obj")
    assert v.shape[1] == known["assets"]["teapot"]["verts"]
    assert f.shape[1] == known["assets"]["teapot"]["faces"]
    edges = oracle.edges_packed(f.astype(np.int64) - 1, v.shape[1])
    assert edges.shape[0] == 3456  # E = 3456 unique edges
    rowptr, colind, vals = oracle.laplacian_csr(edges, v.shape[1])
    assert len(colind) == 2 * 3456 + 1202  # nnz = 2E + V
    loss = oracle.laplacian_loss(v, rowptr, colind, vals)
    assert abs(float(loss) - known["teapot_laplacian_loss"]["value"]) <= 2e-8 + 1e-6 * 0.0589
    # SURVEY.md 8(c) extra oracle value
    assert abs(float(oracle.edge_loss(v, edges)) - 0.093938984) < 1e-6


def test_sphere_asset_is_unit(known):
    v, f = _load_obj("sphere.obj")
    assert v.shape[1] == known["assets"]["sphere"]["verts"] and f.shape[1] == known["assets"]["sphere"]["faces"]
    r = np.sqrt((v.astype(np.float64) ** 2).sum(0))
    assert r.min() > 0.99999 and r.max() < 1.00001


def test_edges_match_sorted_unique(oracle, known):
    """test/rep.jl:135-156: edges == sort/unique of all face edges; faces_to_edges order (e23,e31,e12)."""
    k = known["three_mesh_batch"]
    vl, fl = _mesh_arrays(k["verts"], k["faces"])
    vp, fp0 = _packed(vl, fl)
    edges, f2e = oracle.edges_packed(fp0, vp.shape[1], want_f2e=True)
    e_all = np.concatenate([fp0[[0, 1]].T, fp0[[1, 2]].T, fp0[[2, 0]].T])
    e_all = np.unique(np.sort(e_all, axis=1), axis=0)
    assert np.array_equal(edges, e_all)
    for i in range(f2e.shape[0]):
        assert np.array_equal(edges[f2e[i, 0]], np.sort(fp0[[1, 2], i]))
        assert np.array_equal(edges[f2e[i, 1]], np.sort(fp0[[0, 2], i]))
        assert np.array_equal(edges[f2e[i, 2]], np.sort(fp0[[0, 1], i]))


def test_laplacian_vs_dense(oracle, known):
    """test/rep.jl:158-175 and test/metrics.jl:50-71: sparse Laplacian == dense construction, and
    laplacian_loss == mean(norm(L_dense * verts'))."""
    k = known["three_mesh_batch"]
    vl, fl = _mesh_arrays(k["verts"], k["faces"])
    vp, fp0 = _packed(vl, fl)
    V = vp.shape[1]
    edges = oracle.edges_packed(fp0, V)
    L = np.zeros((V, V))
    for a, b in edges:
        L[a, b] = 1
        L[b, a] = 1
    deg = L.sum(1)
    inv = np.where(deg > 0, 1.0 / np.maximum(deg, 1), deg)
    Ld = np.where(L == 1, inv[:, None], 0.0)
    Ld[np.arange(V), np.arange(V)] = -1
    rowptr, colind, vals = oracle.laplacian_csr(edges, V)
    Ls = np.zeros((V, V))
    for i in range(V):
        Ls[i, colind[rowptr[i]:rowptr[i + 1]]] = vals[rowptr[i]:rowptr[i + 1]]
    assert np.allclose(Ls, Ld, rtol=1e-5, atol=1e-5)
    ref = np.sqrt(((Ld @ vp.T.astype(np.float64)) ** 2).sum(1)).mean()
    assert np.isclose(float(oracle.laplacian_loss(vp, rowptr, colind, vals)), ref, rtol=3.5e-4)
    # edge_loss == mean(norm(v1-v2)^2)  (test/metrics.jl:75-84)
    d = vp[:, edges[:, 0]].astype(np.float64) - vp[:, edges[:, 1]]
    assert np.isclose(float(oracle.edge_loss(vp, edges)), (d ** 2).sum(0).mean(), rtol=1e-6)


def _naive_chamfer(x, y):
    """naive_chamfer of test/metrics.jl:94-107 (dense ||x||^2+||y||^2-2x'y, minimum), float64."""
    x = x.astype(np.float64)
    y = y.astype(np.float64)
    tot = 0.0
    B = x.shape[2]
    for b in range(B):
        P = ((x[:, :, b] ** 2).sum(0)[:, None] + (y[:, :, b] ** 2).sum(0)[None, :]
             - 2 * x[:, :, b].T @ y[:, :, b])
        tot += P.min(1).mean() / B + P.min(0).mean() / B
    return tot


def test_chamfer_vs_naive_dense(oracle):
    """test/metrics.jl:109-111: chamfer_distance(x,y) ~ naive_chamfer on rand(3,1000,2)/(3,500,2)."""
    rng = np.random.default_rng(7)
    x = np.asfortranarray(rng.random((3, 1000, 2), dtype=np.float32))
    y = np.asfortranarray(rng.random((3, 500, 2), dtype=np.float32))
    got = float(oracle.chamfer_distance(x, y))
    assert np.isclose(got, _naive_chamfer(x, y), rtol=3.45e-4)  # isapprox default, sqrt(eps(Float32))
    assert float(oracle.chamfer_distance(x, x)) == 0.0


def test_nn_indices_match_kdtree(oracle):
    """Brute-force first-min indices == exact fp64 cKDTree indices (unique minima), and == the
    oracle's own KD-tree twin of src/metrics/pcloud.jl:54-70 including ties."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(11)
    x = np.asfortranarray(rng.random((3, 1000, 2), dtype=np.float32))
    y = np.asfortranarray(rng.random((3, 500, 2), dtype=np.float32))
    ix, iy = oracle.nn1(x, y)
    kx, ky = oracle.nn1(x, y, kdtree=True)
    assert np.array_equal(ix, kx) and np.array_equal(iy, ky)
    for b in range(2):
        _, j = cKDTree(y[:, :, b].T.astype(np.float64)).query(x[:, :, b].T.astype(np.float64))
        assert np.array_equal(j, ix[:, b])
    # heavy ties: points on a coarse lattice
    xt = np.asfortranarray(rng.integers(0, 4, (3, 300, 1)).astype(np.float32))
    yt = np.asfortranarray(rng.integers(0, 4, (3, 200, 1)).astype(np.float32))
    a = oracle.nn1(xt, yt)
    b_ = oracle.nn1(xt, yt, kdtree=True)
    assert np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1])


def test_knn_matches_scipy(oracle):
    """knn sorted by (distance, index): same sets/order as an exact fp64 tree on random data;
    drop_first drops the query itself (src/models/dgcnn.jl:6)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    x = np.asfortranarray(rng.random((3, 256, 2), dtype=np.float32))
    idx, dist = oracle.knn(x, 20, drop_first=True)
    for b in range(2):
        pts = x[:, :, b].T.astype(np.float64)
        _, j = cKDTree(pts).query(pts, k=21)
        assert np.array_equal(j[:, 1:], idx[:, :, b].T)
    assert np.all(np.diff(dist, axis=0) >= 0)
    # 64-D (second EdgeConv, src/models/dgcnn.jl:121)
    f = np.asfortranarray(rng.standard_normal((64, 128, 1)).astype(np.float32))
    idx64, _ = oracle.knn(f, 10, drop_first=True)
    pts = f[:, :, 0].T.astype(np.float64)
    _, j = cKDTree(pts).query(pts, k=11)
    assert np.array_equal(j[:, 1:], idx64[:, :, 0].T)


def test_sample_points_sphere_radius(oracle):
    """test/transforms/mesh_func.jl:4-14: samples of the unit icosphere have radius ~ 1 (rtol 1e-2)."""
    v, f = _load_obj("sphere.obj")
    V, F = v.shape[1], f.shape[1]
    vpad = np.asfortranarray(np.stack([v, v], axis=2))
    fpad = np.asfortranarray(np.stack([f.astype(np.int64) - 1] * 2, axis=2))
    s, fi, r1, r2 = oracle.sample_points_seeded(vpad, fpad, [F, F], 1000, seed=1234, return_draws=True)
    r = np.sqrt((s.astype(np.float64) ** 2).sum(0))
    assert np.allclose(r, 1.0, rtol=1e-2, atol=1e-5)
    assert fi.min() >= 0 and fi.max() < F and r1.min() >= 0 and r1.max() < 1
    # the two meshes use different Philox counters -> different draws
    assert not np.array_equal(fi[:, 0], fi[:, 1])
    # explicit-draw form reproduces the same points
    s2 = oracle.sample_points_explicit(vpad, fpad, fi, r1, r2)
    assert np.array_equal(s, s2)


def test_sample_points_face_histogram(oracle):
    """Faces are drawn proportionally to area (Categorical(area/sum area), mesh_func.jl:32-47):
    chi-square of the face histogram against the area distribution on the teapot."""
    v, f = _load_obj("teapot.obj")
    F = f.shape[1]
    f0 = f.astype(np.int64) - 1
    vpad = np.asfortranarray(v[:, :, None])
    fpad = np.asfortranarray(f0[:, :, None])
    n = 200000
    _, fi, _, _ = oracle.sample_points_seeded(vpad, fpad, [F], n, seed=99, return_draws=True)
    area = oracle.faces_areas_packed(v, f0).astype(np.float64)
    p = area / area.sum()
    cnt = np.bincount(fi[:, 0], minlength=F)
    keep = p * n >= 5
    chi2 = (((cnt - p * n) ** 2)[keep] / (p * n)[keep]).sum()
    dof = keep.sum() - 1
    assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof)
    assert cnt[~keep].sum() <= (p[~keep].sum() * n) * 3 + 20


def test_face_probs_last_column_quirk(oracle):
    """mesh_func.jl:36-37: the 1-sum fix-up is added at the last PADDED column."""
    a = np.zeros((1, 4, 2), np.float32, order="F")
    a[0, :, 0] = [1, 1, 1, 1]
    a[0, :2, 1] = [3, 1]
    p = oracle.face_probs(a)
    assert np.allclose(p[0, :, 0], 0.25) and np.isclose(p[0, :, 0].sum(), 1.0, atol=1e-15)
    assert np.allclose(p[0, :2, 1], [0.75, 0.25]) and p[0, 3, 1] >= 0
    z = oracle.face_probs(np.zeros((1, 3, 1), np.float32, order="F"))
    assert np.allclose(z[0, :, 0], [0, 0, 1])  # degenerate mesh: all mass on the last column


def test_philox_known_answer(oracle):
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors: zero and pi-digits cases)."""
    assert list(oracle.philox(0, 0, 0, 0, 0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    got = oracle.philox(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)
    assert list(got) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_chamfer_bwd_matches_finite_difference(oracle):
    """Adjoint with constant indices == finite differences of the loss (test/utils.jl:1-21 style),
    away from NN switches (tiny step)."""
    rng = np.random.default_rng(3)
    x = np.asfortranarray(rng.random((3, 40, 2)).astype(np.float32))
    y = np.asfortranarray(rng.random((3, 30, 2)).astype(np.float32))
    loss, ix, iy, _ = oracle.chamfer_distance(x, y, 0.7, 1.3, return_all=True)
    gx, gy = oracle.chamfer_bwd(x, y, ix, iy, 0.7, 1.3)

    def f64(xx, yy):  # same loss in float64 with the indices frozen
        tot = 0.0
        for b in range(2):
            dA = ((xx[:, :, b] - yy[:, ix[:, b], b]) ** 2).sum() / (40 * 2)
            dB = ((yy[:, :, b] - xx[:, iy[:, b], b]) ** 2).sum() / (30 * 2)
            tot += 0.7 * dA + 1.3 * dB
        return tot

    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    h = 1e-6
    for (d, i, b) in [(0, 0, 0), (2, 17, 1), (1, 39, 0)]:
        xp = x64.copy(); xp[d, i, b] += h
        xm = x64.copy(); xm[d, i, b] -= h
        assert np.isclose((f64(xp, y64) - f64(xm, y64)) / (2 * h), gx[d, i, b], rtol=1e-3, atol=1e-6)
    for (d, j, b) in [(0, 3, 0), (1, 29, 1)]:
        yp = y64.copy(); yp[d, j, b] += h
        ym = y64.copy(); ym[d, j, b] -= h
        assert np.isclose((f64(x64, yp) - f64(x64, ym)) / (2 * h), gy[d, j, b], rtol=1e-3, atol=1e-6)


def test_mesh_loss_bwd_finite_difference(oracle):
    v, f = _load_obj("teapot.obj")
    f0 = f.astype(np.int64) - 1
    edges = oracle.edges_packed(f0, v.shape[1])
    rowptr, colind, vals = oracle.laplacian_csr(edges, v.shape[1])
    ge = oracle.edge_loss_bwd(v, edges, 0.1)
    gl = oracle.laplacian_loss_bwd(v, rowptr, colind, vals)
    v64 = v.astype(np.float64)

    def edge64(vv):
        d = vv[:, edges[:, 0]] - vv[:, edges[:, 1]]
        return ((np.sqrt((d ** 2).sum(0)) - 0.1) ** 2).mean()

    Ld = np.zeros((v.shape[1], v.shape[1]))
    for i in range(v.shape[1]):
        Ld[i, colind[rowptr[i]:rowptr[i + 1]]] = vals[rowptr[i]:rowptr[i + 1]]

    def lap64(vv):
        return np.sqrt(((Ld @ vv.T) ** 2).sum(1)).mean()

    h = 1e-6
    for (d, i) in [(0, 5), (1, 600), (2, 1201)]:
        vp = v64.copy(); vp[d, i] += h
        vm = v64.copy(); vm[d, i] -= h
        assert np.isclose((edge64(vp) - edge64(vm)) / (2 * h), ge[d, i], rtol=2e-3, atol=1e-7)
        assert np.isclose((lap64(vp) - lap64(vm)) / (2 * h), gl[d, i], rtol=2e-3, atol=1e-7)


def test_committed_golden_vectors(oracle):
    """The committed fixture (tests/golden/make_golden.py) still reproduces: guards the oracle
    against silent drift between rounds."""
    g = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    loss, ix, iy, sums = oracle.chamfer_distance(g["cx"], g["cy"], return_all=True)
    assert np.array_equal(ix, g["c_ix"]) and np.array_equal(iy, g["c_iy"])
    assert np.float32(loss) == g["c_loss"]
    idx, dist = oracle.knn(g["kx"], 20, drop_first=True)
    assert np.array_equal(idx, g["k_idx"]) and np.array_equal(dist, g["k_dist"])
    s = oracle.sample_points_seeded(g["s_verts"], g["s_faces0"], g["s_faces_len"], 64, seed=int(g["s_seed"]))
    assert np.array_equal(s, g["s_out"])
    # round 4: the large vectors (chunk tail + lattice ties, split-plan shape, kNN at C4's size in both spaces)
    for tag in ("g", "p"):
        loss, ix, iy, _ = oracle.chamfer_distance(g[tag + "_x"], g[tag + "_y"], return_all=True)
        assert np.array_equal(ix, g[tag + "_ix"]) and np.array_equal(iy, g[tag + "_iy"]) and np.float32(loss) == g[tag + "_loss"]
    for tag in ("k3", "k64"):
        idx, dist = oracle.knn(g[tag + "_x"], 20, drop_first=True)
        assert np.array_equal(idx, g[tag + "_idx"]) and np.array_equal(dist, g[tag + "_dist"])


def test_edge_features_structure(oracle):
    """EdgeConv input features (src/models/dgcnn.jl:36-51): shape contract of test/models.jl:29-35 and the
    two layouts hold the same numbers; entry [f, k, n, b] = X[f,n,b] and [F+f, ...] = X[f,idx] - X[f,n]."""
    rng = np.random.default_rng(5)
    F, N, B, K = 5, 40, 2, 6
    x = np.asfortranarray(rng.standard_normal((F, N, B)).astype(np.float32))
    idx = oracle.knn(x, K, drop_first=True, want_dist=False)
    cat = oracle.edge_features(x, idx, layout=0)
    mlp = oracle.edge_features(x, idx, layout=1)
    assert cat.shape == (2 * F, K, N, B) and mlp.shape == (K * N, 2 * F, B)
    for (f, k, n, b) in ((0, 0, 0, 0), (4, 5, 39, 1), (2, 3, 17, 0)):
        assert cat[f, k, n, b] == x[f, n, b]
        assert cat[F + f, k, n, b] == x[f, idx[k, n, b], b] - x[f, n, b]
        assert mlp[k + K * n, f, b] == cat[f, k, n, b] and mlp[k + K * n, F + f, b] == cat[F + f, k, n, b]
    # adjoint: <edge_features'(x) dx, g> with the graph frozen == <dx, bwd(g)>
    g = rng.standard_normal(mlp.shape).astype(np.float32)
    gx = oracle.edge_features_bwd(g, F, N, B, K, layout=1)
    dx = np.asfortranarray(rng.standard_normal(x.shape).astype(np.float32))
    # only the repeated-X terms move: d out = cat(dX, -dX)
    dcat = np.concatenate([np.broadcast_to(dx.reshape(F, 1, N, B, order="F"), (F, K, N, B))] * 2, axis=0).copy()
    dcat[F:] *= -1
    dmlp = np.asfortranarray(np.transpose(dcat, (1, 2, 0, 3))).reshape(N * K, 2 * F, B, order="F")
    assert np.isclose(float((dmlp.astype(np.float64) * g).sum()), float((dx.astype(np.float64) * gx).sum()), rtol=1e-4)


def test_pointcloud_to_voxel_vs_kdtree(oracle):
    """pointcloud_to_voxel (src/conversions.jl:91-131): the reference's tests pin only type and shape
    (test/conversions.jl:36-38), so the restatement is cross-checked against an independent Float64
    KD-tree (scipy) evaluation of the same formula, including the lattice order and the 1-based shift."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(11)
    res, N, B = 16, 300, 2
    p = np.asfortranarray((rng.random((3, N, B), dtype=np.float32) * 3 - 1).astype(np.float32))
    vox = oracle.pointcloud_to_voxel(p, res)
    assert vox.shape == (res, res, res, B) and vox.dtype == np.float32
    assert set(np.unique(vox)) <= {0.0, 1.0} and 0 < vox.sum() < vox.size
    ax = (np.arange(1, res + 1) + 0.5) / res
    gx, gy, gz = np.meshgrid(ax, ax, ax, indexing="ij")            # x outer ... z inner
    grid = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1)  # C-order ravel: z fastest
    for b in range(B):
        pb = p[:, :, b]
        cloud = ((pb - pb.min()) / (pb.max() - pb.min())).astype(np.float32).T.astype(np.float64)
        d, _ = cKDTree(cloud).query(grid, k=1)
        near_thr = np.abs(d * d - 0.6 / res ** 2) < 1e-12
        ref = (d * d <= 0.6 / res ** 2).astype(np.float32)
        got = vox[:, :, :, b].ravel(order="F")                     # first dim (fastest) = z
        assert np.array_equal(got[~near_thr], ref[~near_thr])


def test_pairwise_float32_mean_follows_base_mapreduce(oracle):
    """The oracle's restatement of the reference's `mean` (Base.mapreduce_impl: ranges of < 1024 + 1 elements summed left to
    right from a[i] + a[i+1], longer ones split at ifirst + (ilast - ifirst) >> 1) against an independent numpy recursion, at
    lengths around the block size and for a (3, N, B) problem whose leaves do not start at multiples of 3."""
    rng = np.random.default_rng(5)

    def psum(a, i, j):
        if i == j:
            return a[i]
        if j - i < 1024:
            v = np.float32(a[i] + a[i + 1])
            for k in range(i + 2, j + 1):
                v = np.float32(v + a[k])
            return v
        m = i + ((j - i) >> 1)
        return np.float32(psum(a, i, m) + psum(a, m + 1, j))

    for (N, M, B) in ((341, 342, 1), (1000, 500, 2), (5, 2049, 3)):
        x = np.asfortranarray(rng.random((3, N, B)).astype(np.float32))
        y = np.asfortranarray(rng.random((3, M, B)).astype(np.float32))
        loss64, ix, iy, _ = oracle.chamfer_distance(x, y, 0.5, 2.0, return_all=True)
        got = oracle.chamfer_loss_pairwise(x, y, ix, iy, 0.5, 2.0)
        parts = []
        for a, c, idx in ((x, y, ix), (y, x, iy)):
            T = np.stack([(a[:, :, b] - c[:, idx[:, b], b]) ** 2 for b in range(B)], axis=2).astype(np.float32)
            t = T.reshape(-1, order="F")
            parts.append(np.float32(np.float32(psum(t, 0, t.size - 1) / np.float32(t.size)) * np.float32(3.0)))
        want = np.float32(np.float32(np.float32(0.5) * parts[0]) + np.float32(np.float32(2.0) * parts[1]))
        assert got == want
        assert np.isclose(got, loss64, rtol=2e-6, atol=0)
