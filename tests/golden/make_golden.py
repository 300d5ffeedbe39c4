"""Generates tests/golden/oracle_vectors.npz: small seeded input/output vectors of the CPU oracle
(oracle/flux3d_oracle.c), cross-checked where an independent exact method exists (scipy cKDTree in
float64).  The reference (Julia) cannot run in this image, so these pin the ORACLE, not the
reference; the reference's own known answers are in ref_known_answers.json.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
import flux3d_jl_amd as fx  # noqa: E402  (host-only helpers: synth, load_obj)
from scipy.spatial import cKDTree  # noqa: E402

out = {}
# chamfer: N != M, B = 2 like test/metrics.jl:109-110
cx = fx.synth.uniform_cloud(fx.synth.SEED_A, 3, 300, 2)
cy = fx.synth.uniform_cloud(fx.synth.SEED_B, 3, 170, 2)
loss, ix, iy, _ = oracle.chamfer_distance(cx, cy, return_all=True)
for b in range(2):
    _, j = cKDTree(cy[:, :, b].T.astype(np.float64)).query(cx[:, :, b].T.astype(np.float64))
    assert np.array_equal(j, ix[:, b])
out.update(cx=cx, cy=cy, c_ix=ix, c_iy=iy, c_loss=np.float32(loss))
# knn k=20 drop-first
kx = fx.synth.uniform_cloud(0x5EED0004, 3, 200, 2)
idx, dist = oracle.knn(kx, 20, drop_first=True)
out.update(kx=kx, k_idx=idx, k_dist=dist)
# sampler on a tiny 2-mesh batch (ragged)
v1 = np.asfortranarray(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32).T)
f1 = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]], np.int64).T
v2 = np.asfortranarray(np.array([[0, 0, 0], [2, 0, 0], [0, 3, 0]], np.float32).T)
f2 = np.array([[0, 1, 2]], np.int64).T
sv = np.zeros((3, 4, 2), np.float32, order="F"); sv[:, :4, 0] = v1; sv[:, :3, 1] = v2
sf = np.zeros((3, 4, 2), np.int64, order="F"); sf[:, :4, 0] = f1; sf[:, :1, 1] = f2
seed = 20260928
s = oracle.sample_points_seeded(sv, sf, [4, 1], 64, seed=seed)
out.update(s_verts=sv, s_faces0=sf, s_faces_len=np.array([4, 1], np.int64), s_seed=np.int64(seed), s_out=s)
# ---- round 4: the boundaries that matter, still without a live oracle on the GPU box --------------------------
def unique_nn(xq, yc):
    """float64 cKDTree nearest neighbour of every query + a mask of the queries whose minimum is unique by a margin well above
    Float32 rounding (the oracle's tie-break by index only matters elsewhere)."""
    d, j = cKDTree(yc.T.astype(np.float64)).query(xq.T.astype(np.float64), k=2)
    return j[:, 0], (d[:, 1] - d[:, 0]) > 1e-5 * np.maximum(d[:, 1], 1e-30)


rng = np.random.default_rng(20260929)
# (a) one LDS image + an exact tail (N = 4097 against M = 4096), two clouds; half of the second cloud pair sits on a lattice
#     (exact ties everywhere: the lowest index must win across lane tiles, chunks and the tail)
gx = np.asfortranarray(rng.random((3, 4097, 2)).astype(np.float32))
gy = np.asfortranarray(rng.random((3, 4096, 2)).astype(np.float32))
gx[:, 2000:, 1] = rng.integers(0, 12, (3, 2097)).astype(np.float32) / np.float32(12)
gy[:, 1000:3500, 1] = rng.integers(0, 12, (3, 2500)).astype(np.float32) / np.float32(12)
gl, gix, giy, _ = oracle.chamfer_distance(gx, gy, return_all=True)
for b in range(2):
    j, ok = unique_nn(gx[:, :, b], gy[:, :, b])
    assert np.array_equal(j[ok], gix[ok, b]) and ok.sum() > 1500
    j, ok = unique_nn(gy[:, :, b], gx[:, :, b])
    assert np.array_equal(j[ok], giy[ok, b])
out.update(g_x=gx, g_y=gy, g_ix=gix, g_iy=giy, g_loss=np.float32(gl))
# (b) a split-plan shape: one pair of large clouds, candidate chunks spread over blocks + the merge kernel
sx = np.asfortranarray(rng.standard_normal((3, 9000, 1)).astype(np.float32))
sy = np.asfortranarray((rng.standard_normal((3, 12000, 1)) * 1.5 + 0.25).astype(np.float32))
sl, six, siy, _ = oracle.chamfer_distance(sx, sy, return_all=True)
j, ok = unique_nn(sx[:, :, 0], sy[:, :, 0])
assert np.array_equal(j[ok], six[ok, 0]) and ok.mean() > 0.99
out.update(p_x=sx, p_y=sy, p_ix=six, p_iy=siy, p_loss=np.float32(sl))
# (c) kNN k = 20 (+ the dropped self) at C4's cloud size, D = 3 (two clouds) and D = 64 (one cloud): the matrix-core kernels
k3 = fx.synth.uniform_cloud(0x5EED0004, 3, 1024, 2)
k3i, k3d = oracle.knn(k3, 20, drop_first=True)
k64 = np.asfortranarray(rng.standard_normal((64, 1024, 1)).astype(np.float32))
k64i, k64d = oracle.knn(k64, 20, drop_first=True)
for arr, idx in ((k3, k3i), (k64, k64i)):
    for b in range(arr.shape[2]):
        d, j = cKDTree(arr[:, :, b].T.astype(np.float64)).query(arr[:, :, b].T.astype(np.float64), k=22)
        gap = np.diff(d, axis=1).min(axis=1) > 1e-6 * d[:, -1]      # all 22 distances distinct by a margin: the order is unambiguous
        assert gap.mean() > 0.9 and np.array_equal(j[gap, 1:21], idx[:, gap, b].T)
out.update(k3_x=k3, k3_idx=k3i, k3_dist=k3d, k64_x=k64, k64_idx=k64i, k64_dist=k64d)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz"), **out)
print("written", {k: getattr(v, "shape", v) for k, v in out.items()})
