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
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz"), **out)
print("written", {k: getattr(v, "shape", v) for k, v in out.items()})
