#!/usr/bin/env python3
"""Mesh-side ops on large synthetic sheets (a check that the teapot-sized, launch-bound kernels scale): device time and
algorithmic GB/s (DESIGN.md 3.3 / 3.4 byte counts) for areas, both losses and adjoints, the sampling CDF and the draw."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import flux3d_jl_amd as fx  # noqa: E402
from bench_ops import gpu_time  # noqa: E402


def sheet(nx, ny, seed):
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.arange(nx + 1, dtype=np.float64), np.arange(ny + 1, dtype=np.float64), indexing="ij")
    v = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 0) + rng.uniform(-0.3, 0.3, (3, gx.size))
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    a = (i * (ny + 1) + j).ravel()
    f = np.concatenate([np.stack([a, a + ny + 1, a + ny + 2]), np.stack([a, a + ny + 2, a + 1])], 1)
    return np.asfortranarray(v.astype(np.float32)), np.asfortranarray(f.astype(np.int64) + 1)


for nx, B in [(34, 8), (150, 4), (500, 1), (1000, 1)]:
    ms = [sheet(nx, nx, s) for s in range(B)]
    m = fx.gpu(fx.TriMesh([v for v, _ in ms], [f for _, f in ms]))
    V, F = m.V * B, m.F * B
    E = m.dev("edges").shape[0]
    nnz = m.dev("lap_colind").shape[0]
    out = {}
    out["areas"] = (gpu_time(lambda: fx.compute_faces_areas_packed(m), reps=10, inner=4)[0], 12 * V + 12 * F + 4 * F)
    out["laplacian_loss"] = (gpu_time(lambda: fx.laplacian_loss(m, sync=False), reps=10, inner=4)[0], 12 * V + 8 * nnz + 4 * V)
    out["edge_loss"] = (gpu_time(lambda: fx.edge_loss(m, sync=False), reps=10, inner=4)[0], 12 * V + 8 * E)
    out["mesh_losses fwd+bwd"] = (gpu_time(lambda: (fx.mesh_losses(m, sync=False), fx.mesh_losses_grad(m, reuse_forward=True)), reps=10, inner=4)[0],
                                  2 * (12 * V + 8 * nnz) + 16 * V + 12 * V)

    def fresh():
        for k in [k for k in m._dev if isinstance(k, tuple) and k[0] == "face_cdf"]:
            del m._dev[k]
        return m
    n = 100000
    out[f"sample_points n={n} (CDF + draw)"] = (gpu_time(lambda: fx.sample_points(fresh(), n, seed=3), reps=10, inner=4)[0], 12 * V + 12 * F + 8 * F + 12 * n * B)
    out[f"sample_points n={n} (draw)"] = (gpu_time(lambda: fx.sample_points(m, n, seed=3), reps=10, inner=4)[0], 12 * n * B)
    print(f"--- {B} sheet(s) of {m.V} vertices / {m.F} faces (E = {E})")
    for k, (us, by) in out.items():
        print(f"{k:40s} {us:9.1f} us   {by / us / 1e3:8.1f} GB/s", flush=True)
