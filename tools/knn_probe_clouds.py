"""Writes the clouds of tools/knn_distribution_time.py (B = 32 x 1024 points, D = 3, row-major points) as raw Float32 files for
tools/knn_probe (CLOUD=<file>):   python tools/knn_probe_clouds.py <directory>"""
import os
import sys

import numpy as np

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
rng = np.random.default_rng(9)
B, N, D = 32, 1024, 3


def make(kind):
    if kind == "uniform":
        return rng.random((B, N, D))
    if kind == "clusters":
        c = rng.standard_normal((B, 16, D)) * 3
        return np.take_along_axis(c, rng.integers(0, 16, (B, N, 1)).repeat(D, 2), 1) + rng.standard_normal((B, N, D)) * 1e-3
    if kind == "lattice":
        return rng.integers(0, 8, (B, N, D)) * 0.125
    if kind == "dupes":
        x = rng.random((B, N, D))
        x[:, N // 2:, :] = x[:, : N // 2, :]
        return x
    raise KeyError(kind)


for kind in ("uniform", "clusters", "lattice", "dupes"):
    make(kind).astype(np.float32).tofile(os.path.join(out, f"knn_{kind}.f32"))
