#!/usr/bin/env python3
"""kNN at C4's shape (B = 32 x 1024 points, D = 3 and D = 64) over k: where the matrix-core kernels (k + drop <= 32), the wave
kernels (<= 64) and the general selection kernel take over."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import flux3d_jl_amd as fx
from bench_ops import gpu_time
rng = np.random.default_rng(5)
for D in (3, 64):
    x = fx.gpu(np.asfortranarray(rng.standard_normal((D, 1024, 32)).astype(np.float32)))
    for k in (5, 10, 20, 31, 32, 40, 63, 64, 100, 127, 128):
        mn, md = gpu_time(lambda: fx.knn(x, k, drop_first=True, return_dist=False), reps=8, inner=4)
        print(f"D={D:<3d} B=32 N=1024 k={k:<4d} min {mn:9.1f} us", flush=True)
