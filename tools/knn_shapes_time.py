#!/usr/bin/env python3
"""Self-kNN graphs over a spread of (D, N, B, k) outside the BASELINE configs: per-call device time and pairs per second --
a check for launch-plan cliffs (few clouds with many rows, wide features)."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import flux3d_jl_amd as fx
from bench_ops import gpu_time
rng = np.random.default_rng(5)
for D, N, B, k in [(3, 1024, 32, 20), (3, 4096, 8, 20), (3, 16384, 1, 20), (3, 16384, 4, 20), (3, 32768, 1, 16), (64, 1024, 32, 20), (64, 4096, 4, 20), (64, 8192, 1, 20), (16, 2048, 8, 10), (128, 1024, 8, 20)]:
    x = fx.gpu(np.asfortranarray(rng.standard_normal((D, N, B)).astype(np.float32)))
    try:
        mn, md = gpu_time(lambda: fx.knn(x, k, drop_first=True, return_dist=False), reps=10, inner=4)
        print(f"D={D:<3d} N=M={N:<6d} B={B:<3d} k={k:<3d} min {mn:9.1f} us   {B*N*N/mn/1e6:8.3f} T pairs/s", flush=True)
    except Exception as e:
        print(D, N, B, k, "ERR", str(e)[:100])
