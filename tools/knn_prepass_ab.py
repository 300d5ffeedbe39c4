#!/usr/bin/env python3
"""Pre-pass of fx3d_knn_ws as two launches (default) against one launch with a cloud-local meeting (option knn_prepass_fused): same-box A/B."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import flux3d_jl_amd as fx
from flux3d_jl_amd import _lib
from bench_ops import gpu_time
rng = np.random.default_rng(5)
for (D, N, B, k) in ((64, 1024, 32, 20), (32, 1024, 32, 20), (128, 1024, 8, 20), (64, 2048, 16, 20), (64, 512, 64, 10)):
    x = fx.gpu(np.asfortranarray(rng.standard_normal((D, N, B)).astype(np.float32)))
    ref = None
    for rnd in range(3):
        row = []
        for split in (1, 0):
            _lib.set_option("knn_prepass_fused", 1 - split)
            idx = fx.knn(x, k, drop_first=True, return_dist=False).to_host()
            if ref is None: ref = idx
            ok = np.array_equal(idx, ref)
            mn, md = gpu_time(lambda: fx.knn(x, k, drop_first=True, return_dist=False), reps=8, inner=4)
            row.append(f"{'two launches' if split else 'fused'}: {mn:7.1f}{'' if ok else ' MISMATCH'}")
        print(f"D={D} N={N} B={B} k={k}: " + "   ".join(row), flush=True)
_lib.set_option("knn_prepass_fused", 0)
