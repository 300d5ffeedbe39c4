#!/usr/bin/env python3
"""D = 3 kNN, 32 < k + drop <= 48: the wide geometry (one block of four waves per CU) against the compact one (two blocks per CU),
same box, alternating (option knn_d3_no_compact)."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import flux3d_jl_amd as fx
from flux3d_jl_amd import _lib
from bench_ops import gpu_time
rng = np.random.default_rng(5)
x = fx.gpu(np.asfortranarray(rng.standard_normal((3, 1024, 32)).astype(np.float32)))
for k in (40, 47, 48, 52, 56, 60, 63):
    row = []
    ref = None
    for nomid in (1, 0, 1, 0):
        _lib.set_option("knn_d3_no_compact", nomid)
        idx = fx.knn(x, k, drop_first=True, return_dist=False).to_host()
        if ref is None: ref = idx
        mn, md = gpu_time(lambda: fx.knn(x, k, drop_first=True, return_dist=False), reps=8, inner=4)
        row.append(f"{'wide' if nomid else 'mid '}: {mn:6.1f}{'' if np.array_equal(idx, ref) else ' MISMATCH'}")
    print(f"D=3 k={k}: " + "  ".join(row), flush=True)
_lib.set_option("knn_d3_no_compact", 0)

# ---- larger clouds (several image chunks in the compact geometry)
for (N, B) in ((2048, 16), (4096, 8), (8192, 4), (1600, 20)):
    x = fx.gpu(np.asfortranarray(rng.standard_normal((3, N, B)).astype(np.float32)))
    for k in (32, 40, 47):
        row = []; ref = None
        for mode in (1, 0, 1, 0):   # 1 = wide only, 2 = compact for one-image clouds only (as committed), 0 = compact with chunks
            _lib.set_option("knn_d3_no_compact", mode)
            idx = fx.knn(x, k, drop_first=True, return_dist=False).to_host()
            if ref is None: ref = idx
            mn, md = gpu_time(lambda: fx.knn(x, k, drop_first=True, return_dist=False), reps=6, inner=3)
            row.append(f"{('compact', 'wide')[mode]}: {mn:6.1f}{'' if np.array_equal(idx, ref) else ' MISMATCH'}")
        print(f"D=3 N=M={N} B={B} k={k}: " + "  ".join(row), flush=True)
_lib.set_option("knn_d3_no_compact", 0)
