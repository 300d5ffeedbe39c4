#!/usr/bin/env python3
"""Candidate slices of fx3d_knn_ws (few clouds with many rows): per-call time of self-kNN graphs with the slicing off (1), forced
(2 / 4 / 8) and automatic (0) -- the calibration of knn_slices() (csrc/knn.hip) and its evidence."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import flux3d_jl_amd as fx
from flux3d_jl_amd import _lib
from bench_ops import gpu_time
rng = np.random.default_rng(5)
shapes = [(3, 8192, 1, 20), (3, 16384, 1, 20), (3, 32768, 1, 16), (3, 4096, 4, 20), (3, 16384, 4, 20), (3, 8192, 2, 40),
          (64, 2048, 2, 20), (64, 4096, 1, 20), (64, 4096, 4, 20), (64, 8192, 1, 20), (64, 8192, 4, 20), (64, 2048, 16, 20), (16, 8192, 2, 10), (128, 4096, 2, 20)]
for D, N, B, k in shapes:
    x = fx.gpu(np.asfortranarray(rng.standard_normal((D, N, B)).astype(np.float32)))
    row = []
    ref = None
    for S in (1, 2, 4, 8, 0):
        _lib.set_option("knn_slices", S)
        try:
            idx = fx.knn(x, k, drop_first=True, return_dist=False).to_host()
            if ref is None:
                ref = idx
            same = bool(np.array_equal(idx, ref))
            mn, md = gpu_time(lambda: fx.knn(x, k, drop_first=True, return_dist=False), reps=6, inner=3)
            row.append(f"S={S}: {mn:8.1f}{'' if same else ' MISMATCH'}")
        except Exception as e:  # noqa: BLE001
            row.append(f"S={S}: ERR {str(e)[:60]}")
    _lib.set_option("knn_slices", 0)
    print(f"D={D:<3d} N=M={N:<6d} B={B:<3d} k={k:<3d} us/call  " + "  ".join(row), flush=True)
