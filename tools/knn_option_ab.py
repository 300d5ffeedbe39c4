#!/usr/bin/env python3
"""Same-box A/B of one integer option of the feature-space kNN (default: knn_row_stages -- the exact phase staged by row blocks
against 16-dimension column slices of the whole cloud), several shapes, three alternating rounds; the index lists of the two
settings are compared with each other (parity with the oracle: tests/, tools/fuzz_parity.py).
   usage: python tools/knn_option_ab.py [option [value_a value_b]]"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import flux3d_jl_amd as fx
from flux3d_jl_amd import _lib
from bench_ops import gpu_time
opt = sys.argv[1] if len(sys.argv) > 1 else "knn_row_stages"
va, vb = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 0)
rng = np.random.default_rng(5)
for (D, N, B, k) in ((64, 1024, 32, 20), (32, 1024, 32, 20), (16, 1024, 32, 20), (64, 512, 64, 10), (64, 1000, 32, 20), (48, 768, 32, 16), (64, 1024, 32, 31)):
    x = fx.gpu(np.asfortranarray(rng.standard_normal((D, N, B)).astype(np.float32)))
    ref = None
    for rnd in range(3):
        row = []
        for v in (va, vb):
            _lib.set_option(opt, v)
            idx = fx.knn(x, k, drop_first=True, return_dist=False).to_host()
            if ref is None: ref = idx
            ok = np.array_equal(idx, ref)
            mn, md = gpu_time(lambda: fx.knn(x, k, drop_first=True, return_dist=False), reps=8, inner=4)
            row.append(f"{opt}={v}: {mn:7.1f}{'' if ok else ' MISMATCH'}")
        print(f"D={D} N={N} B={B} k={k}: " + "   ".join(row), flush=True)
_lib.set_option(opt, 0)
