#!/usr/bin/env python3
"""EdgeConv feature build (K*N, 2F, B) at C4' (F = 64, k = 20, B = 32 x 1024: 335 MB written): the feature loop split over
blockIdx.z (option edge_fsplit = features per block; F = the single loop of rounds 1-3, 0 = automatic) x streaming / ordinary
stores (edge_no_nt), same box, three alternating rounds; every variant's tensor compared with the first one's (parity with the
oracle: tests/)."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import flux3d_jl_amd as fx
from flux3d_jl_amd import _lib
from bench_ops import gpu_time
rng = np.random.default_rng(1)
for (F, N, B, k) in ((64, 1024, 32, 20), (32, 1024, 32, 20), (128, 1024, 8, 20)):
    x = fx.gpu(np.asfortranarray(rng.standard_normal((F, N, B)).astype(np.float32)))
    idx = fx.knn(x, k, drop_first=True, return_dist=False)
    nbytes = 4 * (2 * F * k * N * B + k * N * B + F * N * B)
    first = None
    for rnd in range(3):
        row = []
        for fs in (F, F // 2, F // 4, F // 8, 4, 0):
            for nnt in (1, 0):
                _lib.set_option("edge_fsplit", fs); _lib.set_option("edge_no_nt", nnt)
                o = fx.edge_features(x, idx, layout="mlp")
                if rnd == 0:
                    h = o.to_host()
                    if first is None:
                        first = h
                    elif not np.array_equal(h, first): row.append("MISMATCH")
                mn, md = gpu_time(lambda: fx.edge_features(x, idx, layout="mlp"), reps=6, inner=4)
                row.append(f"fs={fs}{'' if nnt else '+nt'}: {mn:6.1f} us {nbytes / mn / 1e6:5.2f} TB/s")
        print(f"F={F} N={N} B={B} k={k} | " + " | ".join(row), flush=True)
_lib.set_option("edge_fsplit", 0); _lib.set_option("edge_no_nt", 0)
