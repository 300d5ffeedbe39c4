#!/usr/bin/env python3
"""A/B of the small-problem exact kernel (nn1_tiny_kernel, option nn1_tiny_mpairs) against the fp16-filter kernel:
per-call time (min of 100 single calls between HIP events, as bench.py's `configs`) of chamfer_distance forward over shapes
from the reference harness's n = 64 (benchmarks/metrics.jl:40) to C5's N = 1024 shard.  Also checks that both paths
return the same indices.   usage: python tools/nn1_tiny_ab.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux3d_jl_amd as fx  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402
from bench import _per_call_ms  # noqa: E402

SHAPES = [(64, 64, 1), (256, 256, 1), (1024, 1024, 1), (1024, 1024, 2), (2048, 2048, 1), (1024, 1024, 4), (4096, 4096, 1),
          (1024, 1024, 8), (2048, 2048, 4), (1024, 1024, 16), (1024, 1024, 32), (5000, 5000, 1), (300, 4096, 2), (512, 512, 32)]
loss_dev = fx.DeviceArray.empty((1,), np.float32)
print(f"{'N':>6} {'M':>6} {'B':>3} {'Mpairs':>8} {'f16 us':>8} {'tiny us':>8}  same")
for N, M, B in SHAPES:
    rng = np.random.default_rng(N + M + B)
    x = fx.gpu(np.asfortranarray(rng.random((3, N, B)).astype(np.float32)))
    y = fx.gpu(np.asfortranarray(rng.random((3, M, B)).astype(np.float32)))
    res = {}
    for name, v in (("f16", 0), ("tiny", 1 << 20)):
        with _lib.option("nn1_tiny_mpairs", v):
            l, ix, iy = fx.chamfer_distance(x, y, return_indices=True)
            t = _per_call_ms(fx, lambda: fx.chamfer_distance(x, y, loss_out=loss_dev, sync=False))
            res[name] = (t["min_ms"] * 1e3, l, ix.to_host(), iy.to_host())
    same = res["f16"][1] == res["tiny"][1] and np.array_equal(res["f16"][2], res["tiny"][2]) and np.array_equal(res["f16"][3], res["tiny"][3])
    print(f"{N:6d} {M:6d} {B:3d} {2e-6 * B * N * M:8.1f} {res['f16'][0]:8.1f} {res['tiny'][0]:8.1f}  {same}")
