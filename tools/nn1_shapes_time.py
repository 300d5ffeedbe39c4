#!/usr/bin/env python3
"""chamfer_distance forward over a spread of shapes (uniform clouds): per-call device time and point pairs per second --
a check for launch-plan cliffs outside the BASELINE configs."""
import os
import sys

import ctypes as C

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import flux3d_jl_amd as fx  # noqa: E402
from bench_ops import gpu_time  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402

rng = np.random.default_rng(11)
out = fx.DeviceArray.empty((1,), np.float32)
for N, M, B in [(1024, 1024, 2), (1024, 1024, 32), (1024, 1024, 256), (2048, 2048, 32), (4096, 4096, 32), (4096, 4096, 256),
                (8192, 8192, 8), (4096, 1024, 32), (1024, 4096, 32), (100, 100, 64), (333, 777, 16), (5000, 5000, 8),
                (10000, 10000, 4), (20000, 3000, 2), (50000, 50000, 1), (4097, 4097, 32), (6000, 6000, 32)]:
    x = fx.gpu(np.asfortranarray(rng.random((3, N, B), dtype=np.float32)))
    y = fx.gpu(np.asfortranarray(rng.random((3, M, B), dtype=np.float32)))
    mn, md = gpu_time(lambda: fx.chamfer_distance(x, y, loss_out=out, sync=False), reps=12, inner=6)
    buf = C.create_string_buffer(256); _lib.call("fx3d_nn1_plan_describe", N, M, B, 3, buf, 256)
    print(f"N={N:<6d} M={M:<6d} B={B:<4d} min {mn:9.1f} us  median {md:9.1f} us  {B * N * M / mn / 1e6:7.2f} T pairs/s  | {buf.value.decode()}", flush=True)
