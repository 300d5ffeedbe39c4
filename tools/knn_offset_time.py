"""Feature-space kNN (C4': D = 64, k = 20, B = 32 x 1024) on data with a common offset of 0 ... 100 standard deviations:
the fp16 filter centres every cloud per dimension, so the time must not depend on the offset (before centring: 2.3x at 3
sigma, 47x from 10 sigma on, every query on the exact fallback).   python tools/knn_offset_time.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx
rng = np.random.default_rng(3)
base = rng.standard_normal((64, 1024, 32)).astype(np.float32)
for shift in (0.0, 1.0, 3.0, 10.0, 30.0, 100.0):
    for relu in (False, True):
        x = base + np.float32(shift)
        if relu:
            x = np.maximum(x, 0)
        dx = fx.gpu(np.asfortranarray(x.astype(np.float32)))
        for _ in range(2):
            fx.knn(dx, 20, drop_first=True, return_dist=False)
        fx.synchronize()
        e0, e1 = fx.Event(), fx.Event()
        e0.record()
        for _ in range(5):
            fx.knn(dx, 20, drop_first=True, return_dist=False)
        e1.record(); e1.synchronize()
        print(f"shift {shift:6.1f} relu={relu}: {e0.elapsed_ms(e1) / 5 * 1e3:9.1f} us", flush=True)
