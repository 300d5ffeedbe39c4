"""kNN (k = 20, self) at the C4 shapes with ONE point of every cloud scaled by 1 ... 1000: the time must not depend on it
(before the mean centring + per-candidate error folding of the feature-space kernel: x10 -> 5.6 times slower, x30 -> 20 times).
  python tools/knn_outlier_time.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx
rng = np.random.default_rng(3)
for D in (3, 64):
    base = rng.standard_normal((D, 1024, 32)).astype(np.float32)
    for fac in (1.0, 3.0, 10.0, 30.0, 100.0, 1000.0):
        x = base.copy()
        x[:, 0, :] *= np.float32(fac)
        dx = fx.gpu(np.asfortranarray(x))
        for _ in range(2):
            fx.knn(dx, 20, drop_first=True, return_dist=False)
        fx.synchronize()
        e0, e1 = fx.Event(), fx.Event()
        e0.record()
        for _ in range(5):
            fx.knn(dx, 20, drop_first=True, return_dist=False)
        e1.record(); e1.synchronize()
        print(f"D={D} one point scaled x{fac:7.1f}: {e0.elapsed_ms(e1) / 5 * 1e3:9.1f} us", flush=True)
