"""chamfer_distance forward at the C2 shape (B = 32, N = M = 4096) and kNN D = 3 at C4 with ONE point of every cloud scaled
by 1 ... 1000 (a stray far point sets the bounding box, hence the fp16 scale of the filter).   python tools/nn1_outlier_time.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402

rng = np.random.default_rng(5)
bx = rng.standard_normal((3, 4096, 32)).astype(np.float32)
by = rng.standard_normal((3, 4096, 32)).astype(np.float32)
for fac in (1.0, 3.0, 10.0, 30.0, 100.0, 1000.0, 1e5):
    x, y = bx.copy(), by.copy()
    x[:, 0, :] *= np.float32(fac)
    y[:, 1, :] *= np.float32(fac)
    dx, dy = fx.gpu(np.asfortranarray(x)), fx.gpu(np.asfortranarray(y))
    out = fx.DeviceArray.empty((1,), np.float32)
    for _ in range(3):
        fx.chamfer_distance(dx, dy, loss_out=out, sync=False)
    fx.synchronize()
    e0, e1 = fx.Event(), fx.Event()
    e0.record()
    for _ in range(10):
        fx.chamfer_distance(dx, dy, loss_out=out, sync=False)
    e1.record()
    e1.synchronize()
    t_ch = e0.elapsed_ms(e1) * 100
    dk = fx.gpu(np.asfortranarray(x[:, :1024, :]))
    for _ in range(2):
        fx.knn(dk, 20, drop_first=True, return_dist=False)
    fx.synchronize()
    e0.record()
    for _ in range(10):
        fx.knn(dk, 20, drop_first=True, return_dist=False)
    e1.record()
    e1.synchronize()
    print(f"one point x{fac:9.1f}: chamfer C2 {t_ch:8.1f} us   kNN D=3 C4 {e0.elapsed_ms(e1) * 100:8.1f} us", flush=True)
