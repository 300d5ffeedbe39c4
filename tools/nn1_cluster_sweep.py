"""chamfer_distance forward at the C2 shape on clustered clouds: 40 cluster centres ~ N(0, 3^2), jitter sigma swept, the
second cloud with the SAME centres or its own.  Shows where the filter's band stops resolving the points of a cluster
(sigma / extent below ~1e-3) and what queries outside the other cloud's range cost.   python tools/nn1_cluster_sweep.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402

B, N = 32, 4096


def timed(x, y):
    dx, dy = fx.gpu(x), fx.gpu(y)
    out = fx.DeviceArray.empty((1,), np.float32)
    for _ in range(5):
        fx.chamfer_distance(dx, dy, loss_out=out, sync=False)
    fx.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = fx.Event(), fx.Event()
        e0.record()
        for _ in range(10):
            fx.chamfer_distance(dx, dy, loss_out=out, sync=False)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_ms(e1) * 100)
    return best


for sigma in (1e-1, 1e-2, 1e-3, 1e-4):
    for same_centres in (True, False):
        rng = np.random.default_rng(11)
        ca = rng.standard_normal((3, 40, B)) * 3
        cb = ca if same_centres else rng.standard_normal((3, 40, B)) * 3
        x = np.asfortranarray((ca[:, rng.integers(0, 40, N), :] + rng.standard_normal((3, N, B)) * sigma).astype(np.float32))
        y = np.asfortranarray((cb[:, rng.integers(0, 40, N), :] + rng.standard_normal((3, N, B)) * sigma).astype(np.float32))
        print(f"sigma {sigma:7.0e}  centres {'shared' if same_centres else 'own   '}: {timed(x, y):8.1f} us", flush=True)
