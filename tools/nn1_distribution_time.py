"""chamfer_distance forward at the C2 shape (B = 32, N = M = 4096, D = 3) on data distributions that stress the filter's
band and FIFO: uniform, Gaussian, tight clusters, lattice (exact ties everywhere), duplicated points, a far outlier
(scale set by it), identical clouds.  Prints microseconds per call.   python tools/nn1_distribution_time.py [kind,kind,...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402

rng = np.random.default_rng(7)
B, N = 32, 4096


def make(kind):
    if kind == "uniform":
        return rng.random((3, N, B))
    if kind == "normal+100":
        return rng.standard_normal((3, N, B)) + 100.0
    if kind == "clusters":
        c = rng.standard_normal((3, 40, B)) * 3
        return c[:, rng.integers(0, 40, N), :] + rng.standard_normal((3, N, B)) * 1e-3
    if kind == "lattice":
        return rng.integers(0, 16, (3, N, B)) * 0.0625
    if kind == "dupes":
        x = rng.random((3, N, B))
        x[:, N // 2:, :] = x[:, : N // 2, :]
        return x
    if kind == "outlier":
        x = rng.random((3, N, B)) * 1e-2
        x[:, 0, :] = 1e4
        return x
    raise KeyError(kind)


KINDS = sys.argv[1].split(",") if len(sys.argv) > 1 else ("uniform", "normal+100", "clusters", "lattice", "dupes", "outlier")
for kind in KINDS:
    for same in (False, True):
        x = np.asfortranarray(make(kind).astype(np.float32))
        y = x if same else np.asfortranarray(make(kind).astype(np.float32))
        dx, dy = fx.gpu(x), fx.gpu(y)
        out = fx.DeviceArray.empty((1,), np.float32)
        for _ in range(5):
            fx.chamfer_distance(dx, dy, loss_out=out, sync=False)
        fx.synchronize()
        best = 1e30
        for _ in range(3):  # min of three groups of ten back-to-back calls (the first group of a process pays one-time set-up)
            e0, e1 = fx.Event(), fx.Event()
            e0.record()
            for _ in range(10):
                fx.chamfer_distance(dx, dy, loss_out=out, sync=False)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_ms(e1) * 100)
        print(f"{kind:12s} {'A==B' if same else 'A!=B'}: {best:9.1f} us  loss {float(out.item()):.6g}", flush=True)
