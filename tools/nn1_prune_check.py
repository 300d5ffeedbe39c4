"""Spatial pruning of nn1_f16_kernel (round 6): the pruned launch against the unpruned one (option nn1_prune = 0) -- indices and
loss bit for bit -- on the shapes and distributions the pruning reacts to, and both timed.   python tools/nn1_prune_check.py [--oracle]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402

rng = np.random.default_rng(11)


def make(kind, n, b):
    if kind == "uniform":
        return rng.random((3, n, b))
    if kind == "normal":
        return rng.standard_normal((3, n, b))
    if kind == "sphere":
        v = rng.standard_normal((3, n, b))
        return v / np.linalg.norm(v, axis=0, keepdims=True)
    if kind == "clusters":
        c = rng.standard_normal((3, 40, b)) * 3
        return np.stack([c[:, rng.integers(0, 40, n), i] for i in range(b)], -1) + rng.standard_normal((3, n, b)) * 1e-2
    if kind == "lattice":
        return rng.integers(0, 16, (3, n, b)) * 0.0625
    if kind == "dupes":
        x = rng.random((3, n, b))
        x[:, n // 2:, :] = x[:, : n // 2, :]
        return x
    if kind == "outlier":
        x = rng.random((3, n, b)) * 1e-2
        x[:, 0, :] = 1e4
        return x
    if kind == "shifted":   # the query cloud mostly outside the candidates' box
        return rng.random((3, n, b)) + 0.8
    if kind == "line":
        t = rng.random((1, n, b))
        return np.concatenate([t, t, t], 0)
    raise KeyError(kind)


def timed(dx, dy, out):
    for _ in range(3):
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


use_oracle = "--oracle" in sys.argv
if use_oracle:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle  # noqa: E402
bad = 0
cases = [("uniform", 4096, 4096, 64), ("uniform", 4096, 4096, 256), ("sphere", 4096, 4096, 32), ("normal", 4096, 4096, 32), ("clusters", 4096, 4096, 32),
         ("lattice", 4096, 4096, 32), ("dupes", 4096, 4096, 32), ("outlier", 4096, 4096, 32), ("shifted", 4096, 4096, 32), ("uniform", 4000, 4090, 32),
         ("uniform", 4096, 4096, 32), ("uniform", 4096, 4096, 2), ("normal", 4096, 4096, 8), ("sphere", 4096, 4096, 8), ("clusters", 4096, 4096, 8),
         ("lattice", 4096, 4096, 4), ("dupes", 4096, 4096, 4), ("outlier", 4096, 4096, 4), ("line", 4096, 4096, 4), ("uniform", 4000, 3000, 8),
         ("uniform", 1024, 4096, 8), ("uniform", 2048, 2048, 16), ("uniform", 1500, 1100, 8), ("sphere", 3333, 4095, 5), ("uniform", 4097, 4096, 2)]
for kind, n, m, b in cases:
    for same in (False, True):
        if same and n != m:
            continue
        x = np.asfortranarray(make(kind, n, b).astype(np.float32))
        y = x if same else np.asfortranarray((make(kind, m, b) + (0.0 if kind != "shifted" else -0.8)).astype(np.float32))
        dx, dy = fx.gpu(x), fx.gpu(y)
        out = fx.DeviceArray.empty((1,), np.float32)
        res = {}
        for pr in (1, 0):
            with _lib.option("nn1_prune", pr):
                loss, ix, iy = fx.chamfer_distance(dx, dy, return_indices=True)
                res[pr] = (loss, ix.to_host(), iy.to_host(), timed(dx, dy, out))
        ok = res[1][0] == res[0][0] and np.array_equal(res[1][1], res[0][1]) and np.array_equal(res[1][2], res[0][2])
        msg = ""
        if use_oracle and b * n * m <= 8 * 4096 * 4096:
            _, ox, oy, _ = oracle.chamfer_distance(x, y, return_all=True)
            ok2 = np.array_equal(res[1][1], ox) and np.array_equal(res[1][2], oy)
            msg = f" oracle {'ok' if ok2 else 'MISMATCH'}"
            ok = ok and ok2
        bad += not ok
        print(f"{kind:9s} {n:5d}x{m:5d} B={b:3d} {'A==B' if same else 'A!=B'}: pruned {res[1][3]:8.1f} us  unpruned {res[0][3]:8.1f} us  "
              f"{'identical' if ok else 'DIFFERENT'} (loss {res[1][0]:.8g} / {res[0][0]:.8g}){msg}", flush=True)
print("FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)
