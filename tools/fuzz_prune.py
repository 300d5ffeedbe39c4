#!/usr/bin/env python3
"""Randomised parity sweep of the spatially pruned fp16 nearest-neighbour kernel (round 6): shapes that take the pruned launch
(one LDS image per cloud, 1024 <= max(N, M) <= 4096, any B), the distributions of tools/fuzz_parity.py plus surfaces, clouds of
different extent / position, identical clouds; indices of fx3d_chamfer_fwd bit for bit against the CPU oracle, the loss bit for
bit against the unpruned launch (option nn1_prune = 0) and equal from call to call.

  python tools/fuzz_prune.py [--seconds 300] [--seed 1]        exit code 1 on the first mismatch (case is printed)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import flux3d_jl_amd as fx  # noqa: E402
import oracle as orc  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402
from fuzz_parity import KINDS, cloud  # noqa: E402


def surface(rng, N, B):
    v = rng.standard_normal((3, N, B))
    v /= np.linalg.norm(v, axis=0, keepdims=True)
    return np.asfortranarray((v * rng.uniform(0.5, 2.0, (3, 1, 1))).astype(np.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0, ncase = time.time(), 0
    kinds = KINDS + ["surface", "surface"]
    while time.time() - t0 < args.seconds:
        kind = kinds[int(rng.integers(0, len(kinds)))]
        B = int(rng.choice([1, 1, 2, 3, 8, 17]))
        big = int(rng.choice([1024, 1500, 2048, 3000, 4000, 4095, 4096, 4096]))
        other = big if rng.random() < 0.4 else int(rng.integers(1, big + 1))
        N, M = (big, other) if rng.random() < 0.5 else (other, big)
        mk = (lambda n: surface(rng, n, B)) if kind == "surface" else (lambda n: cloud(rng, 3, n, B, kind))
        x = mk(N)
        same = N == M and rng.random() < 0.15
        y = x if same else mk(M)
        desc = f"{kind} N={N} M={M} B={B}{' A==B' if same else ''}"
        r = rng.random()
        if not same and r < 0.35:   # clouds of different extent / position
            ratio = float(np.exp(rng.uniform(np.log(0.05), np.log(50.0)))) if r < 0.25 else float(np.exp(rng.uniform(0.0, np.log(1e6))))
            shift = float(rng.choice([0.0, 0.0, 0.3, 1.0, 30.0])) * ratio
            y = np.asfortranarray((y * np.float32(ratio) + np.float32(shift)).astype(np.float32))
            desc += f" y*{ratio:.3g}+{shift:.3g}"
        if rng.random() < 0.05:     # a few non-finite coordinates among the queries / candidates
            y = y.copy(order="F")
            y[int(rng.integers(0, 3)), int(rng.integers(0, M)), int(rng.integers(0, B))] = np.float32(rng.choice([np.inf, -np.inf, np.nan]))
            desc += " +nonfinite"
        dx, dy = fx.gpu(x), fx.gpu(y)
        loss, ix, iy = fx.chamfer_distance(dx, dy, w1=0.7, w2=1.3, return_indices=True)
        loss2 = fx.chamfer_distance(dx, dy, w1=0.7, w2=1.3)
        ox, oy = orc.nn1(x, y)
        ok = np.array_equal(ix.to_host(), ox) and np.array_equal(iy.to_host(), oy)
        if not ok:
            desc += " [indices differ from the oracle]"
        same_bits = lambda a, b: np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)  # noqa: E731
        if ok and not same_bits(loss, loss2):
            ok = False
            desc += f" [loss not reproducible: {loss!r} then {loss2!r}]"
        if ok:
            with _lib.option("nn1_prune", 0):
                loss0 = fx.chamfer_distance(dx, dy, w1=0.7, w2=1.3)
            if not (same_bits(loss, loss0) or np.isclose(loss, loss0, rtol=1e-6, atol=0)):
                ok = False
                desc += f" [loss {loss!r} vs unpruned {loss0!r}]"
        ncase += 1
        if not ok:
            print("MISMATCH:", desc, "seed", args.seed, "case", ncase, flush=True)
            np.savez("/tmp/fuzz_prune_fail.npz", x=x, y=y)
            return 1
    print(f"{ncase} random cases in {time.time() - t0:.0f} s: all bit-identical to the oracle", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
