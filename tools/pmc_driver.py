#!/usr/bin/env python3
"""A few launches of ONE op of the path, for rocprofv3 --pmc passes (counters go in their own runs: tools/profile_round.sh).

  python tools/pmc_driver.py nn1 | nn1_fullmantissa | knn3 | knn64 | edgeconv3 | fit [--reps 6]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("op", choices=["nn1", "nn1_fullmantissa", "knn3", "knn64", "edgeconv3", "fit"])
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    if a.op == "nn1":      # BASELINE config 2
        x = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, 3, 4096, 32))
        y = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, 3, 4096, 32))
        out = fx.DeviceArray.empty((1,), np.float32)
        fn = lambda: fx.chamfer_distance(x, y, loss_out=out, sync=False)  # noqa: E731
    elif a.op == "nn1_fullmantissa":  # the same shape on numpy's default_rng(7) uniforms: a dataset with one slow query
        rng = np.random.default_rng(7)  # (its wave is the launch's tail: same counters, longer kernel -- DESIGN.md 3.1)
        x, y = (fx.gpu(np.asfortranarray(rng.random((3, 4096, 32)).astype(np.float32))) for _ in range(2))
        out = fx.DeviceArray.empty((1,), np.float32)
        fn = lambda: fx.chamfer_distance(x, y, loss_out=out, sync=False)  # noqa: E731
    elif a.op in ("knn3", "edgeconv3"):  # BASELINE config 4
        x = fx.gpu(fx.synth.uniform_cloud(0x5EED0004, 3, 1024, 32))
        fn = (lambda: fx.knn(x, 20, drop_first=True)) if a.op == "knn3" else (lambda: fx.edgeconv_graph(x, 20))
    elif a.op == "knn64":  # the second EdgeConv's feature space
        x = fx.gpu(np.asfortranarray(np.random.default_rng(1).standard_normal((64, 1024, 32)).astype(np.float32)))
        fn = lambda: fx.knn(x, 20, drop_first=True)  # noqa: E731
    else:                  # one fit_mesh iteration (BASELINE config 3's loop body), eager
        g = os.path.join(ROOT, "tests", "golden")
        src, tgt = fx.gpu(fx.load_trimesh(os.path.join(g, "sphere.obj"))), fx.gpu(fx.load_trimesh(os.path.join(g, "teapot.obj")))
        xo = fx.DeviceArray.zeros((3, 2562), np.float32)
        opt = fx.Momentum(1.0, 0.9)
        it = [0]

        def fn():
            _, grad = fx.loss_dolphin(xo, src, tgt, 5000, seed=100 + 2 * it[0], with_grad=True, sync=False)
            opt.update(xo, grad)
            it[0] += 1
    for _ in range(a.reps):
        fn()
    fx.synchronize()


if __name__ == "__main__":
    main()
