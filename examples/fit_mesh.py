#!/usr/bin/env python3
"""The reference's fit_mesh tutorial (examples/fit_mesh.jl) on the MI355X path: deform a sphere towards
a target mesh by optimising per-vertex offsets under chamfer + 0.1 laplacian + edge loss, with every
step (sampling, chamfer forward/backward, sampler adjoint, regularisers, Momentum update) on the device.

  python examples/fit_mesh.py [--iters 500] [--target tests/golden/teapot.obj]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402


def normalized(path):
    v, f = fx.load_obj(path)
    v = (v - v.mean(1, keepdims=True)) / v.std()   # Flux3D.normalize!: zero mean, unit std
    return fx.TriMesh([np.asfortranarray(v.astype(np.float32))], [f])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--target", default=os.path.join(ROOT, "tests", "golden", "teapot.obj"))
    ap.add_argument("--samples", type=int, default=5000)
    ap.add_argument("--graph", action="store_true", help="capture one iteration as a hipGraph and replay it")
    ap.add_argument("--atomics", action="store_true", help="sampling adjoint with float atomics (sums in arrival order) instead of the ordered form")
    ap.add_argument("--separate-step", action="store_true", help="the optimiser step as a launch of its own")
    ap.add_argument("--unfolded", action="store_true", help="the two regularisers as launches of their own (default: they ride in the sampling launches)")
    args = ap.parse_args()
    src = fx.gpu(normalized(os.path.join(ROOT, "tests", "golden", "sphere.obj")))
    tgt = fx.gpu(normalized(args.target))
    x = fx.DeviceArray.zeros((3, src.get_verts_packed().shape[1]), np.float32)
    opt = fx.Momentum(1.0, 0.9)        # examples/fit_mesh.jl:87-88
    t0 = time.perf_counter()
    if args.graph:
        step = fx.FitStepGraph(x, src, tgt, opt, args.samples, ordered=not args.atomics, step_in_launch=not args.separate_step, fold=not args.unfolded)  # runs iteration 1 eagerly, records iteration 2
        for it in range(2, args.iters + 1):
            loss = step.step()
            if it % 50 == 1 or it == args.iters:
                step.synchronize()
                print(f"itr {it:5d}  loss {float(loss.item()):.6f}", flush=True)
        step.synchronize()
    else:
        for it in range(1, args.iters + 1):
            loss, g = fx.loss_dolphin(x, src, tgt, args.samples, with_grad=True, sync=False)
            opt.update(x, g)
            if it % 50 == 1 or it == args.iters:  # the only host round trip of the loop
                print(f"itr {it:5d}  loss {float(loss.item()):.6f}", flush=True)
    fx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.iters} iterations in {dt:.3f} s  ({dt / args.iters * 1e3:.3f} ms / iteration)")


if __name__ == "__main__":
    main()
