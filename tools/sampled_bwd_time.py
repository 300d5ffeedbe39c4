"""The sampling adjoint in pieces (one source mesh = the fit loop's shape, and B = 8): fx3d_sample_points_bwd ordered / atomics,
fx3d_chamfer_bwd, fx3d_chamfer_sampled_bwd ordered / atomics (+ the step).  Min of single calls between events, us."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402

fx.set_device(0)
g = os.path.join(ROOT, "tests", "golden")


def call_us(fn, n=60, warm=5):
    for _ in range(warm):
        fn()
    fx.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = fx.Event(), fx.Event()
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_ms(e1))
    return round(best * 1e3, 2)


for nb, n in ((1, 5000), (8, 5000), (1, 2000)):
    src = fx.gpu(fx.load_trimesh(*[os.path.join(g, "sphere.obj")] * nb))
    tgt = fx.gpu(fx.load_trimesh(*[os.path.join(g, "teapot.obj")] * nb))
    A, fa, r1, r2 = fx.sample_points(src, n, seed=1, return_draws=True)
    Bp = fx.sample_points(tgt, n, seed=2)
    _, ix, iy = fx.chamfer_distance(A, Bp, return_indices=True)
    gA, _ = fx.chamfer_distance_grad(A, Bp, ix, iy)
    out = fx.DeviceArray.zeros((3, src.V, nb), np.float32)
    res = {"B": nb, "n": n}
    res["empty_call"] = call_us(lambda: None)
    res["chamfer_bwd"] = call_us(lambda: fx.chamfer_distance_grad(A, Bp, ix, iy))
    res["sample_bwd_ordered"] = call_us(lambda: fx.sample_points_grad(src, fa, r1, r2, gA, out=out))
    res["sample_bwd_atomics"] = call_us(lambda: fx.sample_points_grad(src, fa, r1, r2, gA, out=out, ordered=False))
    res["sampled_bwd_ordered"] = call_us(lambda: fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=src, draws_a=(fa, r1, r2), out_a=out))
    res["sampled_bwd_atomics"] = call_us(lambda: fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=src, draws_a=(fa, r1, r2), out_a=out, ordered=False))
    print(res, flush=True)

if os.environ.get("FX3D_HIP_LIB", "").endswith("sgprobe.so"):   # a -DFX3D_SG_PROBE build: the gather's phases (block 0, last call)
    import ctypes as C
    from flux3d_jl_amd import _lib
    src = fx.gpu(fx.load_trimesh(os.path.join(g, "sphere.obj")))
    A, fa, r1, r2 = fx.sample_points(src, 5000, seed=1, return_draws=True)
    gA = fx.gpu(np.asfortranarray(np.random.default_rng(0).standard_normal((3, 5000, 1)).astype(np.float32)))
    for _ in range(3):
        fx.sample_points_grad(src, fa, r1, r2, gA)
    fx.synchronize()
    buf = (C.c_uint64 * 16)()
    _lib.load().fx3d_debug_sg_probe(buf)
    t = list(buf)[:10]
    print("gather phases (us): zero+count, scan, place, sort, stage, big faces + entries + vertices:", [round((t[i + 1] - t[i]) / 100.0, 2) for i in range(6)], "total", (t[6] - t[0]) / 100.0)

