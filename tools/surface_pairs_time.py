#!/usr/bin/env python3
"""Which surface-sampled cloud pairs are slow?  chamfer_distance at C2's shape (B = 32, N = M = 4096) with all 32 batch
elements the SAME (mesh i, mesh j) pair, for every pair of {teapot, sphere, 8 ModelNet OFF meshes}; plus each mesh's extent.
  python tools/surface_pairs_time.py        -> table on stdout"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import flux3d_jl_amd as fx  # noqa: E402
import modelnet_chamfer_eval as ev  # noqa: E402
import shutil, tempfile  # noqa: E402

tmp = tempfile.mkdtemp()
for z in ("ModelNet10.zip", "ModelNet40.zip"):
    shutil.copy(os.path.join(ROOT, "tests", "golden", "modelnet", z), tmp)
meshes = [("teapot",) + tuple(fx.load_obj(os.path.join(ROOT, "tests", "golden", "teapot.obj"))),
          ("sphere",) + tuple(fx.load_obj(os.path.join(ROOT, "tests", "golden", "sphere.obj")))] + ev.listing(tmp)
shutil.rmtree(tmp, ignore_errors=True)
B, n = 32, 4096
clouds = []
for name, v, f in meshes:
    m = fx.gpu(fx.TriMesh([v] * B, [f] * B))
    clouds.append((fx.sample_points(m, n, seed=11), fx.sample_points(m, n, seed=12)))
    print(f"{name:44s} V={v.shape[1]:6d} F={f.shape[1]:6d} min={v.min(axis=1)} max={v.max(axis=1)}")
loss = fx.DeviceArray.empty((1,), np.float32)


def t(a, b, reps=20):
    s = fx.Stream.create()
    with fx.stream(s):
        for _ in range(3):
            fx.chamfer_distance(a, b, loss_out=loss, sync=False)
        e0, e1 = fx.Event(), fx.Event()
        e0.record(s)
        for _ in range(reps):
            fx.chamfer_distance(a, b, loss_out=loss, sync=False)
        e1.record(s)
        s.synchronize()
    return e0.elapsed_ms(e1) * 1e3 / reps


print("us per call, row = cloud A's mesh, column = cloud B's mesh (diagonal: two samplings of the same mesh)")
for i in range(len(meshes)):
    print(f"{meshes[i][0][:28]:28s}", " ".join(f"{t(clouds[i][0], clouds[j][1]):7.1f}" for j in range(len(meshes))))
