"""BASELINE configs[2]'s shape: the fit_mesh iteration on eight teapot-class meshes (eight sources against eight targets, 5000 draws
each), replayed as a hipGraph; for `rocprofv3 --kernel-trace` + tools/rocprof_summary.py timeline (one period = from face_cdf)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402

fx.set_device(0)
t = os.path.join(ROOT, "tests", "golden", "teapot.obj")
ordered = "--atomics" not in sys.argv
fold = "--nofold" not in sys.argv
src, tgt = fx.gpu(fx.load_trimesh(*[t] * 8)), fx.gpu(fx.load_trimesh(*[t] * 8))
x = fx.DeviceArray.zeros((3, int(src.dev("verts_packed").shape[1])), np.float32)
step = fx.FitStepGraph(x, src, tgt, fx.Momentum(1.0, 0.9), num_samples=5000, ordered=ordered, fold=fold)
for _ in range(20):
    step.step()
step.synchronize()
e0, e1 = fx.Event(), fx.Event()
e0.record(step.stream)
for _ in range(500):
    step.step()
e1.record(step.stream); e1.synchronize()
print("B = 8, ordered", ordered, "fold", fold, ":", round(e0.elapsed_ms(e1) * 2, 1), "us per iteration, loss", float(step.loss.item()))
