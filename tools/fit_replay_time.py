"""The fit_mesh iteration replayed (FitStepGraph), scatter and ordered sampling adjoint, the regularisers folded into the sampling launches or not: per-iteration events and 500 back-to-back replays;
the tutorial's pair (sphere -> teapot centred and scaled into the sphere's box, examples/fit_mesh.jl:46-54)."""
import os, sys, numpy as np
sys.path.insert(0, ".")
import flux3d_jl_amd as fx
fx.set_device(0)
g = os.path.join("tests", "golden")
for order in (((True, True),), ((True, False), (False, False), (True, True))):
  for ordered, fold in order:
    tv, tf = fx.load_obj(os.path.join(g, "teapot.obj"))
    tv = tv - tv.mean(1, keepdims=True)
    tv = np.asfortranarray((tv / np.abs(tv).max()).astype(np.float32))
    src, tgt = fx.gpu(fx.load_trimesh(os.path.join(g, "sphere.obj"))), fx.gpu(fx.TriMesh([tv], [tf]))
    xg = fx.DeviceArray.zeros((3, int(src.dev("verts_packed").shape[1])), np.float32)
    step = fx.FitStepGraph(xg, src, tgt, fx.Momentum(1.0, 0.9), num_samples=5000, ordered=ordered, fold=fold)
    for _ in range(20): step.step()
    step.synchronize()
    ev = [fx.Event() for _ in range(101)]
    ev[0].record(step.stream)
    for i in range(100):
        step.step(); ev[i + 1].record(step.stream)
    step.synchronize()
    ts = np.array([ev[i].elapsed_ms(ev[i + 1]) for i in range(100)])
    e0, e1 = fx.Event(), fx.Event()
    e0.record(step.stream)
    for i in range(500): step.step()
    e1.record(step.stream); e1.synchronize()
    print("ordered", ordered, "fold", fold, "per-iteration events: min", round(ts.min()*1e3,1), "median", round(float(np.median(ts))*1e3,1), "| 500 back to back:", round(e0.elapsed_ms(e1)*2,1), "us per iteration")
