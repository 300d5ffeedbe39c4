import os, sys, numpy as np
sys.path.insert(0, ".")
import flux3d_jl_amd as fx
fx.set_device(0)
g = os.path.join("tests", "golden")
tv, tf = fx.load_obj(os.path.join(g, "teapot.obj"))
tv = tv - tv.mean(1, keepdims=True)
tv = np.asfortranarray((tv / np.abs(tv).max()).astype(np.float32))
src, tgt = fx.gpu(fx.load_trimesh(os.path.join(g, "sphere.obj"))), fx.gpu(fx.TriMesh([tv], [tf]))
x = fx.DeviceArray.zeros((3, src.V), np.float32)
step = fx.FitStepGraph(x, src, tgt, fx.Momentum(1.0, 0.9), num_samples=5000)
for it in (0, 100, 500, 2000):
    while step.iterations < max(it, 1):
        step.step()
    step.synchronize()
    m = fx.offset(src, x)
    _, fi, _, _ = fx.sample_points(m, 5000, seed=3, return_draws=True)
    c = np.bincount(fi.to_host()[:, 0], minlength=5120)
    print("iteration", step.iterations, "faces with >8 draws:", int((c > 8).sum()), "max", int(c.max()), ">16:", int((c > 16).sum()), ">32:", int((c > 32).sum()), "draws on them:", int(c[c > 8].sum()))
