import os, sys, numpy as np
sys.path.insert(0, ".")
import flux3d_jl_amd as fx
import bench
fx.set_device(0)
loss_dev = fx.DeviceArray.empty((1,), np.float32)
for norm in (True, False):
    sa, sb = bench.surface_clouds(fx, normalise=norm)
    r = bench._per_call_ms(fx, lambda: fx.chamfer_distance(sa, sb, loss_out=loss_dev, sync=False))
    print("surface clouds normalise", norm, round(r["min_ms"] * 1e3, 1), "us")
