import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import flux3d_jl_amd as fx
import bench
for norm, tag in ((True, "unit"), (False, "own")):
    a, b = bench.surface_clouds(fx, normalise=norm)
    a.to_host().ravel(order="F").astype(np.float32).tofile(f"/tmp/sa_{tag}.bin")
    b.to_host().ravel(order="F").astype(np.float32).tofile(f"/tmp/sb_{tag}.bin")
print("dumped")
