"""Host enqueue time vs total time per step of the sharded chamfer evaluation at world size 1 (C2 shape): is the overlapped
form (collective on a second stream) bound by the host or by the GPU?  Measured: serial 8.9 us of host time per step (57.2 us
total), overlapped 25 us (63 us total) -- neither is host bound; the overlapped form pays ~6 us of cross-stream event handling
per step on the GPU side, which buys hiding the all-reduce latency at N > 1.  (An enqueue worker thread for the collective
half was tried and changed nothing: 24.4 vs 25.9 us of host time, 62.9 vs 63.2 us per step.)"""
import time, sys, numpy as np
sys.path.insert(0, "/root/repo")
import flux3d_jl_amd as fx
from flux3d_jl_amd import _lib
from flux3d_jl_amd.distributed import NativeComm, NativeShardedChamfer
comm = NativeComm(0, 1)
x = fx.gpu(fx.synth.uniform_cloud(1, 3, 4096, 32)); y = fx.gpu(fx.synth.uniform_cloud(2, 3, 4096, 32))
s = fx.Stream.create()
for name, kw in (("serial", dict()), ("overlap", dict(overlap=True))):
    sh = NativeShardedChamfer(comm, **kw)
    with fx.stream(s):
        for _ in range(50): sh(x, y, 32, sync=False)
        sh.synchronize(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(500): sh(x, y, 32, sync=False)
        t1 = time.perf_counter()
        sh.synchronize(); s.synchronize()
        t2 = time.perf_counter()
    print(f"{name:18s}: host enqueue {1e6*(t1-t0)/500:.1f} us/step, total {1e6*(t2-t0)/500:.1f} us/step")
