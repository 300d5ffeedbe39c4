#!/usr/bin/env python3
"""C2's chamfer forward, K evaluations back to back: launched one by one against ONE replayed hipGraph of the K launches
(what a caller with a fixed evaluation loop can do): microseconds per evaluation, same box, alternating."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import flux3d_jl_amd as fx

K = 20
a = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, 3, 4096, 32))
b = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, 3, 4096, 32))
loss = fx.DeviceArray.empty((1,), np.float32)
s = fx.Stream.create()


def step():
    fx.chamfer_distance(a, b, loss_out=loss, sync=False)


with fx.stream(s):
    for _ in range(50):
        step()
    s.synchronize()
    g = fx.Graph()
    with g.capture(s):
        for _ in range(K):
            step()

    def timed(fn, reps):
        e0, e1 = fx.Event(), fx.Event()
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s)
        e1.synchronize()
        return e0.elapsed_ms(e1) * 1000.0

    for fn in (lambda: [step() for _ in range(K)], lambda: g.launch()):   # burn-in
        for _ in range(100):
            fn()
    s.synchronize()
    for rnd in range(4):
        eager = min(timed(lambda: [step() for _ in range(K)], 25) / (25 * K) for _ in range(3))
        graph = min(timed(lambda: g.launch(), 25) / (25 * K) for _ in range(3))
        print(f"round {rnd}: launched one by one {eager:6.2f} us per evaluation   one graph of {K} launches {graph:6.2f} us", flush=True)
