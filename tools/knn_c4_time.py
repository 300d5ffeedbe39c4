#!/usr/bin/env python3
"""C4 / C4' kNN timing for same-box A/Bs (FX3D_HIP_LIB selects the library): min / median of 200 single calls between
HIP events after a burn-in, back-to-back rate of 200 calls, D = 3 and D = 64, k = 20 (+ drop), B = 32 x 1024."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux3d_jl_amd as fx  # noqa: E402
from bench import _per_call_ms  # noqa: E402

c4 = fx.gpu(fx.synth.uniform_cloud(0x5EED0004, 3, 1024, 32))
f64 = fx.gpu(np.asfortranarray(np.random.default_rng(1).standard_normal((64, 1024, 32)).astype(np.float32)))
for name, x in (("C4  D=3 ", c4), ("C4' D=64", f64)):
    r = _per_call_ms(fx, lambda: fx.knn(x, 20, drop_first=True), n=200)
    s = fx.Stream.create()
    with fx.stream(s):
        e0, e1 = fx.Event(), fx.Event()
        e0.record(s)
        for _ in range(200):
            fx.knn(x, 20, drop_first=True)
        e1.record(s)
        s.synchronize()
    print(f"{name}: min {r['min_ms'] * 1e3:6.2f} us  median {r['median_ms'] * 1e3:6.2f} us  back-to-back {e0.elapsed_ms(e1) * 5:6.2f} us")
