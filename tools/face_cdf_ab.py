#!/usr/bin/env python3
"""Per-call device time of fx3d_sample_points_cdf (face_cdf_kernel) on the fit loop's source mesh (sphere, 5120 faces,
B = 1) and on C3's batch (8 teapots).  With FX3D_CDF_PROBE_READ=1 and a library built with `make EXTRA=-DFX3D_CDF_PROBE`:
block 0's phase stamps (areas | chunk totals | total | divisions | chunk prefixes | offsets + fix-up | output)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import flux3d_jl_amd as fx  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402
from flux3d_jl_amd.transforms import _verts_padded_dev, EPS  # noqa: E402
from bench_ops import gpu_time  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
cases = {"sphere B=1": fx.gpu(fx.load_trimesh(os.path.join(GOLD, "sphere.obj"))),
         "teapot B=8": fx.gpu(fx.load_trimesh(*[os.path.join(GOLD, "teapot.obj")] * 8))}
for name, m in cases.items():
    verts, faces = _verts_padded_dev(m), m.dev("faces_padded")
    nb = C.c_size_t(0)
    _lib.call("fx3d_sample_points_workspace_bytes", m.F, m.N, C.byref(nb))
    ws = fx.DeviceArray.empty((nb.value,), np.uint8)

    def run():
        _lib.call("fx3d_sample_points_cdf", verts.ptr, m.V, faces.ptr, m.F, m.dev("faces_len").ptr, m.N, float(EPS),
                  ws.ptr, ws.nbytes, fx.current_stream().handle)
    mn, md = gpu_time(run, reps=40, inner=16)
    print(f"{name}  F={m.F}  min {mn:.2f} us  median {md:.2f} us", flush=True)
    if os.environ.get("FX3D_CDF_PROBE_READ"):  # library built with -DFX3D_CDF_PROBE: wall-clock stamps (10 ns) of block 0
        up = lambda v: (v + 31) // 32 * 32  # noqa: E731  (csrc/sampler.hip: CdfWs -- the stamps sit 40 doubles into the misc slots)
        nchp = up(up(m.F) // 32)
        Fp = up(m.F) + nchp + up(nchp // 32) + 32 + 3 * up((nchp // 32 + 31) // 32) + 40
        for _ in range(3):
            run()
        fx.synchronize()
        st = ws.to_host().view(np.int64)[Fp:Fp + 10]
        print(f"{name} area pass: staged +{(st[8] - st[0]) / 100:.2f}, gathered +{(st[9] - st[8]) / 100:.2f}, areas stored +{(st[1] - st[9]) / 100:.2f}")
        st = st[:8]
        print(f"{name} phases (us):", " ".join(f"{(b - a) / 100:.2f}" for a, b in zip(st, st[1:])), flush=True)
