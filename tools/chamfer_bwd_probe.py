"""Phase stamps of chamfer_bwd_gather_kernel (a -DFX3D_BG_PROBE build: tools/build_variant.sh probe chamfer_bwd "-DFX3D_BG_PROBE";
FX3D_HIP_LIB=.../libflux3d_hip_probe.so): per block the wall-clock counter at the phase boundaries; prints, relative to the
launch's first stamp, the median / max end of every phase over the blocks, in microseconds."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux3d_jl_amd as fx  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402

fx.set_device(0)
lib = C.CDLL(_lib.LIB_PATH)
for (Bc, Np) in ((256, 4096), (32, 4096), (2, 1024)):
    a = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, 3, Np, Bc))
    b = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, 3, Np, Bc))
    _, ia, ib = fx.chamfer_distance(a, b, return_indices=True)
    for _ in range(3):
        fx.chamfer_distance_grad(a, b, ia, ib)
    fx.synchronize()
    buf = np.zeros((1024, 8), np.uint64)
    assert lib.fx3d_debug_bg_probe(buf.ctypes.data_as(C.c_void_p)) == 0
    nblk = int((buf[:, 0] > 0).sum())
    t = buf[:nblk].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0  # 100 MHz
    print(f"B={Bc} N={Np}: {nblk} blocks; start median {np.median(us[:,0]):.2f} max {us[:,0].max():.2f}")
    for ph, name in enumerate(["start", "loads+image", "count", "scan", "place", "long rows", "owners+stores issued"]):
        if ph == 0:
            continue
        d = us[:, ph] - us[:, ph - 1]
        print(f"   phase {ph} {name:22s}: duration median {np.median(d):6.2f} max {d.max():6.2f}   end median {np.median(us[:,ph]):6.2f} max {us[:,ph].max():6.2f}")
