"""Chamfer adjoint (fx3d_chamfer_bwd) at the shapes of BASELINE's configs: back-to-back calls between two events.
bytes = the algorithmic traffic of tools/hbm_roofline.py (rows of both clouds read twice + indices + both gradients)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux3d_jl_amd as fx  # noqa: E402

fx.set_device(0)
for (Bc, Np) in ((256, 4096), (32, 4096), (8, 5000), (32, 1024), (2, 1024), (1, 16384)):
    a = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, 3, Np, Bc))
    b = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, 3, Np, Bc))
    _, ia, ib = fx.chamfer_distance(a, b, return_indices=True)
    for _ in range(5):
        fx.chamfer_distance_grad(a, b, ia, ib)
    fx.synchronize()
    import ctypes as C
    from flux3d_jl_amd import _lib
    _lib.call("fx3d_profile_enable", 1)
    for _ in range(20):
        fx.chamfer_distance_grad(a, b, ia, ib)
    fx.synchronize()
    avg, mn, mx, cnt = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int64(0)
    _lib.call("fx3d_profile_kernel_stats", b"chamfer_bwd", C.byref(avg), C.byref(mn), C.byref(mx), C.byref(cnt))
    _lib.call("fx3d_profile_enable", 0)
    e0, e1 = fx.Event(), fx.Event()
    best = 1e9
    for rep in range(5):
        e0.record()
        for _ in range(20):
            fx.chamfer_distance_grad(a, b, ia, ib)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_ms(e1) / 20)
    nbytes = 2 * (12 * Np * Bc * 2 + 4 * Np * Bc) + 2 * 12 * Np * Bc
    print(json.dumps({"B": Bc, "N": Np, "us_per_call": round(best * 1e3, 2), "kernel_avg_us": round(avg.value * 1e3, 2), "kernel_min_us": round(mn.value * 1e3, 2),
                      "TB_per_s_kernel": round(nbytes / avg.value / 1e9, 3)}), flush=True)
