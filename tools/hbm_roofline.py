#!/usr/bin/env python3
"""The HBM-bound members of the path (edge_loss, laplacian_loss, faces_areas, knn_gather, edge_features,
chamfer backward) at a BANDWIDTH-BOUND size -- at the BASELINE sizes they are 3-9 us launch floors.  Per kernel: its
algorithmic bytes (DESIGN.md 3.3 byte counts), its duration measured by the library's own HIP events around the launch
(fx3d_profile_enable: every launch, on the launch's stream) and the resulting GB/s against 8 TB/s nominal / 6.3 TB/s
achievable (tools/ubench_hbm.hip).  One JSON object per kernel; `--brief` = one dict (bench.py embeds it).

  python tools/hbm_roofline.py [--cells 1400] [--reps 20]
rocprofv3 --kernel-trace / --pmc FETCH_SIZE / WRITE_SIZE passes of this script: tools/profile_round.sh -> profiles/r04_*_hbm_*.txt
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402

HBM_PEAK, HBM_ACHIEVABLE = 8000.0, 6300.0


def sheet(nx, ny, seed=0):
    """A jittered (nx x ny)-cell sheet: (nx+1)(ny+1) vertices, 2 nx ny triangles (the shape of tools/mesh_shapes_time.py)."""
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.arange(nx + 1, dtype=np.float64), np.arange(ny + 1, dtype=np.float64), indexing="ij")
    v = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 0) + rng.uniform(-0.3, 0.3, (3, gx.size))
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    a = (i * (ny + 1) + j).ravel()
    f = np.concatenate([np.stack([a, a + ny + 1, a + ny + 2]), np.stack([a, a + ny + 2, a + 1])], 1)
    return np.asfortranarray(v.astype(np.float32)), np.asfortranarray(f.astype(np.int64) + 1)


def kernel_ms(name, fn, reps):
    for _ in range(3):
        fn()
    fx.synchronize()
    _lib.call("fx3d_profile_enable", 1)
    for _ in range(reps):
        fn()
    fx.synchronize()
    avg, mn, mx, cnt = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int64(0)
    _lib.call("fx3d_profile_kernel_stats", name.encode(), C.byref(avg), C.byref(mn), C.byref(mx), C.byref(cnt))
    _lib.call("fx3d_profile_enable", 0)
    return avg.value, mn.value, cnt.value


def measure(cells=1400, reps=20):
    out = []

    def row(kernel, nbytes, avg, mn, note):
        gbs = nbytes / (avg * 1e-3) / 1e9
        out.append({"kernel": kernel, "algorithmic_bytes": nbytes, "kernel_avg_ms": avg, "kernel_min_ms": mn, "achieved_gbs": gbs,
                    "frac_hbm_peak": gbs / HBM_PEAK, "frac_achievable": gbs / HBM_ACHIEVABLE, "size": note})

    v, f = sheet(cells, cells)
    m = fx.gpu(fx.TriMesh([v], [f]))
    V, F = m.V, m.F
    E = m.dev("edges").shape[0]
    nnz = m.dev("lap_colind").shape[0]
    note = f"one sheet: V={V} F={F} E={E} nnz={nnz}"
    avg, mn, _ = kernel_ms("faces_areas", lambda: fx.compute_faces_areas_packed(m), reps)
    row("faces_areas_packed_kernel", 12 * V + 12 * F + 4 * F, avg, mn, note)
    avg, mn, _ = kernel_ms("edge_loss", lambda: fx.edge_loss(m, sync=False), reps)
    row("edge_loss_kernel", 12 * V + 8 * E, avg, mn, note)
    avg, mn, _ = kernel_ms("laplacian_loss", lambda: fx.laplacian_loss(m, sync=False), reps)
    row("laplacian_loss_kernel", 12 * V + 8 * nnz + 4 * (V + 1), avg, mn, note)
    avg, mn, _ = kernel_ms("mesh_losses", lambda: fx.mesh_losses(m, sync=False), reps)
    row("mesh_losses_kernel (both losses, one launch; + the (4,V) unit rows written)", 2 * 12 * V + 8 * nnz + 4 * (V + 1) + 8 * E + 16 * V, avg, mn, note)
    g = fx.DeviceArray.empty((3, V), np.float32)
    avg, mn, _ = kernel_ms("edge_loss_bwd", lambda: fx.edge_loss_grad(m, out=None), reps)
    row("edge_loss adjoint, gather form (mesh_losses_bwd_gather_kernel<false,true>)", 12 * V + 4 * (V + 1) + 4 * nnz + 12 * V, avg, mn, note)
    # (large meshes: the unit rows once -- lap_unit_rows_kernel, not timed here -- + the gather over them)
    avg, mn, _ = kernel_ms("mesh_losses_bwd", lambda: fx.laplacian_loss_grad(m), reps)
    row("laplacian_loss adjoint, gather over the stored unit rows (mesh_losses_bwd_gather_kernel<true,false>)",
        4 * (V + 1) + 4 * nnz + 16 * V + 12 * V, avg, mn, note)
    fx.mesh_losses(m, sync=False)  # (the fused adjoint reuses the forward's unit rows)
    avg, mn, _ = kernel_ms("mesh_losses_bwd", lambda: fx.mesh_losses_grad(m, reuse_forward=True), reps)
    row("both adjoints in one launch (mesh_losses_bwd_gather_kernel<true,true>)", 4 * (V + 1) + 4 * nnz + 16 * V + 12 * V + 12 * V, avg, mn, note)
    del g, m

    # EdgeConv's HBM-bound siblings at C4' (F = 64, k = 20, B = 32 x 1024): the gather and the feature build
    rng = np.random.default_rng(1)
    x = fx.gpu(np.asfortranarray(rng.standard_normal((64, 1024, 32)).astype(np.float32)))
    idx = fx.knn(x, 20, drop_first=True, return_dist=False)
    Fd, N, B, k = 64, 1024, 32, 20
    avg, mn, _ = kernel_ms("knn_gather", lambda: fx.knn_gather(x, idx), reps)
    row("knn_gather4_kernel (F=64 k=20 B=32x1024)", 4 * Fd * N * B + 4 * k * N * B + 4 * Fd * k * N * B, avg, mn, "C4' graph")
    avg, mn, _ = kernel_ms("edge_features", lambda: fx.edge_features(x, idx, layout="mlp"), reps)
    row("edge_features_mlp4_kernel (F=64 k=20 B=32x1024)", 4 * Fd * N * B + 4 * k * N * B + 8 * Fd * k * N * B, avg, mn, "C4' graph")
    del x, idx

    # chamfer backward at a size where it streams: B = 256 clouds of 4096 points (config 5's global batch on one device)
    Bc, Np = 256, 4096
    a = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, 3, Np, Bc))
    b = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, 3, Np, Bc))
    _, ia, ib = fx.chamfer_distance(a, b, return_indices=True)
    avg, mn, _ = kernel_ms("chamfer_bwd", lambda: fx.chamfer_distance_grad(a, b, ia, ib), reps)
    row("chamfer_bwd_gather_kernel (B=256 N=M=4096)", 2 * (12 * Np * Bc * 2 + 4 * Np * Bc) + 2 * 12 * Np * Bc, avg, mn, "B=256 N=M=4096")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=1400, help="the sheet has cells^2 cells: 2 cells^2 faces (1400 -> 3.9 M faces)")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--brief", action="store_true")
    ap.add_argument("--table", action="store_true", help="a readable table instead of JSON lines")
    a = ap.parse_args()
    rows = measure(a.cells, a.reps)
    if a.brief:
        print(json.dumps({r["kernel"]: {"achieved_gbs": round(r["achieved_gbs"], 1), "kernel_avg_ms": r["kernel_avg_ms"]} for r in rows}))
    elif a.table:
        for r in rows:
            print(f"{r['kernel'][:78]:78s} {r['algorithmic_bytes'] / 1e6:8.1f} MB {r['kernel_avg_ms'] * 1e3:8.1f} us {r['achieved_gbs']:7.0f} GB/s "
                  f"= {r['frac_achievable']:.2f} of 6.3 TB/s")
    else:
        for r in rows:
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
