#!/usr/bin/env python3
"""Timing of every op on the hot path at the BASELINE.json configs (beyond the headline bench.py):
C1, C2 (+indices, +backward), C3 (sample_points -> chamfer on B=8 teapot-class meshes, 5000 samples),
C4 (kNN k=20, B=32x1024, D=3 and D=64, + graph gather), mesh losses, and the reference harness's own
degenerate input p_i=(i,i,i)/n (benchmarks/metrics.jl:11-15) for comparison with BASELINE.md.
HIP events on the op's stream, min/median over repeats; CPU = oracle on one core (bounded samples).

  python tools/bench_ops.py [--cpu]      -> one JSON object per line
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def gpu_time(fn, reps=30, warm=3, inner=8):
    """Per-call device time: `inner` calls are enqueued between two events so that the host's launch path
    (ctypes, allocation) overlaps the previous call's kernels instead of being timed as an idle device."""
    s = fx.Stream.create()
    with fx.stream(s):
        for _ in range(warm):
            fn()
        s.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = fx.Event(), fx.Event()
            e0.record()
            for _ in range(inner):
                fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_ms(e1) * 1e3 / inner)
    return float(np.min(ts)), float(np.median(ts))


def cpu_time(fn, reps=1):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best * 1e6


def emit(name, us_min, us_med, **kw):
    d = {"op": name, "gpu_us_min": round(us_min, 2), "gpu_us_median": round(us_med, 2)}
    d.update(kw)
    print(json.dumps(d), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle (1 core)")
    args = ap.parse_args()
    orc = None
    if args.cpu:
        from oracle import oracle as orc

    # ---- chamfer -------------------------------------------------------------------------------------
    for tag, B, N in (("C1 chamfer fwd B=2 N=M=1024", 2, 1024), ("C2 chamfer fwd B=32 N=M=4096", 32, 4096)):
        x = fx.synth.uniform_cloud(fx.synth.SEED_A, 3, N, B)
        y = fx.synth.uniform_cloud(fx.synth.SEED_B, 3, N, B)
        dx, dy = fx.gpu(x), fx.gpu(y)
        out = fx.DeviceArray.empty((1,), np.float32)
        mn, md = gpu_time(lambda: fx.chamfer_distance(dx, dy, loss_out=out, sync=False))
        kw = {"pairs_per_s": B * N * N / (mn * 1e-6)}
        if orc:
            kw["cpu_kdtree_us"] = cpu_time(lambda: orc.chamfer_distance(x, y, kdtree=True))
        emit(tag, mn, md, **kw)
        mn, md = gpu_time(lambda: fx.chamfer_distance(dx, dy, return_indices=True, loss_out=out, sync=False))
        emit(tag + " +indices", mn, md)
        _, ix, iy = fx.chamfer_distance(dx, dy, return_indices=True)
        mn, md = gpu_time(lambda: fx.chamfer_distance_grad(dx, dy, ix, iy))
        emit(tag.replace("fwd", "bwd"), mn, md, bytes=4 * 3 * B * 2 * N * 4)
    # reference harness input: B=1, A == B, p_i = (i,i,i)/n
    for n in (64, 256, 1024, 4096, 16384):
        p = fx.gpu(fx.synth.reference_bench_cloud(n))
        out = fx.DeviceArray.empty((1,), np.float32)
        mn, md = gpu_time(lambda: fx.chamfer_distance(p, p, loss_out=out, sync=False))
        emit(f"reference-harness chamfer fwd n={n}", mn, md, ref_cpu_ms_plot={64: .065, 256: .19, 1024: .8, 4096: 3.2, 16384: 13}[n],
             ref_gpu_ms_plot={64: .33, 256: .38, 1024: .75, 4096: 6, 16384: 85}[n])

    # large single clouds (candidate split across blocks keeps the chip busy when B is small)
    for (B, N) in ((1, 16384), (4, 8192), (1, 65536)):
        x = fx.gpu(fx.synth.uniform_cloud(11, 3, N, B))
        y = fx.gpu(fx.synth.uniform_cloud(12, 3, N, B))
        out = fx.DeviceArray.empty((1,), np.float32)
        mn, md = gpu_time(lambda: fx.chamfer_distance(x, y, loss_out=out, sync=False), reps=10)
        emit(f"chamfer fwd uniform B={B} N=M={N}", mn, md, pairs_per_s=B * N * N / (mn * 1e-6))

    # ---- C4 kNN ---------------------------------------------------------------------------------------
    x = fx.synth.uniform_cloud(0x5EED0004, 3, 1024, 32)
    dx = fx.gpu(x)
    mn, md = gpu_time(lambda: fx.knn(dx, 20, drop_first=True))
    kw = {"pairs_per_s": 32 * 1024 * 1024 / (mn * 1e-6)}
    if orc:
        kw["cpu_bruteforce_us"] = cpu_time(lambda: orc.knn(x[:, :, :4], 20, drop_first=True)) * 8
    emit("C4 kNN k=20 drop-first B=32 N=1024 D=3", mn, md, **kw)
    idx = fx.knn(dx, 20, drop_first=True, return_dist=False)
    mn, md = gpu_time(lambda: fx.knn_gather(dx, idx))
    emit("C4 knn_gather F=3", mn, md, hbm_write_GBps=(4 * 3 * 20 * 1024 * 32) / (mn * 1e-6) / 1e9,
         note="HBM side = the gathered array written once; the 0.4 MB source and the indices are read from L2")
    f = np.asfortranarray(np.random.default_rng(1).standard_normal((64, 1024, 32)).astype(np.float32))
    df = fx.gpu(f)
    mn, md = gpu_time(lambda: fx.knn(df, 20, drop_first=True), reps=10)
    emit("C4' kNN k=20 D=64 (second EdgeConv)", mn, md, pairs_per_s=32 * 1024 * 1024 / (mn * 1e-6))
    idx = fx.knn(df, 20, drop_first=True, return_dist=False)
    mn, md = gpu_time(lambda: fx.knn_gather(df, idx))
    emit("C4' knn_gather F=64", mn, md, hbm_write_GBps=(4 * 64 * 20 * 1024 * 32) / (mn * 1e-6) / 1e9,
         l2_plus_hbm_GBps=(4 * 64 * 20 * 1024 * 32 * 2) / (mn * 1e-6) / 1e9,
         note="168 MB written to HBM; the 168 MB of reads are k = 20 re-reads of an 8 MB source served by the L2 / Infinity "
              "Cache, so read + write bytes per second (l2_plus_hbm_GBps) may exceed the 8 TB/s HBM peak: it is not an HBM rate")

    # EdgeConv graph build up to the MLP input (kNN + cat(X, KNN - X) + permute), src/models/dgcnn.jl:32-51
    for tag, d in (("F=3", dx), ("F=64", df)):
        F = d.shape[0]
        ii = fx.knn(d, 20, drop_first=True, return_dist=False)
        mn, md = gpu_time(lambda: fx.edge_features(d, ii, layout="mlp"), reps=10)
        nbytes = 4 * (2 * F * 20 * 1024 * 32 + 20 * 1024 * 32 + F * 1024 * 32)
        emit(f"EdgeConv edge_features (K*N,2F,B) {tag}", mn, md, hbm_GBps=nbytes / (mn * 1e-6) / 1e9,
             note="algorithmic bytes: features written once + source and indices read once")
        mn, md = gpu_time(lambda: fx.edgeconv_graph(d, 20, layout="mlp"), reps=10)
        emit(f"EdgeConv graph build kNN+features {tag}", mn, md)

    # pointcloud_to_voxel (src/conversions.jl:91-131), res 32, B=32 clouds of 1024 / 4096 points
    for N in (1024, 4096):
        pc = fx.gpu(fx.synth.uniform_cloud(21, 3, N, 32))
        mn, md = gpu_time(lambda: fx.pointcloud_to_voxel(pc, 32))
        kw = {"hbm_GBps": (4 * 32 ** 3 * 32 + 12 * N * 32) / (mn * 1e-6) / 1e9,
              "equiv_nn_pairs_per_s": 32 ** 3 * N * 32 / (mn * 1e-6)}
        if orc and N == 1024:
            ph = fx.synth.uniform_cloud(21, 3, N, 32)
            kw["cpu_bruteforce_us"] = cpu_time(lambda: orc.pointcloud_to_voxel(ph[:, :, :1], 32)) * 32
        emit(f"pointcloud_to_voxel res=32 B=32 N={N}", mn, md, **kw)

    # ---- C3 meshes -------------------------------------------------------------------------------------
    t = os.path.join(GOLD, "teapot.obj")
    m8 = fx.gpu(fx.load_trimesh(*[t] * 8))
    def fresh(m):  # drop the mesh's cached sampling CDF: the C3 figure includes areas -> probabilities -> CDF
        for k in [k for k in m._dev if isinstance(k, tuple) and k[0] == "face_cdf"]:
            del m._dev[k]
        return m

    mn, md = gpu_time(lambda: fx.sample_points(fresh(m8), 5000, seed=3))
    kw = {"draw_only_us_min": gpu_time(lambda: fx.sample_points(m8, 5000, seed=3))[0]}  # CDF kept (unchanged mesh)
    if orc:
        vp, fp = m8.get_verts_padded_host(), m8.get_faces_padded().astype(np.int64) - 1
        kw["cpu_us"] = cpu_time(lambda: orc.sample_points_seeded(vp, fp, m8._faces_len, 5000, 3))
    emit("C3 sample_points B=8 teapot n=5000", mn, md, **kw)
    out = fx.DeviceArray.empty((1,), np.float32)
    m8b = fx.gpu(fx.load_trimesh(*[t] * 8))
    mn, md = gpu_time(lambda: fx.chamfer_distance(fresh(m8), fresh(m8b), 5000, seed=5, loss_out=out, sync=False))
    emit("C3 chamfer_distance(mesh, mesh, 5000) B=8 (2 samplings + chamfer)", mn, md, pairs_per_s=8 * 5000 * 5000 / (mn * 1e-6))
    mn, md = gpu_time(lambda: fx.laplacian_loss(m8, sync=False))
    emit("laplacian_loss B=8 teapot", mn, md)
    mn, md = gpu_time(lambda: fx.edge_loss(m8, sync=False))
    emit("edge_loss B=8 teapot", mn, md)
    mn, md = gpu_time(lambda: fx.laplacian_loss_grad(m8))
    emit("laplacian_loss bwd B=8 teapot", mn, md)
    mn, md = gpu_time(lambda: fx.edge_loss_grad(m8))
    emit("edge_loss bwd B=8 teapot", mn, md)
    mn, md = gpu_time(lambda: fx.compute_faces_areas_packed(m8))
    emit("faces_areas_packed B=8 teapot", mn, md)


if __name__ == "__main__":
    main()
