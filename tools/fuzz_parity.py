#!/usr/bin/env python3
"""Randomised parity sweep beyond the test-suite: nearest neighbours, chamfer sums, kNN (D = 3 and feature space) and
EdgeConv features and the mesh ops (areas, sampler, losses) on random shapes and data distributions (uniform, clustered, lattice with exact ties, duplicated
points, large offsets, wide dynamic range), every index / distance / feature compared bit for bit with the CPU oracle (the chamfer loss to 1e-5 relative).

  python tools/fuzz_parity.py [--seconds 300] [--seed 1]        exit code 1 on the first mismatch (case is printed)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import flux3d_jl_amd as fx  # noqa: E402
import oracle as orc  # noqa: E402


def cloud(rng, D, N, B, kind):
    if kind == "uniform":
        x = rng.random((D, N, B))
    elif kind == "normal":
        x = rng.standard_normal((D, N, B))
    elif kind == "clustered":
        c = rng.standard_normal((D, 1 + N // 50, B)) * 3
        x = c[:, rng.integers(0, c.shape[1], N), :] + rng.standard_normal((D, N, B)) * 0.05
    elif kind == "tight":
        # tight clusters around centres every cloud of the run shares (a fixed stream): dense bands, the retry pass's RUN items
        c = np.random.default_rng(7).standard_normal((D, 40, 1)) * 3
        x = c[:, rng.integers(0, 40, N), :] + rng.standard_normal((D, N, B)) * rng.choice([1e-1, 1e-2, 1e-3])
    elif kind == "lattice":
        x = rng.integers(0, 6, (D, N, B)).astype(np.float64) * 0.25
    elif kind == "dupes":
        x = rng.random((D, N, B))
        x[:, N // 2:, :] = x[:, : N - N // 2, :]
    elif kind == "offset":
        x = rng.standard_normal((D, N, B)) + rng.choice([5.0, 50.0, 1000.0])
    elif kind == "worstsplit":
        # every coordinate a worst case of the 2-way fp16 split (24-bit mantissa = 11-bit head * 2^13 + an odd 12-bit residual:
        # the low piece rounds at an exact tie), point-symmetric about 0 so that the centring keeps the bit patterns
        h = (N + 1) // 2
        head = rng.integers(1 << 10, 1 << 11, (D, h, B)).astype(np.int64)
        r = (rng.integers(1 << 10, 1 << 11, (D, h, B)).astype(np.int64) * 2 + 1) * rng.choice([-1, 1], (D, h, B))
        e = rng.integers(0, int(rng.integers(1, 6)), (D, h, B))
        half = rng.choice([-1.0, 1.0], (D, h, B)) * (head * (1 << 13) + r) * np.exp2(e - 23.0)
        x = np.concatenate([half, -half], 1)[:, rng.permutation(2 * h)[:N], :]
    else:  # "range": a few far outliers
        x = rng.standard_normal((D, N, B)) * 1e-2
        x[:, rng.integers(0, N, 3), :] *= 1e5
    return np.asfortranarray(x.astype(np.float32))


KINDS = ["uniform", "normal", "clustered", "tight", "lattice", "dupes", "offset", "range", "worstsplit"]


def mesh_batch(rng):
    """Ragged batch of random triangle soups over shared vertices (1-based faces like the reference), with thin,
    zero-area and repeated faces."""
    vl, fl = [], []
    for _ in range(int(rng.integers(1, 5))):
        V, F = int(rng.integers(4, 500)), int(rng.integers(2, 1200))
        v = rng.standard_normal((3, V)).astype(np.float32) * np.float32(rng.choice([1e-3, 1.0, 50.0]))
        f = np.stack([rng.choice(V, 3, replace=False) for _ in range(F)], axis=1).astype(np.uint32) + 1
        if F > 4:
            v[:, f[1, 0] - 1] = v[:, f[0, 0] - 1]
            f[:, 3] = f[:, 2]
        vl.append(np.asfortranarray(v))
        fl.append(np.asfortranarray(f))
    return vl, fl


def mesh_case(rng):
    """Areas, sampler (CDF + draws + points, bit exact), both mesh losses (1e-5 relative)."""
    vl, fl = mesh_batch(rng)
    m = fx.gpu(fx.TriMesh(vl, fl))
    v = m.get_verts_packed_host()
    f0 = m.get_faces_packed().astype(np.int64) - 1
    ok = np.array_equal(fx.compute_faces_areas_packed(m).to_host().ravel(), orc.faces_areas_packed(v, f0))
    e0 = m.get_edges_packed().astype(np.int64) - 1
    rowptr, colind, vals = m.get_laplacian_packed()
    ok = ok and bool(np.isclose(fx.laplacian_loss(m), orc.laplacian_loss(v, rowptr.astype(np.int64), colind.astype(np.int64), vals),
                                rtol=1e-5, atol=1e-30))
    ok = ok and bool(np.isclose(fx.edge_loss(m, 0.1), orc.edge_loss(v, e0, 0.1), rtol=1e-5, atol=1e-30))
    n, seed = int(rng.integers(1, 900)), int(rng.integers(0, 1 << 62))
    out, fi, r1, r2 = fx.sample_points(m, n, seed=seed, return_draws=True)
    eo, efi, er1, er2 = orc.sample_points_seeded(m.get_verts_padded_host(), m.get_faces_padded().astype(np.int64) - 1,
                                                 m._faces_len, n, seed, return_draws=True)
    ok = ok and np.array_equal(fi.to_host(), efi) and np.array_equal(r1.to_host(), er1) and np.array_equal(r2.to_host(), er2)
    ok = ok and np.array_equal(out.to_host(), eo)
    # (round 6) the ordered sampling adjoint: bit for bit the oracle's fixed-order sums, alone and behind the chamfer adjoint's rows
    gs = np.asfortranarray(rng.standard_normal((3, n, m.N)).astype(np.float32))
    fp0 = m.get_faces_padded().astype(np.int64) - 1
    base = np.asfortranarray(rng.standard_normal((3, m.V, m.N)).astype(np.float32)) if rng.random() < 0.5 else None
    dev_base = fx.gpu(base.copy(order="F")) if base is not None else None
    g = fx.sample_points_grad(m, fi, r1, r2, gs, out=dev_base)
    ok = ok and np.array_equal(g.to_host(), orc.sample_points_bwd(fp0, m._faces_len, m.V, efi, er1, er2, gs, base=base))
    if n > 1:
        y = fx.gpu(np.asfortranarray(rng.standard_normal((3, n, m.N)).astype(np.float32)))
        _, ix, iy = fx.chamfer_distance(out, y, return_indices=True)
        ga, _ = fx.chamfer_sampled_grad(out, y, ix, iy, mesh_a=m, draws_a=(fi, r1, r2), w1=0.8, w2=1.2)
        oga, _ = orc.chamfer_bwd(eo, y.to_host(), ix.to_host(), iy.to_host(), 0.8, 1.2, 1.0)
        ok = ok and np.array_equal(ga.to_host(), orc.sample_points_bwd(fp0, m._faces_len, m.V, efi, er1, er2, oga))
    return ok, f"mesh batch of {m.N} (V={m.V} F={m.F}) n={n} seed={seed}"


def fit_case(rng):
    """(round 6) The fit iteration's regularisers as passengers of the sampling launches (fx3d_mesh_reg) against the same calls one
    after the other: samples, both losses, the sum, the gradient and the optimiser's state, bit for bit; the separate calls' gradient
    against the oracle's adjoints."""
    from flux3d_jl_amd.metrics import MeshReg, _chamfer_points
    from flux3d_jl_amd.transforms import sample_points_pair
    nb = int(rng.integers(1, 4))
    V, F = int(rng.integers(4, 400)), int(rng.integers(2, 900))
    vl = [np.asfortranarray(rng.standard_normal((3, V)).astype(np.float32) * np.float32(rng.choice([1e-2, 1.0, 20.0]))) for _ in range(nb)]
    fl = [np.asfortranarray(np.stack([rng.choice(V, 3, replace=False) for _ in range(F)], axis=1).astype(np.uint32) + 1) for _ in range(nb)]
    tvl, tfl = mesh_batch(rng)
    tvl, tfl = [tvl[0]] * nb, [tfl[0]] * nb
    n, s1, s2 = int(rng.integers(2, 900)), int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 62))
    w_lap, w_edge = float(rng.choice([0.1, 0.0, 1.5])), float(rng.choice([1.0, 0.0, 0.3]))
    vel0 = np.asfortranarray(rng.standard_normal((3, V * nb)).astype(np.float32) * 1e-3)
    x0 = np.asfortranarray(rng.standard_normal((3, V * nb)).astype(np.float32) * 1e-2)
    res = []
    for passengers in (False, True):
        src, tgt = fx.gpu(fx.TriMesh(vl, fl)), fx.gpu(fx.TriMesh(tvl, tfl))
        if not fx.sampling_adjoint_is_ordered(src, n):
            return True, "fit case beyond the ordered form (skipped)"
        vel, x, out, ctr = fx.gpu(vel0.copy(order="F")), fx.gpu(x0.copy(order="F")), fx.DeviceArray.zeros((3, V * nb), np.float32), fx.DeviceArray.zeros((1,), np.uint64)
        reg = MeshReg(src, 0.0, w_lap, w_edge) if passengers else None
        A, Bp, fa, r1, r2 = sample_points_pair(src, tgt, n, seed_a=s1, seed_b=s2, return_draws_a=True, reg=reg)
        loss1, ix, iy = _chamfer_points(A, Bp, 1.0, 1.0, return_indices=True, sync=False)
        step = (0.9, 0.7, vel, x, src.dev("verts_packed"), out, ctr, 2)
        if passengers:
            reg.set_base(loss1)
            g = fx.DeviceArray.empty((3, V * nb), np.float32)
            fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=src, draws_a=(fa, r1, r2), out_a=g.reshape(3, V, nb), step=step, reg=reg)
            lap, edge, total = reg.lap, reg.edge, reg.total
        else:
            lap, edge, total = fx.mesh_losses(src, 0.0, w_lap, w_edge, base=loss1, sync=False)
            g = fx.mesh_losses_grad(src, 0.0, w_lap, w_edge, reuse_forward=True)
            fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=src, draws_a=(fa, r1, r2), out_a=g.reshape(3, V, nb), step=step)
        res.append([a.to_host().copy() for a in (A, Bp, lap, edge, total, g, vel, x, out, ctr)])
    ok = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(*res))
    return ok, f"fit case nb={nb} V={V} F={F} n={n} seeds={s1},{s2} w=({w_lap},{w_edge})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0, ncase = time.time(), 0
    while time.time() - t0 < args.seconds:
        kind = KINDS[int(rng.integers(0, len(KINDS)))]
        what = int(rng.integers(0, 6))
        B = int(rng.integers(1, 4))
        if what == 4:
            ok, desc = mesh_case(rng)
        elif what == 5:
            ok, desc = fit_case(rng)
        elif what == 0:  # 1-NN both directions + chamfer
            N, M = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
            if rng.random() < 0.35:  # (round 3) sizes around the launch plan's boundaries: one LDS image + an exact tail of <= 64
                def edge():          # candidates, query remainders of <= 64 folded into the last tile, several chunks
                    base = int(rng.choice([512, 1024, 2048, 4096, 4096, 4096, 6144, 8192]))
                    return max(1, base + int(rng.integers(-3, 70)))
                N, M = edge(), (edge() if rng.random() < 0.6 else int(rng.integers(1, 3000)))
                B = int(rng.integers(1, 3))
            x, y = cloud(rng, 3, N, B, kind), cloud(rng, 3, M, B, kind)
            desc = f"nn1 {kind} N={N} M={M} B={B}"
            if rng.random() < 0.3:   # (round 3) clouds of different extent / position: far queries carry a scale of their own
                ratio = float(np.exp(rng.uniform(0.0, np.log(1e7))))
                shift = float(rng.choice([0.0, 0.0, 1.0, 30.0])) * ratio
                y = np.asfortranarray((y * np.float32(ratio) + np.float32(shift)).astype(np.float32))
                desc += f" y*{ratio:.3g}+{shift:.3g}"
            ix, iy = fx.nearest_neighbors(fx.gpu(x), fx.gpu(y))
            ox, oy = orc.nn1(x, y)
            ok = np.array_equal(ix.to_host(), ox) and np.array_equal(iy.to_host(), oy)
            if ok:
                got = fx.chamfer_distance(fx.gpu(x), fx.gpu(y), w1=0.7, w2=1.3)
                exp = orc.chamfer_distance(x, y, 0.7, 1.3)
                ok = bool(np.isclose(got, exp, rtol=1e-5, atol=0))  # the loss: 1e-5 relative (Float64 partial-sum order)
                if not ok:
                    desc += f" [indices equal; loss {got!r} vs oracle {exp!r}]"
            else:
                desc += " [indices differ]"
            if ok:  # (round 5) the adjoint is atomic-free: bit for bit, on the forward's indices or on arbitrary index maps
                if rng.random() < 0.5:
                    jx, jy = ox, oy
                else:  # many-to-one maps: long inverse lists (the wave path), lists of 5 .. 8 (the networks)
                    tx = int(rng.integers(1, max(2, M // int(rng.choice([1, 1, 3, 40])))))
                    ty = int(rng.integers(1, max(2, N // int(rng.choice([1, 1, 3, 40])))))
                    jx = np.asfortranarray(rng.integers(0, tx, (N, B)).astype(np.int32))
                    jy = np.asfortranarray(rng.integers(0, ty, (M, B)).astype(np.int32))
                    desc += f" bwd on random maps (<{tx}, <{ty})"
                gout = float(rng.choice([1.0, -0.5, 2.0]))
                gx, gy = fx.chamfer_distance_grad(x, y, fx.gpu(jx), fx.gpu(jy), w1=0.7, w2=1.3, gout=gout)
                egx, egy = orc.chamfer_bwd(x, y, jx, jy, 0.7, 1.3, gout)
                ok = np.array_equal(gx.to_host().view(np.uint32), egx.view(np.uint32)) and np.array_equal(gy.to_host().view(np.uint32), egy.view(np.uint32))
                if not ok:
                    desc += " [adjoint differs]"
        elif what == 1:  # kNN D = 3
            N, M = int(rng.integers(1, 1500)), int(rng.integers(2, 6000))
            k = int(rng.integers(1, min(33, M)))
            if rng.random() < 0.4:  # (round 3) 32 < k + drop <= 64: the wide geometry of the matrix-core kernel (M >= 128), below that the wave kernels
                k = int(rng.integers(1, min(65, M)))
            drop = bool(rng.integers(0, 2)) and k + 1 <= min(64, M)
            x, y = cloud(rng, 3, N, B, kind), cloud(rng, 3, M, B, kind)
            desc = f"knn3 {kind} N={N} M={M} B={B} k={k} drop={drop}"
            gi, gd = fx.knn(fx.gpu(x), k, y=fx.gpu(y), drop_first=drop)
            oi, od = orc.knn(x, k, y=y, drop_first=drop)
            ok = np.array_equal(gi.to_host(), oi) and np.array_equal(gd.to_host(), od)
        elif what == 2:  # kNN feature space
            D = int(rng.choice([4, 8, 16, 32, 64, 64, 96, 128, 20, 7]))
            N, M = int(rng.integers(1, 400)), int(rng.integers(64, 1500))
            k = int(rng.integers(1, 32))
            if rng.random() < 0.3:  # (round 3) 32 < k <= 64: candidate slices + verified merge (even M), the wave kernels otherwise
                k = int(rng.integers(32, min(65, M)))
                M += int(rng.integers(0, 2)) * (M & 1)
            x, y = cloud(rng, D, N, B, kind), cloud(rng, D, M, B, kind)
            desc = f"knnF {kind} D={D} N={N} M={M} B={B} k={k}"
            gi, gd = fx.knn(fx.gpu(x), k, y=fx.gpu(y))
            oi, od = orc.knn(x, k, y=y)
            ok = np.array_equal(gi.to_host(), oi) and np.array_equal(gd.to_host(), od)
        else:  # EdgeConv graph build, first layer (fused kernel) and a feature layer
            F = int(rng.choice([3, 3, 16, 64]))
            N = int(rng.integers(70, 700))
            K = int(rng.integers(1, 31))
            if rng.random() < 0.3:  # (round 3) k + 1 in 33 ... 64: the wide / compact D = 3 geometries, the verified slices in feature space
                K = int(rng.integers(31, 64))
            x = cloud(rng, F, N, B, kind)
            desc = f"edgeconv {kind} F={F} N={N} B={B} K={K}"
            lay = int(rng.integers(0, 2))
            out, idx = fx.edgeconv_graph(fx.gpu(x), K, layout=("cat", "mlp")[lay], return_idx=True)
            oi = orc.knn(x, K, drop_first=True, want_dist=False)
            ok = np.array_equal(idx.to_host(), oi) and np.array_equal(out.to_host(), orc.edge_features(x, oi, layout=lay))
        ncase += 1
        if not ok:
            print("MISMATCH:", desc, "seed", args.seed, "case", ncase, flush=True)
            return 1
    print(f"{ncase} random cases in {time.time() - t0:.0f} s: all bit-identical to the oracle", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
