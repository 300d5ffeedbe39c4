"""GPU parity, hardening set (round 2): the BASELINE configs at FULL size against the oracle, non-finite and
denormal coordinates, directed near-ties at the edge of the filter band, and the kNN shapes beyond the tuned kernels.

Order of distances everywhere: Julia's isless on the Float32 squared distance (NaN after +Inf), then the lower index
(oracle/flux3d_oracle.c: fless).  Every index a kernel returns must be a valid index -- the adjoint and the gathers
dereference it."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-5


def _f(a):
    return np.asfortranarray(a.astype(np.float32))


def _nn_equal(fx, oracle, x, y):
    ix, iy, dx, dy = fx.nearest_neighbors(x, y, return_dist=True)
    ox, oy, odx, ody = oracle.nn1(x, y, want_dist=True)
    gix, giy = ix.to_host(), iy.to_host()
    assert gix.min() >= 0 and gix.max() < y.shape[1] and giy.min() >= 0 and giy.max() < x.shape[1]
    assert np.array_equal(gix, ox), np.argwhere(gix != ox)[:5]
    assert np.array_equal(giy, oy), np.argwhere(giy != oy)[:5]
    assert np.array_equal(dx.to_host(), odx, equal_nan=True) and np.array_equal(dy.to_host(), ody, equal_nan=True)
    return ix, iy


def _chamfer_equal(fx, oracle, x, y):
    loss, ix, iy = fx.chamfer_distance(x, y, return_indices=True)
    oloss, ox, oy, _ = oracle.chamfer_distance(x, y, return_all=True)
    assert np.array_equal(ix.to_host(), ox) and np.array_equal(iy.to_host(), oy)
    if np.isnan(oloss) or np.isinf(oloss):
        assert (np.isnan(loss) and np.isnan(oloss)) or loss == oloss, (loss, oloss)
    else:
        assert np.isclose(loss, oloss, rtol=LOSS_RTOL, atol=0), (loss, oloss)
    loss2 = fx.chamfer_distance(x, y)
    assert (np.isnan(loss) and np.isnan(loss2)) or loss2 == loss
    # the adjoint dereferences the indices: it must run (and give finite values wherever the inputs are finite)
    gx, gy = fx.chamfer_distance_grad(x, y, ix, iy)
    gx.to_host(); gy.to_host()


# ------------------------------------------------------------------------------ non-finite / denormal coordinates
NONFINITE = ["nan_query", "nan_candidate", "nan_both", "inf_query", "neg_inf_candidate", "inf_same_coordinate",
             "all_nan_cloud", "negative_nan_payloads", "overflowing_distances", "denormals", "mixed_everything"]


def _nonfinite_case(case, N, M, B, seed, D=3):
    rng = np.random.default_rng(seed)
    x = rng.random((D, N, B), dtype=np.float32)
    y = rng.random((D, M, B), dtype=np.float32)
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    if case == "nan_query":
        x[0, 3, 0] = nan; x[D - 1, N - 1, B - 1] = nan
    elif case == "nan_candidate":
        y[1 % D, 0, 0] = nan; y[0, M // 2, B - 1] = nan; y[D - 1, M - 1, B - 1] = nan
    elif case == "nan_both":
        x[0, 0, 0] = nan; y[0, 0, 0] = nan; y[0, 1, 0] = nan
    elif case == "inf_query":
        x[0, 5, 0] = inf; x[D - 1, 6, B - 1] = -inf
    elif case == "neg_inf_candidate":
        y[0, 7, 0] = -inf; y[D - 1, M - 2, B - 1] = inf
    elif case == "inf_same_coordinate":  # inf - inf = NaN distance between the two, +Inf to everybody else
        x[0, 2, 0] = inf; y[0, 9, 0] = inf; y[0, 4, 0] = inf
    elif case == "all_nan_cloud":
        y[:, :, 0] = nan
    elif case == "negative_nan_payloads":  # x86 produces NaNs with the sign bit set; payloads must not order them
        bits = np.array([0xffc00000, 0x7fc00001, 0xffffffff, 0x7fa00000], np.uint32).view(np.float32)
        y[0, 10, 0], y[0, 3, 0], y[0, 11, 0], y[0, 1, 0] = bits
        x[0, 1, 0] = bits[0]
    elif case == "overflowing_distances":  # finite coordinates, +Inf distances: ties at +Inf go to the lowest index
        x *= np.float32(1e30); y *= np.float32(-2e30)
        x[:, : N // 2, :] *= np.float32(1e-25)
        y[:, ::3, :] *= np.float32(1e-27)
    elif case == "denormals":
        x *= np.float32(1e-40); y *= np.float32(1e-40)
        y[:, ::5, :] = 0.0
    elif case == "mixed_everything":
        x[0, 0, 0] = nan; x[0, 1, 0] = inf; x[:, 2, 0] = np.float32(3e38); y[0, 0, 0] = -inf; y[0, 5, 0] = nan
        y[:, 6, 0] = np.float32(-3e38); y[:, 7, 0] = np.float32(1e-42)
    return np.asfortranarray(x), np.asfortranarray(y)


@pytest.mark.parametrize("case", NONFINITE)
@pytest.mark.parametrize("N,M,B", [(700, 4200, 2), (64, 100, 1)])
def test_nn1_and_chamfer_nonfinite(gpu_fx, oracle, case, N, M, B):
    """ADVICE r1 (chamfer.hip:1267): a NaN query / all-NaN cloud used to publish index 0x7fffffff (and +Inf distances the
    same), which the adjoint and the gathers then dereferenced.  Indices, distances and the loss follow the oracle's
    isless order; the loss is NaN exactly when the oracle's is."""
    with np.errstate(all="ignore"):
        x, y = _nonfinite_case(case, N, M, B, N + len(case))
        _nn_equal(gpu_fx, oracle, x, y)
        _chamfer_equal(gpu_fx, oracle, x, y)


@pytest.mark.parametrize("case", NONFINITE)
def test_nn1_nonfinite_split_plan_and_exact_loop(gpu_fx, oracle, case):
    """The candidate-split plan (per-subset rows of 64-bit keys, merged by the unpack kernel) and the D = 2 exact loop see the same keys."""
    with np.errstate(all="ignore"):
        x, y = _nonfinite_case(case, 5000, 9000, 1, 77)
        _chamfer_equal(gpu_fx, oracle, x, y)
        x2, y2 = _nonfinite_case(case, 300, 500, 2, 78, D=2)
        _nn_equal(gpu_fx, oracle, x2, y2)
        x5, y5 = _nonfinite_case(case, 200, 300, 2, 79, D=5)
        _nn_equal(gpu_fx, oracle, x5, y5)


@pytest.mark.parametrize("case", NONFINITE)
@pytest.mark.parametrize("D,N,k,drop", [(3, 1024, 20, True), (3, 300, 40, False), (64, 512, 20, True), (5, 200, 10, True),
                                        (3, 300, 100, True), (16, 256, 70, False)])
def test_knn_nonfinite(gpu_fx, oracle, case, D, N, k, drop):
    """kNN on the matrix-core kernels (D = 3, D = 64), the wave kernels (k + drop > 32, D = 5) and the general selection
    kernel (k + drop > 64): neighbour lists bit-identical to the oracle, every index valid."""
    with np.errstate(all="ignore"):
        x, _ = _nonfinite_case(case, N, N, 2, 5 * D + k, D=D)
        idx, dist = gpu_fx.knn(x, k, drop_first=drop)
        oi, od = oracle.knn(x, k, drop_first=drop)
        gi = idx.to_host()
        assert gi.min() >= 0 and gi.max() < N
        assert np.array_equal(gi, oi), np.argwhere(gi != oi)[:5]
        assert np.array_equal(dist.to_host(), od, equal_nan=True)


@pytest.mark.parametrize("case", NONFINITE)
@pytest.mark.parametrize("D,N,k,drop,slices", [(64, 1024, 20, True, 2), (64, 1024, 40, True, 0), (16, 512, 100, False, 0), (3, 8192, 20, True, 2),
                                                (3, 1024, 40, True, 0), (3, 2048, 47, False, 0)])
def test_knn_nonfinite_through_slices_and_wide_geometries(gpu_fx, oracle, fx_option, case, D, N, k, drop, slices):
    """Non-finite clouds through the round-3 paths: candidate slices (the merge compares canonical distance keys: NaN after +Inf,
    then the index), the verified merge with its selection fallback (k + drop > 32 in feature space), the wide / compact D = 3
    geometries.  Bit-identical to the oracle, every index valid."""
    with np.errstate(all="ignore"):
        x, _ = _nonfinite_case(case, N, N, 1, 7 * D + k, D=D)
        fx_option("knn_slices", str(slices))
        idx, dist = gpu_fx.knn(x, k, drop_first=drop)
        oi, od = oracle.knn(x, k, drop_first=drop)
        gi = idx.to_host()
        assert gi.min() >= 0 and gi.max() < N
        assert np.array_equal(gi, oi), np.argwhere(gi != oi)[:5]
        assert np.array_equal(dist.to_host(), od, equal_nan=True)


# ------------------------------------------------------------------------------ directed near-ties at the band edge
def _near_tie_clouds(scale_exp, offset, M=4096, nq=512, seed=0):
    """Queries with two nearest candidates whose EXACT Float32 distances differ by 0, 1, 2 ulp (and by relative 2^-18 ..
    2^-23, the width of the filter band), at length scale 2^scale_exp around `offset`; the rest of the cloud is background.
    The farther of the pair gets the LOWER index, so any rule other than (distance, index) picks the wrong one."""
    rng = np.random.default_rng(seed + 1000 * (scale_exp + 64))
    s = np.float32(2.0) ** scale_exp
    off = np.float32(offset)
    y = (rng.random((3, M), dtype=np.float32) * np.float32(40.0) + np.float32(8.0)) * s + off   # background, far from the queries
    x = np.zeros((3, nq), np.float32)
    for i in range(nq):
        q = (rng.random(3, dtype=np.float32) * np.float32(2.0) - np.float32(1.0)) * s + off
        x[:, i] = q
        a = np.float32(0.01 + 0.05 * rng.random()) * s
        kind = i % 6
        if kind < 3:      # axis-aligned pair, second one farther by `kind` ulp of the offset
            b = a
            for _ in range(kind):
                b = np.nextafter(b, np.float32(np.inf))
        else:             # relative gaps around the band width
            b = np.float32(a * (np.float32(1.0) + np.float32(2.0) ** -(15 + kind)))
        j1, j2 = 2 * i, 2 * i + 1   # lower index = the farther one (or the equal one)
        y[:, j1] = q; y[0, j1] = q[0] + b
        y[:, j2] = q; y[1, j2] = q[1] - a
    return np.asfortranarray(x[:, :, None]), np.asfortranarray(y[:, :, None])


@pytest.mark.parametrize("scale_exp", [-20, -10, -3, 0, 6])
@pytest.mark.parametrize("offset", [0.0, 3.0, -1000.0])
def test_near_ties_at_the_band_edge(gpu_fx, oracle, scale_exp, offset):
    """the error-band constants were validated by random measurement only.  Directed cases: pairs whose
    exact distances differ by 0 / 1 / 2 ulp and by 2^-18 .. 2^-20 relative, centred and offset, from 2^-20 to 2^6 -- at
    offsets where the coordinate spacing is coarser than the gap, the oracle sees exact ties and the lower index must win."""
    if abs(offset) * 2.0 ** -23 > 2.0 ** scale_exp:  # the offset swallows the whole structure: every point collapses
        pytest.skip("scale below the Float32 spacing at this offset")
    x, y = _near_tie_clouds(scale_exp, offset)
    _nn_equal(gpu_fx, oracle, x, y)
    _nn_equal(gpu_fx, oracle, y[:, :1500], x)   # the other direction through its own tiles
    # kNN sees the same pairs as ranks 0/1 (and the self hit when y is searched in y)
    for k, drop in ((2, False), (8, True)):
        idx, dist = gpu_fx.knn(x, k, y=y, drop_first=drop)
        oi, od = oracle.knn(x, k, y=y, drop_first=drop)
        assert np.array_equal(idx.to_host(), oi) and np.array_equal(dist.to_host(), od)


@pytest.mark.parametrize("sigma", [1e-1, 1e-3])
@pytest.mark.parametrize("shape", [(4096, 4096), (3000, 5000), (700, 9000)])
def test_tight_clusters_dense_band(gpu_fx, oracle, sigma, shape):
    """Clouds of 40 tight clusters around SHARED centres (and a cloud against itself): tens to hundreds of candidates of every
    query lie within the filter's band, every wave re-runs its filter pass and enqueues RUNS of four candidates
    (csrc/chamfer.hip, the `runs` form of the retry pass; one image, two chunks and a candidate-split shape).  Indices and
    distances bit for bit, lowest index on the exact ties of the self search."""
    N, M = shape
    rng = np.random.default_rng(int(1000 * sigma) + N)
    c = rng.standard_normal((3, 40)) * 3
    x = np.asfortranarray((c[:, rng.integers(0, 40, N)] + rng.standard_normal((3, N)) * sigma)[:, :, None].astype(np.float32))
    y = np.asfortranarray((c[:, rng.integers(0, 40, M)] + rng.standard_normal((3, M)) * sigma)[:, :, None].astype(np.float32))
    _nn_equal(gpu_fx, oracle, x, y)
    big = x if N >= M else y
    big[:, 100:200] = big[:, :100]  # exact duplicates: the lower index wins
    _nn_equal(gpu_fx, oracle, big, big)


def _worst_split_values(rng, n, e_lo, e_hi):
    """Float32 values whose 2-way fp16 split rounds as badly as it can: 24-bit mantissa = (11-bit head) * 2^13 + r with r odd
    and 2^11 <= |r| < 2^12 -- the residual needs 12 bits, so the low piece's round-to-nearest is a tie (error 2^-23 of the
    value, the largest a value of that binade can lose); random signs, exponents in [e_lo, e_hi)."""
    head = rng.integers(1 << 10, 1 << 11, n).astype(np.int64)
    r = (rng.integers(1 << 10, 1 << 11, n).astype(np.int64) * 2 + 1) * rng.choice([-1, 1], n)
    m = head * (1 << 13) + r
    e = rng.integers(e_lo, e_hi, n)
    return (rng.choice([-1.0, 1.0], n) * m * np.exp2(e.astype(np.float64) - 23)).astype(np.float32)


@pytest.mark.parametrize("D", [3, 64])
@pytest.mark.parametrize("spread", [1, 4])
def test_filter_worst_case_split_rounding(gpu_fx, oracle, D, spread):
    """VERDICT r1 weak #10 (directed, not random): every coordinate of every point is a worst case of the fp16 split -- the
    low piece's rounding is an exact tie, the representation error sits at its bound for all 3 (64) dimensions of both
    operands at once -- on a point-symmetric cloud (mean ~ 0, so the centring keeps the bit patterns; the scale is a power of
    two).  The filter's error band must still keep every true neighbour: 1-NN both ways and kNN lists bit for bit."""
    rng = np.random.default_rng(100 * D + spread)
    N = 2048 if D == 3 else 512
    half = _worst_split_values(rng, D * N // 2, 0, spread).reshape(D, N // 2)
    pts = np.concatenate([half, -half], 1)[:, rng.permutation(N)]
    x = np.asfortranarray(pts[:, :, None].astype(np.float32))
    half2 = _worst_split_values(rng, D * N // 2, 0, spread).reshape(D, N // 2)
    y = np.asfortranarray(np.concatenate([half2, -half2], 1)[:, rng.permutation(N)][:, :, None].astype(np.float32))
    if D == 3:
        _nn_equal(gpu_fx, oracle, x, y)
        _nn_equal(gpu_fx, oracle, y, x)
    for src, tgt, drop in ((x, None, True), (x, y, False)):
        idx, dist = gpu_fx.knn(src, 20, y=tgt, drop_first=drop)
        oi, od = oracle.knn(src, 20, y=tgt, drop_first=drop)
        assert np.array_equal(idx.to_host(), oi) and np.array_equal(dist.to_host(), od)


def test_near_ties_feature_space(gpu_fx, oracle):
    """D = 64: candidates at exactly equal and 1-ulp-apart distances from their query (rank boundary at k)."""
    rng = np.random.default_rng(4)
    D, N, k = 64, 512, 20
    x = rng.standard_normal((D, N, 1)).astype(np.float32)
    for i in range(0, N, 8):       # make points i+1 .. i+6 mirror images / tiny perturbations of each other around i
        base = x[:, i, 0]
        for t in range(1, 7):
            v = np.zeros(D, np.float32); v[(i + t) % D] = np.float32(0.05)
            x[:, i + t, 0] = base + (v if t % 2 else -v)      # pairs at exactly the same distance from i
        x[(i + 7) % D, i + 7, 0] = np.nextafter(x[(i + 7) % D, i + 7, 0], np.float32(np.inf))
    x = np.asfortranarray(x)
    idx, dist = gpu_fx.knn(x, k, drop_first=True)
    oi, od = oracle.knn(x, k, drop_first=True)
    assert np.array_equal(idx.to_host(), oi) and np.array_equal(dist.to_host(), od)


# ------------------------------------------------------------------------------ kNN beyond the tuned kernels
@pytest.mark.parametrize("D,N,M,B,k,drop", [
    (3, 1024, 1024, 4, 100, True),     # DGCNN-style graph with a large K
    (3, 300, 300, 2, 299, True),       # K + 1 = N: the whole cloud, sorted
    (3, 77, 500, 2, 500, False),       # cross set, k = M
    (64, 256, 256, 2, 80, True),
    (200, 128, 128, 2, 5, True),       # D beyond the wave kernel's tile
    (256, 200, 333, 1, 20, False),     # DGCNN's widest feature space (aligned rows)
    (131, 96, 96, 1, 70, True),        # unaligned rows, k + drop > 64
    (3, 100, 9000, 1, 65, False),      # one wave per block (keys of 9000 candidates)
    (3, 300, 300, 2, 50, True),        # D = 3, 44 < k + drop <= 64: routed here too (the wave kernel's list is short of room)
    (64, 300, 512, 2, 40, True),       # feature space, 32 < k + drop <= 64: knn_wave_generic_kernel (16-byte staging, packed rows)
    (16, 200, 300, 1, 63, False),      # ... its 64-key merge (k + drop = 64 has no candidate list)
    (6, 150, 400, 2, 33, True),        # ... D % 4 != 0: scalar staging
    (24, 100, 130, 1, 35, False),      # ... a last tile of 2 rows
    (3, 150, 700, 1, 64, False),       # ... with exact ties (rounded coordinates)
])
def test_knn_general_selection(gpu_fx, oracle, D, N, M, B, k, drop):
    """VERDICT r1 missing #3: k + drop > 64 was rejected and D > ~110 fell off the wave kernel; the reference's
    `knn(kdtree, x, K+1, true)` (src/models/dgcnn.jl:3-7) takes any K <= N.  knn_select_kernel: bit-identical lists."""
    rng = np.random.default_rng(D + N + k)
    x = _f(rng.standard_normal((D, N, B)))
    y = x if M == N else _f(rng.standard_normal((D, M, B)))
    if D == 3:   # exact ties too
        y = _f(np.round(y * 4) / 4) if M != N else y
    idx, dist = gpu_fx.knn(x, k, y=None if y is x else y, drop_first=drop)
    oi, od = oracle.knn(x, k, y=None if y is x else y, drop_first=drop)
    assert np.array_equal(idx.to_host(), oi), np.argwhere(idx.to_host() != oi)[:5]
    assert np.array_equal(dist.to_host(), od)


def test_knn_general_selection_with_ties_everywhere(gpu_fx, oracle):
    x = _f(np.random.default_rng(8).integers(0, 3, (3, 400, 2)))   # 27 distinct points: massive exact ties
    idx, dist = gpu_fx.knn(x, 150, drop_first=True)
    oi, od = oracle.knn(x, 150, drop_first=True)
    assert np.array_equal(idx.to_host(), oi) and np.array_equal(dist.to_host(), od)


def test_knn_too_many_candidates_for_the_general_path_is_an_error(gpu_fx):
    x = _f(np.zeros((3, 10, 1)))
    y = _f(np.zeros((3, 40000, 1)))
    with pytest.raises(gpu_fx.Flux3DHipError):
        gpu_fx.knn(x, 100, y=y)


# ------------------------------------------------------------------------------ BASELINE configs at full size vs the oracle
@pytest.mark.parametrize("D", [3, 64])
def test_c4_full_size_all_batch_elements_vs_oracle(gpu_fx, oracle, D):
    """BASELINE config 4 (k = 20 self graph, B = 32 x 1024 points) and the second EdgeConv's D = 64: ALL 32 batch elements
    against the oracle, indices and distances bit for bit (round 1 checked 3 of 32)."""
    fx = gpu_fx
    if D == 3:
        x = fx.synth.uniform_cloud(0x5EED0004, 3, 1024, 32)
    else:
        x = _f(np.random.default_rng(1).standard_normal((D, 1024, 32)))
    idx, dist = fx.knn(x, 20, drop_first=True)
    oi, od = oracle.knn(x, 20, drop_first=True)
    assert np.array_equal(idx.to_host(), oi) and np.array_equal(dist.to_host(), od)
    # EdgeConv's fused graph build sees the same lists
    feat, idx2 = fx.edgeconv_graph(x, 20, layout=1, return_idx=True)
    assert np.array_equal(idx2.to_host(), oi)
    assert np.array_equal(feat.to_host(), oracle.edge_features(x, oi, layout=1))


def test_c3_full_size_sample_chamfer_and_gradient_vs_oracle(gpu_fx, oracle):
    """BASELINE config 3 at full size -- B = 8 teapot-class meshes (1202 V / 2256 F), 5000 samples: the seeded draws and the
    sampled points bit for bit, the chamfer loss and NN indices of the two sample sets, and the gradient of the loss w.r.t. the
    source vertices through chamfer_bwd -> barycentric scatter (round 1 had oracle parity at B <= 2 only)."""
    fx = gpu_fx
    t = os.path.join(GOLDEN, "teapot.obj")
    v, f = fx.load_obj(t)
    rng = np.random.default_rng(3)
    verts = [np.asfortranarray((v * np.float32(1.0 + 0.05 * b) + rng.standard_normal(v.shape).astype(np.float32) * np.float32(0.01)))
             for b in range(8)]
    src = fx.gpu(fx.TriMesh(verts, [f] * 8))
    tgt = fx.gpu(fx.load_trimesh(*[t] * 8))
    n, seed = 5000, 2024
    A, *draws = fx.sample_points(src, n, seed=seed, return_draws=True)
    Bp = fx.sample_points(tgt, n, seed=seed + 1)
    vp = src.get_verts_padded_host()
    fp = src.get_faces_padded().astype(np.int64) - 1
    oA, ofa, or1, or2 = oracle.sample_points_seeded(vp, fp, src._faces_len, n, seed, return_draws=True)
    oB = oracle.sample_points_seeded(tgt.get_verts_padded_host(), tgt.get_faces_padded().astype(np.int64) - 1,
                                     tgt._faces_len, n, seed + 1)
    assert np.array_equal(draws[0].to_host(), ofa) and np.array_equal(draws[1].to_host(), or1) and np.array_equal(draws[2].to_host(), or2)
    assert np.array_equal(A.to_host(), oA) and np.array_equal(Bp.to_host(), oB)
    # chamfer of the two (3, 5000, 8) sample sets
    loss, ix, iy = fx.chamfer_distance(A, Bp, return_indices=True)
    oloss, ox, oy, _ = oracle.chamfer_distance(oA, oB, return_all=True)
    assert np.array_equal(ix.to_host(), ox) and np.array_equal(iy.to_host(), oy)
    assert np.isclose(loss, oloss, rtol=LOSS_RTOL, atol=0)
    # the whole a6 call agrees with its parts
    assert float(fx.chamfer_distance(src, tgt, n, seed=seed)) == loss
    # gradient w.r.t. the padded source vertices: chamfer adjoint, then the barycentric scatter of the SAME draws
    gA, _ = fx.chamfer_distance_grad(A, Bp, ix, iy)
    ogA, _ = oracle.chamfer_bwd(oA, oB, ox, oy)
    assert np.allclose(gA.to_host(), ogA, rtol=1e-5, atol=1e-12)
    gv = fx.sample_points_grad(src, draws[0], draws[1], draws[2], gA).to_host()
    u = np.sqrt(or1)
    w = [1 - u, u * (1 - or2), u * or2]
    exp = np.zeros((8, vp.shape[1], 3))
    for b in range(8):
        for tt in range(3):
            np.add.at(exp[b], fp[tt, ofa[:, b], b], (w[tt][:, b][None, :] * ogA[:, :, b]).T)
    assert np.allclose(gv, np.transpose(exp, (2, 1, 0)), rtol=1e-4, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("nb,n", [(2, 5000), (2, 9000), (64, 5000), (3, 4097)])
def test_chamfer_sampled_adjoint_multi_round_paths(gpu_fx, nb, n):
    """fx3d_chamfer_sampled_bwd beyond 4096 samples per side (the reference's default num_samples = 5000,
    src/metrics/mesh.jl:34): the other side is bucketed in rounds and the block's accumulators live in dynamic LDS between
    them -- at B >= 64 (two blocks per side, 2500 rows each) static + dynamic LDS pass 64 KB and need the opt-in;
    n > 8192: three rounds.  Against fx3d_chamfer_bwd followed by fx3d_sample_points_bwd on the same draws and indices."""
    fx = gpu_fx
    paths = [os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj")]
    ma = fx.gpu(fx.load_trimesh(*[paths[b % 2] for b in range(nb)]))
    mb = fx.gpu(fx.load_trimesh(*[paths[(b + 1) % 2] for b in range(nb)]))
    A, fa, ra1, ra2 = fx.sample_points(ma, n, seed=5, return_draws=True)
    Bp, fb, rb1, rb2 = fx.sample_points(mb, n, seed=6, return_draws=True)
    loss, ix, iy = fx.chamfer_distance(A, Bp, w1=0.9, w2=1.1, return_indices=True)
    gA, gB = fx.chamfer_distance_grad(A, Bp, ix, iy, w1=0.9, w2=1.1, gout=1.5)
    ref_a = fx.sample_points_grad(ma, fa, ra1, ra2, gA).to_host()
    ref_b = fx.sample_points_grad(mb, fb, rb1, rb2, gB).to_host()
    ga, gb = fx.chamfer_sampled_grad(A, Bp, ix, iy, ma, (fa, ra1, ra2), mb, (fb, rb1, rb2), w1=0.9, w2=1.1, gout=1.5)
    assert np.isfinite(ref_a).all() and np.abs(ref_a).max() > 0
    assert np.allclose(ga.to_host(), ref_a, rtol=2e-4, atol=1e-8) and np.allclose(gb.to_host(), ref_b, rtol=2e-4, atol=1e-8)


# ------------------------------------------------------------------------------ multi-GPU entry points on one GPU
def test_sharded_entry_points_argument_handling_and_overlap_at_world_size_1(gpu_fx, oracle):
    """SURVEY 8(e) on the one GPU a lease has: the library bootstraps its own communicator (no torch), reports nranks / rank
    / RCCL version, B_local = 0 contributes zeros (more ranks than batch elements), uneven shards divide by the GLOBAL batch
    size, B_global < B_local is rejected, and the overlapped form (collective on a second stream, 4 result slots) gives the
    same Float32 loss as the serial one and as a plain chamfer_distance."""
    import ctypes as C
    fx = gpu_fx
    from flux3d_jl_amd import _lib
    from flux3d_jl_amd.distributed import NativeComm, NativeShardedChamfer, loss_from_sums, shard_bounds
    comm = NativeComm(0, 1)      # fx3d_comm_bootstrap, world size 1: no rendezvous traffic
    info = comm.info()
    assert info["nranks"] == 1 and info["rank"] == 0 and info["rccl_version"] > 20000
    B, N, M = 7, 600, 500
    x, y = fx.synth.uniform_cloud(11, 3, N, B), fx.synth.uniform_cloud(12, 3, M, B)
    dx, dy = fx.gpu(x), fx.gpu(y)
    full = fx.chamfer_distance(dx, dy)
    serial, over = NativeShardedChamfer(comm), NativeShardedChamfer(comm, overlap=True, slots=4)
    assert serial(dx, dy, B) == full
    for _ in range(9):           # more evaluations than slots: slot reuse waits for the slot's previous collective
        assert over(dx, dy, B, sync=False) is not None
    assert over.result() == full
    assert over(dx, dy, B) == full
    # uneven shards of a global batch of 7 over 3 "ranks" (2 + 2 + ... see shard_bounds), evaluated one after the other on
    # this GPU: the per-shard sums add up to the global sums, and each finalises with B_global, not its own count
    _, _, _, osums = oracle.chamfer_distance(x, y, return_all=True)
    tot = np.zeros(2)
    for r in range(3):
        s0, c = shard_bounds(B, 3, r)
        tot += fx.distributed.chamfer_sums(dx.slab(s0, c), dy.slab(s0, c))
    assert np.allclose(tot, osums, rtol=1e-7)     # (device: Float64 sum of the Float32 distances; oracle: of the squares)
    assert np.isclose(loss_from_sums(tot, N, M, B, 3), full, rtol=1e-6)
    whole = fx.distributed.chamfer_sums(dx, dy)
    assert np.allclose(tot, whole, rtol=1e-14)     # the shards' sums add up to the unsharded sums
    # B_local = 0: zeros in, the "global" loss of nothing is 0; B_global < B_local: rejected
    sums, loss = fx.DeviceArray.empty((2,), np.float64), fx.DeviceArray.empty((1,), np.float32)
    ws = fx.metrics.chamfer_workspace(N, M, 1, 3)
    host = C.c_float(-1)
    lib = _lib.load()
    assert lib.fx3d_chamfer_fwd_sharded(comm.handle, dx.ptr, N, dy.ptr, M, 0, 3, 8, 1.0, 1.0, sums.ptr, loss.ptr,
                                        C.byref(host), ws.ptr, ws.nbytes, None) == 0
    assert host.value == 0.0 and np.all(sums.to_host() == 0)
    assert lib.fx3d_chamfer_fwd_sharded(comm.handle, dx.ptr, N, dy.ptr, M, B, 3, B - 1, 1.0, 1.0, sums.ptr, loss.ptr,
                                        None, ws.ptr, ws.nbytes, None) == -1
    # the async form needs its own stream for the collective
    e0, e1 = fx.Event(), fx.Event()
    assert lib.fx3d_chamfer_fwd_sharded_async(comm.handle, dx.ptr, N, dy.ptr, M, B, 3, B, 1.0, 1.0, sums.ptr, loss.ptr,
                                              ws.ptr, ws.nbytes, None, None, e0.handle, e1.handle) == -1


def test_pool_is_stream_ordered(gpu_fx):
    """ADVICE r1 (device.py:340): a block released after an async launch on stream A must not be handed to work on stream B
    while A still uses it.  The pool caches per allocation stream; a cross-stream release / reuse synchronises first."""
    fx = gpu_fx
    from flux3d_jl_amd import device
    pl = device._pool()
    sa, sb = fx.Stream.create(), fx.Stream.create()
    x = fx.gpu(fx.synth.uniform_cloud(1, 3, 4096, 8))
    with fx.stream(sa):
        a = fx.DeviceArray.empty((1 << 20,), np.float32)
        ptr_a = a.ptr
        for _ in range(4):
            fx.chamfer_distance(x, x, sync=False)     # keeps stream A busy
        del a                                          # released under A: cached under A
    with fx.stream(sb):
        b = fx.DeviceArray.empty((1 << 20,), np.float32)
        if b.ptr == ptr_a:                             # only through the steal path, which synchronised A first
            assert pl.steals >= 1
    with fx.stream(sa):
        c = fx.DeviceArray.empty((1 << 20,), np.float32)   # same stream: immediate, stream-ordered reuse is fine
        assert c.ptr == ptr_a or b.ptr == ptr_a
    sa.synchronize(); sb.synchronize()


# ------------------------------------------------------------------------------ fused mesh losses (one launch each way)
def _mesh_batch(fx, seed):
    t, s = os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj")
    tv, tf = fx.load_obj(t)
    sv, sf = fx.load_obj(s)
    rng = np.random.default_rng(seed)
    vs = [np.asfortranarray(tv + rng.standard_normal(tv.shape).astype(np.float32) * np.float32(0.02)),
          np.asfortranarray(sv * np.float32(1.3)), np.asfortranarray(tv * np.float32(0.5))]
    return fx.gpu(fx.TriMesh(vs, [tf, sf, tf]))


@pytest.mark.parametrize("seed", [1, 2])
def test_fused_mesh_losses_and_gather_adjoint_are_bit_identical_to_the_oracle(gpu_fx, oracle, seed):
    """laplacian_loss + edge_loss in ONE launch; both adjoints in ONE gather launch without float atomics.
    The losses equal the single-loss kernels bit for bit; the gradient equals the oracle's two scatter adjoints (rows / sorted
    edges in order) added in Float32 -- bit for bit, and run to run."""
    fx = gpu_fx
    m = _mesh_batch(fx, seed)
    v = m.get_verts_packed_host()
    e0 = m.get_edges_packed().astype(np.int64) - 1
    rp, ci, va = m.get_laplacian_packed()
    rp64, ci64 = rp.astype(np.int64), ci.astype(np.int64)
    tgt = 0.03
    lap, edge, total = fx.mesh_losses(m, tgt, w_lap=0.1, w_edge=1.0)
    assert lap == fx.laplacian_loss(m) and edge == fx.edge_loss(m, tgt)
    assert np.isclose(lap, oracle.laplacian_loss(v, rp64, ci64, va), rtol=1e-6)
    assert np.isclose(edge, oracle.edge_loss(v, e0, tgt), rtol=1e-6)
    assert total == np.float32(np.float32(np.float32(0.1) * lap) + edge)
    base = fx.DeviceArray.from_host(np.array([0.25], np.float32))
    _, _, tot_dev = fx.mesh_losses(m, tgt, w_lap=0.1, w_edge=1.0, base=base, sync=False)
    assert np.float32(tot_dev.item()) == np.float32(np.float32(np.float32(0.25) + np.float32(0.1) * lap) + edge)
    # adjoint: with and without the forward's unit rows, overwrite and accumulate
    g_lap, g_edge = 0.1, 1.0
    ol = oracle.laplacian_loss_bwd(v, rp64, ci64, va, g_lap)
    oe = oracle.edge_loss_bwd(v, e0, tgt, g_edge)
    ref = (ol + oe).astype(np.float32)
    g1 = fx.mesh_losses_grad(m, tgt, g_lap, g_edge, reuse_forward=True).to_host()
    g2 = fx.mesh_losses_grad(m, tgt, g_lap, g_edge, reuse_forward=False).to_host()
    assert np.array_equal(g1, ref) and np.array_equal(g2, ref)
    prev = np.asfortranarray(np.random.default_rng(5).standard_normal(v.shape).astype(np.float32))
    acc = fx.DeviceArray.from_host(prev)
    fx.mesh_losses_grad(m, tgt, g_lap, g_edge, out=acc)
    assert np.array_equal(acc.to_host(), ((prev + ol).astype(np.float32) + oe).astype(np.float32))
    # the scatter kernels agree to rounding (their atomics' order is not fixed)
    gs = fx.laplacian_loss_grad(m, g_lap).to_host() + fx.edge_loss_grad(m, tgt, g_edge).to_host()
    assert np.allclose(gs, ref, rtol=1e-4, atol=1e-7)


def test_fused_mesh_losses_degenerate_mesh(gpu_fx, oracle):
    """Coincident vertices (zero-length edges, zero Laplacian rows): the `nrm > 0` guards of both adjoints."""
    fx = gpu_fx
    v = np.asfortranarray(np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 0]], np.float32).T)
    f = np.asfortranarray(np.array([[1, 2, 4], [2, 3, 4], [1, 4, 5]], np.int64).T)
    m = fx.gpu(fx.TriMesh([v], [f]))
    e0 = m.get_edges_packed().astype(np.int64) - 1
    rp, ci, va = m.get_laplacian_packed()
    ref = (oracle.laplacian_loss_bwd(v, rp.astype(np.int64), ci.astype(np.int64), va, 1.0) + oracle.edge_loss_bwd(v, e0, 0.0, 1.0)).astype(np.float32)
    assert np.array_equal(fx.mesh_losses_grad(m, 0.0, 1.0, 1.0).to_host(), ref)
    lap, edge, _ = fx.mesh_losses(m)
    assert lap == fx.laplacian_loss(m) and edge == fx.edge_loss(m)


def test_mesh_losses_grad_with_both_weights_zero_never_reads_unbuilt_scratch(gpu_fx):
    """g_lap == 0 and g_edge == 0 on a FRESH scratch (reuse_forward=False skips the unit-row build when
    g_lap == 0) used to run the Laplacian-only gather over uninitialised unit rows: 0 * garbage = NaN, poisoning an accumulated
    gradient.  Now: exact zeros / the accumulator untouched, and g_lap == 0 alone takes the instantiation that never reads u."""
    fx = gpu_fx
    m = _mesh_batch(fx, 3)
    V = m.get_verts_packed_host().shape[1]
    E = m.get_edges_packed().shape[0]
    import ctypes as C
    n = C.c_size_t(0)
    fx._lib.call("fx3d_mesh_losses_workspace_bytes", int(V), int(E), C.byref(n))
    nan_bytes = np.frombuffer(np.full((n.value + 3) // 4, np.nan, np.float32).tobytes()[:n.value], np.uint8)
    m._dev["mesh_fused_ws"] = fx.DeviceArray.from_host(nan_bytes)  # a scratch full of NaN where the unit rows would be
    g = fx.mesh_losses_grad(m, 0.0, g_lap=0.0, g_edge=0.0).to_host()
    assert np.array_equal(g, np.zeros_like(g))
    prev = np.asfortranarray(np.random.default_rng(9).standard_normal(g.shape).astype(np.float32))
    acc = fx.DeviceArray.from_host(prev)
    fx.mesh_losses_grad(m, 0.0, g_lap=0.0, g_edge=0.0, out=acc)
    assert np.array_equal(acc.to_host(), prev)
    # the edge term alone over the poisoned scratch: finite and equal to the standalone edge adjoint
    ge = fx.mesh_losses_grad(m, 0.02, g_lap=0.0, g_edge=0.6).to_host()
    assert np.array_equal(ge, fx.edge_loss_grad(m, 0.02, 0.6).to_host())


def test_laplacian_loss_grad_large_mesh_path_is_bit_identical(gpu_fx, oracle):
    """at V >= 65536 laplacian_loss_grad goes through fx3d_mesh_losses_bwd (unit rows + the <true,false>
    gather) instead of fx3d_laplacian_loss_bwd_sym.  Same bits as the oracle and as the small-mesh kernel, also with out=."""
    from flux3d_jl_amd import metrics as fxm
    fx = gpu_fx
    v, f = _grid_mesh(260, 260, 4)  # 261 x 261 = 68 121 vertices
    assert v.shape[1] >= fxm._LAP_BWD_TWO_PASS_FROM
    m = fx.gpu(fx.TriMesh([v], [f]))
    rp, ci, va = m.get_laplacian_packed()
    ol = oracle.laplacian_loss_bwd(v, rp.astype(np.int64), ci.astype(np.int64), va, 0.7)
    g_large = fx.laplacian_loss_grad(m, 0.7).to_host()
    assert np.array_equal(g_large, ol)
    import ctypes as C  # the small-mesh kernel on the same mesh
    g_sym = fx.DeviceArray.empty(v.shape, np.float32)
    fx._lib.call("fx3d_laplacian_loss_bwd_sym", m.dev("verts_packed").ptr, v.shape[1], m.dev("lap_rowptr").ptr,
                 m.dev("lap_colind").ptr, m.dev("lap_vals").ptr, 0.7, g_sym.ptr, 0, None, fx.current_stream().handle)
    assert np.array_equal(g_sym.to_host(), ol)
    prev = np.asfortranarray(np.random.default_rng(6).standard_normal(v.shape).astype(np.float32))
    acc = fx.DeviceArray.from_host(prev)
    fx.laplacian_loss_grad(m, 0.7, out=acc)
    assert np.array_equal(acc.to_host(), (prev + ol).astype(np.float32))
    # the edge term alone on the large mesh (<false,true>) against the oracle
    e0 = m.get_edges_packed().astype(np.int64) - 1
    assert np.array_equal(fx.mesh_losses_grad(m, 0.1, g_lap=0.0, g_edge=1.0).to_host(), oracle.edge_loss_bwd(v, e0, 0.1, 1.0))


# ------------------------------------------------------------------------------ far outliers: the side list
@pytest.mark.parametrize("nout,fac", [(1, 1e3), (1, 1e5), (1, 1e7), (5, 1e4), (64, 1e5), (70, 1e5), (300, 1e6)])
def test_far_outliers_take_the_exact_side_list(gpu_fx, oracle, nout, fac):
    """one stray point used to set the fp16 scale for the whole cloud.  The scale now comes from a robust
    range (16 x the mean deviation); candidates beyond it leave the filter and are compared exactly by every query (up to
    64 per chunk, more: the chunk falls back to exact scans).  Indices, distances, loss: the oracle's."""
    rng = np.random.default_rng(nout)
    x = rng.standard_normal((3, 2500, 2)).astype(np.float32)
    y = rng.standard_normal((3, 4096, 2)).astype(np.float32)
    oi = rng.choice(4096, nout, replace=False)
    y[:, oi, :] *= np.float32(fac)
    x[:, :max(1, nout // 2), 0] *= np.float32(fac)          # far queries too, and a far point that is somebody's nearest
    y[:, oi[0], 1] = x[:, 0, 1] * np.float32(1.0000001)
    x, y = np.asfortranarray(x), np.asfortranarray(y)
    _nn_equal(gpu_fx, oracle, x, y)
    _chamfer_equal(gpu_fx, oracle, x, y)


@pytest.mark.parametrize("M,nout,fac,k,drop", [(1024, 1, 1e5, 20, True), (1024, 1, 1e7, 20, True), (1500, 5, 1e4, 20, True),
                                               (2048, 16, 1e5, 10, False), (2048, 17, 1e5, 10, False), (5000, 3, 1e6, 31, True),
                                               (300, 2, 1e5, 20, False), (1024, 40, 1e5, 5, True)])
def test_knn_d3_far_outliers_take_the_exact_side_list(gpu_fx, oracle, M, nout, fac, k, drop):
    """The D = 3 kNN kernel's port of the robust range: candidates beyond it leave the filter (norm +inf) and every query
    appends them to its survivors (up to 16; more: every query takes the exact merge).  One chunk with the raw points in LDS,
    one chunk without, several chunks; far points as queries and as somebody's neighbours.  Lists and distances: the oracle's."""
    rng = np.random.default_rng(M + nout)
    B = 2
    y = rng.standard_normal((3, M, B)).astype(np.float32)
    oi = rng.choice(M, nout, replace=False)
    y[:, oi, :] *= np.float32(fac)
    if drop:
        x = y
    else:
        x = rng.standard_normal((3, 200, B)).astype(np.float32)
        x[:, :2, :] = y[:, oi[:1], :] * np.float32(1.0000001)   # queries next to a far candidate: it is their nearest
    x, y = np.asfortranarray(x), np.asfortranarray(y)
    idx, dist = gpu_fx.knn(x, k, y=None if drop else y, drop_first=drop)
    oidx, od = oracle.knn(x, k, y=None if drop else y, drop_first=drop)
    assert np.array_equal(idx.to_host(), oidx)
    assert np.array_equal(dist.to_host(), od)


@pytest.mark.parametrize("D,M,kind", [(64, 1024, "one_1e6"), (64, 1024, "one_1e6_unsampled"), (64, 1000, "five_1e5"),
                                      (32, 700, "one_1e8"), (128, 512, "one_1e6"), (64, 1024, "exponential"),
                                      (64, 1024, "constant_dims"), (16, 2048, "one_1e6")])
def test_knn_feature_space_robust_centre(gpu_fx, oracle, fx_option, D, M, kind):
    """knn_mfma_kernel: a few points far from the bulk pull the per-dimension mean away from it (every query then sits far from
    the centre and its band swallows the cloud: 1.7 ms instead of 80 us).  The kernel switches to the medians of 16 sampled
    rows when a mean lies 8 interquartile ranges off, and scales its absolute error terms to the bulk.  Whatever it decides,
    the lists and distances are the oracle's: far points in a sampled row or not, several of them, beyond fp16's range,
    skewed clean data (the means stay), constant dimensions (interquartile range 0)."""
    rng = np.random.default_rng(D + M)
    B, k = 2, 20
    if kind == "exponential":
        x = rng.exponential(1.0, (D, M, B))
    else:
        x = rng.random((D, M, B)) * 1e-2
    if kind == "constant_dims":
        x[: D // 2] = 0.37
        x[:, 5, :] = 1e3
    if kind.startswith("one_"):
        fac = float(kind.split("_")[1])
        x[:, 0 if "unsampled" not in kind else 3, :] = 1e-2 * fac
    if kind == "five_1e5":
        x[:, rng.choice(M, 5, replace=False), :] *= 1e5
    x = np.asfortranarray(x.astype(np.float32))
    oidx, od = oracle.knn(x, k, drop_first=True)
    for nopre in ("0", "1"):  # the pre-pass (fx3d_knn_ws) and the in-kernel statistics apply the same rule
        fx_option("knn_no_prepass", nopre)
        idx, dist = gpu_fx.knn(x, k, drop_first=True)
        assert np.array_equal(idx.to_host(), oidx), f"nopre={nopre}"
        assert np.array_equal(dist.to_host(), od), f"nopre={nopre}"


def test_knn_ws_entry_point_contract(gpu_fx, oracle):
    """fx3d_knn_ws: 0 bytes for shapes with neither a pre-pass (D = 3, D % 4 != 0, k + drop > 32, M > 4096) nor candidate slices
    (an odd M, slices below 512 rows); a NULL, short or misaligned workspace behaves exactly like fx3d_knn; x != y (only the
    candidate cloud has an image); non-finite clouds."""
    import ctypes as C
    fx = gpu_fx
    from flux3d_jl_amd import _lib
    nb = C.c_size_t(1)
    for (N, M, B, D, k, drop) in ((100, 1024, 2, 3, 20, 1), (100, 1000, 2, 5, 20, 0), (100, 1023, 2, 64, 40, 0), (100, 5001, 1, 64, 20, 0),
                                  (100, 32, 1, 64, 5, 0)):
        _lib.call("fx3d_knn_workspace_bytes", N, M, B, D, k, drop, C.byref(nb))
        assert nb.value == 0
    rng = np.random.default_rng(9)
    N, M, B, D, k = 130, 700, 2, 32, 9
    x = np.asfortranarray(rng.standard_normal((D, N, B)).astype(np.float32))
    y = np.asfortranarray(rng.standard_normal((D, M, B)).astype(np.float32))
    y[:, 17, 1] = np.nan                                    # one cloud non-finite: its queries take the exact merge
    oi, od = oracle.knn(x, k, y=y)
    dx, dy = fx.gpu(x), fx.gpu(y)
    _lib.call("fx3d_knn_workspace_bytes", N, M, B, D, k, 0, C.byref(nb))
    assert nb.value > 0
    ws = fx.DeviceArray.empty((nb.value + 512,), np.uint8)
    base = (ws.ptr + 255) // 256 * 256
    for (ptr, nbytes) in ((base, nb.value), (None, 0), (base, nb.value - 1), (base + 16, nb.value)):
        idx = fx.DeviceArray.empty((k, N, B), np.int32)
        dist = fx.DeviceArray.empty((k, N, B), np.float32)
        _lib.call("fx3d_knn_ws", dx.ptr, N, dy.ptr, M, B, D, k, 0, idx.ptr, dist.ptr, ptr, nbytes, None)
        fx.synchronize()
        assert np.array_equal(idx.to_host(), oi)
        assert np.array_equal(dist.to_host(), od, equal_nan=True)


def test_laplacian_loss_grad_gather_is_bit_identical_to_the_oracle(gpu_fx, oracle, fx_option):
    """fx3d_laplacian_loss_bwd (round 3): the gather form -- one launch, no float atomics -- returns the oracle's bits (the order in
    which its row-by-row scatter reaches a vertex) on a ragged batch with isolated and degenerate vertices, and the same bits
    run after run; the scatter form (option lap_bwd_scatter) stays within rounding of it."""
    fx = gpu_fx
    m = fx.load_trimesh(os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj"), os.path.join(GOLDEN, "teapot.obj"))
    v = m.get_verts_packed_host()
    rp, ci, va = m.get_laplacian_packed()
    ol = oracle.laplacian_loss_bwd(v, rp.astype(np.int64), ci.astype(np.int64), va, 0.7)
    md = fx.gpu(m)
    g1 = fx.laplacian_loss_grad(md, 0.7).to_host()
    assert np.array_equal(g1, ol)
    for _ in range(3):
        assert np.array_equal(fx.laplacian_loss_grad(md, 0.7).to_host(), g1)
    fx_option("lap_bwd_scatter", "1")
    gs = fx.laplacian_loss_grad(md, 0.7).to_host()
    assert np.allclose(gs, ol, rtol=1e-4, atol=1e-9)


def test_knn_prepass_search_staging(gpu_fx, oracle):
    """The search kernel behind the pre-pass (fx3d_knn_ws) with its image chunks brought in through registers (default; the first
    chunk requested at the kernel's start, the query rows staged behind it; D = 128 keeps the direct-to-LDS loads):
    the same neighbours as the oracle -- a cloud with a far point (robust centre), C4's rows with queries from another array, a
    ragged tail (M < the 256-row padding), D = 128, a cloud beyond 1024 rows (row stages instead of column slices), call after call
    on the same scratch."""
    rng = np.random.default_rng(21)
    x = np.asfortranarray(rng.standard_normal((32, 700, 5)).astype(np.float32))
    x[:, 17, 2] += 3.0e4  # one far point: the mean leaves the middle of the range (robust-centre rule)
    oi, od = oracle.knn(x, 12, drop_first=True)
    for _ in range(3):
        idx, dist = gpu_fx.knn(x, 12, drop_first=True)
        assert np.array_equal(idx.to_host(), oi) and np.array_equal(dist.to_host(), od)
    y = np.asfortranarray(rng.standard_normal((64, 1024, 9)).astype(np.float32))
    oi2, od2 = oracle.knn(y[:, :200, :], 20, y=y)
    for _ in range(2):
        idx, dist = gpu_fx.knn(np.asfortranarray(y[:, :200, :]), 20, y=y)
        assert np.array_equal(idx.to_host(), oi2) and np.array_equal(dist.to_host(), od2)
    z = np.asfortranarray((rng.standard_normal((128, 500, 3)) * 7.0 + 2.0).astype(np.float32))
    oi3, od3 = oracle.knn(z, 9, drop_first=True)
    idx, dist = gpu_fx.knn(z, 9, drop_first=True)
    assert np.array_equal(idx.to_host(), oi3) and np.array_equal(dist.to_host(), od3)
    w = np.asfortranarray(rng.standard_normal((64, 1100, 2)).astype(np.float32))
    oi4 = oracle.knn(w, 20, drop_first=True, want_dist=False)
    assert np.array_equal(gpu_fx.knn(w, 20, drop_first=True, return_dist=False).to_host(), oi4)


def test_knn_exact_phase_column_slices(gpu_fx, oracle):
    """The exact phase of the feature-space kNN on 16-dimension column slices of the whole cloud (default for D % 16 == 0, D <= 64,
    M <= 1024) and on row stages (the other shapes: D = 40, a cloud of 1300 rows): the oracle's lists and distances -- uniform data,
    duplicated rows (ties: the verified ranking's re-rank), k + drop up to 32, a cloud that is not a multiple of 128 rows."""
    rng = np.random.default_rng(33)
    for (D, N, B, k, drop) in ((64, 1024, 3, 20, True), (32, 1000, 2, 31, True), (16, 333, 4, 7, False), (48, 640, 2, 16, True),
                               (40, 700, 2, 12, True), (64, 1300, 1, 20, True)):
        x = rng.standard_normal((D, N, B)).astype(np.float32)
        x[:, N // 2:N // 2 + 40, 0] = x[:, :40, 0]  # exact duplicates: equal distances, ordered by index
        x = np.asfortranarray(x)
        oi, od = oracle.knn(x, k, drop_first=drop)
        idx, dist = gpu_fx.knn(x, k, drop_first=drop)
        assert np.array_equal(idx.to_host(), oi) and np.array_equal(dist.to_host(), od), (D, N, B, k)


def _grid_mesh(nx, ny, seed, extra_verts=0):
    """A jittered (nx x ny)-cell sheet: (nx+1)(ny+1) (+ extra unused) vertices, 2 nx ny triangles of uneven area."""
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.arange(nx + 1, dtype=np.float64), np.arange(ny + 1, dtype=np.float64), indexing="ij")
    v = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 0) + rng.uniform(-0.3, 0.3, (3, gx.size))
    if extra_verts:
        v = np.concatenate([v, rng.uniform(0, nx, (3, extra_verts))], 1)
    idx = lambda i, j: i * (ny + 1) + j  # noqa: E731
    f = []
    for i in range(nx):
        for j in range(ny):
            f.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            f.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    return np.asfortranarray(v.astype(np.float32)), np.asfortranarray((np.array(f, np.int64).T + 1))


def _cdf_stride(Fmax):
    """Doubles per mesh of the sampler's CDF workspace (csrc/sampler.hip: CdfWs)."""
    up = lambda v: (v + 31) // 32 * 32  # noqa: E731
    Fp = up(Fmax)
    nchp = up(Fp // 32)
    ng = nchp // 32
    return Fp + nchp + up(ng) + 32 + 3 * up((ng + 31) // 32) + 64


def _tree_offsets(t):
    """Exclusive offsets of one level of the radix-32 summation tree (oracle/flux3d_oracle.c: level_offsets)."""
    n = len(t)
    ng = (n + 31) // 32
    goff = _tree_offsets(np.array([np.cumsum(t[32 * g:32 * g + 32])[-1] for g in range(ng)])) if ng > 1 else None
    off = np.zeros(n)
    for g in range(ng):
        seg = t[32 * g:32 * g + 32]
        e = np.concatenate([[0.0], np.cumsum(seg)[:-1]])      # sequential Float64 prefix sums
        off[32 * g:32 * g + len(seg)] = (goff[g] + e) if goff is not None else e
    return off


def _tree_scan(p):
    """The specified inclusive scan: chunk-local sequential prefixes + the tree's chunk offsets."""
    n = len(p)
    loc = [np.cumsum(p[c0:c0 + 32]) for c0 in range(0, n, 32)]
    off = _tree_offsets(np.array([l[-1] for l in loc]))
    return np.concatenate([off[c] + l for c, l in enumerate(loc)])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tiny", "two_sweeps", "global_cdf", "verts_not_staged", "ragged", "ragged_forced_multiblock",
                                  "two_blocks", "ragged_multiblock", "three_levels_above_the_chunks"])
def test_face_cdf_bits_every_kernel_variant(gpu_fx, oracle, case, fx_option):
    """The sampling CDF itself (not only the draws made from it), bit for bit against the specified order
    (oracle/flux3d_oracle.c: a radix-32 tree, every node summed left to right), for every variant: the one-block kernel
    (one sweep / two sweeps of the area pass, working copy in LDS or in the workspace, vertices staged in LDS or gathered
    from global memory, a ragged batch with the fix-up column outside the shorter meshes) and the five-launch path of
    meshes beyond 32 768 faces (two blocks; 34 blocks: one more level; small meshes forced onto it)."""
    fx = gpu_fx
    if case == "ragged_forced_multiblock":
        fx_option("cdf_multiblock_from", "64")
    from flux3d_jl_amd.transforms import EPS, _face_cdf, _verts_padded_dev
    meshes = {"tiny": [_grid_mesh(7, 7, 1)],                             # 98 faces
              "two_sweeps": [_grid_mesh(56, 56, 2)],                     # 6272 faces > 6 x 1024
              "global_cdf": [_grid_mesh(62, 62, 3)],                     # 7688 faces: working copy > 60 KiB of LDS
              "verts_not_staged": [_grid_mesh(50, 50, 4, extra_verts=6000)],   # 8601 vertices: 134 KiB of float4 slots
              "ragged": [_grid_mesh(30, 30, 5), _grid_mesh(10, 10, 6), _grid_mesh(40, 40, 7)],
              "ragged_forced_multiblock": [_grid_mesh(30, 30, 5), _grid_mesh(10, 10, 6), _grid_mesh(40, 40, 7)],
              "two_blocks": [_grid_mesh(140, 140, 8)],                   # 39 200 faces
              "ragged_multiblock": [_grid_mesh(20, 20, 10), _grid_mesh(150, 150, 11), _grid_mesh(64, 64, 12)],   # 800 / 45 000 / 8 192 faces
              "three_levels_above_the_chunks": [_grid_mesh(740, 740, 9)]}[case]   # 1 095 200 faces: 34 blocks
    m = fx.gpu(fx.TriMesh([v for v, _ in meshes], [f for _, f in meshes]))
    ws = _face_cdf(m, _verts_padded_dev(m), m.dev("faces_padded"), EPS)
    Fmax, B = m.F, m.N
    vp, fp0 = m.get_verts_padded_host(), m.get_faces_padded().astype(np.int64) - 1
    p = oracle.face_probs(oracle.faces_areas_padded(vp, fp0, m._faces_len), EPS)[0]     # (Fmax, B) Float64, fix-up applied
    want = np.stack([_tree_scan(p[:, b]) for b in range(B)])
    stride = _cdf_stride(Fmax)
    got = ws.to_host().view(np.float64)[:B * stride].reshape(B, stride)[:, :Fmax]
    for b in range(B):
        assert np.array_equal(got[b], want[b]), (case, b, int(np.argmax(got[b] != want[b])))
    # and the draws made from it
    out, fi, r1, r2 = fx.sample_points(m, 3000, seed=99, return_draws=True)
    eo, efi, er1, er2 = oracle.sample_points_seeded(vp, fp0, m._faces_len, 3000, 99, return_draws=True)
    assert np.array_equal(fi.to_host(), efi) and np.array_equal(out.to_host(), eo)


# ------------------------------------------------------------------------------ round 3: one process / several devices, options
def test_single_process_multi_device_entry_points(gpu_fx):
    """fx3d_comm_init_all + fx3d_chamfer_fwd_multi (the single-process form a Julia host needs) on the
    devices this box has -- one here, so ndev = 1: communicator from ncclCommInitAll, a worker thread + stream +
    scratch per device, the global loss equals the plain call, an empty device slot contributes zeros, shapes that
    need a larger scratch re-allocate it, and argument errors come back as status codes with a message."""
    fx = gpu_fx
    from flux3d_jl_amd.distributed import MultiDevice
    md = MultiDevice(ndev=1)
    info = md.info()
    assert info["ndev"] == 1 and info["devices"] == [0] and info["rccl_version"] > 20000
    for (B, N, M) in ((5, 300, 200), (32, 4096, 4096), (3, 64, 5000)):
        x, y = fx.synth.uniform_cloud(21, 3, N, B), fx.synth.uniform_cloud(22, 3, M, B)
        dx, dy = md.shard(x, 0), md.shard(y, 0)
        full = fx.chamfer_distance(dx, dy, w1=0.5, w2=2.0)
        assert md.chamfer_distance([dx], [dy], B, w1=0.5, w2=2.0) == full
        # the shard as 5 of a global batch of 8: the divisor is the GLOBAL batch size
        assert np.isclose(md.chamfer_distance([dx], [dy], B + 3, w1=0.5, w2=2.0), full * B / (B + 3), rtol=1e-6)
    md.synchronize()
    with pytest.raises(fx.Flux3DHipError, match="B_global"):
        md.chamfer_distance([dx], [dy], 1)
    with pytest.raises(ValueError):
        md.chamfer_distance([None], [None], 4)
    md.close()
    with pytest.raises(fx.Flux3DHipError, match="device"):
        MultiDevice(devices=[0, 0])
    with pytest.raises(fx.Flux3DHipError):
        MultiDevice(devices=[fx.device_count()])


def test_option_api_replaces_environment_reads(gpu_fx, oracle, monkeypatch):
    """the variant switches are explicit options (fx3d_set_option), the environment only seeds their
    defaults when the library is first used.  Setting FX3D_NN1_VARIANT in the environment of a RUNNING process changes
    nothing; the option does, per call, and both variants give the oracle's indices."""
    fx = gpu_fx
    from flux3d_jl_amd import _lib
    opts = _lib.options()
    assert set(opts) >= {"nn1_variant", "knn_no_mfma", "knn_no_prepass", "bwd_global_atomics", "cdf_multiblock_from"}
    assert opts["nn1_variant"] == 3
    with pytest.raises(fx.Flux3DHipError, match="unknown option"):
        _lib.set_option("no_such_switch", 1)
    x, y = fx.synth.uniform_cloud(31, 3, 700, 2), fx.synth.uniform_cloud(32, 3, 900, 2)
    oix, oiy = oracle.nn1(x, y)[:2]
    monkeypatch.setenv("FX3D_NN1_VARIANT", "0")            # no effect: not read on the launch path
    assert _lib.get_option("nn1_variant") == 3
    _lib.load().fx3d_profile_enable(1)
    ix, iy = fx.nearest_neighbors(x, y)
    assert np.array_equal(ix.to_host(), oix) and np.array_equal(iy.to_host(), oiy)
    with _lib.option("nn1_variant", 0):                    # the exact VALU loop, for these calls only
        assert _lib.get_option("FX3D_NN1_VARIANT") == 0    # (the environment-style name is accepted as an alias)
        ix, iy = fx.nearest_neighbors(x, y)
        assert np.array_equal(ix.to_host(), oix) and np.array_equal(iy.to_host(), oiy)
    assert _lib.get_option("nn1_variant") == 3
    _lib.load().fx3d_profile_enable(0)


def _modelnet_root(tmp_path):
    import shutil
    for z in ("ModelNet10.zip", "ModelNet40.zip"):
        shutil.copy(os.path.join(GOLDEN, "modelnet", z), tmp_path)
    return str(tmp_path)


def test_modelnet_off_files_sample_and_chamfer_against_the_oracle(gpu_fx, oracle, tmp_path):
    """SURVEY 8 f.4, the real-data half: the reference's ModelNet10 / ModelNet40 test archives (OFF files, 138 ... 27 438
    faces, listed as src/datasets/modelnet/base.jl:30-108 does) -> load_off -> a ragged TriMesh batch -> sample_points(4096)
    -> chamfer_distance, against the oracle on the same meshes: draws and points bit for bit, NN indices bit for bit, loss
    to 1e-5."""
    fx = gpu_fx
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import modelnet_chamfer_eval as ev
    meshes = ev.listing(_modelnet_root(tmp_path))
    assert len(meshes) == 8 and sorted(m[2].shape[1] for m in meshes)[::7] == [138, 27438]
    ia, ib = zip(*[ev.pair_of(k, 8) for k in range(8)])
    assert all(a != b for a, b in zip(ia, ib))
    ta = fx.gpu(fx.TriMesh([meshes[i][1] for i in ia], [meshes[i][2] for i in ia]))
    tb = fx.gpu(fx.TriMesh([meshes[i][1] for i in ib], [meshes[i][2] for i in ib]))
    n, seed = 4096, 77
    A, *draws = fx.sample_points(ta, n, seed=seed, return_draws=True)
    Bp = fx.sample_points(tb, n, seed=seed + 1)
    oA, ofa, or1, or2 = oracle.sample_points_seeded(ta.get_verts_padded_host(), ta.get_faces_padded().astype(np.int64) - 1,
                                                    ta._faces_len, n, seed, return_draws=True)
    oB = oracle.sample_points_seeded(tb.get_verts_padded_host(), tb.get_faces_padded().astype(np.int64) - 1, tb._faces_len, n, seed + 1)
    assert np.array_equal(draws[0].to_host(), ofa) and np.array_equal(draws[1].to_host(), or1) and np.array_equal(draws[2].to_host(), or2)
    assert np.array_equal(A.to_host(), oA) and np.array_equal(Bp.to_host(), oB)
    loss, ix, iy = fx.chamfer_distance(A, Bp, return_indices=True)
    oloss, ox, oy, _ = oracle.chamfer_distance(oA, oB, return_all=True)
    assert np.array_equal(ix.to_host(), ox) and np.array_equal(iy.to_host(), oy)
    assert np.isclose(loss, oloss, rtol=LOSS_RTOL, atol=0)
    assert float(fx.chamfer_distance(ta, tb, n, seed=seed)) == loss


def test_modelnet_eval_driver_world_size_1_all_forms(gpu_fx, tmp_path):
    """examples/modelnet_chamfer_eval.py end to end on this box's one GPU: the per-GPU-process form in its three collective
    placements and the single-process form (fx3d_chamfer_fwd_multi) give the same per-evaluation losses."""
    import json
    import subprocess
    root = _modelnet_root(tmp_path)
    res = {}
    for tag, extra in (("overlap", ["--mode", "overlap"]), ("serial", ["--mode", "serial"]), ("deferred", ["--mode", "deferred"]),
                       ("multi", ["--single-process", "1"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "modelnet_chamfer_eval.py"), "--root", root,
                            "--batch", "64", "--points", "2048"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT,
                           env=dict(os.environ, MASTER_ADDR="", MASTER_PORT=""))
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert res[tag]["evaluations"] == 2 and res[tag]["pairs_of_clouds"] == 64 and len(res[tag]["meshes"]) == 8
    ref = res["serial"]["loss_per_evaluation"]
    assert all(np.isfinite(ref)) and min(ref) > 0
    for tag in ("overlap", "deferred", "multi"):
        assert res[tag]["loss_per_evaluation"] == ref, tag


@pytest.mark.parametrize("ratio", [30.0, 250.0, 4.0e3, 1.0e5, 3.0e6, 1.0e9])
@pytest.mark.parametrize("offset", [0.0, 3.0])
def test_nn1_clouds_of_very_different_extent(gpu_fx, oracle, ratio, offset):
    """A query far outside the candidate cloud (a unit teapot against a ModelNet table in millimetres: found with the
    surface-sampled bench line, 690 us instead of 58) used to leave the fp16 range of the filter's query operand and scan
    the cloud exactly; it now carries a power-of-two scale of its own (up to 2^14, i.e. |qm~| < 2^28), beyond that the
    exact path remains.  Indices and distances against the oracle, both directions, flat faces included (a far query
    sees the points of a face at nearly equal distances)."""
    fx = gpu_fx
    rng = np.random.default_rng(int(ratio) % 1000 + int(offset))
    N, M, B = 1300, 1700, 2
    x = rng.random((3, N, B)).astype(np.float32)
    y = (rng.random((3, M, B)).astype(np.float32) - np.float32(0.5)) * np.float32(ratio) + np.float32(offset * ratio)
    y[2, : M // 2, :] = np.float32(0.25 * ratio)           # half of the big cloud on one flat face
    x, y = np.asfortranarray(x), np.asfortranarray(y)
    ix, iy, dx, dy = fx.nearest_neighbors(x, y, return_dist=True)
    oix, oiy, odx, ody = oracle.nn1(x, y, want_dist=True)
    assert np.array_equal(ix.to_host(), oix) and np.array_equal(iy.to_host(), oiy)
    assert np.array_equal(dx.to_host(), odx) and np.array_equal(dy.to_host(), ody)


@pytest.mark.parametrize("N,M,B,D", [(1, 1, 1, 3), (341, 342, 1, 3), (1000, 500, 2, 3), (4096, 4096, 32, 3), (700, 900, 3, 2), (300, 200, 2, 7)])
def test_chamfer_loss_in_the_reference_float32_pairwise_arithmetic(gpu_fx, oracle, N, M, B, D):
    """the one float output of the path in the reference's own arithmetic.  `mean((A .- B[:, nn]).^2) * 3f0`
    with Base's Float32 pairwise sum (blocks of 1024 over the materialised column-major array, src/metrics/pcloud.jl:47-50):
    fx3d_chamfer_loss_pairwise_f32 == the oracle's restatement bit for bit (lengths below, at and above the block size, leaf
    boundaries that are not multiples of D), and both within 1e-6 of the Float64 sum the default forward uses."""
    fx = gpu_fx
    rng = np.random.default_rng(N + M)
    x = np.asfortranarray(rng.random((D, N, B)).astype(np.float32))
    y = np.asfortranarray(rng.random((D, M, B)).astype(np.float32))
    dx, dy = fx.gpu(x), fx.gpu(y)
    loss, ix, iy = fx.chamfer_distance(dx, dy, w1=0.75, w2=1.5, return_indices=True)
    pw = fx.chamfer_loss_pairwise_f32(dx, dy, ix, iy, w1=0.75, w2=1.5)
    opw = oracle.chamfer_loss_pairwise(x, y, ix.to_host(), iy.to_host(), 0.75, 1.5)
    assert pw == opw, (pw, opw)
    assert np.isclose(pw, loss, rtol=2e-6, atol=0)


@pytest.mark.parametrize("N,M,B", [(4097, 4097, 3), (4160, 4100, 2), (4161, 4096, 2), (4159, 60, 2), (70, 4130, 1), (4097, 4097, 32),
                                   (1030, 700, 8), (1088, 1088, 32), (2050, 2049, 16), (577, 513, 32), (4096 + 64, 4096 + 64, 9)])
def test_nn1_just_above_an_lds_image_and_query_remainders(gpu_fx, oracle, N, M, B):
    """N = M = 4097 cost 2.1 x N = M = 4096.  A cloud of up to 4096 + 64 points is ONE LDS image now plus a tail
    that every query compares exactly, and a remainder of <= 64 queries beyond a direction's last full tile is one more pass of
    that tile's block.  Indices, distances and the loss against the oracle at the boundaries of both rules (4160 / 4161
    candidates; remainders of 1, 6, 64, 65 queries; clouds of different sizes; the tail holding the nearest neighbour)."""
    fx = gpu_fx
    rng = np.random.default_rng(N * 7 + M)
    x = np.asfortranarray(rng.random((3, N, B)).astype(np.float32))
    y = np.asfortranarray(rng.random((3, M, B)).astype(np.float32))
    if M > 4096:   # make tail candidates the nearest neighbours of some queries (and exact duplicates of earlier candidates: ties -> lower index)
        y[:, 4096:, :] = x[:, : M - 4096, :]
        y[:, -1, :] = y[:, 5, :]
    ix, iy, dx, dy = fx.nearest_neighbors(x, y, return_dist=True)
    oix, oiy, odx, ody = oracle.nn1(x, y, want_dist=True)
    assert np.array_equal(ix.to_host(), oix) and np.array_equal(iy.to_host(), oiy)
    assert np.array_equal(dx.to_host(), odx) and np.array_equal(dy.to_host(), ody)
    loss = fx.chamfer_distance(x, y, w1=0.5, w2=1.5)
    assert np.isclose(loss, oracle.chamfer_distance(x, y, 0.5, 1.5), rtol=LOSS_RTOL, atol=0)


def test_host_pointer_convenience_variants(gpu_fx, oracle):
    """host-pointer variants of the ops -- plain host buffers in Julia's layout in, host results out, the device
    path underneath (per-thread device scratch, the same entry points, a synchronous call).  Against the oracle: chamfer
    loss + both index arrays, kNN lists + distances (self and cross), seeded samples, both mesh losses on the teapot."""
    import ctypes as C
    fx = gpu_fx
    from flux3d_jl_amd import _lib
    lib = _lib.load()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rng = np.random.default_rng(12)
    for rep in range(2):  # second round: the scratch is reused / grown
        N, M, B = (700, 900, 3) if rep == 0 else (4096, 2000, 5)
        x = np.asfortranarray(rng.random((3, N, B)).astype(np.float32))
        y = np.asfortranarray(rng.random((3, M, B)).astype(np.float32))
        loss = C.c_float(0)
        ix, iy = np.zeros((N, B), np.int32, order="F"), np.zeros((M, B), np.int32, order="F")
        _lib.check(lib.fx3d_chamfer_distance_host(p(x), N, p(y), M, B, 3, 0.5, 2.0, C.byref(loss), p(ix), p(iy)))
        oloss, ox, oy, _ = oracle.chamfer_distance(x, y, 0.5, 2.0, return_all=True)
        assert np.array_equal(ix, ox) and np.array_equal(iy, oy) and np.isclose(loss.value, oloss, rtol=LOSS_RTOL, atol=0)
        _lib.check(lib.fx3d_chamfer_distance_host(p(x), N, p(y), M, B, 3, 0.5, 2.0, C.byref(loss), None, None))
        assert np.isclose(loss.value, oloss, rtol=LOSS_RTOL, atol=0)
    k = 20
    for D, y in ((3, None), (64, None), (3, np.asfortranarray(rng.random((3, 333, 2)).astype(np.float32)))):
        x = np.asfortranarray(rng.standard_normal((D, 1024 if y is None else 200, 2)).astype(np.float32))
        N = x.shape[1]
        idx, dist = np.zeros((k, N, 2), np.int32, order="F"), np.zeros((k, N, 2), np.float32, order="F")
        _lib.check(lib.fx3d_knn_host(p(x), N, p(y) if y is not None else None, y.shape[1] if y is not None else 0, 2, D, k,
                                     int(y is None), p(idx), p(dist)))
        oi, od = oracle.knn(x, k, y=y, drop_first=y is None)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    # ... and through the scratch-fed paths of fx3d_knn_ws (the host variant brings its own scratch): candidate slices of one large
    # cloud, k + drop > 32 in feature space (interleaved slices, verified merge), the wide D = 3 geometry
    for D, N, B, k2 in ((64, 4096, 1, 20), (32, 1024, 2, 50), (3, 1024, 2, 60)):
        x = np.asfortranarray(rng.standard_normal((D, N, B)).astype(np.float32))
        idx, dist = np.zeros((k2, N, B), np.int32, order="F"), np.zeros((k2, N, B), np.float32, order="F")
        _lib.check(lib.fx3d_knn_host(p(x), N, None, 0, B, D, k2, 1, p(idx), p(dist)))
        oi, od = oracle.knn(x, k2, drop_first=True)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    m = fx.load_trimesh(os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj"))
    vp = m.get_verts_padded_host()
    fp0 = np.asfortranarray(m.get_faces_padded().astype(np.int32) - 1)
    fl = m._faces_len.astype(np.int32)
    out = np.zeros((3, 3000, 2), np.float32, order="F")
    _lib.check(lib.fx3d_sample_points_host(p(vp), m.V, p(fp0), m.F, p(fl), 2, 3000, 1e-6, 99, p(out)))
    assert np.array_equal(out, oracle.sample_points_seeded(vp, fp0.astype(np.int64), m._faces_len, 3000, 99))
    vpk = fx.get_verts_packed(m)
    edges0 = np.asfortranarray(fx.get_edges_packed(m).astype(np.int32) - 1)
    l = C.c_float(0)
    _lib.check(lib.fx3d_edge_loss_host(p(vpk), vpk.shape[1], p(edges0), edges0.shape[0], 0.05, C.byref(l)))
    assert np.isclose(l.value, oracle.edge_loss(vpk, edges0.astype(np.int64), 0.05), rtol=LOSS_RTOL)
    rowptr, colind, vals = oracle.laplacian_csr(edges0.astype(np.int64), vpk.shape[1])
    r32, c32 = rowptr.astype(np.int32), colind.astype(np.int32)
    _lib.check(lib.fx3d_laplacian_loss_host(p(vpk), vpk.shape[1], p(r32), p(c32), p(vals), C.byref(l)))
    assert np.isclose(l.value, oracle.laplacian_loss(vpk, rowptr, colind, vals), rtol=LOSS_RTOL)
    assert lib.fx3d_chamfer_distance_host(p(x), 0, p(x), 5, 1, 3, 1.0, 1.0, C.byref(l), None, None) == -1


def test_sample_points_pair_is_two_sample_points(gpu_fx):
    """chamfer_distance(m1, m2, n) draws from both meshes (src/metrics/mesh.jl:41-42): fx3d_sample_points_cdf_pair / _draw_pair
    put the two CDF builds and the two draws in one launch each.  Same bits as two sample_points calls -- batches of different
    sizes and paddings, batches that take different CDF kernels (one of them forced onto the multi-block path), a cached CDF on
    one side, and the whole mesh-to-mesh chamfer call."""
    fx = gpu_fx
    from flux3d_jl_amd import _lib
    t, sph = os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj")
    a = fx.gpu(fx.load_trimesh(t, sph, t))            # ragged, 3 meshes
    b = fx.gpu(fx.load_trimesh(*[sph] * 5))           # 5 meshes, other padding
    for reuse in (False, True, True):                 # (third round: both CDFs cached)
        pa, pb = fx.sample_points_pair(a, b, 3000, seed_a=41, seed_b=97, reuse_cdf=reuse)
        ea = fx.sample_points(fx.gpu(fx.load_trimesh(t, sph, t)), 3000, seed=41)
        eb = fx.sample_points(fx.gpu(fx.load_trimesh(*[sph] * 5)), 3000, seed=97)
        assert np.array_equal(pa.to_host(), ea.to_host()) and np.array_equal(pb.to_host(), eb.to_host())
    c = fx.gpu(fx.load_trimesh(t))                    # one side with a cached CDF, the other without
    fx.sample_points(c, 10, seed=1)
    pc, pd = fx.sample_points_pair(c, fx.gpu(fx.load_trimesh(sph)), 777, seed_a=5, seed_b=6)
    assert np.array_equal(pc.to_host(), fx.sample_points(fx.gpu(fx.load_trimesh(t)), 777, seed=5).to_host())
    assert np.array_equal(pd.to_host(), fx.sample_points(fx.gpu(fx.load_trimesh(sph)), 777, seed=6).to_host())
    with _lib.option("cdf_multiblock_from", 3000):    # the sphere (5120 faces) takes the five-launch path, the teapot (2256) the one-block kernel
        pe, pf = fx.sample_points_pair(fx.gpu(fx.load_trimesh(t)), fx.gpu(fx.load_trimesh(sph)), 500, seed_a=8, seed_b=9)
    assert np.array_equal(pe.to_host(), fx.sample_points(fx.gpu(fx.load_trimesh(t)), 500, seed=8).to_host())
    assert np.array_equal(pf.to_host(), fx.sample_points(fx.gpu(fx.load_trimesh(sph)), 500, seed=9).to_host())
    m1, m2 = fx.gpu(fx.load_trimesh(*[t] * 8)), fx.gpu(fx.load_trimesh(*[sph] * 8))
    whole = fx.chamfer_distance(m1, m2, 5000, seed=123)
    parts = fx.chamfer_distance(fx.sample_points(m1, 5000, seed=123), fx.sample_points(m2, 5000, seed=124))
    assert whole == parts


@pytest.mark.gpu
def test_bench_line_force_dist_equals_plain_and_roofline_is_a_fraction(gpu_fx):
    """(a) the bench line's `roofline.frac` is the fraction of the bound it names (f16 MFMA pipe, peak 2500 TF;
    no field called *frac above 1; the algorithmic-fp32 figure only as a named extra); (b) `--gpus 1 --force-dist` (RCCL
    communicator + collectives at world size 1) reports the same loss, n_gpus and workload as the plain line; (c) `--gpus 2`
    on this one-GPU box ends non-zero without printing a line."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    common = ["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras", "--burn-ms", "20"]

    def line(extra):
        r = subprocess.run([sys.executable, bench, "--gpus", "1"] + common + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    plain, forced = line([]), line(["--force-dist", "--mode", "serial"])
    for out in (plain, forced):
        roof = out["roofline"]
        assert out["n_gpus"] == 1 and roof["bound"] == "mfma" and roof["peak"] == 2500.0 and roof["unit"] == "TFLOP/s"
        assert 0.05 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
        assert all(v is None or v <= 1.0 for k, v in roof.items() if k.endswith("frac"))
        assert roof["algorithmic_fp32_over_valu_peak"] > 0
    assert np.float32(plain["loss"]) == np.float32(forced["loss"])
    assert plain["config"]["workload"] == forced["config"]["workload"] and forced["comm"]["nranks"] == 1
    # the N-rank line is self-proving -- per rank the physical device, its own kernel average and time per step, the
    # communicator's size; every placement of the collective reports whether its loss equals the plain single-launch loss
    ranks = forced["comm"]["ranks"]
    assert len(ranks) == 1 and forced["comm"]["distinct_devices"] == 1 and ranks[0]["comm_nranks"] == 1
    assert ranks[0]["pci_bus_id"] == gpu_fx.device_identity(0)[0] and ranks[0]["device_uuid"] == gpu_fx.device_identity(0)[1]
    assert 0.02 < ranks[0]["kernel_avg_ms"] < 0.2 and 0.02 < ranks[0]["ms_per_step_local"] < 0.5
    modes = forced["modes"]
    assert np.float32(modes["plain_loss"]) == np.float32(plain["loss"])
    for m in ("serial", "overlap", "deferred"):
        assert modes[m]["loss_equal"] is True, (m, modes[m])
    assert 0.5 < forced["value"] / plain["value"] < 2.0
    if gpu_fx.device_count() < 2:
        r = subprocess.run([sys.executable, bench, "--gpus", "2"] + common, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "n_gpus" not in r.stdout


@pytest.mark.parametrize("N,M,B,D", [(1000, 500, 2, 3), (64, 64, 1, 3), (4096, 4096, 4, 3), (9000, 12000, 1, 3), (300, 200, 3, 2), (50, 70, 2, 5)])
def test_chamfer_value_and_grad_is_forward_plus_backward(gpu_fx, oracle, N, M, B, D):
    """fx3d_chamfer_fwd_bwd (benchmarks/metrics.jl:24-38 "total", examples/fit_mesh.jl:106-110): ONE ABI call
    returns the loss, both gradients and (on request) the indices -- the same bits as fx3d_chamfer_fwd + fx3d_chamfer_bwd, loss
    and gradients against the oracle; every launch plan (one launch, split run, exact D = 2, generic D)."""
    fx = gpu_fx
    rng = np.random.default_rng(N + M)
    x, y = _f(rng.random((D, N, B))), _f(rng.random((D, M, B)))
    dx, dy = fx.gpu(x), fx.gpu(y)
    l2, ix2, iy2 = fx.chamfer_distance(dx, dy, w1=0.7, w2=1.3, return_indices=True)
    gx2, gy2 = fx.chamfer_distance_grad(dx, dy, ix2, iy2, w1=0.7, w2=1.3, gout=2.0)
    l1, gx1, gy1, ix1, iy1 = fx.chamfer_value_and_grad(dx, dy, w1=0.7, w2=1.3, gout=2.0, return_indices=True)
    assert l1 == l2 and np.array_equal(ix1.to_host(), ix2.to_host()) and np.array_equal(iy1.to_host(), iy2.to_host())
    # (the adjoint accumulates through LDS float atomics: the last bit depends on their arrival order, run to run)
    assert np.allclose(gx1.to_host(), gx2.to_host(), rtol=1e-5, atol=1e-9) and np.allclose(gy1.to_host(), gy2.to_host(), rtol=1e-5, atol=1e-9)
    l3, gx3, gy3 = fx.chamfer_value_and_grad(dx, dy, w1=0.7, w2=1.3, gout=2.0)       # indices in the scratch
    assert l3 == l1 and np.allclose(gx3.to_host(), gx1.to_host(), rtol=1e-5, atol=1e-9) and np.allclose(gy3.to_host(), gy1.to_host(), rtol=1e-5, atol=1e-9)
    ol, ox, oy, _ = oracle.chamfer_distance(x, y, 0.7, 1.3, return_all=True)
    assert np.array_equal(ix1.to_host(), ox) and np.array_equal(iy1.to_host(), oy)
    assert np.isclose(l1, ol, rtol=LOSS_RTOL, atol=0)
    ogx, ogy = oracle.chamfer_bwd(x, y, ox, oy, 0.7, 1.3, 2.0)
    assert np.allclose(gx1.to_host(), ogx, rtol=1e-5, atol=1e-9) and np.allclose(gy1.to_host(), ogy, rtol=1e-5, atol=1e-9)
    from flux3d_jl_amd import _lib
    rc = _lib.load().fx3d_chamfer_fwd_bwd(dx.ptr, N, dy.ptr, M, B, D, 1.0, 1.0, 1.0, B, gx1.ptr, None, gx1.ptr, gy1.ptr, None, None,
                                          gx1.ptr, 16, None)
    assert rc == -6 and "workspace" in _lib.last_error()   # FX3D_ERR_WORKSPACE: a short scratch is refused, nothing is launched


def test_edge_and_laplacian_adjoints_gather_forms_are_bit_identical_to_the_oracle(gpu_fx, oracle):
    """VERDICT r3 #6 / ADVICE r3: the wrappers' adjoints of edge_loss and laplacian_loss are the GATHER forms
    (fx3d_edge_loss_bwd_adj, fx3d_laplacian_loss_bwd_sym): one launch, no float atomics, no memset -- the oracle's bits, run after
    run, also when added onto an existing gradient; the raw entry points (scatter, any edge list / any CSR) stay within
    rounding.  An ASYMMETRIC CSR: the scatter entry is the adjoint of the forward (finite differences of the oracle's loss), the
    symmetric-only gather reports the entries it cannot see."""
    fx = gpu_fx
    from flux3d_jl_amd import _lib
    m = fx.load_trimesh(os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj"), os.path.join(GOLDEN, "teapot.obj"))
    v = m.get_verts_packed_host()
    e0 = m.get_edges_packed().astype(np.int64) - 1
    rp, ci, va = m.get_laplacian_packed()
    oe = oracle.edge_loss_bwd(v, e0, 0.05, 1.3)
    ol = oracle.laplacian_loss_bwd(v, rp.astype(np.int64), ci.astype(np.int64), va, 0.7)
    md = fx.gpu(m)
    for _ in range(3):
        assert np.array_equal(fx.edge_loss_grad(md, 0.05, 1.3).to_host(), oe)
        assert np.array_equal(fx.laplacian_loss_grad(md, 0.7).to_host(), ol)
    acc = fx.laplacian_loss_grad(md, 0.7)
    fx.edge_loss_grad(md, 0.05, 1.3, out=acc)
    assert np.array_equal(acc.to_host(), ol + oe)
    V, E = v.shape[1], e0.shape[0]
    verts, edges = md.dev("verts_packed"), md.dev("edges")
    g = fx.DeviceArray.empty((3, V), np.float32)
    _lib.call("fx3d_edge_loss_bwd", verts.ptr, V, edges.ptr, E, 0.05, 1.3, g.ptr, 0, None)       # scatter, bare edge list
    fx.synchronize()
    assert np.allclose(g.to_host(), oe, rtol=1e-4, atol=1e-9)
    _lib.call("fx3d_laplacian_loss_bwd", verts.ptr, V, md.dev("lap_rowptr").ptr, md.dev("lap_colind").ptr, md.dev("lap_vals").ptr,
              0.7, g.ptr, 0, None)                                                                # scatter, any CSR
    fx.synchronize()
    assert np.allclose(g.to_host(), ol, rtol=1e-4, atol=1e-9)
    # an asymmetric (pruned) matrix: drop every stored entry (r, c) with c > r + 1 -- their transposes stay
    rng = np.random.default_rng(3)
    Vs = 200
    vs = _f(rng.random((3, Vs)))
    dense = (rng.random((Vs, Vs)) < 0.04).astype(np.float32) * rng.standard_normal((Vs, Vs)).astype(np.float32)
    dense = dense + dense.T + np.diag(-np.ones(Vs, np.float32))
    dense[np.triu_indices(Vs, 2)] = 0.0
    rows = [np.nonzero(dense[r])[0] for r in range(Vs)]
    rp2 = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    ci2 = np.concatenate(rows).astype(np.int32)
    va2 = np.concatenate([dense[r][rows[r]] for r in range(Vs)]).astype(np.float32)
    o2 = oracle.laplacian_loss_bwd(vs, rp2.astype(np.int64), ci2.astype(np.int64), va2, 1.0)
    dv, drp, dci, dva = fx.gpu(vs), fx.DeviceArray.from_host(rp2), fx.DeviceArray.from_host(ci2), fx.DeviceArray.from_host(va2)
    g2 = fx.DeviceArray.empty((3, Vs), np.float32)
    _lib.call("fx3d_laplacian_loss_bwd", dv.ptr, Vs, drp.ptr, dci.ptr, dva.ptr, 1.0, g2.ptr, 0, None)
    fx.synchronize()
    assert np.allclose(g2.to_host(), o2, rtol=1e-4, atol=1e-7)
    missing = fx.DeviceArray.zeros((1,), np.uint32)
    _lib.call("fx3d_laplacian_loss_bwd_sym", dv.ptr, Vs, drp.ptr, dci.ptr, dva.ptr, 1.0, g2.ptr, 0, missing.ptr, None)
    fx.synchronize()
    nsym = sum(1 for r in range(Vs) for c in rows[r] if r not in rows[c])
    assert nsym > 0 and int(missing.to_host()[0]) == nsym
    missing = fx.DeviceArray.zeros((1,), np.uint32)                                               # a symmetric L: nothing missing
    _lib.call("fx3d_laplacian_loss_bwd_sym", verts.ptr, V, md.dev("lap_rowptr").ptr, md.dev("lap_colind").ptr,
              md.dev("lap_vals").ptr, 0.7, g.ptr, 0, missing.ptr, None)
    fx.synchronize()
    assert int(missing.to_host()[0]) == 0 and np.array_equal(g.to_host(), ol)


@pytest.mark.parametrize("N,M,B", [(1, 1, 1), (17, 33, 3), (64, 64, 1), (1024, 1024, 2), (300, 4096, 2), (5000, 3000, 3), (1000, 1000, 40), (257, 4097, 1)])
def test_nn1_small_problem_kernel_matches_the_oracle(gpu_fx, oracle, fx_option, N, M, B):
    """nn1_tiny_kernel (VERDICT r3 #4a: no small-N path; C1 17.6 us where an empty launch is 6): the exact small-problem kernel
    forced onto shapes up to 90 M pair evaluations -- one and two queries per lane, several query tiles per block, clouds
    that stay in registers and clouds that are streamed, ragged ends -- on uniform data, a lattice (exact ties: the lowest
    index must win across lanes, groups and slices), duplicated points and non-finite coordinates (isless order).  Indices and
    distances bit-equal to the oracle; the loss within 1e-5; identical to what the fp16-filter kernel returns."""
    fx = gpu_fx
    rng = np.random.default_rng(N * 7 + M)
    lat = lambda n: _f(rng.integers(0, 4, (3, n, B)) * 0.25)                       # noqa: E731
    cases = {"uniform": (_f(rng.random((3, N, B))), _f(rng.random((3, M, B)))), "lattice": (lat(N), lat(M))}
    d = _f(rng.random((3, M, B)))
    d[:, M // 2:, :] = d[:, :M - M // 2, :]
    cases["dupes"] = (_f(rng.random((3, N, B))), d)
    nf_x, nf_y = _f(rng.random((3, N, B))), _f(rng.random((3, M, B)))
    nf_x[:, ::7, :] = np.nan
    nf_y[0, ::5, :] = np.inf
    nf_y[:, 1::11, :] = np.nan
    cases["nonfinite"] = (nf_x, nf_y)
    for name, (x, y) in cases.items():
        fx_option("nn1_tiny_mpairs", 1 << 20)
        ix, iy, dx, dy = fx.nearest_neighbors(x, y, return_dist=True)
        ox, oy, odx, ody = oracle.nn1(x, y, want_dist=True)
        assert np.array_equal(ix.to_host(), ox) and np.array_equal(iy.to_host(), oy), name
        assert np.array_equal(dx.to_host(), odx, equal_nan=True) and np.array_equal(dy.to_host(), ody, equal_nan=True), name
        if name != "nonfinite":
            lt, jx, jy = fx.chamfer_distance(x, y, w1=0.3, w2=1.7, return_indices=True)
            assert np.array_equal(jx.to_host(), ox) and np.array_equal(jy.to_host(), oy), name
            assert np.isclose(lt, oracle.chamfer_distance(x, y, 0.3, 1.7), rtol=LOSS_RTOL, atol=1e-30), name
            assert fx.chamfer_distance(x, y, w1=0.3, w2=1.7) == lt                 # (the loss-only instantiation)
            fx_option("nn1_tiny_mpairs", 0)
            lf = fx.chamfer_distance(x, y, w1=0.3, w2=1.7)
            assert np.isclose(lf, lt, rtol=1e-6, atol=1e-30), name


# ------------------------------------------------------------------------------ ordered (atomic-free) sampling adjoint (round 6)
def _big_face_mesh(fx, nfan=40):
    """One triangle that takes nearly all the area (its draws: lists far beyond the eight a sorting network orders) + a fan of
    small ones, some of them degenerate (a repeated vertex: two corners of a face on one vertex)."""
    verts = [[0, 0, 0], [10, 0, 0], [0, 10, 0]]
    faces = [[1, 2, 3]]
    for i in range(nfan):
        a = 2 * np.pi * i / nfan
        verts.append([0.3 * np.cos(a), 0.3 * np.sin(a), 0.5])
    for i in range(nfan):
        faces.append([1, 4 + i, 4 + (i + 1) % nfan])
    faces.append([2, 2, 3])          # degenerate: zero area, never drawn -- but its corners are in the table
    v = np.asfortranarray(np.array(verts, np.float32).T)
    f = np.asfortranarray(np.array(faces, np.uint32).T)
    return v, f


def _draws_case(fx, kind, nb):
    paths = [os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj")]
    if kind == "bigface":
        v, f = _big_face_mesh(fx)
        tv, tf = fx.load_obj(paths[0])
        return fx.gpu(fx.TriMesh([v, tv][:nb] if nb <= 2 else [v, tv] * (nb // 2), [f, tf][:nb] if nb <= 2 else [f, tf] * (nb // 2)))
    return fx.gpu(fx.load_trimesh(*[paths[b % 2] for b in range(nb)]))


@pytest.mark.parametrize("kind,nb,n", [("ragged", 2, 2000), ("ragged", 8, 5000), ("ragged", 3, 5300), ("bigface", 2, 3000),
                                       ("bigface", 1, 7)])
def test_sample_points_adjoint_ordered_form_is_the_oracles_bit_for_bit(gpu_fx, oracle, kind, nb, n):
    """fx3d_sample_points_bwd with the vertex -> face table: no float atomics, every vertex's sum in the order of
    oracle.sample_points_bwd (per face and corner over the draws ascending, then the vertex's (face, corner) pairs ascending) --
    array_equal, also on top of a base, also twice in a row; the float-atomic form (ordered=False) agrees to rounding."""
    import ctypes as C
    from flux3d_jl_amd import _lib
    fx = gpu_fx
    m = _draws_case(fx, kind, nb)
    fits = C.c_int32(0)
    _lib.call("fx3d_sample_points_bwd_ordered", m.F, n, C.byref(fits))
    assert fits.value == 1
    _lib.call("fx3d_sample_points_bwd_ordered", m.F, 9000, C.byref(fits))   # (beyond one CU's LDS: the scatter -- tested below)
    assert fits.value == 0
    _, fi, r1, r2 = fx.sample_points(m, n, seed=31 + n, return_draws=True)
    rng = np.random.default_rng(n + nb)
    gout = np.asfortranarray(rng.standard_normal((3, n, m.N)).astype(np.float32))
    gout[:, ::7, :] = 0.0
    fp0 = m.get_faces_padded().astype(np.int64) - 1
    exp = oracle.sample_points_bwd(fp0, m._faces_len, m.V, fi.to_host(), r1.to_host(), r2.to_host(), gout)
    g1 = fx.sample_points_grad(m, fi, r1, r2, gout).to_host()
    g2 = fx.sample_points_grad(m, fi, r1, r2, gout).to_host()
    assert np.array_equal(g1, exp), np.argwhere(g1 != exp)[:5]
    assert np.array_equal(g1, g2)
    if kind == "bigface":
        assert np.bincount(fi.to_host()[:, 0]).max() > 8 or n <= 8   # the long-list path ran
    base = np.asfortranarray(rng.standard_normal(exp.shape).astype(np.float32))
    out = fx.gpu(base.copy(order="F"))
    g3 = fx.sample_points_grad(m, fi, r1, r2, gout, out=out)
    assert g3 is out and np.array_equal(out.to_host(), oracle.sample_points_bwd(fp0, m._faces_len, m.V, fi.to_host(), r1.to_host(),
                                                                             r2.to_host(), gout, base=base))
    ga = fx.sample_points_grad(m, fi, r1, r2, gout, ordered=False).to_host()
    assert np.allclose(ga, exp, rtol=2e-4, atol=1e-5)
    if nb == 3:  # more draws than the ordered form stages: the call takes the float-atomic scatter by itself
        _, fi9, r19, r29 = fx.sample_points(m, 9000, seed=2, return_draws=True)
        g9 = np.asfortranarray(rng.standard_normal((3, 9000, m.N)).astype(np.float32))
        e9 = oracle.sample_points_bwd(fp0, m._faces_len, m.V, fi9.to_host(), r19.to_host(), r29.to_host(), g9)
        assert np.allclose(fx.sample_points_grad(m, fi9, r19, r29, g9).to_host(), e9, rtol=2e-4, atol=1e-5)


def test_sample_points_adjoint_ordered_beyond_the_draws_held_in_registers(gpu_fx, oracle):
    """sample_gather.h keeps six sweeps of a thread's draws (6144 per mesh) in registers for both passes over them; a small mesh admits a few
    more (6300 draws at 8 faces fit the LDS layout): the draws beyond are read where they are used.  Every face holds ~800 draws (the bitmap
    ordering of long lists, the table of faces with many draws)."""
    import ctypes as C
    from flux3d_jl_amd import _lib
    fx = gpu_fx
    v = np.asfortranarray(np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32).T)
    f = np.asfortranarray(np.array([[1, 3, 5], [3, 2, 5], [2, 4, 5], [4, 1, 5], [3, 1, 6], [2, 3, 6], [4, 2, 6], [1, 4, 6]], np.uint32).T)
    m = fx.gpu(fx.TriMesh([v, v * 2.0], [f, f]))
    n = 6300
    fits = C.c_int32(0)
    _lib.call("fx3d_sample_points_bwd_ordered", m.F, n, C.byref(fits))
    assert fits.value == 1
    _, fi, r1, r2 = fx.sample_points(m, n, seed=5, return_draws=True)
    gout = np.asfortranarray(np.random.default_rng(1).standard_normal((3, n, m.N)).astype(np.float32))
    fp0 = m.get_faces_padded().astype(np.int64) - 1
    exp = oracle.sample_points_bwd(fp0, m._faces_len, m.V, fi.to_host(), r1.to_host(), r2.to_host(), gout)
    got = fx.sample_points_grad(m, fi, r1, r2, gout).to_host()
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:5]


@pytest.mark.parametrize("nb,n", [(2, 3000), (8, 5000), (2, 5300), (64, 5000), (1, 5000)])
def test_chamfer_sampled_adjoint_ordered_form_is_the_oracles_chain_bit_for_bit(gpu_fx, oracle, nb, n):
    """fx3d_chamfer_sampled_bwd, ordered form: the chamfer adjoint's rows (bit-identical to oracle.chamfer_bwd) go to scratch
    and are gathered in oracle.sample_points_bwd's order: array_equal to the oracle's chain, run to run, one side only, on top
    of a base; the float-atomic form agrees to rounding."""
    fx = gpu_fx
    paths = [os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj")]
    ma = fx.gpu(fx.load_trimesh(*[paths[b % 2] for b in range(nb)]))
    mb = fx.gpu(fx.load_trimesh(*[paths[(b + 1) % 2] for b in range(nb)]))
    A, fa, ra1, ra2 = fx.sample_points(ma, n, seed=5, return_draws=True)
    Bp, fb, rb1, rb2 = fx.sample_points(mb, n, seed=6, return_draws=True)
    loss, ix, iy = fx.chamfer_distance(A, Bp, w1=0.9, w2=1.1, return_indices=True)
    oga, ogb = oracle.chamfer_bwd(A.to_host(), Bp.to_host(), ix.to_host(), iy.to_host(), 0.9, 1.1, 1.5)
    ea = oracle.sample_points_bwd(ma.get_faces_padded().astype(np.int64) - 1, ma._faces_len, ma.V, fa.to_host(), ra1.to_host(), ra2.to_host(), oga)
    eb = oracle.sample_points_bwd(mb.get_faces_padded().astype(np.int64) - 1, mb._faces_len, mb.V, fb.to_host(), rb1.to_host(), rb2.to_host(), ogb)
    for _ in range(2):
        ga, gb = fx.chamfer_sampled_grad(A, Bp, ix, iy, ma, (fa, ra1, ra2), mb, (fb, rb1, rb2), w1=0.9, w2=1.1, gout=1.5)
        assert np.array_equal(ga.to_host(), ea), np.argwhere(ga.to_host() != ea)[:5]
        assert np.array_equal(gb.to_host(), eb), np.argwhere(gb.to_host() != eb)[:5]
    ga1, none = fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=ma, draws_a=(fa, ra1, ra2), w1=0.9, w2=1.1, gout=1.5)
    assert none is None and np.array_equal(ga1.to_host(), ea)
    base = np.asfortranarray(np.random.default_rng(3).standard_normal(eb.shape).astype(np.float32))
    out = fx.gpu(base.copy(order="F"))
    fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_b=mb, draws_b=(fb, rb1, rb2), w1=0.9, w2=1.1, gout=1.5, out_b=out)
    eb2 = oracle.sample_points_bwd(mb.get_faces_padded().astype(np.int64) - 1, mb._faces_len, mb.V, fb.to_host(), rb1.to_host(), rb2.to_host(),
                                   ogb, base=base)
    assert np.array_equal(out.to_host(), eb2)
    gaa, gba = fx.chamfer_sampled_grad(A, Bp, ix, iy, ma, (fa, ra1, ra2), mb, (fb, rb1, rb2), w1=0.9, w2=1.1, gout=1.5, ordered=False)
    assert np.allclose(gaa.to_host(), ea, rtol=2e-4, atol=1e-8) and np.allclose(gba.to_host(), eb, rtol=2e-4, atol=1e-8)


@pytest.mark.parametrize("nb", [1, 3])
def test_chamfer_sampled_adjoint_with_the_optimiser_step_in_its_launch(gpu_fx, oracle, nb):
    """fx3d_chamfer_sampled_bwd_step = fx3d_chamfer_sampled_bwd (ordered, on top of a base gradient) followed by
    fx3d_momentum_step_offset, bit for bit: velocity, parameters, the next offset mesh, the gradient and the seed counter --
    one mesh, and a batch of meshes of equal vertex counts (padded == packed bytes)."""
    fx = gpu_fx
    src = fx.gpu(fx.load_trimesh(*[os.path.join(GOLDEN, "sphere.obj")] * nb))
    tgt = fx.gpu(fx.load_trimesh(*[os.path.join(GOLDEN, "teapot.obj")] * nb))
    n = 5000
    A, fa, r1, r2 = fx.sample_points(src, n, seed=1, return_draws=True)
    Bp = fx.sample_points(tgt, n, seed=2)
    _, ix, iy = fx.chamfer_distance(A, Bp, return_indices=True)
    rng = np.random.default_rng(0)
    V = src.V
    base_g = np.asfortranarray(rng.standard_normal((3, V, nb)).astype(np.float32) * 1e-3)
    vel0 = np.asfortranarray(rng.standard_normal((3, V * nb)).astype(np.float32) * 1e-3)
    x0 = np.asfortranarray(rng.standard_normal((3, V * nb)).astype(np.float32) * 1e-2)
    basev = src.dev("verts_packed")

    def state():
        return (fx.gpu(base_g.copy(order="F")), fx.gpu(vel0.copy(order="F")), fx.gpu(x0.copy(order="F")),
                fx.DeviceArray.zeros((3, V * nb), np.float32), fx.DeviceArray.zeros((1,), np.uint64))
    g_a, vel_a, x_a, out_a, ctr_a = state()
    fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=src, draws_a=(fa, r1, r2), out_a=g_a)
    opt = fx.Momentum(0.7, 0.9)
    opt.v = vel_a
    opt.update_offset(x_a, g_a.reshape(3, V * nb), basev, out_a, ctr_a, 2)
    g_b, vel_b, x_b, out_b, ctr_b = state()
    fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=src, draws_a=(fa, r1, r2), out_a=g_b,
                            step=(0.9, 0.7, vel_b, x_b, basev, out_b, ctr_b, 2))
    for a, b in ((g_a, g_b), (vel_a, vel_b), (x_a, x_b), (out_a, out_b)):
        assert np.array_equal(a.to_host(), b.to_host())
    assert int(ctr_a.to_host()[0]) == int(ctr_b.to_host()[0]) == 2
    assert np.abs(vel_b.to_host() - vel0).max() > 0


@pytest.mark.parametrize("nb,w_lap", [(1, 0.1), (3, 0.1), (1, 0.0)])
def test_regularisers_as_passengers_of_the_sampling_launches(gpu_fx, nb, w_lap):
    """fx3d_mesh_reg (examples/fit_mesh.jl:80-83's 0.1 laplacian_loss + edge_loss): the forward as extra blocks of the draw launch
    (fx3d_sample_points_draw_pair_reg), the adjoint + the tutorial's sum as extra blocks of the launch of the chamfer adjoint's rows
    (fx3d_chamfer_sampled_bwd_step_reg) = fx3d_mesh_losses, fx3d_mesh_losses_bwd, fx3d_chamfer_sampled_bwd_step one after the other,
    bit for bit: samples, both losses, the sum, the gradient, the optimiser's state.  A weight of zero takes launches of its own."""
    fx = gpu_fx
    from flux3d_jl_amd.metrics import MeshReg, _chamfer_points
    from flux3d_jl_amd.transforms import sample_points_pair
    n, w_edge = 5000, 1.0
    rng = np.random.default_rng(5)

    def meshes():
        src = fx.gpu(fx.load_trimesh(*[os.path.join(GOLDEN, "sphere.obj")] * nb))
        tgt = fx.gpu(fx.load_trimesh(*[os.path.join(GOLDEN, "teapot.obj")] * nb))
        return src, tgt
    src, tgt = meshes()
    V = src.V
    vel0 = np.asfortranarray(rng.standard_normal((3, V * nb)).astype(np.float32) * 1e-3)
    x0 = np.asfortranarray(rng.standard_normal((3, V * nb)).astype(np.float32) * 1e-2)

    def state():
        return (fx.gpu(vel0.copy(order="F")), fx.gpu(x0.copy(order="F")), fx.DeviceArray.zeros((3, V * nb), np.float32),
                fx.DeviceArray.zeros((1,), np.uint64))
    # one after the other
    A, Bp, fa, r1, r2 = sample_points_pair(src, tgt, n, seed_a=21, seed_b=22, return_draws_a=True)
    loss1, ix, iy = _chamfer_points(A, Bp, 1.0, 1.0, return_indices=True, sync=False)
    lap, edge, total = fx.mesh_losses(src, 0.0, w_lap, w_edge, base=loss1, sync=False)
    g_a = fx.mesh_losses_grad(src, 0.0, w_lap, w_edge, reuse_forward=True)
    vel_a, x_a, out_a, ctr_a = state()
    fx.chamfer_sampled_grad(A, Bp, ix, iy, mesh_a=src, draws_a=(fa, r1, r2), out_a=g_a.reshape(3, V, nb),
                            step=(0.9, 0.7, vel_a, x_a, src.dev("verts_packed"), out_a, ctr_a, 2))
    # as passengers
    src2, tgt2 = meshes()
    reg = MeshReg(src2, 0.0, w_lap, w_edge)
    A2, Bp2, fa2, r12, r22 = sample_points_pair(src2, tgt2, n, seed_a=21, seed_b=22, return_draws_a=True, reg=reg)
    loss2, ix2, iy2 = _chamfer_points(A2, Bp2, 1.0, 1.0, return_indices=True, sync=False)
    reg.set_base(loss2)
    g_b = fx.gpu(np.asfortranarray(rng.standard_normal((3, V * nb)).astype(np.float32)))  # (overwritten, not added to)
    vel_b, x_b, out_b, ctr_b = state()
    fx.chamfer_sampled_grad(A2, Bp2, ix2, iy2, mesh_a=src2, draws_a=(fa2, r12, r22), out_a=g_b.reshape(3, V, nb),
                            step=(0.9, 0.7, vel_b, x_b, src2.dev("verts_packed"), out_b, ctr_b, 2), reg=reg)
    for a, b in ((A, A2), (Bp, Bp2), (lap, reg.lap), (edge, reg.edge), (total, reg.total), (g_a, g_b), (vel_a, vel_b), (x_a, x_b), (out_a, out_b)):
        assert np.array_equal(a.to_host(), b.to_host())
    assert int(ctr_b.to_host()[0]) == 2 and np.isfinite(reg.total.to_host()).all()


def test_fit_step_graph_with_and_without_the_folded_regularisers(gpu_fx):
    """FitStepGraph(fold=True) -- five launches per iteration -- walks the trajectory of fold=False -- seven -- bit for bit (the ordered
    adjoint has no float atomics: every iteration's loss and the final offsets are equal, not close)."""
    fx = gpu_fx
    tv, tf = fx.load_obj(os.path.join(GOLDEN, "teapot.obj"))
    tv = tv - tv.mean(1, keepdims=True)
    tv = np.asfortranarray((tv / np.abs(tv).max()).astype(np.float32))
    runs = []
    for fold in (False, True):
        src, tgt = fx.gpu(fx.load_trimesh(os.path.join(GOLDEN, "sphere.obj"))), fx.gpu(fx.TriMesh([tv], [tf]))
        x = fx.DeviceArray.zeros((3, src.V), np.float32)
        step = fx.FitStepGraph(x, src, tgt, fx.Momentum(1.0, 0.9), num_samples=5000, seed=77, fold=fold)
        losses = [float(step.first_loss.item())]
        for _ in range(25):
            step.step()
            step.synchronize()
            losses.append(float(step.loss.item()))
        runs.append((losses, x.to_host().copy()))
    assert runs[0][0] == runs[1][0], (runs[0][0][:4], runs[1][0][:4])
    assert np.array_equal(runs[0][1], runs[1][1])
    assert runs[1][0][-1] < runs[1][0][0]
