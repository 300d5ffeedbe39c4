"""CPU-only tests of the host side: the C-ABI library loads and exports every declared symbol,
the TriMesh/PointCloud mirror reproduces the reference's layout tables, the host topology builders
agree with the oracle, and compute calls fail loudly without a GPU (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _mesh_lists(k, R=np.int64):
    vl = [np.asfortranarray(np.array(v, np.float32).T) for v in k["verts"]]
    fl = [np.asfortranarray(np.array(f, R).T) for f in k["faces"]]
    return vl, fl


def test_abi_exports_every_declared_symbol(fx):
    hdr = open(os.path.join(ROOT, "include", "flux3d_hip.h")).read()
    declared = set(re.findall(r"FX3D_API\s+[\w\s\*]+?\b(fx3d_\w+)\s*\(", hdr))
    assert len(declared) >= 40
    lib = ctypes.CDLL(fx.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    from flux3d_jl_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_version_and_last_error(fx):
    assert "gfx950" in fx.__version__
    from flux3d_jl_amd import _lib
    rc = _lib.load().fx3d_chamfer_workspace_bytes(0, 1, 1, 3, ctypes.byref(ctypes.c_size_t()))
    assert rc == -1 and "empty" in _lib.last_error()


def test_no_cpu_fallback(fx):
    """Without a device every compute entry point raises (status != 0); nothing is computed on
    the host.  (On the GPU box this test is skipped: a device is present.)"""
    if fx.functional():
        pytest.skip("GPU present")
    x = np.zeros((3, 8, 1), np.float32, order="F")
    with pytest.raises(fx.Flux3DHipError):
        fx.chamfer_distance(x, x)
    with pytest.raises(fx.Flux3DHipError):
        fx.knn(x, 2)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "flux3d.jl_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".jl")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn
                assert "flux3d_oracle" not in txt or "oracle/flux3d_oracle.c" in txt, fn


def test_pointcloud_shapes(fx):
    """src/rep/pcloud.jl:30-55: rank-2 input is lifted to (D,N,1); Float64 is cast."""
    p = fx.PointCloud(np.random.rand(3, 16))
    assert p.points.shape == (3, 16, 1) and p.points.dtype == np.float32 and p.points.flags.f_contiguous
    assert fx.npoints(p) == 16
    p2 = fx.PointCloud(np.random.rand(3, 16, 4), np.random.rand(3, 16, 4))
    assert p2.normals.shape == (3, 16, 4)
    with pytest.raises(ValueError):
        fx.PointCloud(np.random.rand(3, 16, 1), np.random.rand(3, 15, 1))
    assert p2[1].shape == (3, 16)


@pytest.mark.parametrize("R", [np.int64, np.uint32])
def test_trimesh_layouts(fx, known, R):
    """test/rep.jl:112-133: list/packed/padded verts, packed face offsets, pad value 0."""
    vl, fl = _mesh_lists(known["three_mesh_batch"], R)
    m = fx.TriMesh(vl, fl)
    assert (m.N, m.V, m.F, m.equalised) == (3, 5, 7, False)
    assert all(np.array_equal(a, b) for a, b in zip(vl, fx.get_verts_list(m)))
    assert np.array_equal(np.concatenate(vl, axis=1), fx.get_verts_packed(m))
    pad = fx.get_verts_padded(m)
    for i, v in enumerate(vl):
        assert np.array_equal(pad[:, : v.shape[1], i], v) and np.all(pad[:, v.shape[1]:, i] == 0)
    packed = fx.get_faces_packed(m)
    assert packed.dtype == R
    cur = off = 0
    for i, f in enumerate(fl):
        assert np.array_equal(packed[:, cur:cur + f.shape[1]], f + off)
        cur += f.shape[1]
        off += vl[i].shape[1]
    fp = fx.get_faces_padded(m)
    for i, f in enumerate(fl):
        assert np.array_equal(fp[:, : f.shape[1], i], f) and np.all(fp[:, f.shape[1]:, i] == 0)
    with pytest.raises(ValueError):
        fx.TriMesh(vl, fl[:2])
    with pytest.raises(ValueError):
        fx.TriMesh([vl[0]], [np.array([[1], [2], [9]], R)])


def test_converter_tables(fx, known):
    """test/rep.jl:392-489."""
    from flux3d_jl_amd import rep
    k = known["converters"]
    for T in (np.float64, np.float32, np.int64, np.uint32):
        lst = [np.asfortranarray(np.array(a, T).T) for a in k["list"]]
        packed = np.concatenate(lst, axis=1)
        padded = np.zeros((3, 4, 3), T)
        for i, a in enumerate(lst):
            padded[:, : a.shape[1], i] = a
        items_len, first, to_list = rep._auxiliary_mesh(lst)
        assert list(items_len) == k["items_len"] and list(first) == k["packed_first_idx"]
        assert np.all(to_list[:4] == 1) and np.all(to_list[4:6] == 2) and np.all(to_list[6:] == 3)
        assert np.array_equal(rep._list_to_padded(lst, 0), padded)
        assert np.array_equal(rep._list_to_packed(lst), packed)
        assert np.array_equal(rep._packed_to_padded(packed, items_len, 0), padded)
        assert all(np.array_equal(a, b) for a, b in zip(rep._packed_to_list(packed, items_len), lst))
        assert all(np.array_equal(a, b) for a, b in zip(rep._padded_to_list(padded, items_len), lst))
        assert np.array_equal(rep._padded_to_packed(padded, items_len), packed)
        assert rep._list_to_padded(lst, 0).dtype == T


def test_topology_builders_match_oracle(fx, oracle, known):
    """fx3d_build_edges_packed / fx3d_build_laplacian_csr (host C++) vs the oracle and the
    reference's identities (test/rep.jl:135-175)."""
    vl, fl = _mesh_lists(known["three_mesh_batch"])
    m = fx.TriMesh(vl, fl)
    edges = fx.get_edges_packed(m)  # 1-based (E,2)
    fp0 = fx.get_faces_packed(m).astype(np.int64) - 1
    oe, of2e = oracle.edges_packed(fp0, 12, want_f2e=True)
    assert np.array_equal(edges.astype(np.int64) - 1, oe)
    assert np.array_equal(fx.get_faces_to_edges_packed(m).astype(np.int64) - 1, of2e)
    d = fx.get_edges_to_key(m)
    for i, (a, b) in enumerate(edges):
        assert d[(int(a), int(b))] == i + 1
    rowptr, colind, vals = fx.get_laplacian_packed(m)
    orp, oci, ov = oracle.laplacian_csr(oe, 12)
    assert np.array_equal(rowptr, orp) and np.array_equal(colind, oci) and np.array_equal(vals, ov)
    # teapot + sphere batch
    mm = fx.load_trimesh(os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj"))
    e2 = fx.get_edges_packed(mm).astype(np.int64) - 1
    oe2 = oracle.edges_packed(fx.get_faces_packed(mm).astype(np.int64) - 1, 1202 + 2562)
    assert np.array_equal(e2, oe2)
    r2 = fx.get_laplacian_packed(mm)
    o2 = oracle.laplacian_csr(oe2, 1202 + 2562)
    assert all(np.array_equal(a, b) for a, b in zip(r2, o2))


def test_degenerate_faces_laplacian(fx, oracle):
    """benchmarks/metrics.jl:17-22 generate_trimesh: faces (i,i,i) -> self edges; sparse() sums the
    duplicate entries so L == 0 and laplacian_loss == 0."""
    n = 16
    v = np.asfortranarray((np.cumsum(np.ones((3, n)), axis=1) / n).astype(np.float32))
    f = np.asfortranarray(np.tile(np.arange(1, n + 1), (3, 1)).astype(np.int64))
    m = fx.TriMesh([v], [f])
    rowptr, colind, vals = fx.get_laplacian_packed(m)
    assert len(colind) == n and np.all(vals == 0.0)
    oe = oracle.edges_packed(f - 1, n)
    orp, oci, ov = oracle.laplacian_csr(oe, n)
    assert np.array_equal(rowptr, orp) and np.array_equal(vals, ov)
    assert float(oracle.laplacian_loss(v, orp, oci, ov)) == 0.0


def test_synth_generator_is_stable(fx):
    """The documented SplitMix64 stream: fixed first values, shard == slice of the global batch."""
    u = fx.synth.splitmix_uniform(fx.synth.SEED_A, 4)
    assert u.dtype == np.float32 and np.all((u >= 0) & (u < 1))
    # independent scalar restatement of the recurrence
    def sm(seed, k):
        M = (1 << 64) - 1
        z = (seed + (k + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        return np.float32((z >> 40) * 2.0 ** -24)
    assert [sm(fx.synth.SEED_A, k) for k in range(4)] == list(u)
    full = fx.synth.uniform_cloud(fx.synth.SEED_A, 3, 8, 4)
    part = fx.synth.uniform_cloud(fx.synth.SEED_A, 3, 8, 2, batch_offset=2)
    assert np.array_equal(full[:, :, 2:], part)
    p = fx.synth.reference_bench_cloud(8)
    assert np.allclose(p[:, 3], 0.5)


def test_off_loader_roundtrip(fx, tmp_path):
    """OFF (ModelNet's format, src/datasets/modelnet/base.jl:100-101) parses to the same arrays as the OBJ
    twin of the same mesh, including the glued-header variant and polygon faces."""
    v, f = fx.load_obj(os.path.join(GOLDEN, "sphere.obj"))
    lines = ["OFF", f"{v.shape[1]} {f.shape[1]} 0"]
    lines += [" ".join(repr(float(c)) for c in v[:, i]) for i in range(v.shape[1])]
    lines += ["3 " + " ".join(str(int(c) - 1) for c in f[:, j]) for j in range(f.shape[1])]
    p = tmp_path / "sphere.off"
    p.write_text("\n".join(lines) + "\n")
    v2, f2 = fx.load_off(str(p))
    assert np.array_equal(v, v2) and np.array_equal(f, f2)
    q = tmp_path / "quad.off"
    q.write_text("OFF4 1 0\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    v3, f3 = fx.load_off(str(q))
    assert v3.shape == (3, 4) and np.array_equal(f3, np.array([[1, 1], [2, 3], [3, 4]], np.uint32))
    m = fx.load_trimesh(str(p), str(q))
    assert m.N == 2 and m.V == v.shape[1]


def test_graph_capture_needs_a_created_stream(fx):
    """Graph.capture refuses the default stream before touching the device (hipGraph capture is per stream), and the
    Julia shim binds every entry point of the header."""
    g = fx.Graph()
    with pytest.raises(ValueError):
        g.capture(fx.current_stream())
    shim = open(os.path.join(ROOT, "flux3d.jl_amd", "julia", "Flux3DHip.jl")).read()
    from flux3d_jl_amd import _lib
    missing = [name for name in _lib.SIGNATURES if name not in shim]  # one @ccall per ABI entry point
    assert not missing, missing


def test_header_is_plain_c_and_links_from_c(fx, tmp_path):
    """The boundary is a C ABI: include/flux3d_hip.h compiles as pedantic C99 and examples/c_abi_example.c (clouds ->
    device -> chamfer forward, no Python, no torch) links against the shared library; without a GPU it reports that
    and exits cleanly."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "c_abi_example")
    libdir = os.path.dirname(fx.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_example.c"), "-o", exe, "-L", libdir, "-lflux3d_hip",
                    "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "flux3d_hip" in r.stdout and ("chamfer_distance(A, B)" in r.stdout or "no MI355X visible" in r.stdout)


def test_option_api_without_a_gpu(fx):
    from flux3d_jl_amd import _lib
    assert _lib.load().fx3d_option_count() == len(_lib.options()) == 12
    with _lib.option("knn_no_mfma", 1):
        assert _lib.get_option("knn_no_mfma") == 1
    assert _lib.get_option("knn_no_mfma") == 0
    # the header's list of the switches and README's are the library's table (a stale list survived one pruning)
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "flux3d_hip.h")).read()
    block = hdr[hdr.index("variant switches"):hdr.index("fx3d_option_count / fx3d_option_name enumerate them")]
    listed = set(re.findall(r"\b([a-z0-9]+(?:_[a-z0-9]+)+)\b", block)) - {"process_wide"}
    assert set(_lib.options()) <= listed, sorted(set(_lib.options()) - listed)
    assert not {n for n in listed if n.startswith(("knn_", "nn1_", "edge", "lap_", "cdf_", "mesh_", "bwd_"))} - set(_lib.options())
    #  no source comment may cite an option that no longer exists: `option <name>` / `options <name>` in csrc/
    import glob
    import re
    known = set(_lib.options())
    for path in glob.glob(os.path.join(ROOT, "flux3d.jl_amd", "csrc", "*")):
        for m in re.finditer(r"\boption\s+([a-z0-9_]+)", open(path, errors="ignore").read()):
            name = m.group(1)
            if "_" in name and name.split("_")[0] in ("knn", "nn1", "edgeconv", "lap", "cdf", "mesh", "bwd"):
                assert name in known, f"{os.path.basename(path)} cites option {name}, which the library does not have"


def test_knn_scratch_plan_without_a_gpu(fx):
    """fx3d_knn_workspace_bytes is a function of the shape, the CU count (256 when no device is visible) and the options: zero for the
    BASELINE D = 3 shape, the pre-pass slabs for C4', slice lists for few clouds with many rows, slices + flags + the interleaved copy
    for k + drop in 33 ... 128 in feature space; option knn_slices = 1 switches every slicing off."""
    import ctypes as C
    from flux3d_jl_amd import _lib
    lib = _lib.load()

    def ws(N, M, B, D, k, drop):
        nb = C.c_size_t(0)
        assert lib.fx3d_knn_workspace_bytes(N, M, B, D, k, drop, C.byref(nb)) == 0
        return nb.value

    assert ws(1024, 1024, 32, 3, 20, 1) == 0
    c4p = ws(1024, 1024, 32, 64, 20, 1)
    assert 32 * 1024 * 64 * 2 < c4p < 3 * 32 * 1024 * 64 * 2                    # the fp16 images + norms + statistics
    one = ws(8192, 8192, 1, 64, 20, 1)
    assert one > 8192 * 64 * 2 + 2 * 2 * 21 * 8192 * 4                          # images of the virtual clouds + >= 2 slices' lists
    wide = ws(1024, 1024, 32, 64, 40, 1)
    assert wide > 1024 * 64 * 32 * 4 + 2 * 2 * 32 * 1024 * 32 * 4               # interleaved copy + two slices' lists of 32
    assert ws(1024, 1023, 32, 64, 40, 1) == 0                                    # an odd cloud cannot be sliced: the wave kernels
    with _lib.option("knn_slices", 1):
        assert ws(8192, 8192, 1, 64, 20, 1) == 0                                 # M > 4096 without slices: no pre-pass either
        assert ws(1024, 1024, 32, 64, 40, 1) == 0
        assert ws(1024, 1024, 32, 64, 20, 1) == c4p


def test_bench_refuses_more_gpus_than_visible():
    """VERDICT r3 #1b: `python bench.py --gpus N` with no launcher must run N ranks or FAIL -- never print a line whose n_gpus
    differs from --gpus.  On a box with fewer than N devices (this container has none) the launcher exits non-zero before any
    JSON line exists; a WORLD_SIZE that contradicts --gpus is refused too."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "device(s) visible" in r.stderr, (r.returncode, r.stdout, r.stderr)
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=dict(env, WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "WORLD_SIZE=1" in r.stderr, (r.returncode, r.stdout, r.stderr)
    r = subprocess.run([sys.executable, bench, "--gpus", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    # (round 6) the in-process launcher -- one process driving N devices, also rank 0's fallback when the process-per-GPU bootstrap
    # fails -- obeys the same rule: too few devices, no line; under a launcher only rank 0 acts, the other ranks leave quietly
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--launcher", "inproc", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "n_gpus" not in r.stdout, (r.returncode, r.stdout, r.stderr)
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--launcher", "inproc", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="8", RANK="3", LOCAL_RANK="3"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "", (r.returncode, r.stdout, r.stderr)


def test_vertex_face_table_and_the_oracles_sampling_adjoint(fx, oracle):
    """fx3d_build_vertex_faces (host C++): a CSR over the vertices of every mesh of a padded batch whose entries face * 4 + corner
    ascend -- against numpy; and the oracle's ordered sampling adjoint (the order the device reproduces) against a float64
    scatter of the same draws (src/transforms/mesh_func.jl:64-73)."""
    import ctypes as C
    from flux3d_jl_amd import _lib
    m = fx.load_trimesh(os.path.join(GOLDEN, "teapot.obj"), os.path.join(GOLDEN, "sphere.obj"))
    fp = np.asfortranarray((m.get_faces_padded().astype(np.int64) - 1).clip(min=0).astype(np.int32))
    fl = np.ascontiguousarray(m._faces_len, dtype=np.int32)
    rowptr = np.zeros((m.V + 1, m.N), np.int32, order="F")
    ent = np.zeros((3 * m.F, m.N), np.int32, order="F")
    _lib.call("fx3d_build_vertex_faces", fp.ctypes.data, fl.ctypes.data, int(m.V), int(m.F), int(m.N), rowptr.ctypes.data, ent.ctypes.data)
    for b in range(m.N):
        assert rowptr[0, b] == 0 and rowptr[-1, b] == 3 * fl[b]
        e = ent[: 3 * fl[b], b]
        owner = np.repeat(np.arange(m.V), np.diff(rowptr[:, b]))
        assert np.array_equal(fp[e & 3, e >> 2, b], owner)               # every entry names a corner that holds its vertex
        assert np.all(np.diff(e)[np.diff(owner) == 0] > 0)                # ascending within a vertex
        assert len(np.unique(e)) == len(e)                                # every (face, corner) once
    bad = fp.copy(order="F"); bad[0, 0, 0] = m.V
    with pytest.raises(_lib.Flux3DHipError):
        _lib.call("fx3d_build_vertex_faces", bad.ctypes.data, fl.ctypes.data, int(m.V), int(m.F), int(m.N), rowptr.ctypes.data, ent.ctypes.data)
    # the oracle's adjoint vs float64
    rng = np.random.default_rng(5)
    n = 4000
    fi = np.asfortranarray(np.stack([rng.integers(0, fl[b], n) for b in range(m.N)], axis=1).astype(np.int32))
    fi[:50, 0] = 7  # a face drawn many times
    r1 = np.asfortranarray(rng.random((n, m.N), dtype=np.float32)); r2 = np.asfortranarray(rng.random((n, m.N), dtype=np.float32))
    gout = np.asfortranarray(rng.standard_normal((3, n, m.N)).astype(np.float32))
    got = oracle.sample_points_bwd(m.get_faces_padded().astype(np.int64) - 1, m._faces_len, m.V, fi, r1, r2, gout)
    exp = np.zeros((3, m.V, m.N))
    for b in range(m.N):
        u = np.sqrt(r1[:, b]).astype(np.float64); v = r2[:, b].astype(np.float64)
        w = [1 - u, u * (1 - v), u * v]
        for t in range(3):
            np.add.at(exp[:, :, b].T, fp[t, fi[:, b], b], (w[t][None, :] * gout[:, :, b]).T)
    assert np.allclose(got, exp, rtol=1e-4, atol=1e-5)
    base = np.asfortranarray(rng.standard_normal(got.shape).astype(np.float32))
    assert np.allclose(oracle.sample_points_bwd(m.get_faces_padded().astype(np.int64) - 1, m._faces_len, m.V, fi, r1, r2, gout, base=base),
                       exp + base, rtol=1e-4, atol=1e-5)
