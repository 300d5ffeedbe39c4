"""Geometric metrics: chamfer_distance, laplacian_loss, edge_loss (+ their adjoints).

Mirror of src/metrics/pcloud.jl and src/metrics/mesh.jl over the HIP C ABI.  Every function
accepts host data (numpy / host PointCloud / host TriMesh: uploaded, computed on the GPU, scalar
returned) or device data (:class:`DeviceArray`-backed): there is no CPU code path.
"""
import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, current_stream, is_device, workspace
from .rep import PointCloud, TriMesh


def _as_dev_points(a):
    """Float32 (D,N,B) device array from PointCloud / ndarray / DeviceArray (rank-2 lifted to
    (D,N,1), src/metrics/pcloud.jl:28-37; non-Float32 cast, :14-19)."""
    if isinstance(a, PointCloud):
        a = a.points
    if is_device(a):
        if a.dtype != np.float32:
            raise TypeError("device point arrays must be Float32")
        return a.reshape(a.shape + (1,)) if a.ndim == 2 else a
    a = np.asfortranarray(np.asarray(a, dtype=np.float32))
    if a.ndim == 2:
        a = a.reshape(a.shape[0], a.shape[1], 1, order="F")
    if a.ndim != 3:
        raise ValueError("points must be (D,N) or (D,N,B)")
    return DeviceArray.from_host(a)


def _check_pair(x, y):
    D, N, B = x.shape
    D2, M, B2 = y.shape
    if D != D2:
        raise ValueError(f"DimensionMismatch: point dimensionality differs ({D} vs {D2})")
    if B != B2:
        raise ValueError(f"DimensionMismatch: batch sizes differ ({B} vs {B2})")
    return D, N, M, B


def nearest_neighbors(x, y, return_dist=False):
    """_nearest_neighbors(x, y) (src/metrics/pcloud.jl:54-86): for every point of x the index of
    its nearest point in y (same batch element) and vice versa.  Returns device int32 arrays
    (N,B) and (M,B), 0-based (the reference returns CartesianIndex(j, b), 1-based)."""
    x, y = _as_dev_points(x), _as_dev_points(y)
    D, N, M, B = _check_pair(x, y)
    ix = DeviceArray.empty((N, B), np.int32)
    iy = DeviceArray.empty((M, B), np.int32)
    dx = DeviceArray.empty((N, B), np.float32) if return_dist else None
    dy = DeviceArray.empty((M, B), np.float32) if return_dist else None
    _lib.call("fx3d_nn1", x.ptr, N, y.ptr, M, B, D, ix.ptr, iy.ptr, dx.ptr if dx else None,
              dy.ptr if dy else None, current_stream().handle)
    return (ix, iy, dx, dy) if return_dist else (ix, iy)


def chamfer_workspace(N, M, B, D):
    n = C.c_size_t(0)
    _lib.call("fx3d_chamfer_workspace_bytes", N, M, B, D, C.byref(n))
    return workspace(n.value, "chamfer")


def _chamfer_points(x, y, w1, w2, return_indices=False, loss_out=None, sync=True):
    x, y = _as_dev_points(x), _as_dev_points(y)
    D, N, M, B = _check_pair(x, y)
    ws = chamfer_workspace(N, M, B, D)
    loss_dev = loss_out if loss_out is not None else DeviceArray.empty((1,), np.float32)
    ix = DeviceArray.empty((N, B), np.int32) if return_indices else None
    iy = DeviceArray.empty((M, B), np.int32) if return_indices else None
    host = C.c_float(0)
    _lib.call("fx3d_chamfer_fwd", x.ptr, N, y.ptr, M, B, D, float(w1), float(w2), loss_dev.ptr,
              C.byref(host) if sync else None, ix.ptr if ix else None, iy.ptr if iy else None,
              ws.ptr, ws.nbytes, current_stream().handle)
    loss = np.float32(host.value) if sync else loss_dev
    return (loss, ix, iy) if return_indices else loss


def chamfer_distance(A, B, num_samples=5000, w1=1.0, w2=1.0, return_indices=False, seed=None,
                     loss_out=None, sync=True, reuse_cdf=True):
    """chamfer_distance(A, B; w1, w2) for PointCloud / arrays (src/metrics/pcloud.jl:11-26) and
    chamfer_distance(m1::TriMesh, m2::TriMesh, num_samples=5000; w1, w2) (src/metrics/mesh.jl:34-44).

    Returns the Float32 loss (host scalar; with ``sync=False`` the 1-element device array).
    ``return_indices=True`` also returns the nearest-neighbour index arrays (device)."""
    if isinstance(A, TriMesh) or isinstance(B, TriMesh):
        if not (isinstance(A, TriMesh) and isinstance(B, TriMesh)):
            raise TypeError("chamfer_distance: both arguments must be TriMesh")
        from .transforms import sample_points_pair
        s1 = None if seed is None else seed
        s2 = None if seed is None else seed + 1
        PA, PB = sample_points_pair(A, B, num_samples, seed_a=s1, seed_b=s2, reuse_cdf=reuse_cdf)  # one CDF launch, one draw launch
        return _chamfer_points(PA, PB, w1, w2, return_indices, loss_out, sync)
    return _chamfer_points(A, B, w1, w2, return_indices, loss_out, sync)


def chamfer_loss_pairwise_f32(A, B, idx_a, idx_b, w1=1.0, w2=1.0, sync=True):
    """The chamfer loss in the reference's own arithmetic -- Float32 pairwise `mean` (Base, blocks of 1024) over the
    materialised squared differences, src/metrics/pcloud.jl:47-50 -- from the forward's indices
    (``chamfer_distance(..., return_indices=True)``).  :func:`chamfer_distance` itself sums in Float64 (within 1e-6 of
    this); use this one where the reference's last bit matters."""
    x, y = _as_dev_points(A), _as_dev_points(B)
    D, N, M, Bn = _check_pair(x, y)
    n = C.c_size_t(0)
    _lib.call("fx3d_chamfer_pairwise_workspace_bytes", N, M, Bn, D, C.byref(n))
    ws = workspace(n.value, "chamfer_pairwise")
    loss_dev = DeviceArray.empty((1,), np.float32)
    host = C.c_float(0)
    _lib.call("fx3d_chamfer_loss_pairwise_f32", x.ptr, N, y.ptr, M, Bn, D, idx_a.ptr, idx_b.ptr, float(w1), float(w2),
              loss_dev.ptr, C.byref(host) if sync else None, ws.ptr, ws.nbytes, current_stream().handle)
    return np.float32(host.value) if sync else loss_dev


def chamfer_distance_grad(A, B, idx_a, idx_b, w1=1.0, w2=1.0, gout=1.0, B_global=None):
    """Adjoint of _chamfer_distance w.r.t. both clouds with the indices held constant
    (Zygote through src/metrics/pcloud.jl:47-48; `@ignore` at :45).  Returns device (D,N,B), (D,M,B)."""
    x, y = _as_dev_points(A), _as_dev_points(B)
    D, N, M, Bn = _check_pair(x, y)
    gx = DeviceArray.empty(x.shape, np.float32)
    gy = DeviceArray.empty(y.shape, np.float32)
    _lib.call("fx3d_chamfer_bwd", x.ptr, N, y.ptr, M, Bn, D, idx_a.ptr, idx_b.ptr, float(w1),
              float(w2), float(gout), int(B_global or Bn), gx.ptr, gy.ptr, current_stream().handle)
    return gx, gy


def chamfer_value_and_grad(A, B, w1=1.0, w2=1.0, gout=1.0, B_global=None, return_indices=False, loss_out=None, sync=True,
                           out=None):
    """``loss, (gA, gB) = withgradient(chamfer_distance, A, B)`` in ONE ABI call (fx3d_chamfer_fwd_bwd): the forward with its
    nearest-neighbour indices and the adjoint are queued back to back, the indices stay in scratch (benchmarks/metrics.jl:24-38
    times exactly this as "total"; examples/fit_mesh.jl:106-110).  ``out``: (gA, gB) device arrays to overwrite.  Returns
    (loss, gA, gB) (+ idx_a, idx_b with ``return_indices``); loss is a host Float32, or the 1-element device array with
    ``sync=False``."""
    x, y = _as_dev_points(A), _as_dev_points(B)
    D, N, M, Bn = _check_pair(x, y)
    n = C.c_size_t(0)
    _lib.call("fx3d_chamfer_fwd_bwd_workspace_bytes", N, M, Bn, D, C.byref(n))
    ws = workspace(n.value, "chamfer_fwd_bwd")
    loss_dev = loss_out if loss_out is not None else DeviceArray.empty((1,), np.float32)
    gx, gy = out if out is not None else (DeviceArray.empty(x.shape, np.float32), DeviceArray.empty(y.shape, np.float32))
    ix = DeviceArray.empty((N, Bn), np.int32) if return_indices else None
    iy = DeviceArray.empty((M, Bn), np.int32) if return_indices else None
    host = C.c_float(0)
    _lib.call("fx3d_chamfer_fwd_bwd", x.ptr, N, y.ptr, M, Bn, D, float(w1), float(w2), float(gout), int(B_global or Bn),
              loss_dev.ptr, C.byref(host) if sync else None, gx.ptr, gy.ptr, ix.ptr if ix else None, iy.ptr if iy else None,
              ws.ptr, ws.nbytes, current_stream().handle)
    loss = np.float32(host.value) if sync else loss_dev
    return (loss, gx, gy, ix, iy) if return_indices else (loss, gx, gy)


def sampling_adjoint_is_ordered(m, n):
    """Whether the ordered (atomic-free, bit-reproducible) sampling adjoint takes meshes of this shape with ``n`` draws each
    (fx3d_sample_points_bwd_ordered: the draws and their tables must fit one CU's LDS); otherwise the calls scatter with float atomics."""
    f = C.c_int32(0)
    _lib.call("fx3d_sample_points_bwd_ordered", int(m.F), int(n), C.byref(f))
    return f.value != 0


def chamfer_sampled_grad(A, B, idx_a, idx_b, mesh_a=None, draws_a=None, mesh_b=None, draws_b=None, w1=1.0, w2=1.0,
                         gout=1.0, B_global=None, out_a=None, out_b=None, ordered=True, step=None, reg=None):
    """Adjoint of ``chamfer_distance(m_a::TriMesh, m_b::TriMesh, n)`` (src/metrics/mesh.jl:34-44) w.r.t. the padded vertices of
    ``mesh_a`` and / or ``mesh_b`` in ONE launch (fx3d_chamfer_sampled_bwd): ``A`` / ``B`` are the sampled clouds of the forward,
    ``draws_*`` = (face_idx, r1, r2) of :func:`sample_points` (``return_draws=True``), ``idx_*`` the forward's neighbour indices.
    A mesh that is None is skipped.  ``out_*``: (3,Vmax,B) device arrays the gradient is ADDED to (default: fresh).
    ``ordered`` (default): no float atomics -- every vertex's sum in a fixed order, the same bits on every run (meshes whose draws
    fit one CU's LDS -- ~5300 draws at 5120 faces --, otherwise, or ``ordered=False``, the float-atomic scatter).
    ``step`` = (rho, eta, vel, params, base, out, counter, inc): the Momentum step + offset of the fit_mesh loop applied to
    ``mesh_a``'s finished gradient rows in the same launch (meshes of equal vertex counts, ordered form).
    ``reg`` (with ``step``): a :class:`MeshReg` of ``mesh_a`` whose forward rode in the draw launch -- its adjoint is written to
    ``out_a`` by spare blocks of the first launch and the sampling adjoint adds on top (``out_a`` is overwritten, not added to).
    Returns (g_a, g_b) (None for a skipped side)."""
    x, y = _as_dev_points(A), _as_dev_points(B)
    D, N, M, Bn = _check_pair(x, y)
    if D != 3:
        raise ValueError("chamfer_sampled_grad: sampled clouds are (3, n, B)")
    if (out_a is None) != (out_b is None) and mesh_a is not None and mesh_b is not None:
        raise ValueError("chamfer_sampled_grad: pass both out arrays or neither")
    accumulate = ((out_a is not None) or (out_b is not None)) and reg is None
    if reg is not None and step is None:
        raise ValueError("chamfer_sampled_grad(reg=...) rides with step=...")
    def fits(m, n):
        if m is None:
            return True
        f = C.c_int32(0)
        _lib.call("fx3d_sample_points_bwd_ordered", m.F, n, C.byref(f))
        return f.value != 0
    ordered = bool(ordered) and fits(mesh_a, N) and fits(mesh_b, M)
    nb = C.c_size_t(0)
    _lib.call("fx3d_chamfer_sampled_bwd_workspace_bytes", N, M, Bn, C.byref(nb))
    ws = workspace(nb.value, "chamfer_sampled_bwd") if ordered else None

    def side(m, draws, out):
        if m is None:
            return [None, 0, 0, None, None, None, None], [None, None], None
        fi, r1, r2 = draws
        g = out if out is not None else DeviceArray.empty((3, m.V, m.N), np.float32)
        vf = [m.dev("vf_rowptr").ptr, m.dev("vf_ent").ptr] if ordered else [None, None]
        return [m.dev("faces_padded").ptr, m.V, m.F, fi.ptr, r1.ptr, r2.ptr, g.ptr], vf, g
    sa, vfa, ga = side(mesh_a, draws_a, out_a)
    sb, vfb, gb = side(mesh_b, draws_b, out_b)
    if step is not None:
        if mesh_a is None or mesh_b is not None or not mesh_a.verts_aliased or not ordered:
            raise ValueError("chamfer_sampled_grad(step=...): gradient w.r.t. mesh_a only, meshes of equal vertex counts, ordered form")
        rho, eta, vel, params, base, out, counter, inc = step
        if reg is not None:
            _lib.call("fx3d_chamfer_sampled_bwd_step_reg", x.ptr, N, y.ptr, M, Bn, idx_a.ptr, idx_b.ptr, float(w1), float(w2), float(gout),
                      *sa, int(accumulate), *vfa, float(rho), float(eta), vel.ptr, params.ptr, base.ptr, out.ptr,
                      counter.ptr if counter is not None else None, int(inc), ws.ptr, ws.nbytes, reg.ptr, current_stream().handle)
            return ga, None
        _lib.call("fx3d_chamfer_sampled_bwd_step", x.ptr, N, y.ptr, M, Bn, idx_a.ptr, idx_b.ptr, float(w1), float(w2), float(gout),
                  *sa, int(accumulate), *vfa, float(rho), float(eta), vel.ptr, params.ptr, base.ptr, out.ptr,
                  counter.ptr if counter is not None else None, int(inc), ws.ptr, ws.nbytes, current_stream().handle)
        return ga, None
    _lib.call("fx3d_chamfer_sampled_bwd", x.ptr, N, y.ptr, M, Bn, idx_a.ptr, idx_b.ptr, float(w1), float(w2), float(gout),
              int(B_global or Bn), *sa, *sb, int(accumulate), *vfa, *vfb, ws.ptr if ws is not None else None,
              ws.nbytes if ws is not None else 0, current_stream().handle)
    return ga, gb


def _mesh_ws(count):
    n = C.c_size_t(0)
    _lib.call("fx3d_mesh_loss_workspace_bytes", int(count), C.byref(n))
    return workspace(n.value, "mesh")


def laplacian_loss(m, sync=True):
    """laplacian_loss(m::TriMesh) (src/metrics/mesh.jl:9-15)."""
    verts = m.dev("verts_packed")
    V = verts.shape[1]
    ws = _mesh_ws(V)
    loss_dev = DeviceArray.empty((1,), np.float32)
    host = C.c_float(0)
    _lib.call("fx3d_laplacian_loss", verts.ptr, V, m.dev("lap_rowptr").ptr, m.dev("lap_colind").ptr,
              m.dev("lap_vals").ptr, loss_dev.ptr, C.byref(host) if sync else None, ws.ptr,
              ws.nbytes, current_stream().handle)
    return np.float32(host.value) if sync else loss_dev


_LAP_BWD_TWO_PASS_FROM = 1 << 16


def laplacian_loss_grad(m, gout=1.0, out=None):
    """Adjoint of laplacian_loss w.r.t. the packed verts: device (3, sumV).  ``out``: add to this array instead."""
    verts = m.dev("verts_packed")
    V = verts.shape[1]
    if V >= _LAP_BWD_TWO_PASS_FROM:
        # large meshes: unit rows once (16 V bytes of scratch) + one gather -- the scratch-free kernel below recomputes every
        # neighbour's row (7 x the traffic: 0.55 TB/s at 2 M vertices against 3.4 for this form; same bits)
        return mesh_losses_grad(m, g_lap=gout, g_edge=0.0, out=out)
    g = DeviceArray.empty((3, V), np.float32) if out is None else out
    # the mesh's L is the Laplacian of an undirected edge list (src/rep/mesh.jl:957-1002): structurally symmetric, so the
    # atomic-free gather form applies (fx3d_laplacian_loss_bwd itself takes any CSR and scatters)
    _lib.call("fx3d_laplacian_loss_bwd_sym", verts.ptr, V, m.dev("lap_rowptr").ptr,
              m.dev("lap_colind").ptr, m.dev("lap_vals").ptr, float(gout), g.ptr, int(out is not None),
              None, current_stream().handle)
    return g


def edge_loss(m, target_length=0.0, sync=True):
    """edge_loss(m::TriMesh, target_length=0.0) (src/metrics/mesh.jl:24-32)."""
    verts = m.dev("verts_packed")
    V = verts.shape[1]
    edges = m.dev("edges")
    E = edges.shape[0]
    ws = _mesh_ws(E)
    loss_dev = DeviceArray.empty((1,), np.float32)
    host = C.c_float(0)
    _lib.call("fx3d_edge_loss", verts.ptr, V, edges.ptr, E, float(target_length), loss_dev.ptr,
              C.byref(host) if sync else None, ws.ptr, ws.nbytes, current_stream().handle)
    return np.float32(host.value) if sync else loss_dev


def edge_loss_grad(m, target_length=0.0, gout=1.0, out=None):
    """Adjoint of edge_loss w.r.t. the packed verts: device (3, sumV).  ``out``: add to this array instead."""
    verts = m.dev("verts_packed")
    V = verts.shape[1]
    edges = m.dev("edges")
    g = DeviceArray.empty((3, V), np.float32) if out is None else out
    # gather over the vertex adjacency (the Laplacian's rowptr / colind of the same edge list): one launch, no float
    # atomics, no memset node, bit-identical to the oracle (fx3d_edge_loss_bwd is the scatter for a bare edge list)
    _lib.call("fx3d_edge_loss_bwd_adj", verts.ptr, V, m.dev("lap_rowptr").ptr, m.dev("lap_colind").ptr, edges.shape[0],
              float(target_length), float(gout), g.ptr, int(out is not None), current_stream().handle)
    return g


def _mesh_fused_ws(m, V, E, must_exist=False):
    """Scratch of the fused mesh-loss pair, owned by the mesh object: the forward leaves the Laplacian's unit rows
    in it and the adjoint of the SAME mesh (same vertices) reads them back.  A host mesh has no such cache (its
    vertices are uploaded per call, the scratch comes fresh from the pool): ``must_exist`` -- the adjoint's
    ``reuse_forward`` -- is an error there instead of a read of uninitialised rows."""
    ws = m._dev.get("mesh_fused_ws") if m.on_device else None
    if ws is None:
        if must_exist:
            raise ValueError("mesh_losses_grad(reuse_forward=True) needs the forward's scratch: call mesh_losses on this very "
                             "DEVICE mesh first (gpu(m)); a host mesh keeps nothing between calls")
        n = C.c_size_t(0)
        _lib.call("fx3d_mesh_losses_workspace_bytes", int(V), int(E), C.byref(n))
        ws = DeviceArray.empty((n.value,), np.uint8)
        if m.on_device:
            m._dev["mesh_fused_ws"] = ws
    return ws


def mesh_losses(m, target_length=0.0, w_lap=0.1, w_edge=1.0, base=None, sync=True):
    """laplacian_loss(m) and edge_loss(m, target_length) (src/metrics/mesh.jl:9-32) in ONE launch, plus the fit_mesh
    objective's weighted sum ``(base + w_lap*lap) + w_edge*edge`` (examples/fit_mesh.jl:80-83; ``base``: a 1-element
    device array, e.g. the chamfer term, or None).  Returns (lap, edge, total): host Float32 when ``sync`` else
    1-element device arrays."""
    verts = m.dev("verts_packed")
    V = verts.shape[1]
    edges = m.dev("edges")
    E = edges.shape[0]
    ws = _mesh_fused_ws(m, V, E)
    out = DeviceArray.empty((3,), np.float32)
    f4 = out.dtype.itemsize
    _lib.call("fx3d_mesh_losses", verts.ptr, V, m.dev("lap_rowptr").ptr, m.dev("lap_colind").ptr, m.dev("lap_vals").ptr,
              edges.ptr, E, float(target_length), float(w_lap), float(w_edge), base.ptr if base is not None else None,
              out.ptr, out.ptr + f4, out.ptr + 2 * f4, ws.ptr, ws.nbytes, current_stream().handle)
    if sync:
        h = out.to_host()
        return np.float32(h[0]), np.float32(h[1]), np.float32(h[2])
    return out.slab(0, 1), out.slab(1, 1), out.slab(2, 1)


class MeshReg:
    """``w_lap * laplacian_loss(m) + w_edge * edge_loss(m, target_length)`` of a DEVICE mesh as passengers of the fit iteration's
    sampling launches (include/flux3d_hip.h: fx3d_mesh_reg; examples/fit_mesh.jl:80-83): hand it to
    :func:`flux3d_hip.transforms.sample_points_pair` (the forward rides in the draw launch: ``lap`` / ``edge``) and, with ``base``
    set to the chamfer term, to :func:`chamfer_sampled_grad` (the adjoint rides in the launch of the chamfer adjoint's rows:
    ``total`` = (base + w_lap lap) + w_edge edge).  The bits of :func:`mesh_losses` / :func:`mesh_losses_grad`, two launches less."""

    def __init__(self, m, target_length=0.0, w_lap=0.1, w_edge=1.0):
        if not m.on_device or not m.verts_aliased:
            raise ValueError("MeshReg: a device mesh whose padded vertices are its packed ones (one mesh, or equal vertex counts)")
        verts, edges = m.dev("verts_packed"), m.dev("edges")
        V, E = verts.shape[1], edges.shape[0]
        ws = _mesh_fused_ws(m, V, E)
        self.out = DeviceArray.empty((3,), np.float32)
        self.lap, self.edge, self.total = self.out.slab(0, 1), self.out.slab(1, 1), self.out.slab(2, 1)
        self._keep = (verts, edges, ws, m)
        self.c = _lib.MeshRegStruct(verts.ptr, V, m.dev("lap_rowptr").ptr, m.dev("lap_colind").ptr, m.dev("lap_vals").ptr, edges.ptr, E,
                                    float(target_length), float(w_lap), float(w_edge), None, self.lap.ptr, self.edge.ptr,
                                    self.total.ptr, ws.ptr, ws.nbytes)

    def set_base(self, base):
        self._base = base
        self.c.base_dev = base.ptr if base is not None else None

    @property
    def ptr(self):
        return C.addressof(self.c)


def mesh_losses_grad(m, target_length=0.0, g_lap=0.1, g_edge=1.0, out=None, reuse_forward=False):
    """``g_lap * d laplacian_loss/dv + g_edge * d edge_loss/dv`` w.r.t. the packed verts, device (3, sumV), in ONE gather
    launch without float atomics (bit-reproducible; bit-identical to the oracle's adjoints).  ``out``: add to this array.
    ``reuse_forward``: :func:`mesh_losses` ran on this very mesh object since its vertices were set."""
    verts = m.dev("verts_packed")
    V = verts.shape[1]
    E = m.dev("edges").shape[0]
    ws = _mesh_fused_ws(m, V, E, must_exist=bool(reuse_forward))
    g = DeviceArray.empty((3, V), np.float32) if out is None else out
    _lib.call("fx3d_mesh_losses_bwd", verts.ptr, V, m.dev("lap_rowptr").ptr, m.dev("lap_colind").ptr, m.dev("lap_vals").ptr,
              E, float(target_length), float(g_lap), float(g_edge), int(bool(reuse_forward)), g.ptr, int(out is not None),
              ws.ptr, ws.nbytes, current_stream().handle)
    return g
