"""sample_points and face areas (src/transforms/mesh_func.jl:21-82, src/rep/mesh.jl:765-836)."""
import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, current_stream, workspace

EPS = 1e-6  # src/transforms/utils.jl:4

_seed_counter = [0x5EED5A4D]


def compute_faces_areas_packed(m):
    """compute_faces_areas_packed (src/rep/mesh.jl:765-780): device (sumF,) Float32."""
    verts, faces = m.dev("verts_packed"), m.dev("faces_packed")
    F = faces.shape[1]
    out = DeviceArray.empty((F,), np.float32)
    _lib.call("fx3d_faces_areas_packed", verts.ptr, verts.shape[1], faces.ptr, F, out.ptr,
              current_stream().handle)
    return out


def compute_faces_areas_padded(m):
    """compute_faces_areas_padded (src/rep/mesh.jl:799-808): device (1,Fmax,B), zero padded."""
    verts, faces = m.dev("verts_padded") if not m.on_device else m.get_verts_padded(), m.dev("faces_padded")
    out = DeviceArray.empty((1, m.F, m.N), np.float32)
    _lib.call("fx3d_faces_areas_padded", verts.ptr, m.V, faces.ptr, m.F, m.dev("faces_len").ptr, m.N,
              out.ptr, current_stream().handle)
    return out


def compute_faces_areas_list(m):
    """compute_faces_areas_list (src/rep/mesh.jl:826-836): list of host (1,F_i) arrays."""
    a = compute_faces_areas_packed(m).to_host()
    out, cur = [], 0
    for n in m._faces_len:
        out.append(np.asfortranarray(a[cur:cur + n].reshape(1, -1)))
        cur += int(n)
    return out


def _verts_padded_dev(m):
    return m.get_verts_padded() if m.on_device else m.dev("verts_padded")


def _face_cdf(m, verts, faces, eps, reuse=True):
    """The mesh's sampling CDF (areas -> Float64 probabilities -> prefix sums, src/transforms/mesh_func.jl:27-39)
    on the device.  It depends only on the vertices, so it is kept with the mesh's device vertex mirrors:
    computed once for a mesh that is sampled again and again (the target of a fitting loop), dropped with them
    when the vertices are replaced (set_verts_packed; offset / with_verts_packed start from empty mirrors)."""
    key = ("face_cdf", float(eps))
    ws = m._dev.get(key) if (m.on_device and reuse) else None
    if ws is None:
        nb = C.c_size_t(0)
        _lib.call("fx3d_sample_points_workspace_bytes", m.F, m.N, C.byref(nb))
        ws = DeviceArray.empty((nb.value,), np.uint8)
        _lib.call("fx3d_sample_points_cdf", verts.ptr, m.V, faces.ptr, m.F, m.dev("faces_len").ptr, m.N, float(eps),
                  ws.ptr, ws.nbytes, current_stream().handle)
        if m.on_device:
            m._dev[key] = ws
    return ws


def sample_points(m, num_samples=5000, eps=EPS, seed=None, return_draws=False,
                  face_idx=None, r1=None, r2=None, seed_dev=None, reuse_cdf=True):
    """sample_points(m::TriMesh, num_samples=5000; eps) (src/transforms/mesh_func.jl:21-58).

    Returns a device ``(3, num_samples, B)`` Float32 array (the mesh's storage type in the
    reference, ``::S{T,3}``).  Draws come from the device Philox stream keyed by ``seed`` (a fresh
    seed per call when None, like the reference's global RNG); or pass explicit ``face_idx`` (n,B)
    0-based mesh-local, ``r1``, ``r2`` (n,B) to reproduce `_sample_points` for given draws.
    ``return_draws=True`` also returns (face_idx, r1, r2) device arrays for the adjoint.
    ``seed_dev``: optional device uint64 added to ``seed`` by the kernel (a captured graph advances it between
    replays, see fit.FitStepGraph).  ``reuse_cdf=False``: recompute areas -> probabilities -> CDF on this call even
    if the mesh object still holds them from an earlier one (what the reference does on every call; same result)."""
    verts = _verts_padded_dev(m)
    faces = m.dev("faces_padded")
    n, B = int(num_samples), m.N
    out = DeviceArray.empty((3, n, B), np.float32)
    st = current_stream().handle
    if face_idx is not None:
        fi = face_idx if isinstance(face_idx, DeviceArray) else DeviceArray.from_host(np.asarray(face_idx, np.int32))
        a = r1 if isinstance(r1, DeviceArray) else DeviceArray.from_host(np.asarray(r1, np.float32))
        b = r2 if isinstance(r2, DeviceArray) else DeviceArray.from_host(np.asarray(r2, np.float32))
        _lib.call("fx3d_sample_points_explicit", verts.ptr, m.V, faces.ptr, m.F, B, n, fi.ptr, a.ptr,
                  b.ptr, out.ptr, st)
        return (out, fi, a, b) if return_draws else out
    if seed is None:
        _seed_counter[0] = (_seed_counter[0] * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        seed = _seed_counter[0]
    ws = _face_cdf(m, verts, faces, eps, reuse=reuse_cdf)
    fo = DeviceArray.empty((n, B), np.int32) if return_draws else None
    a = DeviceArray.empty((n, B), np.float32) if return_draws else None
    b = DeviceArray.empty((n, B), np.float32) if return_draws else None
    _lib.call("fx3d_sample_points_draw", verts.ptr, m.V, faces.ptr, m.F, m.dev("faces_len").ptr, B, n,
              int(seed) & ((1 << 64) - 1), seed_dev.ptr if seed_dev is not None else None, ws.ptr, ws.nbytes, out.ptr,
              fo.ptr if fo else None, a.ptr if a else None, b.ptr if b else None, st)
    return (out, fo, a, b) if return_draws else out


def sample_points_pair(ma, mb, num_samples=5000, eps=EPS, seed_a=None, seed_b=None, reuse_cdf=True, seed_dev=None,
                       return_draws_a=False, reg=None):
    """``(sample_points(ma, n; seed_a), sample_points(mb, n; seed_b))`` -- what chamfer_distance(m1, m2, n) draws
    (src/metrics/mesh.jl:41-42) -- with both CDF builds in one launch and both draws in one launch
    (fx3d_sample_points_cdf_pair / _draw_pair): identical results, two launch-bound kernels less per evaluation.
    ``return_draws_a``: also (face_idx, r1, r2) of the first mesh's draws (the fitting loop's adjoint needs them).
    ``reg``: a :class:`flux3d_hip.metrics.MeshReg` of ``ma`` -- the forward of its two regularisers rides in the draw launch."""
    if seed_a is None or seed_b is None:
        _seed_counter[0] = (_seed_counter[0] * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        seed_a = _seed_counter[0] if seed_a is None else seed_a
        seed_b = (_seed_counter[0] + 1) % (1 << 64) if seed_b is None else seed_b
    n = int(num_samples)
    st = current_stream().handle
    sides = []
    for m in (ma, mb):
        verts, faces = _verts_padded_dev(m), m.dev("faces_padded")
        key = ("face_cdf", float(eps))
        ws = m._dev.get(key) if (m.on_device and reuse_cdf) else None
        sides.append([m, verts, faces, ws, key])
    need = [sd for sd in sides if sd[3] is None]
    for sd in need:
        nb = C.c_size_t(0)
        _lib.call("fx3d_sample_points_workspace_bytes", sd[0].F, sd[0].N, C.byref(nb))
        sd[3] = DeviceArray.empty((nb.value,), np.uint8)
    if len(need) == 2:
        (m0, v0, f0, w0, _), (m1, v1, f1, w1, _) = need
        _lib.call("fx3d_sample_points_cdf_pair", v0.ptr, m0.V, f0.ptr, m0.F, m0.dev("faces_len").ptr, m0.N, w0.ptr, w0.nbytes,
                  v1.ptr, m1.V, f1.ptr, m1.F, m1.dev("faces_len").ptr, m1.N, w1.ptr, w1.nbytes, float(eps), st)
    elif len(need) == 1:
        m0, v0, f0, w0, _ = need[0]
        _lib.call("fx3d_sample_points_cdf", v0.ptr, m0.V, f0.ptr, m0.F, m0.dev("faces_len").ptr, m0.N, float(eps), w0.ptr, w0.nbytes, st)
    for sd in need:
        if sd[0].on_device:
            sd[0]._dev[sd[4]] = sd[3]
    outs = [DeviceArray.empty((3, n, sd[0].N), np.float32) for sd in sides]
    (m0, v0, f0, w0, _), (m1, v1, f1, w1, _) = sides
    mask = (1 << 64) - 1
    fo = DeviceArray.empty((n, m0.N), np.int32) if return_draws_a else None
    ra = DeviceArray.empty((n, m0.N), np.float32) if return_draws_a else None
    rb = DeviceArray.empty((n, m0.N), np.float32) if return_draws_a else None
    _lib.call("fx3d_sample_points_draw_pair_reg" if reg is not None else "fx3d_sample_points_draw_pair",
              v0.ptr, m0.V, f0.ptr, m0.F, m0.dev("faces_len").ptr, m0.N, n, int(seed_a) & mask,
              w0.ptr, w0.nbytes, outs[0].ptr, fo.ptr if fo else None, ra.ptr if ra else None, rb.ptr if rb else None,
              v1.ptr, m1.V, f1.ptr, m1.F, m1.dev("faces_len").ptr, m1.N, n, int(seed_b) & mask, w1.ptr, w1.nbytes, outs[1].ptr,
              None, None, None, seed_dev.ptr if seed_dev is not None else None, *([reg.ptr] if reg is not None else []), st)
    return (outs[0], outs[1], fo, ra, rb) if return_draws_a else (outs[0], outs[1])


def sample_points_grad(m, face_idx, r1, r2, gout, out=None, ordered=True):
    """Adjoint of sample_points w.r.t. the padded verts for fixed draws: device (3,Vmax,B).
    ``out``: add into this (3,Vmax,B) array instead of starting from zero (no memset node).
    ``ordered`` (default): the atomic-free form -- every vertex's sum in a fixed order, bit-identical to the oracle's adjoint
    and from run to run (meshes whose draws fit one CU's LDS: up to ~5300 draws at 5120 faces; larger ones, or ``ordered=False``:
    float atomics)."""
    n, B = face_idx.shape
    g = DeviceArray.empty((3, m.V, m.N), np.float32) if out is None else out
    gout = gout if isinstance(gout, DeviceArray) else DeviceArray.from_host(np.asarray(gout, np.float32))
    fits = C.c_int32(0)
    _lib.call("fx3d_sample_points_bwd_ordered", m.F, n, C.byref(fits))
    use = bool(ordered) and fits.value != 0
    _lib.call("fx3d_sample_points_bwd", m.dev("faces_padded").ptr, m.V, m.F, B, n, face_idx.ptr,
              r1.ptr, r2.ptr, gout.ptr, g.ptr, int(out is not None), m.dev("vf_rowptr").ptr if use else None,
              m.dev("vf_ent").ptr if use else None, current_stream().handle)
    return g


def lincomb(a, x, b, y, c=0.0, z=None, out=None):
    """out = a*x + b*y (+ c*z): Float32 device arrays of equal size (fx3d_lincomb)."""
    out = out if out is not None else DeviceArray.empty(x.shape, np.float32)
    _lib.call("fx3d_lincomb", x.size, float(a), x.ptr, float(b), y.ptr, float(c), z.ptr if z is not None else None,
              out.ptr, current_stream().handle)
    return out


def offset(m, offset_verts_packed):
    """offset(m::TriMesh, offset_verts_packed) (src/transforms/mesh_func.jl:435-438): a new mesh whose
    packed vertices are `verts + offset`; stays on the device, topology caches are shared."""
    verts = m.dev("verts_packed")
    off = offset_verts_packed if isinstance(offset_verts_packed, DeviceArray) else \
        DeviceArray.from_host(np.asarray(offset_verts_packed, np.float32))
    if off.shape != verts.shape:
        raise ValueError("mesh and offset_verts size mismatch")
    return m.with_verts_packed(lincomb(1.0, verts, 1.0, off))
