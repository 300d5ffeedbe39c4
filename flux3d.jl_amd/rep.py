"""PointCloud / TriMesh: host-side mirror of the reference's types for the hot path.

* ``PointCloud``  -- src/rep/pcloud.jl:25-84: ``points`` is ``(D, N, B)`` Float32 (column-major,
  i.e. an F-ordered numpy array or a :class:`DeviceArray`), optional ``normals``.
* ``TriMesh``     -- src/rep/mesh.jl:70-231, 344-567, 838-1002: batched heterogeneous triangle
  meshes in list / packed ``(3, sumV)`` / padded ``(3, Vmax, B)`` form with the same lazy validity
  flags; faces are integer and stay on the host in the reference (:87-97) -- here they also get a
  cached int32 0-based device mirror because the kernels gather through them.

Faces keep the reference's 1-based numbering at this level (``index_base=1``), so fixtures copied
from the reference's tests read the same; the C ABI takes 0-based int32.
"""
import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, current_stream, is_device


def _f32(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float32))


class PointCloud:
    """PointCloud(points, normals=None) -- src/rep/pcloud.jl:25-55."""

    def __init__(self, points, normals=None):
        if isinstance(points, PointCloud):
            points, normals = points.points, points.normals
        self.points = self._lift(points)
        self.normals = None if normals is None else self._lift(normals)
        if self.normals is not None and self.normals.shape[1] != self.points.shape[1]:
            raise ValueError("number of points and normals must match in PointCloud.")

    @staticmethod
    def _lift(a):
        if is_device(a):
            if a.dtype != np.float32:
                raise TypeError("PointCloud device storage must be Float32")
            return a.reshape(a.shape + (1,)) if a.ndim == 2 else a
        a = _f32(a)
        if a.ndim == 2:  # (D,N) -> (D,N,1), src/rep/pcloud.jl:30-44
            a = a.reshape(a.shape[0], a.shape[1], 1, order="F")
        if a.ndim != 3:
            raise ValueError("points must be (D,N) or (D,N,B)")
        return a

    def __getitem__(self, index):  # p[i] -> points[:, :, i]  (:66), 0-based here
        pts = self.points.to_host() if is_device(self.points) else self.points
        return pts[:, :, index]

    @property
    def on_device(self):
        return is_device(self.points)

    def _to_device(self):
        if self.on_device:
            return self
        return PointCloud(DeviceArray.from_host(self.points),
                          None if self.normals is None else DeviceArray.from_host(self.normals))

    def _to_host(self):
        if not self.on_device:
            return self
        return PointCloud(self.points.to_host(),
                          None if self.normals is None else self.normals.to_host())

    def __repr__(self):
        D, N, B = self.points.shape
        return (f"PointCloud{{Float32}} Structure:\n    Batch size: {B}\n    Points: {N}\n"
                f"    Normals {0 if self.normals is None else self.normals.shape[1]}\n"
                f"    Storage type: {type(self.points).__name__}")


def npoints(p):
    """npoints(p::PointCloud), src/rep/pcloud.jl:84."""
    return p.points.shape[1]


# ------------------------------------------------------------------------------------ converters
def _list_to_packed(lst):
    """src/rep/utils.jl:95-101."""
    return np.asfortranarray(np.concatenate(lst, axis=1))


def _list_to_padded(lst, pad_value=0, pad_size=None):
    """src/rep/utils.jl:51-93."""
    if pad_size is None:
        pad_size = (max(x.shape[0] for x in lst), max(x.shape[1] for x in lst))
    out = np.full((pad_size[0], pad_size[1], len(lst)), pad_value, dtype=lst[0].dtype, order="F")
    for i, x in enumerate(lst):
        out[: x.shape[0], : x.shape[1], i] = x
    return out


def _packed_to_padded(packed, items_len, pad_value=0):
    """src/rep/utils.jl:119-139."""
    M = int(max(items_len))
    out = np.full((packed.shape[0], M, len(items_len)), pad_value, dtype=packed.dtype, order="F")
    cur = 0
    for i, n in enumerate(items_len):
        out[:, :n, i] = packed[:, cur:cur + n]
        cur += n
    return out


def _packed_to_list(packed, items_len):
    """src/rep/utils.jl:141-157."""
    out, cur = [], 0
    for n in items_len:
        out.append(np.asfortranarray(packed[:, cur:cur + n]))
        cur += n
    return out


def _padded_to_packed(padded, items_len):
    """src/rep/utils.jl:159-181."""
    return np.asfortranarray(np.concatenate([padded[:, :n, i] for i, n in enumerate(items_len)], axis=1))


def _padded_to_list(padded, items_len):
    """src/rep/utils.jl:183-206."""
    return [np.asfortranarray(padded[:, :n, i]) for i, n in enumerate(items_len)]


def _auxiliary_mesh(lst):
    """src/rep/utils.jl:31-49 (1-based first indices, like the reference)."""
    items_len = np.array([x.shape[1] for x in lst], dtype=np.int64)
    first = np.concatenate([[1], 1 + np.cumsum(items_len)[:-1]]).astype(np.int64)
    to_list = np.repeat(np.arange(1, len(lst) + 1), items_len).astype(np.int64)
    return items_len, first, to_list


# ---------------------------------------------------------------------------------------- TriMesh
_IDX_TYPES = {np.dtype(np.int32): 0, np.dtype(np.uint32): 1, np.dtype(np.int64): 2}


def index_upload(arr, index_base, clamp_pad=False, limit=0):
    """Host index array of one of the reference's types (UInt32 / Int64, or Int32; src/rep/mesh.jl:70-98) -> device int32
    0-based of the same (column-major) shape through fx3d_index_upload: the conversion runs on the device.  ``limit`` > 0:
    values outside [0, limit) raise (counted on the device)."""
    a = np.asfortranarray(arr)
    if a.dtype not in _IDX_TYPES:
        a = a.astype(np.int64)
    t = _IDX_TYPES[a.dtype]
    out = DeviceArray.empty(a.shape, np.int32)
    if a.size == 0:
        return out
    n = C.c_size_t(0)
    _lib.call("fx3d_index_upload_workspace_bytes", t, int(a.size), C.byref(n))
    ws = DeviceArray.empty((n.value,), np.uint8)
    bad = DeviceArray.zeros((1,), np.uint32)
    _lib.call("fx3d_index_upload", a.ctypes.data, t, int(index_base), int(a.size), int(bool(clamp_pad)), int(limit), out.ptr, bad.ptr,
              ws.ptr, ws.nbytes, current_stream().handle)
    nbad = int(bad.to_host()[0])
    if nbad:
        raise ValueError(f"index_upload: {nbad} indices outside [0, {limit}) after subtracting index_base={index_base}")
    return out


class TriMesh:
    """TriMesh(verts_list, faces_list; offset=-1) -- src/rep/mesh.jl:70-187.

    ``verts_list[i]`` is ``(3, V_i)`` float (numpy or DeviceArray), ``faces_list[i]`` is
    ``(3, F_i)`` integer, 1-based mesh-local vertex ids (``index_base=1``) like the reference.
    """

    def __init__(self, verts_list, faces_list, offset=-1, index_base=1, faces_dtype=None):
        if len(verts_list) != len(faces_list):
            raise ValueError(
                f"batch size of verts and faces should match, {len(verts_list)} != {len(faces_list)}")
        self._device = any(is_device(v) for v in verts_list)
        host_verts = [_f32(v.to_host() if is_device(v) else v) for v in verts_list]
        for v in host_verts:
            if v.ndim != 2 or v.shape[0] != 3:
                raise ValueError("each verts array must be (3, V)")
        R = faces_dtype or np.asarray(faces_list[0]).dtype
        if not np.issubdtype(R, np.integer):
            R = np.int64
        self.R = np.dtype(R)
        self.index_base = int(index_base)
        self._faces_list = [np.asfortranarray(np.asarray(f, dtype=self.R)) for f in faces_list]
        self._verts_len = np.array([v.shape[1] for v in host_verts], dtype=np.int64)
        self._faces_len = np.array([f.shape[1] for f in self._faces_list], dtype=np.int64)
        for f, nv in zip(self._faces_list, self._verts_len):
            if f.ndim != 2 or f.shape[0] != 3:
                raise ValueError("each faces array must be (3, F)")
            if f.size and (f.min() < self.index_base or f.max() >= nv + self.index_base):
                raise ValueError("face index outside the mesh's vertex range")
        self.N = len(host_verts)
        self.V = int(self._verts_len.max())
        self.F = int(self._faces_len.max())
        self.equalised = bool(np.all(self._verts_len == self.V) and np.all(self._faces_len == self.F))
        # padded (3,Vmax,B) and packed (3,sumV) vertex arrays are the same bytes (one mesh, or equal vertex counts): no copy between them
        self.verts_aliased = bool(np.all(self._verts_len == self.V))
        self.valid = self._faces_len > 0
        self.offset = int(offset)

        # verts: list is the primary host form (always valid, src/rep/mesh.jl:208-231)
        self._verts_list = host_verts
        self._verts_packed = None
        self._verts_padded = None
        self._verts_packed_valid = False
        self._verts_padded_valid = False
        self._verts_list_valid = True
        # topology caches (never invalidated: faces are immutable, :87-97).  They live in dicts SHARED by every
        # TriMesh derived from this one (gpu/cpu, offset): `offset(m, x)` is deepcopy + offset! upstream
        # (src/transforms/mesh_func.jl:435-438) and must not rebuild edges / Laplacian per optimiser step
        self._topo = {}       # faces_packed, faces_padded, edges_packed, faces_to_edges_packed, laplacian_packed
        self._topo_dev = {}   # int32 0-based device mirrors of the integer data
        # device mirrors of the vertex positions (per mesh)
        self._dev = {}
        if self._device:
            self._dev["verts_packed"] = DeviceArray.from_host(self.get_verts_packed_host())

    def _topo_prop(name):  # noqa: N805  (attribute backed by the shared topology dict)
        return property(lambda self: self._topo.get(name), lambda self, v: self._topo.__setitem__(name, v))

    _faces_packed = _topo_prop("faces_packed")
    _faces_padded = _topo_prop("faces_padded")
    _edges_packed = _topo_prop("edges_packed")
    _faces_to_edges_packed = _topo_prop("faces_to_edges_packed")
    _laplacian_packed = _topo_prop("laplacian_packed")  # (rowptr, colind, vals) host CSR, 0-based
    del _topo_prop

    # ---- verts -----------------------------------------------------------------------------
    def get_verts_list(self):
        if self._device and not self._verts_list_valid:
            self._verts_list = _packed_to_list(self._dev["verts_packed"].to_host(), self._verts_len)
            self._verts_list_valid = True
        return self._verts_list

    def get_verts_packed_host(self):
        if self._device and "verts_packed" in self._dev and not self._verts_list_valid:
            return self._dev["verts_packed"].to_host()
        if not self._verts_packed_valid:
            self._verts_packed = _list_to_packed(self.get_verts_list())
            self._verts_packed_valid = True
        return self._verts_packed

    def get_verts_padded_host(self):
        if not self._verts_padded_valid or self._device:
            self._verts_padded = _packed_to_padded(self.get_verts_packed_host(), self._verts_len, 0)
            self._verts_padded_valid = True
        return self._verts_padded

    def get_verts_packed(self):
        """get_verts_packed (src/rep/mesh.jl:344-347): (3, sumV), storage type of the mesh."""
        return self._dev["verts_packed"] if self._device else self.get_verts_packed_host()

    def get_verts_padded(self):
        """get_verts_padded (src/rep/mesh.jl:366-369): (3, Vmax, B) zero padded."""
        if not self._device:
            return self.get_verts_padded_host()
        if "verts_padded" not in self._dev:
            if self.verts_aliased:  # one mesh, or equal vertex counts: (3,sumV) and (3,V,B) are the same bytes
                self._dev["verts_padded"] = self._dev["verts_packed"].reshape(3, self.V, self.N)
            else:
                self._dev["verts_padded"] = self._packed_to_padded_dev(self._dev["verts_packed"])
        return self._dev["verts_padded"]

    def _packed_to_padded_dev(self, packed):
        """_packed_to_padded (src/rep/utils.jl:119-139) without leaving the device."""
        out = DeviceArray.empty((3, self.V, self.N), np.float32)
        _lib.call("fx3d_packed_to_padded", packed.ptr, self._verts_len.ctypes.data, self.N, self.V, out.ptr,
                  current_stream().handle)
        return out

    def padded_to_packed_dev(self, padded):
        """_padded_to_packed (src/rep/utils.jl:159-181) on the device: (3,Vmax,B) -> (3,sumV)."""
        if self.verts_aliased:
            return padded.reshape(3, self.V * self.N)
        out = DeviceArray.empty((3, int(self._verts_len.sum())), np.float32)
        _lib.call("fx3d_padded_to_packed", padded.ptr, self._verts_len.ctypes.data, self.N, self.V, out.ptr,
                  current_stream().handle)
        return out

    def with_verts_packed(self, new_packed):
        """A TriMesh with the same topology (shared caches and device mirrors) and new device vertex
        positions -- what `offset(m, x)` (deepcopy + offset!, src/transforms/mesh_func.jl:409-438) needs."""
        m = TriMesh.__new__(TriMesh)
        m.__dict__.update(self.__dict__)
        m._dev = {}  # vertex mirrors are per mesh; _topo / _topo_dev stay shared
        m._device = True
        m._dev["verts_packed"] = new_packed
        m._verts_list_valid = False
        m._verts_packed_valid = False
        m._verts_padded_valid = False
        return m

    def set_verts_packed(self, new):
        """`m._verts_packed = v` (setproperty!, src/rep/mesh.jl:208-231): replaces the vertex
        positions and invalidates the other forms.  Topology caches are kept."""
        if is_device(new):
            assert new.shape == (3, int(self._verts_len.sum())) and new.dtype == np.float32
            self._device = True
            self._dev = {"verts_packed": new}  # every vertex-derived mirror (padded form, sampling CDF) is stale
            self._verts_list_valid = False
            self._verts_packed_valid = False
            self._verts_padded_valid = False
        else:
            new = _f32(new)
            assert new.shape == (3, int(self._verts_len.sum()))
            self._verts_list = _packed_to_list(new, self._verts_len)
            self._verts_list_valid = True
            self._verts_packed, self._verts_packed_valid = new, True
            self._verts_padded_valid = False
            if self._device:
                self._dev = {"verts_packed": DeviceArray.from_host(new)}

    # ---- faces (host, reference numbering) ------------------------------------------------------
    def get_faces_list(self):
        return self._faces_list

    def get_faces_packed(self):
        """get_faces_packed (src/rep/mesh.jl:413-416, _compute_faces_packed :884-896): (3, sumF),
        ids offset by the cumulative vertex count of the preceding meshes."""
        if self._faces_packed is None:
            offs = np.concatenate([[0], np.cumsum(self._verts_len)[:-1]])
            self._faces_packed = np.asfortranarray(np.concatenate(
                [f.astype(np.int64) + o for f, o in zip(self._faces_list, offs)], axis=1).astype(self.R))
        return self._faces_packed

    def get_faces_padded(self):
        """get_faces_padded (src/rep/mesh.jl:435-438): (3, Fmax, B), pad value 0."""
        if self._faces_padded is None:
            self._faces_padded = _list_to_padded(self._faces_list, 0, (3, self.F))
        return self._faces_padded

    # ---- topology (host builders of the C ABI; cached forever like the reference) ---------------
    def _build_edges(self):
        faces = np.asfortranarray(self.get_faces_packed(), dtype=np.int64)
        F = faces.shape[1]
        V = int(self._verts_len.sum())
        f2e = np.zeros((F, 3), np.int64, order="F")
        E = C.c_int64(0)
        buf = np.zeros(6 * F, np.int64)  # capacity 3F rows; written densely as (E,2) column-major
        _lib.call("fx3d_build_edges_packed", faces.ctypes.data, F, V, self.index_base,
                  buf.ctypes.data, f2e.ctypes.data, C.byref(E))
        E = E.value
        edges = np.asfortranarray(buf[: 2 * E].reshape((E, 2), order="F")).astype(self.R)
        self._edges_packed = edges
        self._faces_to_edges_packed = f2e.astype(self.R)

    def get_edges_packed(self):
        """get_edges_packed (src/rep/mesh.jl:482-485): (E,2) sorted unique, reference numbering."""
        if self._edges_packed is None:
            self._build_edges()
        return self._edges_packed

    def get_faces_to_edges_packed(self):
        if self._faces_to_edges_packed is None:
            self._build_edges()
        return self._faces_to_edges_packed

    def get_edges_to_key(self):
        """get_edges_to_key: Dict edge tuple -> key (src/rep/mesh.jl:939-940), key in reference numbering."""
        e = self.get_edges_packed()
        return {(int(a), int(b)): i + self.index_base for i, (a, b) in enumerate(e)}

    def get_laplacian_packed(self):
        """get_laplacian_packed (src/rep/mesh.jl:559-565) as 0-based CSR (rowptr, colind, vals)."""
        if self._laplacian_packed is None:
            e = np.asfortranarray(self.get_edges_packed(), dtype=np.int64)
            E, V = e.shape[0], int(self._verts_len.sum())
            rowptr = np.zeros(V + 1, np.int32)
            colind = np.zeros(2 * E + V, np.int32)
            vals = np.zeros(2 * E + V, np.float32)
            nnz = C.c_int64(0)
            _lib.call("fx3d_build_laplacian_csr", e.ctypes.data, E, V, self.index_base,
                      rowptr.ctypes.data, colind.ctypes.data, vals.ctypes.data, C.byref(nnz))
            self._laplacian_packed = (rowptr, colind[: nnz.value].copy(), vals[: nnz.value].copy())
        return self._laplacian_packed

    def laplacian_dense(self):
        rowptr, colind, vals = self.get_laplacian_packed()
        V = len(rowptr) - 1
        L = np.zeros((V, V), np.float32)
        for i in range(V):
            L[i, colind[rowptr[i]:rowptr[i + 1]]] = vals[rowptr[i]:rowptr[i + 1]]
        return L

    # ---- device mirrors of the integer data ---------------------------------------------------------
    def dev(self, name):
        """Cached device copies: faces_packed / faces_padded / faces_len (int32 0-based), edges
        (E,2) int32 0-based, lap_rowptr / lap_colind / lap_vals, and verts_* float32."""
        store = self._dev if name.startswith("verts") else self._topo_dev
        if name in store:
            return store[name]
        b = self.index_base
        if name == "verts_packed":
            arr = DeviceArray.from_host(self.get_verts_packed_host())
        elif name == "verts_padded":
            arr = DeviceArray.from_host(self.get_verts_padded_host())
        elif name == "faces_packed":   # the reference's own arrays (UInt32 / Int64, 1-based) cross the ABI untouched: the
            arr = index_upload(self.get_faces_packed(), b, limit=int(np.sum(self._verts_len)))   # library converts on the device
        elif name == "faces_padded":    # (pad entries -- value 0 in the reference -- become 0: never dereferenced)
            arr = index_upload(self.get_faces_padded(), b, clamp_pad=True, limit=int(self.V))
        elif name == "faces_len":
            arr = DeviceArray.from_host(self._faces_len.astype(np.int32))
        elif name == "edges":
            arr = index_upload(self.get_edges_packed(), b, limit=int(np.sum(self._verts_len)))
        elif name in ("vf_rowptr", "vf_ent"):  # vertex -> (face, corner) table of the ordered sampling adjoint (padded batch)
            fp = np.asfortranarray(self.get_faces_padded().astype(np.int64) - b).astype(np.int32)
            fp[fp < 0] = 0  # (padding faces: never read, faces_len bounds the walk)
            fp = np.asfortranarray(fp)
            fl = np.ascontiguousarray(self._faces_len, dtype=np.int32)
            rowptr = np.zeros((self.V + 1, self.N), np.int32, order="F")
            ent = np.zeros((3 * self.F, self.N), np.int32, order="F")
            _lib.call("fx3d_build_vertex_faces", fp.ctypes.data, fl.ctypes.data, int(self.V), int(self.F), int(self.N),
                      rowptr.ctypes.data, ent.ctypes.data)
            store["vf_rowptr"] = DeviceArray.from_host(rowptr)
            store["vf_ent"] = DeviceArray.from_host(ent)
            return store[name]
        elif name in ("lap_rowptr", "lap_colind", "lap_vals"):
            rowptr, colind, vals = self.get_laplacian_packed()
            store["lap_rowptr"] = DeviceArray.from_host(rowptr)
            store["lap_colind"] = DeviceArray.from_host(colind)
            store["lap_vals"] = DeviceArray.from_host(vals)
            return store[name]
        else:
            raise KeyError(name)
        if name.startswith("verts") and not self._device:
            return arr  # host mesh: do not cache vertex uploads (verts may change)
        store[name] = arr
        return arr

    # ---- gpu / cpu (functor(::TriMesh) moves only the verts, src/rep/mesh.jl:189-190) ---------------
    @property
    def on_device(self):
        return self._device

    def _to_device(self):
        if self._device:
            return self
        m = TriMesh(self.get_verts_list(), self._faces_list, offset=self.offset,
                    index_base=self.index_base, faces_dtype=self.R)
        m._share_topology(self)
        m._device = True
        m._dev["verts_packed"] = DeviceArray.from_host(m.get_verts_packed_host())
        return m

    def _to_host(self):
        if not self._device:
            return self
        m = TriMesh(self.get_verts_list(), self._faces_list, offset=self.offset,
                    index_base=self.index_base, faces_dtype=self.R)
        m._share_topology(self)
        return m

    def _share_topology(self, other):
        self._topo, self._topo_dev = other._topo, other._topo_dev

    def __getitem__(self, i):
        return self.get_verts_list()[i], self._faces_list[i]

    def __repr__(self):
        return (f"TriMesh{{Float32, {self.R}, {'DeviceArray' if self._device else 'Array'}}} Structure:\n"
                f"    Batch size: {self.N}\n    Max verts: {self.V}\n    Max faces: {self.F}\n"
                f"    offset: {self.offset}\n    Storage type: {'DeviceArray' if self._device else 'Array'}")


# accessor functions with the reference's names
def get_verts_packed(m): return m.get_verts_packed()
def get_verts_padded(m): return m.get_verts_padded()
def get_verts_list(m): return m.get_verts_list()
def get_faces_packed(m): return m.get_faces_packed()
def get_faces_padded(m): return m.get_faces_padded()
def get_faces_list(m): return m.get_faces_list()
def get_edges_packed(m): return m.get_edges_packed()
def get_faces_to_edges_packed(m): return m.get_faces_to_edges_packed()
def get_edges_to_key(m): return m.get_edges_to_key()
def get_laplacian_packed(m): return m.get_laplacian_packed()


def load_obj(path):
    """Minimal Wavefront OBJ reader (``v`` and triangular ``f`` records, ``a/b/c`` index forms).
    Stands in for MeshIO behind load_trimesh (src/rep/mesh.jl:244-262) for the test assets."""
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                verts.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                ids = [int(t.split("/")[0]) for t in line.split()[1:]]
                for k in range(1, len(ids) - 1):  # fan-triangulate polygons
                    faces.append([ids[0], ids[k], ids[k + 1]])
    v = np.asfortranarray(np.array(verts, dtype=np.float32).T)
    f = np.asfortranarray(np.array(faces, dtype=np.uint32).T)
    return v, f


def load_off(path):
    """Minimal OFF reader (the ModelNet on-disk format read by load_trimesh at
    src/datasets/modelnet/base.jl:100-101): header ``OFF`` (ModelNet also ships files with the counts glued
    to it, ``OFF490 518 0``), ``nv nf ne``, nv vertex lines, nf face lines ``k i0 .. ik-1`` (0-based;
    polygons are fan-triangulated).  Returns the same (verts (3,V) Float32, faces (3,F) UInt32 1-based)."""
    with open(path) as fh:
        toks = fh.read().split()
    if not toks or not toks[0].upper().startswith("OFF"):
        raise ValueError(f"{path}: not an OFF file")
    head = toks[0][3:]
    toks = ([head] if head else []) + toks[1:]
    nv, nf = int(toks[0]), int(toks[1])
    pos = 3
    v = np.array(toks[pos:pos + 3 * nv], dtype=np.float32).reshape(nv, 3)
    pos += 3 * nv
    faces = []
    for _ in range(nf):
        k = int(toks[pos])
        ids = [int(t) + 1 for t in toks[pos + 1:pos + 1 + k]]
        pos += 1 + k
        for j in range(1, k - 1):
            faces.append([ids[0], ids[j], ids[j + 1]])
    return (np.asfortranarray(v.T), np.asfortranarray(np.array(faces, dtype=np.uint32).reshape(-1, 3).T))


def load_trimesh(*paths):
    """load_trimesh(fn...) (src/rep/mesh.jl:244-262) for .obj / .off files: one batched TriMesh."""
    flat = []
    for p in paths:
        flat.extend(p if isinstance(p, (list, tuple)) else [p])
    vs, fs = zip(*[(load_off(p) if str(p).lower().endswith(".off") else load_obj(p)) for p in flat])
    return TriMesh(list(vs), list(fs))
