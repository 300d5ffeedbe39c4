"""ctypes binding of libflux3d_hip.so (include/flux3d_hip.h).

This is the Python twin of julia/Flux3DHip.jl's ``@ccall`` layer: one thin wrapper per C-ABI entry
point, status codes turned into exceptions.  There is NO fallback: if the shared library is
missing or a call fails (e.g. no GPU), an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FX3D_HIP_LIB overrides the in-tree library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("FX3D_HIP_LIB") or os.path.join(_HERE, "lib", "libflux3d_hip.so")


class Flux3DHipError(RuntimeError):
    """Raised for any non-zero fx3d_status (mirrors the reference's `error(msg)`)."""

    def __init__(self, code, msg):
        super().__init__(f"[fx3d status {code}] {msg}")
        self.code = code


c_i32, c_i64, c_f32, c_f64, c_u64 = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint64
vp, sz = C.c_void_p, C.c_size_t

class MeshRegStruct(C.Structure):
    """include/flux3d_hip.h: fx3d_mesh_reg (the fit iteration's regularisers as passengers of its sampling launches)."""
    _fields_ = [("verts", vp), ("V", c_i64), ("rowptr", vp), ("colind", vp), ("vals", vp), ("edges", vp), ("E", c_i64),
                ("target", c_f32), ("w_lap", c_f32), ("w_edge", c_f32), ("base_dev", vp), ("loss_lap_dev", vp),
                ("loss_edge_dev", vp), ("total_dev", vp), ("ws", vp), ("ws_bytes", sz)]


# name -> argtypes  (restype is int32 status unless listed in _RESTYPES)
SIGNATURES = {
    "fx3d_version": [],
    "fx3d_last_error": [C.c_char_p, sz],
    "fx3d_set_option": [C.c_char_p, c_i32],
    "fx3d_get_option": [C.c_char_p, C.POINTER(c_i32)],
    "fx3d_option_count": [],
    "fx3d_option_name": [c_i32],
    "fx3d_device_count": [C.POINTER(c_i32)],
    "fx3d_set_device": [c_i32],
    "fx3d_get_device": [C.POINTER(c_i32)],
    "fx3d_device_name": [c_i32, C.c_char_p, sz],
    "fx3d_nn1_plan_describe": [c_i32, c_i32, c_i32, c_i32, C.c_char_p, sz],
    "fx3d_device_identity": [c_i32, C.c_char_p, sz, C.c_void_p],
    "fx3d_device_sync": [],
    "fx3d_malloc": [C.POINTER(vp), sz],
    "fx3d_free": [vp],
    "fx3d_memcpy_h2d": [vp, vp, sz, vp],
    "fx3d_memcpy_d2h": [vp, vp, sz, vp],
    "fx3d_memcpy_d2d": [vp, vp, sz, vp],
    "fx3d_memset": [vp, c_i32, sz, vp],
    "fx3d_stream_create": [C.POINTER(vp)],
    "fx3d_stream_destroy": [vp],
    "fx3d_stream_sync": [vp],
    "fx3d_graph_begin_capture": [vp],
    "fx3d_graph_end_capture": [vp, C.POINTER(vp)],
    "fx3d_graph_launch": [vp, vp],
    "fx3d_graph_destroy": [vp],
    "fx3d_counter_add": [vp, c_u64, vp],
    "fx3d_event_create": [C.POINTER(vp)],
    "fx3d_event_create_sync": [C.POINTER(vp)],
    "fx3d_event_destroy": [vp],
    "fx3d_event_record": [vp, vp],
    "fx3d_event_sync": [vp],
    "fx3d_stream_wait_event": [vp, vp],
    "fx3d_event_elapsed_ms": [vp, vp, C.POINTER(c_f32)],
    "fx3d_profile_enable": [c_i32],
    "fx3d_profile_kernel_stats": [C.c_char_p, C.POINTER(c_f64), C.POINTER(c_f64), C.POINTER(c_f64),
                                  C.POINTER(c_i64)],
    "fx3d_nn1": [vp, c_i32, vp, c_i32, c_i32, c_i32, vp, vp, vp, vp, vp],
    "fx3d_chamfer_workspace_bytes": [c_i32, c_i32, c_i32, c_i32, C.POINTER(sz)],
    "fx3d_chamfer_sums": [vp, c_i32, vp, c_i32, c_i32, c_i32, vp, vp, vp, vp, sz, vp],
    "fx3d_chamfer_finalize": [vp, c_i32, c_i32, c_i64, c_i32, c_f32, c_f32, vp, vp],
    "fx3d_chamfer_finalize_many": [vp, c_i32, c_i32, c_i32, c_i64, c_i32, c_f32, c_f32, vp, vp],
    "fx3d_chamfer_fwd": [vp, c_i32, vp, c_i32, c_i32, c_i32, c_f32, c_f32, vp, C.POINTER(c_f32),
                         vp, vp, vp, sz, vp],
    "fx3d_chamfer_pairwise_workspace_bytes": [c_i32, c_i32, c_i32, c_i32, C.POINTER(sz)],
    "fx3d_chamfer_loss_pairwise_f32": [vp, c_i32, vp, c_i32, c_i32, c_i32, vp, vp, c_f32, c_f32, vp, C.POINTER(c_f32),
                                       vp, sz, vp],
    "fx3d_chamfer_bwd": [vp, c_i32, vp, c_i32, c_i32, c_i32, vp, vp, c_f32, c_f32, c_f32, c_i64,
                         vp, vp, vp],
    "fx3d_chamfer_fwd_bwd_workspace_bytes": [c_i32, c_i32, c_i32, c_i32, C.POINTER(sz)],
    "fx3d_chamfer_fwd_bwd": [vp, c_i32, vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_i64, vp, C.POINTER(c_f32),
                             vp, vp, vp, vp, vp, sz, vp],
    "fx3d_chamfer_sampled_bwd_workspace_bytes": [c_i32, c_i32, c_i32, C.POINTER(sz)],
    "fx3d_chamfer_sampled_bwd": [vp, c_i32, vp, c_i32, c_i32, vp, vp, c_f32, c_f32, c_f32, c_i64,
                                 vp, c_i32, c_i32, vp, vp, vp, vp, vp, c_i32, c_i32, vp, vp, vp, vp, c_i32,
                                 vp, vp, vp, vp, vp, sz, vp],
    "fx3d_chamfer_sampled_bwd_step": [vp, c_i32, vp, c_i32, c_i32, vp, vp, c_f32, c_f32, c_f32, vp, c_i32, c_i32, vp, vp, vp, vp, c_i32,
                                      vp, vp, c_f32, c_f32, vp, vp, vp, vp, vp, C.c_uint64, vp, sz, vp],
    "fx3d_chamfer_sampled_bwd_step_reg": [vp, c_i32, vp, c_i32, c_i32, vp, vp, c_f32, c_f32, c_f32, vp, c_i32, c_i32, vp, vp, vp, vp, c_i32,
                                          vp, vp, c_f32, c_f32, vp, vp, vp, vp, vp, C.c_uint64, vp, sz, vp, vp],
    "fx3d_knn": [vp, c_i32, vp, c_i32, c_i32, c_i32, c_i32, c_i32, vp, vp, vp],
    "fx3d_knn_workspace_bytes": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, C.POINTER(sz)],
    "fx3d_knn_ws": [vp, c_i32, vp, c_i32, c_i32, c_i32, c_i32, c_i32, vp, vp, vp, sz, vp],
    "fx3d_knn_gather": [vp, c_i32, c_i32, c_i32, c_i32, vp, vp, vp],
    "fx3d_edge_features": [vp, c_i32, c_i32, c_i32, c_i32, vp, c_i32, vp, vp],
    "fx3d_edge_features_bwd": [vp, c_i32, c_i32, c_i32, c_i32, c_i32, vp, vp],
    "fx3d_edgeconv_graph": [vp, c_i32, c_i32, c_i32, c_i32, c_i32, vp, vp, vp],
    "fx3d_index_convert": [vp, c_i32, c_i32, c_i64, c_i32, c_i64, vp, vp, vp],
    "fx3d_index_upload_workspace_bytes": [c_i32, c_i64, C.POINTER(sz)],
    "fx3d_index_upload": [vp, c_i32, c_i32, c_i64, c_i32, c_i64, vp, vp, vp, sz, vp],
    "fx3d_faces_areas_packed": [vp, c_i64, vp, c_i64, vp, vp],
    "fx3d_faces_areas_padded": [vp, c_i32, vp, c_i32, vp, c_i32, vp, vp],
    "fx3d_sample_points_explicit": [vp, c_i32, vp, c_i32, c_i32, c_i32, vp, vp, vp, vp, vp],
    "fx3d_sample_points_workspace_bytes": [c_i32, c_i32, C.POINTER(sz)],
    "fx3d_sample_points": [vp, c_i32, vp, c_i32, vp, c_i32, c_i32, c_f64, c_u64, vp, vp, vp, vp,
                           vp, sz, vp],
    "fx3d_sample_points_cdf": [vp, c_i32, vp, c_i32, vp, c_i32, c_f64, vp, sz, vp],
    "fx3d_sample_points_draw": [vp, c_i32, vp, c_i32, vp, c_i32, c_i32, c_u64, vp, vp, sz, vp, vp, vp, vp, vp],
    "fx3d_sample_points_cdf_pair": [vp, c_i32, vp, c_i32, vp, c_i32, vp, sz, vp, c_i32, vp, c_i32, vp, c_i32, vp, sz, c_f64, vp],
    "fx3d_sample_points_draw_pair": [vp, c_i32, vp, c_i32, vp, c_i32, c_i32, c_u64, vp, sz, vp, vp, vp, vp,
                                     vp, c_i32, vp, c_i32, vp, c_i32, c_i32, c_u64, vp, sz, vp, vp, vp, vp, vp, vp],
    "fx3d_sample_points_draw_pair_reg": [vp, c_i32, vp, c_i32, vp, c_i32, c_i32, c_u64, vp, sz, vp, vp, vp, vp,
                                         vp, c_i32, vp, c_i32, vp, c_i32, c_i32, c_u64, vp, sz, vp, vp, vp, vp, vp, vp, vp],
    "fx3d_build_vertex_faces": [vp, vp, c_i32, c_i32, c_i32, vp, vp],
    "fx3d_sample_points_bwd_ordered": [c_i32, c_i32, C.POINTER(c_i32)],
    "fx3d_sample_points_bwd": [vp, c_i32, c_i32, c_i32, c_i32, vp, vp, vp, vp, vp, c_i32, vp, vp, vp],
    "fx3d_voxel_workspace_bytes": [c_i32, C.POINTER(sz)],
    "fx3d_pointcloud_to_voxel": [vp, c_i32, c_i32, c_i32, vp, vp, sz, vp],
    "fx3d_lincomb": [c_i64, c_f32, vp, c_f32, vp, c_f32, vp, vp, vp],
    "fx3d_momentum_step": [c_i64, c_f32, c_f32, vp, vp, vp, vp],
    "fx3d_momentum_step_offset": [c_i64, c_f32, c_f32, vp, vp, vp, vp, vp, vp, C.c_uint64, vp],
    "fx3d_packed_to_padded": [vp, vp, c_i32, c_i32, vp, vp],
    "fx3d_padded_to_packed": [vp, vp, c_i32, c_i32, vp, vp],
    "fx3d_mesh_loss_workspace_bytes": [c_i64, C.POINTER(sz)],
    "fx3d_edge_loss": [vp, c_i64, vp, c_i64, c_f32, vp, C.POINTER(c_f32), vp, sz, vp],
    "fx3d_edge_loss_bwd": [vp, c_i64, vp, c_i64, c_f32, c_f32, vp, c_i32, vp],
    "fx3d_edge_loss_bwd_adj": [vp, c_i64, vp, vp, c_i64, c_f32, c_f32, vp, c_i32, vp],
    "fx3d_laplacian_loss": [vp, c_i64, vp, vp, vp, vp, C.POINTER(c_f32), vp, sz, vp],
    "fx3d_laplacian_loss_bwd": [vp, c_i64, vp, vp, vp, c_f32, vp, c_i32, vp],
    "fx3d_laplacian_loss_bwd_sym": [vp, c_i64, vp, vp, vp, c_f32, vp, c_i32, vp, vp],
    "fx3d_mesh_losses_workspace_bytes": [c_i64, c_i64, C.POINTER(sz)],
    "fx3d_mesh_losses": [vp, c_i64, vp, vp, vp, vp, c_i64, c_f32, c_f32, c_f32, vp, vp, vp, vp, vp, sz, vp],
    "fx3d_mesh_losses_bwd": [vp, c_i64, vp, vp, vp, c_i64, c_f32, c_f32, c_f32, c_i32, vp, c_i32, vp, sz, vp],
    "fx3d_comm_unique_id": [vp],
    "fx3d_comm_init_rank": [C.POINTER(vp), c_i32, vp, c_i32],
    "fx3d_comm_bootstrap": [C.POINTER(vp), c_i32, c_i32, C.c_char_p],
    "fx3d_comm_exchange_id": [vp, c_i32, c_i32, C.c_char_p],
    "fx3d_comm_info": [vp, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32)],
    "fx3d_comm_destroy": [vp],
    "fx3d_comm_allreduce_sum_f64": [vp, vp, c_i64, vp],
    "fx3d_comm_allreduce_max_f64": [vp, vp, c_i64, vp],
    "fx3d_comm_init_all": [C.POINTER(vp), c_i32, vp],
    "fx3d_multi_destroy": [vp],
    "fx3d_multi_info": [vp, C.POINTER(c_i32), vp, C.POINTER(c_i32)],
    "fx3d_multi_sync": [vp],
    "fx3d_chamfer_fwd_multi": [vp, C.POINTER(vp), c_i32, C.POINTER(vp), c_i32, vp, c_i32, c_i64, c_f32, c_f32,
                               C.POINTER(c_f32), C.POINTER(vp)],
    "fx3d_chamfer_fwd_sharded": [vp, vp, c_i32, vp, c_i32, c_i32, c_i32, c_i64, c_f32, c_f32, vp, vp,
                                 C.POINTER(c_f32), vp, sz, vp],
    "fx3d_chamfer_fwd_sharded_async": [vp, vp, c_i32, vp, c_i32, c_i32, c_i32, c_i64, c_f32, c_f32, vp, vp, vp, sz,
                                       vp, vp, vp, vp],
    "fx3d_chamfer_distance_host": [vp, c_i32, vp, c_i32, c_i32, c_i32, c_f32, c_f32, C.POINTER(c_f32), vp, vp],
    "fx3d_knn_host": [vp, c_i32, vp, c_i32, c_i32, c_i32, c_i32, c_i32, vp, vp],
    "fx3d_sample_points_host": [vp, c_i32, vp, c_i32, vp, c_i32, c_i32, c_f64, c_u64, vp],
    "fx3d_edge_loss_host": [vp, c_i64, vp, c_i64, c_f32, C.POINTER(c_f32)],
    "fx3d_laplacian_loss_host": [vp, c_i64, vp, vp, vp, C.POINTER(c_f32)],
    "fx3d_build_edges_packed": [vp, c_i64, c_i64, c_i32, vp, vp, C.POINTER(c_i64)],
    "fx3d_build_laplacian_csr": [vp, c_i64, c_i64, c_i32, vp, vp, vp, C.POINTER(c_i64)],
}
_RESTYPES = {"fx3d_version": C.c_char_p, "fx3d_last_error": sz, "fx3d_option_count": c_i32, "fx3d_option_name": C.c_char_p}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C flux3d.jl_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, c_i32)
    _lib = lib
    return lib


def last_error():
    lib = load()
    buf = C.create_string_buffer(512)
    lib.fx3d_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc):
    if rc != 0:
        raise Flux3DHipError(rc, last_error())


def call(name, *args):
    check(getattr(load(), name)(*args))


def set_option(name, value):
    """fx3d_set_option: choose one of the kernels' alternative code paths (include/flux3d_hip.h "variant switches")."""
    call("fx3d_set_option", name.encode(), int(value))


def get_option(name):
    v = c_i32(0)
    call("fx3d_get_option", name.encode(), C.byref(v))
    return v.value


def options():
    """{name: value} of every option the library knows."""
    lib = load()
    return {lib.fx3d_option_name(i).decode(): get_option(lib.fx3d_option_name(i).decode()) for i in range(lib.fx3d_option_count())}


class option:
    """``with option("knn_no_mfma", 1): ...`` -- set for the block, restored afterwards (process-wide: not for concurrent use)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False
