"""flux3d.jl_amd -- MI355X (gfx950) implementation of Flux3D.jl's geometric-metric hot path.

Host-side mirror of the reference API for this path (same names and argument meaning as
src/metrics/*.jl, src/transforms/mesh_func.jl:21-82, src/rep/{pcloud,mesh}.jl, the kNN call sites of
src/models/dgcnn.jl) over the C ABI of ``lib/libflux3d_hip.so`` (``include/flux3d_hip.h``).  The
Julia twin of this layer is ``julia/Flux3DHip.jl``.  Importing the package loads the shared
library and fails loudly if it is missing; there is no CPU implementation behind these functions.

The directory name carries a dot; import it as ``flux3d_jl_amd`` (repo-root alias module).
"""
from . import _lib

_lib.load()  # fail loudly (ImportError) if the HIP library has not been built

from ._lib import Flux3DHipError, LIB_PATH  # noqa: E402
from .device import (DeviceArray, Event, Graph, Stream, cpu, current_stream, device_count, device_identity,  # noqa: E402
                     device_name, empty_cache, functional, gpu, set_device, stream, synchronize)
from .rep import (PointCloud, TriMesh, get_edges_packed, get_edges_to_key, get_faces_list,  # noqa: E402
                  get_faces_packed, get_faces_padded, get_faces_to_edges_packed,
                  get_laplacian_packed, get_verts_list, get_verts_packed, get_verts_padded,
                  load_obj, load_off, load_trimesh, npoints)
from .metrics import (chamfer_distance, chamfer_distance_grad, chamfer_loss_pairwise_f32, chamfer_sampled_grad, chamfer_value_and_grad, edge_loss, edge_loss_grad,  # noqa: E402
                      laplacian_loss, laplacian_loss_grad, mesh_losses, mesh_losses_grad, MeshReg, nearest_neighbors,
                      sampling_adjoint_is_ordered)
from .transforms import (EPS, compute_faces_areas_list, compute_faces_areas_packed,  # noqa: E402
                         compute_faces_areas_padded, lincomb, offset, sample_points, sample_points_grad, sample_points_pair)
from .fit import FitStepGraph, Momentum, loss_dolphin  # noqa: E402
from .graph import (create_knn_graph, edge_features, edge_features_grad, edgeconv_graph, knn,  # noqa: E402
                    knn_gather)
from .conversions import pointcloud_to_voxel  # noqa: E402
from . import synth  # noqa: E402

use_hip = [functional()]  # the `Flux3D.use_cuda[]` analogue (src/Flux3D.jl:52-61)

__version__ = _lib.load().fx3d_version().decode()
