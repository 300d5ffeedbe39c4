"""k-NN graph construction for DGCNN's EdgeConv (src/models/dgcnn.jl:3-9,36)."""
import numpy as np

import ctypes as C

from . import _lib
from .device import DeviceArray, current_stream, workspace
from .metrics import _as_dev_points


def knn(x, k, y=None, drop_first=False, return_dist=True):
    """`knn(KDTree(y), x, k, true)` for every batch element (NearestNeighbors.jl call at
    src/models/dgcnn.jl:5-6).  x:(D,N,B), y:(D,M,B) (default: x itself).  Returns device
    ``idx (k,N,B)`` int32 0-based sorted by (distance, index) and ``dist (k,N,B)`` squared
    distances.  ``drop_first`` drops the rank-0 hit (the reference's `[2:K+1]`)."""
    x = _as_dev_points(x)
    y = x if y is None else _as_dev_points(y)
    D, N, B = x.shape
    M = y.shape[1]
    if y.shape[0] != D or y.shape[2] != B:
        raise ValueError("DimensionMismatch between x and y")
    idx = DeviceArray.empty((k, N, B), np.int32)
    dist = DeviceArray.empty((k, N, B), np.float32) if return_dist else None
    nb = C.c_size_t(0)
    _lib.call("fx3d_knn_workspace_bytes", N, M, B, D, int(k), int(bool(drop_first)), C.byref(nb))
    if nb.value:  # feature space: statistics + fp16 image of the candidate clouds built once per cloud (pre-pass)
        ws = workspace(nb.value, "knn")
        _lib.call("fx3d_knn_ws", x.ptr, N, y.ptr, M, B, D, int(k), int(bool(drop_first)), idx.ptr,
                  dist.ptr if dist else None, ws.ptr, ws.nbytes, current_stream().handle)
    else:
        _lib.call("fx3d_knn", x.ptr, N, y.ptr, M, B, D, int(k), int(bool(drop_first)), idx.ptr,
                  dist.ptr if dist else None, current_stream().handle)
    return (idx, dist) if return_dist else idx


def knn_gather(x, idx):
    """X[:, idxs] for every point: (F,k,N,B) (src/models/dgcnn.jl:6, cat at :36)."""
    x = _as_dev_points(x)
    F, N, B = x.shape
    k = idx.shape[0]
    out = DeviceArray.empty((F, k, N, B), np.float32)
    _lib.call("fx3d_knn_gather", x.ptr, N, B, F, k, idx.ptr, out.ptr, current_stream().handle)
    return out


def create_knn_graph(X, K):
    """EdgeConv's graph build (src/models/dgcnn.jl:36): for X (F,N,B) the K nearest neighbours of
    every point in feature space, self excluded -> neighbour features (F,K,N,B) on the device."""
    idx = knn(X, K, drop_first=True, return_dist=False)
    return knn_gather(X, idx)


_LAYOUTS = {"cat": 0, "mlp": 1, 0: 0, 1: 1}


def edge_features(X, idx, layout="mlp"):
    """EdgeConv's input features (src/models/dgcnn.jl:36-51) from X (F,N,B) and idx (K,N,B):
    ``cat(X, KNNGraph - X, dims=1)``.  layout "cat": (2F,K,N,B), the array at :45; layout "mlp":
    (K*N, 2F, B), the array handed to the 1x1 convolution after the permute + reshape (:48-51).
    The gathered graph and the K copies of X are never materialised."""
    X = _as_dev_points(X)
    F, N, B = X.shape
    k = idx.shape[0]
    lay = _LAYOUTS[layout]
    out = DeviceArray.empty((2 * F, k, N, B) if lay == 0 else (k * N, 2 * F, B), np.float32)
    _lib.call("fx3d_edge_features", X.ptr, N, B, F, k, idx.ptr, lay, out.ptr, current_stream().handle)
    return out


def edge_features_grad(gout, F, N, B, K, layout="mlp"):
    """Adjoint of :func:`edge_features` w.r.t. X; the graph is ``@nograd`` upstream
    (src/models/dgcnn.jl:9) so only the repeated-X terms carry gradient."""
    gx = DeviceArray.empty((F, N, B), np.float32)
    _lib.call("fx3d_edge_features_bwd", gout.ptr, N, B, F, K, _LAYOUTS[layout], gx.ptr, current_stream().handle)
    return gx


def edgeconv_graph(X, K, layout="mlp", return_idx=False):
    """The whole graph build of ``(m::EdgeConv)(X)`` up to the MLP input (src/models/dgcnn.jl:32-51) in
    one library call: self-kNN in feature space (self dropped) + features."""
    X = _as_dev_points(X)
    F, N, B = X.shape
    lay = _LAYOUTS[layout]
    nb = C.c_size_t(0)
    _lib.call("fx3d_knn_workspace_bytes", N, N, B, F, int(K), 1, C.byref(nb))
    if nb.value:  # feature space: the search with its pre-pass (fx3d_knn_ws), then the features -- what fx3d_edgeconv_graph
        idx = knn(X, K, drop_first=True, return_dist=False)  # does in one call, minus the per-block image builds
        out = edge_features(X, idx, layout)
        return (out, idx) if return_idx else out
    idx = DeviceArray.empty((K, N, B), np.int32)
    out = DeviceArray.empty((2 * F, K, N, B) if lay == 0 else (K * N, 2 * F, B), np.float32)
    _lib.call("fx3d_edgeconv_graph", X.ptr, N, B, F, int(K), lay, idx.ptr, out.ptr, current_stream().handle)
    return (out, idx) if return_idx else out
