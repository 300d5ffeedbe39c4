"""Representation conversions that sit on the 1-NN path (src/conversions.jl:91-131)."""
import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, current_stream, workspace
from .metrics import _as_dev_points


def pointcloud_to_voxel(pcloud, resolution=32):
    """`pointcloud_to_voxel(pcloud, resolution)` (src/conversions.jl:91-131): occupancy grid
    (res,res,res,B) Float32 on the device; a voxel is set iff the nearest (min/max-normalised) cloud
    point of its lattice centre is within sqrt(0.6)/res.  Same lattice convention as the reference
    (centres (i+0.5)/res for i = 1..res; first array dimension = the innermost loop variable)."""
    x = _as_dev_points(pcloud)
    D, N, B = x.shape
    if D != 3:
        raise ValueError("pointcloud_to_voxel needs 3-D points")
    res = int(resolution)
    nb = C.c_size_t(0)
    _lib.call("fx3d_voxel_workspace_bytes", B, C.byref(nb))
    ws = workspace(nb.value, tag="voxel")
    out = DeviceArray.empty((res, res, res, B), np.float32)
    _lib.call("fx3d_pointcloud_to_voxel", x.ptr, N, B, res, out.ptr, ws.ptr, ws.nbytes, current_stream().handle)
    return out
