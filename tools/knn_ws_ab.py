"""fx3d_knn (every block of the search kernel builds the fp16 image of its cloud) against fx3d_knn_ws (pre-pass: the image is
built once per cloud) at C4' (B = 32 x 1024, D = 64, k = 20 + self), same process, min of 5 x 20 back-to-back calls."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flux3d_jl_amd as fx
from flux3d_jl_amd import _lib
from flux3d_jl_amd.device import DeviceArray, current_stream
rng = np.random.default_rng(1)
x = fx.gpu(np.asfortranarray(rng.standard_normal((64, 1024, 32)).astype(np.float32)))
idx = DeviceArray.empty((20, 1024, 32), np.int32)
def run(ws):
    if ws is not None:
        _lib.call("fx3d_knn_ws", x.ptr, 1024, x.ptr, 1024, 32, 64, 20, 1, idx.ptr, None, ws.ptr, ws.nbytes, current_stream().handle)
    else:
        _lib.call("fx3d_knn", x.ptr, 1024, x.ptr, 1024, 32, 64, 20, 1, idx.ptr, None, current_stream().handle)
for mode in ("knn", "knn_ws"):
    ws = None
    if mode == "knn_ws":
        nb = C.c_size_t(0); _lib.call("fx3d_knn_workspace_bytes", 1024, 1024, 32, 64, 20, 1, C.byref(nb))
        ws = DeviceArray.empty((nb.value,), np.uint8)
    for _ in range(5): run(ws)
    fx.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = fx.Event(), fx.Event(); e0.record()
        for _ in range(20): run(ws)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_ms(e1) / 20 * 1e3)
    print(f"{mode:7s}: {best:6.1f} us")
