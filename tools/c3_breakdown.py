"""C3 (chamfer_distance(mesh, mesh, 5000), B = 8 teapots) in pieces: the draw launch, the chamfer launch on the sampled clouds
under the planner's choice and under forced plans (option nn1_nosplit), the whole call.  Kernel time = the library's
events around the launch (fx3d_profile_*), call time = min of single calls between events."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402
from flux3d_jl_amd.transforms import sample_points_pair  # noqa: E402

fx.set_device(0)
g = os.path.join(ROOT, "tests", "golden")
tv, tf = fx.load_obj(os.path.join(g, "teapot.obj"))
rng = np.random.default_rng(0)
n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
if len(sys.argv) > 2 and sys.argv[2] == "offset":  # a source mesh on its way to the target (fit_mesh.jl): noisy against scaled
    m8 = fx.gpu(fx.TriMesh([np.asfortranarray(tv + rng.standard_normal(tv.shape).astype(np.float32) * 0.01) for _ in range(8)], [tf] * 8))
    m8b = fx.gpu(fx.TriMesh([np.asfortranarray(tv * np.float32(1.1)) for _ in range(8)], [tf] * 8))
else:                                                # bench.py's C3: the same mesh on both sides, different draws
    m8 = fx.gpu(fx.TriMesh([tv] * 8, [tf] * 8))
    m8b = fx.gpu(fx.TriMesh([tv] * 8, [tf] * 8))


def call_us(fn, n=60, warm=5):
    for _ in range(warm):
        fn()
    fx.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = fx.Event(), fx.Event()
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_ms(e1))
    return round(best * 1e3, 2)


def kern_us(name, fn, reps=30):
    for _ in range(3):
        fn()
    fx.synchronize()
    _lib.call("fx3d_profile_enable", 1)
    for _ in range(reps):
        fn()
    fx.synchronize()
    avg, mn, mx, cnt = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int64(0)
    _lib.call("fx3d_profile_kernel_stats", name.encode(), C.byref(avg), C.byref(mn), C.byref(mx), C.byref(cnt))
    _lib.call("fx3d_profile_enable", 0)
    return round(avg.value * 1e3, 2), round(mn.value * 1e3, 2)


loss_dev = fx.DeviceArray.empty((1,), np.float32)
out = {"n": n3}
buf = C.create_string_buffer(256)
_lib.call("fx3d_nn1_plan_describe", n3, n3, 8, 3, buf, 256)
out["plan"] = buf.value.decode()
UA = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_A, 3, n3, 8)); UB = fx.gpu(fx.synth.uniform_cloud(fx.synth.SEED_B, 3, n3, 8))
out["nn1_kernel_uniform_clouds_us(avg,min)"] = kern_us("nn1", lambda: fx.chamfer_distance(UA, UB, loss_out=loss_dev, sync=False))
out["whole_call_cdf_rebuilt_us"] = call_us(lambda: fx.chamfer_distance(m8, m8b, n3, seed=5, loss_out=loss_dev, sync=False, reuse_cdf=False))
out["whole_call_cdf_cached_us"] = call_us(lambda: fx.chamfer_distance(m8, m8b, n3, seed=5, loss_out=loss_dev, sync=False))
out["draw_pair_call_us"] = call_us(lambda: sample_points_pair(m8, m8b, n3, seed_a=5, seed_b=6))
PA, PB = sample_points_pair(m8, m8b, n3, seed_a=5, seed_b=6)
out["chamfer_on_samples_call_us"] = call_us(lambda: fx.chamfer_distance(PA, PB, loss_out=loss_dev, sync=False))
out["nn1_kernel_us(avg,min)"] = kern_us("nn1", lambda: fx.chamfer_distance(PA, PB, loss_out=loss_dev, sync=False))
if os.environ.get("C3_BRIEF"):
    print(json.dumps({k: out[k] for k in ("plan", "nn1_kernel_uniform_clouds_us(avg,min)", "chamfer_on_samples_call_us")}))
    sys.exit(0)
for opts in ({"nn1_nosplit": 1},):
    for k, v in opts.items():
        _lib.set_option(k, v)
    out["chamfer_call_us " + json.dumps(opts)] = call_us(lambda: fx.chamfer_distance(PA, PB, loss_out=loss_dev, sync=False))
    for k in opts:
        _lib.set_option(k, 0)
print(json.dumps(out, indent=1))
