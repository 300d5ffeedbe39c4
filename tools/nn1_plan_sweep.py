"""Forced launch plans of the fp16 1-NN kernel at the fit iteration's chamfer shapes (N = M = 5000 surface samples, B = 1 and 8):
a library whose planner reads FX3D_NN1_FORCE_PLAN="chunks split passes" (the eleven-line patch of make_plan's search loop kept at the
end of this file: apply, `bash tools/build_variant.sh sweep chamfer "-DFX3D_PLAN_SWEEP"`, run with FX3D_HIP_LIB=.../libflux3d_hip_sweep.so; the
shipped planner has no such switch).  Min of single calls between events (us; the wrapper's allocations and ~4.4 us of call overhead
included).  Result (profiles/r06_v13_nn1_plan_sweep.txt): the planner's own choices -- 12 subsets of 448 candidates at one mesh, 3 x 1728
with two passes at eight -- are the fastest plans of the kernel at both shapes."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flux3d_jl_amd as fx  # noqa: E402
from flux3d_jl_amd import _lib  # noqa: E402

fx.set_device(0)
g = os.path.join(ROOT, "tests", "golden")


def call_us(fn, n=40, warm=4):
    for _ in range(warm):
        fn()
    fx.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = fx.Event(), fx.Event()
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_ms(e1))
    return round(best * 1e3, 2)


for nb in (1, 8):
    src = fx.gpu(fx.load_trimesh(*[os.path.join(g, "sphere.obj")] * nb))
    tgt = fx.gpu(fx.load_trimesh(*[os.path.join(g, "teapot.obj")] * nb))
    A, Bp = fx.sample_points(src, 5000, seed=1), fx.sample_points(tgt, 5000, seed=2)
    ref = None
    for force in ["", "2 0 1", "2 1 1", "2 1 2", "3 0 1", "3 1 1", "3 1 2", "4 1 1", "4 1 2", "6 1 1", "6 1 2", "8 1 1", "8 1 2", "12 1 1", "12 1 2", "16 1 1"]:
        if force:
            os.environ["FX3D_NN1_FORCE_PLAN"] = force
        else:
            os.environ.pop("FX3D_NN1_FORCE_PLAN", None)
        buf = C.create_string_buffer(256)
        _lib.call("fx3d_nn1_plan_describe", 5000, 5000, nb, 3, buf, 256)
        t = call_us(lambda: fx.chamfer_distance(A, Bp, return_indices=True, sync=False))
        _, ix, iy = fx.chamfer_distance(A, Bp, return_indices=True)
        sig = (ix.to_host().tobytes(), iy.to_host().tobytes())
        ref = ref or sig
        print(f"B={nb} force='{force}' {' '.join(buf.value.decode().split()[2:9])} : {t} us {'' if sig == ref else 'INDICES DIFFER'}", flush=True)


PLANNER_PATCH = r"""
diff --git a/flux3d.jl_amd/csrc/chamfer.hip b/flux3d.jl_amd/csrc/chamfer.hip
index 3da777b..9acd76e 100644
--- a/flux3d.jl_amd/csrc/chamfer.hip
+++ b/flux3d.jl_amd/csrc/chamfer.hip
@@ -2109,6 +2109,10 @@ Plan make_plan(int N, int M, int B, int D, bool allow_split = true) {
         const int cminc = (maxc + cmax - 1) / cmax;
         double best = 1e30;
         b_chunk = (maxc + gran - 1) / gran * gran < cmax ? (maxc + gran - 1) / gran * gran : cmax; b_tpb = 1; b_split = 1;
+#ifdef FX3D_PLAN_SWEEP  // (tools/nn1_plan_sweep.py: "chunks split passes" forces the plan)
+        int f_nch = 0, f_split = 0, f_tpb = 0;
+        if (const char *fp = getenv("FX3D_NN1_FORCE_PLAN")) sscanf(fp, "%d %d %d", &f_nch, &f_split, &f_tpb);
+#endif
         for (int nch = cminc; nch <= cminc * 8 && nch <= 64; ++nch) {
             int ch = ((maxc + nch - 1) / nch + gran - 1) / gran * gran;
             if (ch > cmax) continue;
@@ -2117,6 +2121,9 @@ Plan make_plan(int N, int M, int B, int D, bool allow_split = true) {
                 if (split && (!allow_split || anch == 1 || opt(OPT_NN1_NOSPLIT))) continue;
                 for (int tpb = 1; tpb <= 8; tpb *= 2) {
                     if (!split && anch > 1 && tpb > 1) continue;
+#ifdef FX3D_PLAN_SWEEP
+                    if (f_nch && (nch != f_nch || split != f_split || tpb != f_tpb)) continue;
+#endif
                     const long long tiles = ((long long)maxc + 512 * tpb - 1) / (512 * tpb);
                     const long long blocks = 2ll * B * tiles * (split ? anch : 1);
                     const double per_chunk = 5.0 * ch / 4096.0 + 0.5 + tpb * (9.7 * ch / 4096.0 + 0.8);
"""
