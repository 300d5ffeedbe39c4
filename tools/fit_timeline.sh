#!/bin/bash
# The fit loop's captured iteration node by node: rocprofv3 kernel trace of examples/fit_mesh.py --graph and one
# steady-state period (tools/rocprof_summary.py timeline), plus the untraced per-iteration time.
#   usage (on the GPU box, from the repo root):  bash tools/fit_timeline.sh <tag>   -> gpurun_out/<tag>/fit_*.txt
set -u
TAG=${1:-fit}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
python "$R/examples/fit_mesh.py" --graph --iters 2000 > "$OUT/fit_graph.log" 2>&1
tail -1 "$OUT/fit_graph.log"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_fit" -o r -- python "$R/examples/fit_mesh.py" --graph --iters 600 > "$OUT/fit_under_rocprof.log" 2>&1
python "$R/tools/rocprof_summary.py" timeline "$OUT/kt_fit/r_results.db" face_cdf > "$OUT/fit_timeline.txt" 2>&1
rm -rf "$OUT/kt_fit"
cat "$OUT/fit_timeline.txt"
