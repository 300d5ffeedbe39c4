#!/bin/bash
# Same-box A/B of several BUILDS of the library (no runtime option needed): keep each build as
# flux3d.jl_amd/lib/libflux3d_hip_<name>.so (the in-place library is "default"), then on the GPU box:
#   bash tools/ab_two_libs.sh <pmc_driver op: nn1 | knn3 | knn64 | ...> <kernel name pattern> <reps> <name> [<name> ...]
# FX3D_HIP_LIB selects the library the Python loader opens; prints calls, rocprofv3 kernel-trace average and minimum (ns) of the
# kernel, three alternating rounds.
OP=${1:-nn1}; PAT=${2:-nn1_f16}; REPS=${3:-200}; shift 3
NAMES=${@:-base default}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for rep in 1 2 3; do
for v in $NAMES; do
  if [ $v = default ]; then unset FX3D_HIP_LIB; else export FX3D_HIP_LIB=$ROOT/flux3d.jl_amd/lib/libflux3d_hip_$v.so; fi
  rm -rf /tmp/kt_$v; rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o r -- python $ROOT/tools/pmc_driver.py $OP --reps $REPS > /dev/null 2>&1
  echo "$v: $(python $ROOT/tools/rocprof_summary.py stats /tmp/kt_$v/r_results.db | grep "$PAT" | awk -F'|' '{print $2, $4, $5}')"
done
done
