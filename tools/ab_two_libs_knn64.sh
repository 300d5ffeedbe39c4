#!/bin/bash
# Same-box A/B of TWO BUILDS of the library at C4' (no runtime option needed): build the baseline, copy it to
# flux3d.jl_amd/lib/libflux3d_hip_base.so, build the variant in place, then on the GPU box:  bash tools/ab_two_libs_knn64.sh
# (FX3D_HIP_LIB selects the library the Python loader opens; rocprofv3 kernel-trace average / minimum of knn_mfma_kernel in ns,
# two alternating rounds).  Round 4: the decode with a plain running position instead of the packed per-stage positions: 52.55 /
# 52.61 vs 52.32 / 52.51 us -- inside the noise, not kept.
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
for v in base variant; do
  if [ $v = base ]; then export FX3D_HIP_LIB=$GRAFT_REPO_ROOT/flux3d.jl_amd/lib/libflux3d_hip_base.so; else unset FX3D_HIP_LIB; fi
  rm -rf /tmp/kt_$v; rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o r -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py knn64 --reps 60 > /dev/null 2>&1
  echo "$v: $(python $GRAFT_REPO_ROOT/tools/rocprof_summary.py stats /tmp/kt_$v/r_results.db | grep knn_mfma | awk -F'|' '{print $4, $5}')"
done
done
