#!/bin/bash
# SQ counters + kernel-trace of tools/pmc_driver.py nn1 under the library options given as VAR=VALUE arguments, e.g.
#   bash tools/pmc_ab.sh FX3D_NN1_PRUNE=0      (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for kv in "$@"; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pab_*; 
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pab_sq -o r -- python $ROOT/tools/pmc_driver.py nn1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pab_sq2 -o r -- python $ROOT/tools/pmc_driver.py nn1 > /dev/null 2>&1
python $ROOT/tools/rocprof_summary.py pmc /tmp/pab_sq/r_results.db /tmp/pab_sq2/r_results.db | grep -v "^# json" | grep -v "^{" | grep nn1_f16
rocprofv3 --kernel-trace --stats -d /tmp/pab_kt -o r -- python $ROOT/tools/pmc_driver.py nn1 --reps 200 > /dev/null 2>&1
python $ROOT/tools/rocprof_summary.py stats /tmp/pab_kt/r_results.db | grep nn1_f16
