#!/bin/bash
# Reproduces the evidence kept under profiles/: rocprofv3 kernel-trace stats of bench.py and of the per-op
# benchmark, and the PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_*; counters in their own runs, no trace domains).
#   usage (on the GPU box, from the repo root):  bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>/*
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras"   # only the headline launches in the trace
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o r -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
python "$ROOT/tools/rocprof_summary.py" stats "$OUT/kt/r_results.db" > "$OUT/kernel_trace_stats.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_ops" -o r -- python "$ROOT/tools/bench_ops.py" > "$OUT/ops_under_rocprof.jsonl" 2> "$OUT/ops_under_rocprof.err"
python "$ROOT/tools/rocprof_summary.py" stats "$OUT/kt_ops/r_results.db" > "$OUT/ops_kernel_trace_stats.txt" 2>&1
# PMC passes: one op per run (tools/pmc_driver.py), counters in their own runs (no trace domains next to --pmc)
pmc_pass() {  # <op> <tag> <counters...>
  local op=$1 tag=$2; shift 2
  rocprofv3 --pmc "$@" -d "$OUT/pmc_${op}_$tag" -o r -- python "$ROOT/tools/pmc_driver.py" "$op" > /dev/null 2>&1
}
for op in nn1 knn3 knn64; do
  pmc_pass $op fetch FETCH_SIZE
  pmc_pass $op write WRITE_SIZE
  pmc_pass $op sq SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
  pmc_pass $op sq2 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pmc_pass $op l1l2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
  pmc_pass $op l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
  python "$ROOT/tools/rocprof_summary.py" pmc "$OUT"/pmc_${op}_*/r_results.db > "$OUT/pmc_$op.txt" 2>&1
done
# the HBM-bound members at a bandwidth-bound size (tools/hbm_roofline.py): kernel trace, then FETCH / WRITE in their own passes
python "$ROOT/tools/hbm_roofline.py" --table > "$OUT/hbm_roofline.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_hbm" -o r -- python "$ROOT/tools/hbm_roofline.py" --reps 5 > /dev/null 2>&1
python "$ROOT/tools/rocprof_summary.py" stats "$OUT/kt_hbm/r_results.db" > "$OUT/hbm_kernel_trace_stats.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_hbm_fetch" -o r -- python "$ROOT/tools/hbm_roofline.py" --reps 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_hbm_write" -o r -- python "$ROOT/tools/hbm_roofline.py" --reps 5 > /dev/null 2>&1
python "$ROOT/tools/rocprof_summary.py" pmc "$OUT"/pmc_hbm_*/r_results.db > "$OUT/pmc_hbm.txt" 2>&1
cp "$OUT/pmc_nn1.txt" "$OUT/pmc.txt"
python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
python "$ROOT/tools/bench_ops.py" --cpu > "$OUT/bench_ops.jsonl" 2> "$OUT/bench_ops.err"
rm -rf "$OUT"/kt/*.db "$OUT"/kt_ops/*.db "$OUT"/kt_hbm/*.db "$OUT"/pmc_*/*.db 2>/dev/null
ls -la "$OUT"
