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
BENCH="python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o r -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
python "$ROOT/tools/rocprof_summary.py" stats "$OUT/kt/r_results.db" > "$OUT/kernel_trace_stats.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/kt_ops" -o r -- python "$ROOT/tools/bench_ops.py" > "$OUT/ops_under_rocprof.jsonl" 2> "$OUT/ops_under_rocprof.err"
python "$ROOT/tools/rocprof_summary.py" stats "$OUT/kt_ops/r_results.db" > "$OUT/ops_kernel_trace_stats.txt" 2>&1
PB="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o r -- $PB > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o r -- $PB > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES -d "$OUT/pmc_sq" -o r -- $PB > /dev/null 2>&1
python "$ROOT/tools/rocprof_summary.py" pmc "$OUT/pmc_fetch/r_results.db" "$OUT/pmc_write/r_results.db" "$OUT/pmc_sq/r_results.db" > "$OUT/pmc.txt" 2>&1
python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
python "$ROOT/tools/bench_ops.py" --cpu > "$OUT/bench_ops.jsonl" 2> "$OUT/bench_ops.err"
rm -rf "$OUT"/kt/*.db "$OUT"/kt_ops/*.db "$OUT"/pmc_*/*.db 2>/dev/null
ls -la "$OUT"
