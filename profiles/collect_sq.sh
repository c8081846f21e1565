#!/bin/bash
# profiles/collect_sq.sh <tag> -- SQ-level stall/issue counters for the default bench workload
# (separate --pmc passes, no trace domains); summarised by profiles/summarize.py.
set -u
TAG=${1:-sq}
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-neighbours --steps 1 --warmup 0"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_CYCLES"; do
  NAME=$(echo "$SET" | tr ' ' '+')
  timeout 600 rocprofv3 --pmc $SET -d "$ROOT/$OUT/pmc_$NAME" -o run --output-format csv -- $BENCH > "$ROOT/$OUT/pmc_$NAME.log" 2>&1
done
cd "$ROOT" && python3 profiles/summarize.py "$OUT" > "$OUT/summary.json" 2> "$OUT/summarize.err"
tail -c 600 "$OUT/summary.json"
