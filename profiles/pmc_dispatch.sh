#!/bin/bash
# profiles/pmc_dispatch.sh <tag> <kernel-name filter (regex)> <bench.py arguments...>   -- run on the GPU box (gpurun), from anywhere.
# Per DISPATCH of the kernels that match the filter: duration (rocprofv3 --kernel-trace) and PMC counters (separate
# --pmc passes, never combined with a trace domain; the program directly after `--`), joined by dispatch order
# (bench.py --debug fan_lanes=1: one stream, the same launch sequence in every pass).  Output:
# gpurun_out/pmc_<tag>/dispatches.txt (copy it to profiles/<round>/ to have it judged).
set -u
TAG=$1; FILTER=$2; shift 2
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
O=gpurun_out/pmc_$TAG; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py $* --no-cpu-baseline --no-end-to-end --no-neighbours --no-verify --steps 1 --warmup 0 --debug fan_lanes=1"
timeout 900 rocprofv3 --kernel-trace -d "$ROOT/$O/trace" -o run --output-format csv -- $BENCH > "$ROOT/$O/trace.log" 2>&1
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  NAME=$(echo "$SET" | tr ' ' '+')
  timeout 900 rocprofv3 --pmc $SET -d "$ROOT/$O/pmc_$NAME" -o run --output-format csv -- $BENCH > "$ROOT/$O/pmc_$NAME.log" 2>&1
done
cd "$ROOT" && python3 profiles/pmc_dispatch.py "$O" "$FILTER" "$*" > "$O/dispatches.txt" 2> "$O/dispatch.err"
find "$O" -name '*.csv' -size +2M -delete
cat "$O/dispatches.txt"
