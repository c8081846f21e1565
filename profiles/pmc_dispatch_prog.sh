#!/bin/bash
# profiles/pmc_dispatch_prog.sh <tag> <kernel-name filter (regex)> <python script and its arguments...>   -- run on the GPU box.
# profiles/pmc_dispatch.sh for a program other than bench.py (the same passes, the same table): e.g.
#   bash profiles/pmc_dispatch_prog.sh neighbours 'nw_|short' tests/manual/gpu_neighbours.py
set -u
TAG=$1; FILTER=$2; shift 2
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
O=gpurun_out/pmc_$TAG; rm -rf "$O"; mkdir -p "$O"
PROG="python3 $ROOT/$*"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d "$ROOT/$O/trace" -o run --output-format csv -- $PROG > "$ROOT/$O/trace.log" 2>&1
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  NAME=$(echo "$SET" | tr ' ' '+')
  timeout 900 rocprofv3 --pmc $SET -d "$ROOT/$O/pmc_$NAME" -o run --output-format csv -- $PROG > "$ROOT/$O/pmc_$NAME.log" 2>&1
done
cd "$ROOT" && python3 profiles/pmc_dispatch.py "$O" "$FILTER" "(program: $*)" > "$O/dispatches.txt" 2> "$O/dispatch.err"
find "$O" -name '*.csv' -size +2M -delete
cat "$O/dispatches.txt"
