#!/bin/bash
# profiles/collect.sh <tag>  -- run on the GPU box (through gpurun) from the repo root.
# Collects, for the default bench workload (config 3, 10 000 loci):
#   * rocprofv3 --kernel-trace --stats        -> gpurun_out/prof_<tag>/trace
#   * rocprofv3 --pmc <one counter set> (separate passes, never combined with a trace domain)
#                                             -> gpurun_out/prof_<tag>/pmc_<set>
# profiles/summarize.py then reduces the CSVs to the small files committed under profiles/<round>/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
# A plan's launches normally alternate between two streams (a launch then shares the GPU with its neighbour and its own
# duration says little); the profiled runs keep them on ONE stream -- bench.py --debug fan_lanes=1 -> ltr_ctx_set_debug --
# so that every launch of a kernel is the launch bench.py times with HIP events in its per-kernel pass (roofline.kernel_ms).
WORKLOAD=${2:-config3}
BENCH="python3 $ROOT/bench.py --workload $WORKLOAD --no-cpu-baseline --no-end-to-end --no-neighbours --steps 2 --warmup 1 --debug fan_lanes=1"
timeout 600 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/trace" -o run --output-format csv -- $BENCH > "$ROOT/$OUT/trace.log" 2>&1
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  NAME=$(echo "$SET" | tr ' ' '+')
  timeout 600 rocprofv3 --pmc $SET -d "$ROOT/$OUT/pmc_$NAME" -o run --output-format csv -- $BENCH > "$ROOT/$OUT/pmc_$NAME.log" 2>&1
done
cd "$ROOT" && python3 profiles/summarize.py "$OUT" "$WORKLOAD" > "$OUT/summary.json" 2> "$OUT/summarize.err"
# keep the merge-back small: the raw per-dispatch traces are not needed once summarised
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
tail -c 2000 "$OUT/summary.json"
