#!/bin/bash
# profiles/assemble_r06.sh [job dir] [profile dir]  -- HERE, after `gpurun -- bash tests/manual/r06_final_job.sh <job>` and
# `gpurun -- bash tests/manual/r06_profile_job.sh <prof>` have merged their output into gpurun_out/: reduces the collections to the
# files committed under profiles/r06/ (see its README.md).
set -e
cd "$(dirname "$0")/.."
R=profiles/r06; O=gpurun_out/${1:-r06_final}; P=gpurun_out/${2:-r06_prof}; mkdir -p $R
last_json() { python3 -c "import sys; l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; open(sys.argv[2],'w').write(l)" "$1" "$2"; }
for f in $O/bench_*.json; do b=$(basename $f); case $b in bench_detail*) cp $f $R/$b;; *) last_json $f $R/$b || echo "no line in $b";; esac; done
for f in plan_size.log plan_size_general_model.log plan_size_general_model_launch_per_class.log plan_size_catalogue.log shard_balance.log fuzz.log nw_rate.log adapter_latency.log long_vntr_first_pass.log; do grep -v amdgpu.ids $O/$f > $R/$f || true; done
cp $O/short_fuzz.log $R/short_path_fuzz.log; cp $O/nw_fuzz.log $R/nw_fuzz.log; cp $O/gputests.log $R/gpu_tests.log
grep -v "launched\|upload:\|tables built\|plan: create\|amdgpu.ids" $O/e2e_trace_catalogue.log > $R/e2e_trace_catalogue.log || true
if [ -d gpurun_out/prof_r06 ]; then
  python3 profiles/make_traffic.py gpurun_out/prof_r06/summary.json $R | tail -1
  for t in shard1250 config5_thresholds config5_certificates config5hifi_exact config5hifi_thresholds neighbours; do [ -f gpurun_out/pmc_$t/dispatches.txt ] && cp gpurun_out/pmc_$t/dispatches.txt $R/pmc_dispatch_$t.txt; done
  for w in 1250 10000; do [ -f $P/wave_clock_$w.log ] && grep -v amdgpu.ids $P/wave_clock_$w.log | cut -c1-1200 > $R/wave_clock_$w.log; done
  f=$(find $P/trace_config5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $R/kernel_stats_config5.csv
fi
ls $R
