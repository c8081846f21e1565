#!/bin/bash
# profiles/assemble_r05.sh [job dir] [profile dir]  -- HERE, after `gpurun -- bash tests/manual/r05_job.sh <job>` and
# `gpurun -- bash tests/manual/r05_profile_job.sh <prof>` have merged their output into gpurun_out/: reduces the collections to the
# files committed under profiles/r05/ (see its README.md).
set -e
cd "$(dirname "$0")/.."
R=profiles/r05; O=gpurun_out/${1:-r05_final}; P=gpurun_out/${2:-r05_prof_final}; mkdir -p $R
python3 profiles/make_traffic.py gpurun_out/prof_r05/summary.json $R | tail -1
T=$(mktemp -d); python3 profiles/make_traffic.py gpurun_out/prof_r05_catalogue/summary.json $T | tail -1
cp $T/kernel_stats.csv $R/kernel_stats_catalogue.csv; cp $T/pmc_traffic.json $R/pmc_traffic_catalogue.json; rm -rf $T
last_json() { python3 -c "import sys; l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; open(sys.argv[2],'w').write(l)" "$1" "$2"; }
for w in config3 catalogue config2 config5 config5hifi config3skew config3_exact_only config5hifi_exact_only config4_2ranks_one_gpu config4_8ranks_one_gpu; do last_json $O/bench_$w.json $R/bench_$w.json; done
for w in config3 catalogue config5hifi; do cp $O/bench_detail_$w.json $R/bench_detail_$w.json; done
cp gpurun_out/pmc_shard1250/dispatches.txt $R/pmc_dispatch_shard1250.txt
cp gpurun_out/pmc_exact/dispatches.txt $R/pmc_dispatch_exact_only.txt
cp gpurun_out/pmc_neighbours/dispatches.txt $R/pmc_dispatch_neighbours.txt
for f in plan_size.log plan_size_catalogue.log shard_balance.log fuzz.log nw_rate.log prep_ahead_ab.log; do cp $O/$f $R/$f; done
cp $O/chain_ab.log $R/chain/chain_ab_final_build.log
cp $O/short_fuzz.log $R/short_path_fuzz.log; cp $O/nw_fuzz.log $R/nw_fuzz.log; cp $O/gputests.log $R/gpu_tests.log
grep -v "launched\|upload:\|tables built\|plan: create" $O/e2e_trace_catalogue.log > $R/e2e_trace_catalogue.log; grep -v "launched\|upload:\|tables built\|plan: create" $O/e2e_trace_config3.log > $R/e2e_trace_config3.log
grep -v amdgpu.ids $P/wave_clock_1250.log | cut -c1-1200 > $R/wave_clock_1250.log; grep -v amdgpu.ids $P/wave_clock_10000.log | cut -c1-1200 > $R/wave_clock_10000.log
cp "$(find $P/trace_neighbours -name '*kernel_stats.csv' | head -1)" $R/kernel_stats_neighbours.csv
ls $R
