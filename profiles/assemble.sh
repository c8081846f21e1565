#!/bin/bash
# profiles/assemble.sh <round>  -- HERE, after `gpurun -- bash tests/manual/final_job.sh <round>` has merged its output into
# gpurun_out/: reduces the collections to the files committed under profiles/<round>/ (see its README.md).
set -e
cd "$(dirname "$0")/.."
RD=${1:-r04}; R=profiles/$RD; O=gpurun_out/final; mkdir -p $R
python3 profiles/make_traffic.py gpurun_out/prof_$RD/summary.json $R | tail -1
T=$(mktemp -d); python3 profiles/make_traffic.py gpurun_out/prof_${RD}_catalogue/summary.json $T | tail -1
cp $T/kernel_stats.csv $R/kernel_stats_catalogue.csv; cp $T/pmc_traffic.json $R/pmc_traffic_catalogue.json; rm -rf $T
last_json() { python3 -c "import sys; l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; open(sys.argv[2],'w').write(l)" "$1" "$2"; }
for w in config3 catalogue config2 config5 config5hifi config3skew config3_exact_only config5hifi_exact_only; do last_json $O/bench_$w.json $R/bench_$w.json; done
last_json $O/bench_2ranks_one_gpu.json $R/bench_config4_2ranks_one_gpu.json
last_json $O/bench_8ranks_one_gpu.json $R/bench_config4_8ranks_one_gpu.json
cp gpurun_out/pmc_c5hifi/dispatches.txt $R/pmc_dispatch_config5hifi.txt
cp gpurun_out/pmc_exact/dispatches.txt $R/pmc_dispatch_exact_only.txt
cp $O/plan_size.log $R/plan_size.log; cp $O/shard_balance.log $O/multi_ab.log $O/launch_size.log $R/
grep -v "launched\|upload:\|tables built\|plan: create" $O/e2e_trace_catalogue.log > $R/e2e_trace_catalogue.log; grep -v "launched\|upload:\|tables built\|plan: create" $O/e2e_trace_config3.log > $R/e2e_trace_config3.log
for w in config3 catalogue config5hifi; do cp $O/bench_detail_$w.json $R/bench_detail_$w.json; done; cp $O/trio.log $R/real_reads_trio.log; cp $O/trio.vcf.gz $R/real_reads_trio.vcf.gz
for t in c5hifi exact neighbours; do cp "$(find $O/trace_$t -name '*kernel_stats.csv' | head -1)" $R/kernel_stats_$t.csv; done
cp $O/gputests.log $R/gpu_tests.log
cp gpurun_out/pmc_neighbours/dispatches.txt $R/pmc_dispatch_neighbours.txt
cp $O/fuzz.log $R/fuzz.log; cp $O/short_fuzz.log $R/short_path_fuzz.log
ls $R
