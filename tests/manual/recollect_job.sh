cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/gputests.log
bash profiles/collect.sh r04 > $O/collect.log 2>&1
bash profiles/collect.sh r04_catalogue catalogue > $O/collect_catalogue.log 2>&1
timeout 300 python tests/manual/gpu_calc_hap_aln_probs_rate.py 30000 catalogue trace > $O/e2e_trace_catalogue.log 2>&1
timeout 300 python tests/manual/gpu_calc_hap_aln_probs_rate.py 6000 config3 trace > $O/e2e_trace_config3.log 2>&1
timeout 400 python tests/manual/gpu_staging_ab.py 30000 catalogue > $O/staging_ab.log 2>&1
timeout 600 python bench.py --workload catalogue --no-cpu-baseline --no-neighbours --steps 5 --warmup 1 > $O/bench_catalogue.json 2> $O/bench_catalogue.err; cp bench_detail_catalogue.json $O/
timeout 300 python examples/real_reads_trio.py $O/trio.vcf.gz > $O/trio.log 2>&1
find gpurun_out/prof_* -name "*kernel_trace.csv" -size +2M -delete
tail -2 $O/gputests.log; tail -1 $O/e2e_trace_catalogue.log | cut -c1-260; tail -3 $O/trio.log | cut -c1-200
