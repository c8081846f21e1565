# round-end measurement set: GPU tests, smoke, the default bench line, the other workloads, profiles
cd $GRAFT_REPO_ROOT; R=${1:-r04}; O=gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_config3.json 2> $O/bench_config3.err; cp bench_detail.json $O/bench_detail_config3.json
for w in catalogue config2 config5 config5hifi config3skew; do timeout 900 python bench.py --workload $w --no-cpu-baseline --no-neighbours --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; cp bench_detail_$w.json $O/; done
timeout 600 python bench.py --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $O/bench_config3_exact_only.json 2> $O/bench_config3_exact_only.err
timeout 600 python bench.py --workload config5hifi --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $O/bench_config5hifi_exact_only.json 2> $O/bench_config5hifi_exact_only.err
timeout 900 python bench.py --gpus 2 --one-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
timeout 900 python bench.py --gpus 8 --one-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_8ranks_one_gpu.json 2> $O/bench_8ranks.err
timeout 600 python tests/manual/gpu_plan_size.py > $O/plan_size.log 2>&1
timeout 600 python tests/manual/gpu_shard_balance.py 8 > $O/shard_balance.log 2>&1
timeout 600 python tests/manual/gpu_multi_ab.py config3 1 8 16 > $O/multi_ab.log 2>&1
timeout 300 python tests/manual/gpu_multi_ab.py catalogue 1 8 >> $O/multi_ab.log 2>&1
timeout 300 python tests/manual/gpu_launch_size.py 930 > $O/launch_size.log 2>&1
timeout 300 python tests/manual/gpu_calc_hap_aln_probs_rate.py 30000 catalogue trace > $O/e2e_trace_catalogue.log 2>&1
timeout 300 python tests/manual/gpu_calc_hap_aln_probs_rate.py 6000 config3 trace > $O/e2e_trace_config3.log 2>&1
timeout 600 python examples/real_reads_trio.py $O/trio.vcf.gz > $O/trio.log 2>&1
bash profiles/collect.sh $R > $O/collect.log 2>&1
bash profiles/collect.sh ${R}_catalogue catalogue > $O/collect_catalogue.log 2>&1
bash profiles/pmc_dispatch.sh c5hifi 'wg_kernel|wgx' --workload config5hifi > $O/pmc_c5hifi.log 2>&1
bash profiles/pmc_dispatch.sh exact 'ltr_dp_kernel<|wgx' --pair-packing 4 > $O/pmc_exact.log 2>&1
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$O/trace_c5hifi -o run --output-format csv -- python3 $ROOT/bench.py --workload config5hifi --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $ROOT/$O/trace_c5hifi.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$O/trace_exact -o run --output-format csv -- python3 $ROOT/bench.py --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 2 --warmup 1 > $ROOT/$O/trace_exact.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$O/trace_neighbours -o run --output-format csv -- python3 $ROOT/tests/manual/gpu_neighbours.py > $ROOT/$O/trace_neighbours.log 2>&1
cd $ROOT; bash profiles/pmc_dispatch_prog.sh neighbours 'nw_|short' tests/manual/gpu_neighbours.py > $O/pmc_neighbours.log 2>&1
for s in 21 22; do timeout 300 python tests/manual/gpu_fuzz.py 90 $s 2>&1 | tail -1; done > $O/fuzz.log 2>&1
for s in 3 4; do timeout 300 python tests/manual/gpu_short_fuzz.py 60 $s 2>&1 | tail -1; done > $O/short_fuzz.log 2>&1
cd $ROOT; find $O gpurun_out/prof_* gpurun_out/pmc_* -name "*kernel_trace.csv" -size +2M -delete
tail -3 $O/gputests.log; tail -2 $O/smoke.log; python - <<'P'
import json,glob
for f in sorted([f for f in glob.glob("gpurun_out/final/bench_*.json") if "detail" not in f]):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'mism', d.get('oracle_check',{}).get('mismatches'), 'e2e', d.get('loci_per_s_end_to_end'), 'plan_create_s', round(d.get('plan_create_s') or 0,4))
    except Exception as e: print(f, 'ERR', e)
P
