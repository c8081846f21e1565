# round-6 measurement set: GPU tests, smoke, the default bench line, the other workloads, plan-size curves (both indel models), the N-rank
# paths on one GPU (resident and end to end), per-locus latency, host-thread budgets, fuzz
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_final}; mkdir -p $O
ASYM="-1.2,-0.3,-0.9,-0.5,-0.0001,-5.0,-4.0"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_config3.json 2> $O/bench_config3.err; cp bench_detail.json $O/bench_detail_config3.json
for w in catalogue config2 config5 config5hifi config3skew; do timeout 900 python bench.py --workload $w --no-cpu-baseline --no-neighbours --steps 5 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; cp bench_detail_$w.json $O/; done
timeout 400 python bench.py --workload config5 --no-cpu-baseline --no-neighbours --no-end-to-end --steps 5 --warmup 2 --debug wg_first_pass=1 > $O/bench_config5_certificates_first.json 2> $O/bench_config5_cf.err
timeout 600 python bench.py --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $O/bench_config3_exact_only.json 2> $O/bench_config3_exact_only.err
timeout 600 python bench.py --workload config5hifi --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $O/bench_config5hifi_exact_only.json 2> $O/bench_config5hifi_exact_only.err
timeout 600 python bench.py --workload config5hifi --no-cpu-baseline --no-end-to-end --no-neighbours --steps 5 --warmup 2 --debug wg_first_pass=2 > $O/bench_config5hifi_thresholds_first.json 2> $O/bench_config5hifi_tf.err
timeout 900 python bench.py --params=$ASYM --no-cpu-baseline --no-neighbours --steps 5 --warmup 2 > $O/bench_config3_general_model.json 2> $O/bench_config3_general_model.err
timeout 900 python bench.py --gpus 2 --one-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_config4_2ranks_one_gpu.json 2> $O/bench_2ranks.err
timeout 900 python bench.py --gpus 8 --one-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_config4_8ranks_one_gpu.json 2> $O/bench_8ranks.err
timeout 900 python bench.py --gpus 8 --one-gpu --end-to-end --steps 2 --warmup 1 > $O/bench_e2e_8ranks_one_gpu.json 2> $O/bench_e2e_8ranks.err
timeout 600 python bench.py --end-to-end --steps 3 --warmup 1 > $O/bench_e2e_1rank.json 2> $O/bench_e2e_1rank.err
timeout 600 python bench.py --end-to-end --steps 3 --warmup 1 --host-threads 2 > $O/bench_e2e_1rank_2threads.json 2> $O/bench_e2e_1rank_2t.err
timeout 600 python bench.py --workload catalogue --end-to-end --steps 3 --warmup 1 --loci 30000 > $O/bench_e2e_catalogue_1rank.json 2> $O/bench_e2e_cat.err
timeout 600 python bench.py --workload catalogue --end-to-end --steps 3 --warmup 1 --loci 30000 --host-threads 4 > $O/bench_e2e_catalogue_1rank_4threads.json 2> $O/bench_e2e_cat4.err
timeout 600 python tests/manual/gpu_plan_size.py > $O/plan_size.log 2>&1
timeout 900 python tests/manual/gpu_plan_size.py config3 10000 0 $ASYM > $O/plan_size_general_model.log 2>&1
timeout 900 python tests/manual/gpu_plan_size.py config3 10000 1 $ASYM > $O/plan_size_general_model_launch_per_class.log 2>&1
timeout 600 python tests/manual/gpu_plan_size.py catalogue 100000 > $O/plan_size_catalogue.log 2>&1
timeout 600 python tests/manual/gpu_shard_balance.py 8 > $O/shard_balance.log 2>&1
timeout 300 python tests/manual/gpu_adapter_latency.py 60 > $O/adapter_latency.log 2>&1
for tr in 3000 5000 7400; do for k in 1 2; do timeout 300 python tests/manual/gpu_long_vntr_rate.py $tr 48 -1 $k 2>&1 | grep -v amdgpu.ids; done; done > $O/long_vntr_first_pass.log 2>&1
timeout 300 python tests/manual/gpu_calc_hap_aln_probs_rate.py 30000 catalogue trace > $O/e2e_trace_catalogue.log 2>&1
timeout 200 python tests/manual/gpu_nw_rate.py 3000 trace 2>&1 | grep -v amdgpu.ids > $O/nw_rate.log
for s in 81 82 83; do timeout 400 python tests/manual/gpu_fuzz.py 150 $s 2>&1 | tail -1; done > $O/fuzz.log 2>&1
for s in 5 6; do timeout 300 python tests/manual/gpu_short_fuzz.py 45 $s 2>&1 | tail -1; done > $O/short_fuzz.log 2>&1
for s in 21 22; do timeout 300 python tests/manual/gpu_nw_fuzz.py 60 $s 2>&1 | tail -1; done > $O/nw_fuzz.log 2>&1
tail -3 $O/gputests.log; tail -2 $O/smoke.log; grep -v amdgpu.ids $O/plan_size.log $O/plan_size_general_model.log $O/plan_size_catalogue.log | cut -c1-260; cat $O/fuzz.log $O/short_fuzz.log $O/nw_fuzz.log $O/adapter_latency.log $O/long_vntr_first_pass.log; python - <<P
import json,glob
for f in sorted([f for f in glob.glob("$O/bench_*.json") if "detail" not in f]):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'whole', d['roofline'].get('whole_pass_frac'), d['roofline']['kernel'][:50], 'mism', d.get('oracle_check',{}).get('mismatches'), 'e2e', d.get('loci_per_s_end_to_end'), 'e2efrac', d.get('end_to_end_frac_of_resident'), (d.get('wg_first_pass') or {}).get('kernels'), d.get('host_threads_per_rank'), d.get('loci_per_s'))
    except Exception as e: print(f, 'ERR', e)
P
