# round 6, second GPU run: threshold kernels remapped (wide four-wave classes on eight waves, the eight-wave list in two launches),
# compact plans (one-locus latency), --end-to-end over ranks
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_job2}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/gputests.log
timeout 400 python bench.py --workload config5hifi --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $O/bench_config5hifi_exact_only.json 2> $O/bench_config5hifi_exact_only.err
timeout 400 python bench.py --workload config5hifi --no-cpu-baseline --no-neighbours --no-end-to-end --steps 5 --warmup 2 --debug wg_first_pass=2 > $O/bench_config5hifi_fp2.json 2> $O/bench_config5hifi_fp2.err
timeout 400 python bench.py --workload config5 --no-cpu-baseline --no-neighbours --steps 5 --warmup 2 > $O/bench_config5.json 2> $O/bench_config5.err
for tr in 3000 5000 7400; do timeout 300 python tests/manual/gpu_long_vntr_rate.py $tr 48 -1 2 2>&1 | grep -v amdgpu.ids; done > $O/long_vntr_first_pass.log 2>&1
timeout 300 python tests/manual/gpu_long_vntr_rate.py 5000 48 -1 2 wgt_keep_waves=1 2>&1 | grep -v amdgpu.ids >> $O/long_vntr_first_pass.log
timeout 300 python tests/manual/gpu_long_vntr_rate.py 4200 48 -1 2 2>&1 | grep -v amdgpu.ids >> $O/long_vntr_first_pass.log
timeout 300 python tests/manual/gpu_long_vntr_rate.py 4200 48 -1 2 wgt_keep_waves=1 2>&1 | grep -v amdgpu.ids >> $O/long_vntr_first_pass.log
timeout 300 python bench.py --workload config2 --no-cpu-baseline --no-neighbours --steps 20 --warmup 5 > $O/bench_config2.json 2> $O/bench_config2.err
timeout 300 python bench.py --workload config2 --no-cpu-baseline --no-neighbours --steps 20 --warmup 5 --debug compact_plan=-1 > $O/bench_config2_nocompact.json 2> $O/bench_config2_nocompact.err
timeout 300 python tests/manual/gpu_adapter_latency.py 60 > $O/adapter_latency.log 2>&1
timeout 900 python bench.py --gpus 8 --one-gpu --end-to-end --steps 2 --warmup 1 > $O/bench_e2e_8ranks_one_gpu.json 2> $O/bench_e2e_8ranks.err
timeout 900 python bench.py --gpus 2 --one-gpu --end-to-end --steps 2 --warmup 1 > $O/bench_e2e_2ranks_one_gpu.json 2> $O/bench_e2e_2ranks.err
timeout 600 python bench.py --end-to-end --steps 3 --warmup 1 > $O/bench_e2e_1rank.json 2> $O/bench_e2e_1rank.err
tail -4 $O/gputests.log; cat $O/long_vntr_first_pass.log $O/adapter_latency.log; python - <<P
import json,glob
for f in sorted([f for f in glob.glob("$O/bench_*.json") if "detail" not in f]):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'whole', d['roofline'].get('whole_pass_frac'), d['roofline']['kernel'][:60], 'mism', d.get('oracle_check',{}).get('mismatches'), 'e2e', d.get('loci_per_s_end_to_end'), d.get('wg_first_pass',{}).get('kernels'), d.get('host_threads_per_rank'), d.get('gather_check'), d.get('loci_per_s'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
P
