# round 6, first GPU run: the workgroup threshold kernels (first pass / exact lists), the learnt first pass, the plan kernel of the
# general model, host-thread budgets -- tests, then rates
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_job1}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/gputests.log
ASYM="-1.2,-0.3,-0.9,-0.5,-0.0001,-5.0,-4.0"
for k in 0 1 2; do timeout 400 python bench.py --workload config5 --no-cpu-baseline --no-neighbours --no-end-to-end --steps 5 --warmup 2 --debug wg_first_pass=$k > $O/bench_config5_fp$k.json 2> $O/bench_config5_fp$k.err; done
for k in 0 2; do timeout 400 python bench.py --workload config5hifi --no-cpu-baseline --no-neighbours --no-end-to-end --steps 5 --warmup 2 --debug wg_first_pass=$k > $O/bench_config5hifi_fp$k.json 2> $O/bench_config5hifi_fp$k.err; done
timeout 400 python bench.py --workload config5hifi --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $O/bench_config5hifi_exact_only.json 2> $O/bench_config5hifi_exact_only.err
for tr in 3000 5000 7400; do for k in 1 2; do timeout 300 python tests/manual/gpu_long_vntr_rate.py $tr 48 -1 $k 2>&1 | grep -v amdgpu.ids; done; done > $O/long_vntr_first_pass.log 2>&1
for tr in 5000; do for k in 1 2; do timeout 300 python tests/manual/gpu_long_vntr_rate.py $tr 8 -1 $k 2>&1 | grep -v amdgpu.ids; done; done >> $O/long_vntr_first_pass.log 2>&1
timeout 900 python tests/manual/gpu_plan_size.py config3 10000 0 $ASYM 2>&1 | grep -v amdgpu.ids > $O/plan_size_asym.log
timeout 900 python tests/manual/gpu_plan_size.py config3 10000 1 $ASYM 2>&1 | grep -v amdgpu.ids > $O/plan_size_asym_per_class.log
tail -4 $O/gputests.log; cat $O/long_vntr_first_pass.log $O/plan_size_asym.log $O/plan_size_asym_per_class.log; python - <<P
import json,glob
for f in sorted([f for f in glob.glob("$O/bench_*.json") if "detail" not in f]):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'whole', d['roofline'].get('whole_pass_frac'), d['roofline']['kernel'], 'mism', d.get('oracle_check',{}).get('mismatches'), d.get('wg_first_pass'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
P
