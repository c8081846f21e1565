# round 6, fifth GPU run: the eight-wave threshold launches one after the other instead of side by side; band skipping in the narrow exact kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_job5}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_wg.py tests/test_gpu_align.py tests/test_gpu_scale.py tests/test_gpu_host_path.py -m gpu -q -x 2>&1 | tail -5 > $O/gputests.log
timeout 600 python bench.py --workload config5hifi --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $O/bench_config5hifi_exact_only.json 2> $O/e1.err
timeout 600 python bench.py --workload config5hifi --no-cpu-baseline --no-end-to-end --no-neighbours --steps 5 --warmup 2 --debug wg_first_pass=2 > $O/bench_config5hifi_thresholds_first.json 2> $O/e2.err
timeout 600 python bench.py --workload config5hifi --no-cpu-baseline --no-end-to-end --no-neighbours --steps 5 --warmup 2 > $O/bench_config5hifi.json 2> $O/e3.err
timeout 400 python bench.py --workload config5 --no-cpu-baseline --no-neighbours --no-end-to-end --steps 5 --warmup 2 > $O/bench_config5.json 2> $O/e4.err
timeout 400 python bench.py --workload config5 --no-cpu-baseline --no-neighbours --no-end-to-end --steps 5 --warmup 2 --debug wg_first_pass=1 > $O/bench_config5_certificates_first.json 2> $O/e5.err
timeout 600 python bench.py --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 > $O/bench_config3_exact_only.json 2> $O/e6.err
for s in 91; do timeout 400 python tests/manual/gpu_fuzz.py 120 $s 2>&1 | tail -1; done > $O/fuzz.log 2>&1
tail -3 $O/gputests.log; cat $O/fuzz.log; python - <<P
import json,glob
for f in sorted([f for f in glob.glob("$O/bench_*.json") if "detail" not in f]):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'whole', d['roofline'].get('whole_pass_frac'), d['roofline']['kernel'][:50], 'mism', d.get('oracle_check',{}).get('mismatches'), (d.get('wg_first_pass') or {}).get('kernels'))
    except Exception as e: print(f, 'ERR', e)
P
