cd $GRAFT_REPO_ROOT; O=gpurun_out/final5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputests.log
timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_config3.json 2> $O/bench_config3.err
timeout 900 python bench.py --workload config3skew --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_config3skew.json 2> $O/bench_config3skew.err
tail -3 $O/gputests.log; python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/final5/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'mism', d.get('oracle_check',{}).get('mismatches'), 'e2e', d.get('loci_per_s_end_to_end'), 'plan_create_s', d.get('plan_create_s'))
    except Exception as e: print(f, 'ERR', e)
P
