cd $GRAFT_REPO_ROOT; O=gpurun_out/final4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputests.log
timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_config3.json 2> $O/bench_config3.err
timeout 900 python tests/manual/gpu_chunk_sweep.py config3skew 6000 > $O/sweep_skew.log 2>&1
LTR_DEBUG=1 timeout 600 python tests/manual/gpu_calc_hap_aln_probs_rate.py 6000 config3skew > $O/rate_skew.log 2> $O/rate_skew.err
grep -v "^\[ltr  " $O/rate_skew.err | tail -14 > $O/rate_skew.timeline; grep "^\[ltr  " $O/rate_skew.err | grep -v "launched" | tail -12 > $O/rate_skew.plan; rm $O/rate_skew.err
tail -3 $O/gputests.log; grep " N " $O/sweep_skew.log; cat $O/rate_skew.log $O/rate_skew.timeline $O/rate_skew.plan; python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/final4/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'mism', d.get('oracle_check',{}).get('mismatches'), 'e2e', d.get('loci_per_s_end_to_end'), 'plan_create_s', d.get('plan_create_s'))
    except Exception as e: print(f, 'ERR', e)
P
