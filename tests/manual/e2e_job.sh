cd $GRAFT_REPO_ROOT; O=gpurun_out/final8; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gputests.log
timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_config3.json 2> $O/bench_config3.err
for w in config2 config5 config5hifi config3skew; do timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 600 python tests/manual/gpu_chunk_sweep.py config3 6000 > $O/sweep.log 2>&1
tail -2 $O/gputests.log; python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/final8/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'whole', round(d['roofline']['whole_pass_frac'],3), 'mism', d.get('oracle_check',{}).get('mismatches'), 'e2e', d.get('loci_per_s_end_to_end'))
    except Exception as e: print(f, 'ERR', e)
P
grep " N " $O/sweep.log | head -10
