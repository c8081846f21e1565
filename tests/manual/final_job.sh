# round-end measurement set: GPU tests, the default bench line, the other workloads, smoke, profiles
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_config3.json 2> $O/bench_config3.err
for w in config2 config5 config5hifi config3skew; do timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 600 python bench.py --pair-packing 4 --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 > $O/bench_config3_exact_only.json 2> $O/bench_config3_exact_only.err
bash profiles/collect.sh r02d > $O/collect.log 2>&1
tail -3 $O/gputests.log; cat $O/smoke.log | tail -2; python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/final/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.2f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'mism', d.get('oracle_check',{}).get('mismatches'), 'e2e', d.get('loci_per_s_end_to_end'))
    except Exception as e: print(f, 'ERR', e)
P
