cd $GRAFT_REPO_ROOT; O=gpurun_out/final9; mkdir -p $O
bash profiles/collect.sh r02f > $O/collect.log 2>&1
timeout 900 python bench.py > $O/bench_config3.json 2> $O/bench_config3.err
for w in config2 config5 config5hifi config3skew; do timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 600 python bench.py --pair-packing 4 --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 > $O/bench_config3_exact_only.json 2> $O/bench_config3_exact_only.err
timeout 900 python bench.py --gpus 2 --one-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
tail -c 300 $O/collect.log; python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/final9/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'mism', d.get('oracle_check',{}).get('mismatches'), 'e2e', d.get('loci_per_s_end_to_end'), d.get('single_gpu_check'))
    except Exception as e: print(f, 'ERR', e)
P
