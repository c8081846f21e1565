cd $GRAFT_REPO_ROOT; O=gpurun_out/onegpu; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k two_ranks 2>&1 | tail -30 > $O/test.log
timeout 900 python bench.py --gpus 2 --one-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
cat $O/test.log; python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/onegpu/bench_2ranks_one_gpu.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('n_gpus','value','ms_per_step','scaling','debug_one_gpu','single_gpu_check','oracle_check','weak_scaling')})
P
tail -5 $O/bench_2ranks.err
