# after a kernel change late in the round: the GPU suite, a fuzz run, the counters + kernel stats of the default bench (config 3), the
# default and the catalogue bench lines with those counters attached, the plan-size curve.  The other files of profiles/r05 stay as they are.
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r05_partial}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/gputests.log
timeout 200 python tests/manual/gpu_fuzz.py 90 91 2>&1 | tail -1 > $O/fuzz.log
bash profiles/collect.sh r05 > $O/collect.log 2>&1
python3 profiles/make_traffic.py gpurun_out/prof_r05/summary.json profiles/r05 | tail -1 >> $O/collect.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_config3.json 2> $O/bench_config3.err; cp bench_detail.json $O/bench_detail_config3.json
timeout 900 python bench.py --workload catalogue --no-cpu-baseline --no-neighbours --steps 5 --warmup 1 > $O/bench_catalogue.json 2> $O/bench_catalogue.err; cp bench_detail_catalogue.json $O/
timeout 600 python tests/manual/gpu_plan_size.py > $O/plan_size.log 2>&1
timeout 300 python tests/manual/gpu_plan_size.py catalogue 100000 > $O/plan_size_catalogue.log 2>&1
cat $O/gputests.log $O/fuzz.log; grep -v amdgpu.ids $O/plan_size.log $O/plan_size_catalogue.log | cut -c1-170
python - <<P
import json
for f in ("bench_config3","bench_catalogue"):
    d=json.loads([l for l in open("$O/%s.json"%f) if l.startswith('{')][-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('whole_pass_frac'), d['roofline'].get('traffic'), d.get('end_to_end_frac_of_resident'), d.get('oracle_check'))
P
