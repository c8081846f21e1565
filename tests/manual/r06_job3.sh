# round 6, third GPU run: band skipping in the threshold bodies (A/B against builds without it), merged threshold launches, the general
# model's threshold redo inside the plan kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_job3}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/gputests.log
ASYM="-1.2,-0.3,-0.9,-0.5,-0.0001,-5.0,-4.0"
for lib in default wgt_noband wgt_lb8; do
  if [ $lib = default ]; then unset LTR_GPU_LIB; else export LTR_GPU_LIB=$PWD/abtest/$lib.so; fi
  for tr in 3000 5000 7400; do timeout 300 python tests/manual/gpu_long_vntr_rate.py $tr 48 -1 2 2>&1 | grep -v amdgpu.ids | sed "s/^/[$lib] /"; done
  timeout 400 python bench.py --workload config5hifi --no-cpu-baseline --no-neighbours --no-end-to-end --no-verify --steps 5 --warmup 2 --debug wg_first_pass=2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$lib] config5hifi fp2 %.2f ms %.3e' % (d['ms_per_step'], d['value']))"
  timeout 400 python bench.py --workload config5 --no-cpu-baseline --no-neighbours --no-end-to-end --no-verify --steps 5 --warmup 2 --debug wg_first_pass=2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$lib] config5 fp2 %.3f ms' % d['ms_per_step'])"
done > $O/band_ab.log 2>&1
for lib in default exact_noband; do
  if [ $lib = default ]; then unset LTR_GPU_LIB; else export LTR_GPU_LIB=$PWD/abtest/$lib.so; fi
  timeout 400 python bench.py --workload config5hifi --pair-packing 4 --no-cpu-baseline --no-end-to-end --no-neighbours --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$lib] config5hifi exact-only %.2f ms %.3e whole %.3f mism %s' % (d['ms_per_step'], d['value'], d['roofline']['whole_pass_frac'], d.get('oracle_check',{}).get('mismatches')))"
done >> $O/band_ab.log 2>&1
unset LTR_GPU_LIB
timeout 900 python tests/manual/gpu_plan_size.py config3 10000 0 $ASYM 2>&1 | grep -v amdgpu.ids > $O/plan_size_asym.log
timeout 600 python tests/manual/gpu_plan_size.py config3 10000 0 2>&1 | grep -v amdgpu.ids > $O/plan_size.log
tail -4 $O/gputests.log; cat $O/band_ab.log $O/plan_size_asym.log $O/plan_size.log
