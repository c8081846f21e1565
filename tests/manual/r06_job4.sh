# round 6, fourth GPU run (short): where the general model's 1250-locus plan ends (wave clocks), the extended fuzz, the seeded path kernel by kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_job4}; mkdir -p $O
ASYM="-1.2,-0.3,-0.9,-0.5,-0.0001,-5.0,-4.0"
timeout 300 python tests/manual/gpu_wave_clock.py config3 8 $ASYM 2>&1 | grep -v amdgpu.ids > $O/wave_clock_1250_asym.log
timeout 300 python tests/manual/gpu_wave_clock.py config3 8 2>&1 | grep -v amdgpu.ids > $O/wave_clock_1250.log
for s in 71 72; do timeout 400 python tests/manual/gpu_fuzz.py 120 $s 2>&1 | tail -2; done > $O/fuzz.log 2>&1
timeout 300 python -m pytest tests/test_gpu_wg.py tests/test_gpu_scale.py -m gpu -q -x -k "threshold or learnt or config5 or plan_kernel" 2>&1 | tail -4 > $O/gputests_subset.log
timeout 300 python - > $O/short_split.log 2>&1 <<P
import sys, json
sys.path.insert(0, '.')
import bench
from longtr_amd import _lib
ctx = _lib.Context(0)
info = ctx.device_info()
peak = info["n_cu"] * 64 * info["clock_mhz"] * 1e6 / 1e12
out = bench.neighbours(ctx, peak)
print(json.dumps(out["short_path"], indent=1)); print(json.dumps(out["nw"], indent=1))
P
cat $O/fuzz.log $O/gputests_subset.log; head -4 $O/wave_clock_1250_asym.log | cut -c1-700; sed -n 5,60p $O/wave_clock_1250_asym.log | cut -c1-160; tail -30 $O/short_split.log
