for w in catalogue config3; do for r in 0 1 2; do
timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-neighbours --no-end-to-end --no-verify --debug pack_rule=$r > gpurun_out/r04_pr_${w}_$r.log 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_pr_${w}_$r.log").read().strip().splitlines()[-1])
print("$w rule $r ms", round(d["ms_per_step"],2), "loci/s", round(d["loci_per_s"]))
import shutil; shutil.copy(d["detail"], "gpurun_out/r04_pr_${w}_$r.detail.json")
PY
done; done
