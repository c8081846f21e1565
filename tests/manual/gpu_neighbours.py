"""Manual GPU check: bench.py's `neighbours` block alone (NW + seeded stutter path kernels)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from longtr_amd import _lib
ctx = _lib.Context(0)
info = ctx.device_info()
peak = info["n_cu"] * 64 * info["clock_mhz"] * 1e6 / 1e12
print(json.dumps(bench.neighbours(ctx, peak), indent=1))
