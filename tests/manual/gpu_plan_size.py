"""Manual GPU check: resident-plan rate of ONE rank's share of BASELINE config 3 under strong scaling -- the 10 000 loci
cost-sharded N ways from the generator's headers (as bench.py --gpus N does), shard 0 scored on this GPU -- for
N = 1, 2, 4, 8, 16: the single-GPU ceiling of the scaling curve (rate(N) / rate(1)), before any gather.
    python tests/manual/gpu_plan_size.py [workload] [n_loci] [plan_kernel knob: 0 rule (default), 1 off = round 4's launches] [a,b,c,d,e,f,g]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _abi, _lib, shard, synth

WL = sys.argv[1] if len(sys.argv) > 1 else "config3"
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
hdr = synth.config_headers(WL, n_loci=NL)
costs = shard.header_time_costs(hdr)
ctx = _lib.Context(0)
KNOB = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx.set_debug("plan_kernel", KNOB)
if len(sys.argv) > 4:
    ctx.set_params(_abi.make_params(tuple(float(x) for x in sys.argv[4].split(","))))
    print("alignment params", sys.argv[4], flush=True)
base = None
for n in (1, 2, 4, 8, 16):
    parts = shard.shard_by_cost(costs, n)
    rates, loads = [], []
    for r in sorted({0, n - 1}):
        loci, _ = synth.config_loci(WL, n_loci=NL, ids=parts[r])
        batch, _ = synth.pack_loci(loci)
        plan = ctx.plan(batch)
        plan.execute(); plan.wait()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(4): plan.execute()
            plan.wait(); ts.append((time.perf_counter() - t0) / 4)
        rates.append(plan.cells / min(ts)); loads.append(min(ts))
        plan.close()
    if base is None: base = rates[0]
    print(f"[plan_kernel {KNOB}] {WL} {NL} loci over {n} ranks: shard of {len(parts[0])} loci: {rates[0]:.3e} cells/s = {rates[0]/base:.3f} of the full plan's rate; "
          f"pass time of shards 0 / {n-1}: " + " / ".join(f"{x*1e3:.2f} ms" for x in loads), flush=True)
