"""Manual GPU check: the plan kernel (one launch per plan) against round 4's launches on shards of a workload, and its knobs.
    python tests/manual/gpu_plan_ab.py [workload] [shards ...]   e.g. config3 8 16"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _lib, shard, synth

WL = sys.argv[1] if len(sys.argv) > 1 else "config3"
NS = [int(x) for x in sys.argv[2:]] or [8, 16]
NL = synth._DEFAULT_N[WL]
hdr = synth.config_headers(WL, n_loci=NL)
costs = shard.header_time_costs(hdr)
ctx = _lib.Context(0)
for n in NS:
    parts = shard.shard_by_cost(costs, n)
    loci, _ = synth.config_loci(WL, n_loci=NL, ids=parts[0])
    batch, _ = synth.pack_loci(loci)
    ref = None
    for name, knobs in (("classic", {"plan_kernel": 1}), ("plan kernel", {"plan_kernel": -1}), ("plan kernel, no shares", {"plan_kernel": -1, "plan_share": 1}), ("classic", {"plan_kernel": 1}), ("plan kernel", {"plan_kernel": -1}), ("by rule", {})):
        ctx.set_debug("reset", 0)
        for k, v in knobs.items():
            ctx.set_debug(k, v)
        plan = ctx.plan(batch)
        plan.execute(); ll = plan.fetch()[0]
        if ref is None: ref = ll
        same = bool(np.array_equal(ll.view(np.uint64), ref.view(np.uint64)))
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            for _ in range(4): plan.execute()
            plan.wait(); ts.append((time.perf_counter() - t0) / 4)
        one = []
        for _ in range(4):
            t0 = time.perf_counter(); plan.execute(); plan.wait(); one.append(time.perf_counter() - t0)
        print(f"{WL} shard of {len(parts[0])} loci ({plan.num_pairs} pairs), {name}: {min(ts)*1e3:.2f} ms per pass back to back, {min(one)*1e3:.2f} ms alone, "
              f"{plan.cells/min(ts):.3e} cells/s, bits equal {same}", flush=True)
        plan.close()
ctx.set_debug("reset", 0)
