"""Manual GPU check: multi-width launches against a launch per class (ltr_ctx_set_debug "no_multi") on shards of a workload.
    python tests/manual/gpu_multi_ab.py [workload] [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _lib, shard, synth

WL = sys.argv[1] if len(sys.argv) > 1 else "config3"
NS = [int(x) for x in sys.argv[2:]] or [1, 8, 16]
NL = synth._DEFAULT_N[WL]
hdr = synth.config_headers(WL, n_loci=NL)
costs = shard.header_time_costs(hdr)
ctx = _lib.Context(0)
for n in NS:
    ids = shard.shard_by_cost(costs, n)[0]
    loci, _ = synth.config_loci(WL, n_loci=NL, ids=ids)
    batch, _ = synth.pack_loci(loci)
    ref = None
    for nm in (1, 0, 1, 0):
        ctx.set_debug("no_multi", nm)
        plan = ctx.plan(batch)
        plan.execute(); ll, _ = plan.fetch()
        if ref is None: ref = ll.copy()
        same = bool(np.array_equal(ref.view(np.uint64), ll.view(np.uint64)))
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            for _ in range(4): plan.execute()
            plan.wait(); ts.append((time.perf_counter() - t0) / 4)
        st = [k for k in plan.kernel_stats() if k["pairs"] and k["family"] != "exact"]
        print(f"{WL} shard of {len(ids)} loci, no_multi {nm}: {min(ts)*1e3:.2f} ms per pass, {plan.cells/min(ts):.3e} cells/s, {len(st)} certificate launches, bits equal {same}", flush=True)
        plan.close()
