"""Manual GPU check: the steal budget of the one-wave launches (ltr_ctx_set_debug "steal_budget") on shards of config 3.
    python tests/manual/gpu_steal_sweep.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from longtr_amd import _lib, shard, synth
import oracle_lib as ol

NS = [int(x) for x in sys.argv[1:]] or [1, 8, 16]
hdr = synth.config_headers("config3", n_loci=10000)
costs = shard.header_time_costs(hdr)
ctx = _lib.Context(0)
for n in NS:
    ids = shard.shard_by_cost(costs, n)[0]
    loci, _ = synth.config_loci("config3", n_loci=10000, ids=ids)
    batch, _ = synth.pack_loci(loci)
    ref = None
    for budget in (0, 1, 2, 4, 1000000, 0, 2):
        ctx.set_debug("steal_budget", budget)
        plan = ctx.plan(batch)
        plan.execute(); ll, _ = plan.fetch()
        if ref is None: ref = ll.copy()
        same = bool(np.array_equal(ref.view(np.uint64), ll.view(np.uint64)))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(4): plan.execute()
            plan.wait(); ts.append((time.perf_counter() - t0) / 4)
        print(f"shard of {len(ids)} loci, steal budget {budget}: {min(ts)*1e3:.2f} ms per pass, {plan.cells/min(ts):.3e} cells/s, bits equal {same}", flush=True)
        plan.close()
