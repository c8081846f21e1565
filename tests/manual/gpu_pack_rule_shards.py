import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from longtr_amd import _lib, shard, synth
hdr = synth.config_headers("config3", n_loci=10000)
costs = shard.header_time_costs(hdr)
ctx = _lib.Context(0)
for n in (8, 16, 4):
    ids = shard.shard_by_cost(costs, n)[0]
    loci, _ = synth.config_loci("config3", n_loci=10000, ids=ids)
    batch, _ = synth.pack_loci(loci)
    ref = None
    for rule in (0, 1, 2, 0, 2):
        ctx.set_debug("pack_rule", rule)
        plan = ctx.plan(batch)
        plan.execute(); ll, _ = plan.fetch()
        if ref is None: ref = ll.copy()
        same = bool(np.array_equal(ref.view(np.uint64), ll.view(np.uint64)))
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            for _ in range(4): plan.execute()
            plan.wait(); ts.append((time.perf_counter() - t0) / 4)
        st = [k for k in plan.kernel_stats() if k["pairs"] and k["family"] == "packed"]
        lanes = {}
        for k in st:
            for lp, w, npairs in k.get("ranges", []): lanes[lp] = lanes.get(lp, 0) + npairs
        print(f"shard of {len(ids)} loci, pack_rule {rule}: {min(ts)*1e3:.2f} ms per pass, {plan.cells/min(ts):.3e} cells/s, bits equal {same}, packed pairs by lanes {dict(sorted(lanes.items()))}", flush=True)
        plan.close()
