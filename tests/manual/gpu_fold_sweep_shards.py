"""Manual GPU check: the fold threshold (ltr_ctx_set_debug "fold_rounds") on shards of config 3 -- fewer, longer launches
against slack columns.    python tests/manual/gpu_fold_sweep_shards.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _lib, shard, synth

NS = [int(x) for x in sys.argv[1:]] or [1, 8, 16]
hdr = synth.config_headers("config3", n_loci=10000)
costs = shard.header_time_costs(hdr)
ctx = _lib.Context(0)
for n in NS:
    ids = shard.shard_by_cost(costs, n)[0]
    loci, _ = synth.config_loci("config3", n_loci=10000, ids=ids)
    batch, _ = synth.pack_loci(loci)
    ref = None
    for fr in (6, 12, 18, 24, 36, 48, 6):
        ctx.set_debug("fold_rounds", fr)
        plan = ctx.plan(batch)
        plan.execute(); ll, _ = plan.fetch()
        if ref is None: ref = ll.copy()
        same = bool(np.array_equal(ref.view(np.uint64), ll.view(np.uint64)))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(4): plan.execute()
            plan.wait(); ts.append((time.perf_counter() - t0) / 4)
        st = [k for k in plan.kernel_stats() if k["pairs"] and k["family"] != "exact"]
        print(f"shard of {len(ids)} loci, fold_rounds {fr}: {min(ts)*1e3:.2f} ms per pass, {plan.cells/min(ts):.3e} cells/s, {len(st)} certificate launches, bits equal {same}", flush=True)
        plan.close()
