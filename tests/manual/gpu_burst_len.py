"""Manual GPU check: does the rate of a small resident plan depend on how many passes are queued back to back (clock ramp,
launch gaps)?  Shard 0 of config 3 cost-sharded N ways; K passes per timed burst.    python tests/manual/gpu_burst_len.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from longtr_amd import _lib, shard, synth
NS = [int(x) for x in sys.argv[1:]] or [1, 8, 16]
hdr = synth.config_headers("config3", n_loci=10000)
costs = shard.header_time_costs(hdr)
ctx = _lib.Context(0)
for n in NS:
    ids = shard.shard_by_cost(costs, n)[0]
    loci, _ = synth.config_loci("config3", n_loci=10000, ids=ids)
    batch, _ = synth.pack_loci(loci)
    plan = ctx.plan(batch)
    plan.execute(); plan.wait()
    for K in (1, 2, 4, 16, 64 if n > 1 else 8):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(K): plan.execute()
            plan.wait(); ts.append((time.perf_counter() - t0) / K)
        print(f"shard of {len(ids)} loci, {K:3d} passes per burst: {min(ts)*1e3:.2f} ms per pass, {plan.cells/min(ts):.3e} cells/s", flush=True)
    plan.close()
