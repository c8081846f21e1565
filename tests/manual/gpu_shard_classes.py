import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _lib, shard, synth
hdr = synth.config_headers("config3", n_loci=10000); costs = np.asarray(shard.header_time_costs(hdr))
parts = shard.shard_by_cost(costs, 8); ctx = _lib.Context(0)
for r in (0, 4):
    loci, _ = synth.config_loci("config3", n_loci=10000, ids=parts[r]); batch, _ = synth.pack_loci(loci)
    plan = ctx.plan(batch); plan.execute(); plan.wait()
    t0 = time.perf_counter(); [plan.execute() for _ in range(4)]; plan.wait(); dt = (time.perf_counter() - t0) / 4
    plan.set_timing(True); plan.execute(); plan.fetch()
    st = [k for k in plan.kernel_stats() if k["pairs"]]
    print(f"shard {r}: {dt*1e3:.2f} ms; sum of launches {sum(k['ms'] for k in st):.2f} ms")
    print("   ", [(k["family"][:4], k["lanes_per_pair"], k["strip_width"], k["pairs"], round(k["ms"], 2)) for k in st if k["ms"] > 0.4])
    plan.close()
