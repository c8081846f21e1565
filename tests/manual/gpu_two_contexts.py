"""Manual GPU check: what the start and the end of a plan cost.  Shard 0 of config 3 cost-sharded N ways as ONE resident plan
executed back to back, against TWO contexts (own streams, own side lanes) with a resident plan of the same batch each, executes
alternating: the head of pass k + 1 can fill the end of pass k without sharing a launch stream with it.
    python tests/manual/gpu_two_contexts.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from longtr_amd import _lib, shard, synth
NS = [int(x) for x in sys.argv[1:]] or [1, 8, 16]
hdr = synth.config_headers("config3", n_loci=10000)
costs = shard.header_time_costs(hdr)
ctxs = [_lib.Context(0), _lib.Context(0)]
K = 8
for n in NS:
    ids = shard.shard_by_cost(costs, n)[0]
    loci, _ = synth.config_loci("config3", n_loci=10000, ids=ids)
    batch, _ = synth.pack_loci(loci)
    plans = [c.plan(batch) for c in ctxs]
    res = {}
    for depth in (1, 2, 1, 2):
        for w in range(2): plans[w % depth].execute()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K): plans[k % depth].execute()
        torch.cuda.synchronize()
        res.setdefault(depth, []).append((time.perf_counter() - t0) / K)
    print(f"config3 shard of {len(ids)} loci: one plan {min(res[1])*1e3:.2f} ms per pass ({plans[0].cells/min(res[1]):.3e} cells/s); two contexts alternating "
          f"{min(res[2])*1e3:.2f} ms per pass ({plans[0].cells/min(res[2]):.3e} cells/s)", flush=True)
    for p in plans: p.close()
