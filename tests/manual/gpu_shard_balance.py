"""Manual GPU check: how evenly the N = 8 shards of BASELINE config 4 run -- every shard's resident plan timed on this GPU --
for the greedy LPT assignment on the modelled costs and for dealing the cost-sorted loci out in snake order.
    python tests/manual/gpu_shard_balance.py [n_ranks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _lib, shard, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hdr = synth.config_headers("config3", n_loci=10000)
costs = np.asarray(shard.header_time_costs(hdr))
ctx = _lib.Context(0)

def snake(costs, world):
    order = np.argsort(-costs, kind="stable")
    shards = [[] for _ in range(world)]
    for i, l in enumerate(order):
        k = i % (2 * world)
        shards[k if k < world else 2 * world - 1 - k].append(int(l))
    return [sorted(s) for s in shards]

def stratified(costs, world, strata):
    """Greedy LPT over the loci ordered by stratum (strip-width class of the locus's reads), largest first inside a stratum:
    every rank gets the same share of EVERY launch class, not only the same total."""
    import heapq
    order = np.lexsort((-costs, strata))
    heap = [(0.0, r) for r in range(world)]; heapq.heapify(heap)
    shards = [[] for _ in range(world)]
    for l in order:
        load, r = heapq.heappop(heap); shards[r].append(int(l)); heapq.heappush(heap, (load + float(costs[l]), r))
    return [sorted(s) for s in shards]

strata = -((hdr[:, 0] + 19) // 64)                     # ~ strip width of the locus's reads (TR + pads + flanks columns), widest first
for name, parts in (("lpt", shard.shard_by_cost(costs, N)), ("stratified", stratified(costs, N, strata))):
    times, model, cells = [], [], []
    for r in range(N):
        loci, _ = synth.config_loci("config3", n_loci=10000, ids=parts[r])
        batch, _ = synth.pack_loci(loci)
        plan = ctx.plan(batch)
        plan.execute(); plan.wait()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(4): plan.execute()
            plan.wait(); ts.append((time.perf_counter() - t0) / 4)
        times.append(min(ts) * 1e3); model.append(float(costs[parts[r]].sum())); cells.append(plan.cells)
        plan.close()
    m = np.asarray(model)
    print(f"{name}: loci per shard {[len(p) for p in parts]}; modelled cost spread {m.max()/m.mean()-1:+.4f}; pass ms {[round(t, 2) for t in times]}; "
          f"max/mean {max(times)/np.mean(times):.4f}; cells/1e10 {[round(c / 1e10, 3) for c in cells]}; Tcells/s {[round(c / t / 1e9, 3) for c, t in zip(cells, times)]}", flush=True)
