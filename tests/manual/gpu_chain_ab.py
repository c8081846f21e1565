"""Manual GPU check: the chained walk of the one-wave classes (ltr_dp_chain.hpp) against the plain one, per strip width.
    python tests/manual/gpu_chain_ab.py [workload] [shards]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_util
from longtr_amd import _lib, shard, synth
WL = sys.argv[1] if len(sys.argv) > 1 else "config3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
NL = synth._DEFAULT_N[WL]
parts = shard.shard_by_cost(shard.header_time_costs(synth.config_headers(WL, n_loci=NL)), N)
loci, _ = synth.config_loci(WL, n_loci=NL, ids=parts[0])
batch, _ = synth.pack_loci(loci)
ctx = _lib.Context(0)
ref = None
variants = [("plain", {}), ("chained 11..20", {"chain": 1}), ("chained 11..14", {"chain": 1, "chain_max_w": 14}), ("chained 11..15", {"chain": 1, "chain_max_w": 15}), ("chained 15..15", {"chain": 1, "chain_min_w": 15, "chain_max_w": 15}), ("chained 16..20", {"chain": 1, "chain_min_w": 16}),
            ("plain", {}), ("chained 11..20", {"chain": 1})]
for name, knobs in variants:
    ctx.set_debug("reset", 0)
    for k, v in knobs.items():
        ctx.set_debug(k, v)
    plan = ctx.plan(batch)
    plan.execute(); ll = plan.fetch()[0]
    if ref is None:
        ref = ll
        res = parity_util.stratified_oracle_check(batch, ll, ctx.params, n_loci_target=40)
        print("plain vs oracle:", res["mismatches"], "of", res["checked_pairs"], flush=True)
    bad = int((ll.view(np.uint64) != ref.view(np.uint64)).sum())
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(3): plan.execute()
        plan.wait(); ts.append((time.perf_counter() - t0) / 3)
    print(f"{WL} shard of {len(parts[0])} loci ({plan.num_pairs} pairs), {name}: {min(ts)*1e3:.2f} ms per pass, {plan.cells/min(ts):.3e} cells/s, pairs that differ from plain: {bad}", flush=True)
    plan.close()
ctx.set_debug("reset", 0)
