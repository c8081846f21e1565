"""Manual GPU run for profiling: shard 0 of config 3 cost-sharded N ways, a few resident passes.
    python tests/manual/gpu_one_shard.py [N] [passes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from longtr_amd import _lib, shard, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hdr = synth.config_headers("config3", n_loci=10000)
ids = shard.shard_by_cost(shard.header_time_costs(hdr), N)[0]
loci, _ = synth.config_loci("config3", n_loci=10000, ids=ids)
batch, _ = synth.pack_loci(loci)
ctx = _lib.Context(0)
plan = ctx.plan(batch)
for _ in range(P):
    plan.execute(); plan.wait(); time.sleep(0.05)
print("cells", plan.cells)
