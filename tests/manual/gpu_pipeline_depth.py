"""Manual GPU check: what the END of a plan costs.  Shard 0 of config 3 cost-sharded N ways; K passes of ONE resident plan
back to back on one stream (every pass ends in the tail of its last launches and its exact lists) against K passes dealt
over TWO resident plans of the same batch on two streams (the head of pass k + 1 fills the end of pass k).
    python tests/manual/gpu_pipeline_depth.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from longtr_amd import _lib, shard, synth

NS = [int(x) for x in sys.argv[1:]] or [1, 8, 16]
hdr = synth.config_headers("config3", n_loci=10000)
costs = shard.header_time_costs(hdr)
ctx = _lib.Context(0)
K = 8
for n in NS:
    ids = shard.shard_by_cost(costs, n)[0]
    loci, _ = synth.config_loci("config3", n_loci=10000, ids=ids)
    batch, _ = synth.pack_loci(loci)
    plans = [ctx.plan(batch), ctx.plan(batch)]
    outs = [torch.empty(max(p.ll_size, 1), dtype=torch.float64, device="cuda:0") for p in plans]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    res = {}
    for depth in (1, 2, 1, 2):
        for w in range(2):
            plans[w % depth].execute(outs[w % depth].data_ptr(), streams[w % depth].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            plans[k % depth].execute(outs[k % depth].data_ptr(), streams[k % depth].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        res.setdefault(depth, []).append(dt)
    same = bool(torch.equal(outs[0], outs[1]))
    print(f"config3 shard of {len(ids)} loci: one plan {min(res[1])*1e3:.2f} ms per pass ({plans[0].cells/min(res[1]):.3e} cells/s), two plans on two streams "
          f"{min(res[2])*1e3:.2f} ms per pass ({plans[0].cells/min(res[2]):.3e} cells/s), outputs equal {same}", flush=True)
    for p in plans:
        p.close()
