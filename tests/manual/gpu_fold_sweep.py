import sys, time, json
sys.path.insert(0,'/root/repo')
import numpy as np
from longtr_amd import _lib, synth
loci,_=synth.config_loci("catalogue", n_loci=int(sys.argv[1]) if len(sys.argv)>1 else 100000)
batch,_=synth.pack_loci(loci)
ctx=_lib.Context(0)
ref=None
for fr in (3, 6, 12, 24, 48):
    ctx.set_debug("fold_rounds", fr)
    plan=ctx.plan(batch); plan.execute(); ll,_=plan.fetch()
    if ref is None: ref=ll.copy()
    bad=int((ll.view(np.uint64)!=ref.view(np.uint64)).sum())
    ts=[]
    for _ in range(3):
        t0=time.perf_counter()
        for _ in range(5): plan.execute()
        plan.wait(); ts.append((time.perf_counter()-t0)/5)
    st=[k for k in plan.kernel_stats() if k["pairs"] and k["family"]!="exact"]
    print("fold_rounds", fr, "ms %.2f"%(min(ts)*1e3), "cells/s %.3e"%(plan.cells/min(ts)), "launch classes", len(st), "bad", bad, flush=True)
    plan.close()
