"""Manual GPU check (not a pytest file): per-locus latency of the drop-in calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from longtr_amd import _abi, _lib, synth
ctx = _lib.Context(0)
loci, _ = synth.config_loci("config2")
batch, _ = synth.pack_loci(loci)
for name, fn in [("align_batch (plan create + execute + fetch + destroy)", lambda: ctx.align_batch(batch))]:
    fn(); fn()
    t0 = time.perf_counter(); N = 50
    for _ in range(N): fn()
    print(f"{name}: {(time.perf_counter()-t0)/N*1e3:.3f} ms per call, {batch.ll_size} pairs")
plan = ctx.plan(batch)
plan.execute(); plan.fetch()
t0 = time.perf_counter()
for _ in range(200): plan.execute()
plan.fetch()
print(f"plan.execute only: {(time.perf_counter()-t0)/200*1e3:.3f} ms")
plan.execute(); print("kernel ms of one execute (HIP events, ctrl reset excluded):", plan.last_kernel_ms())
for mode in (0, 2, 3):
    ctx.set_pair_packing(mode)
    p3 = ctx.plan(batch); p3.execute(); p3.fetch()
    t0 = time.perf_counter()
    for _ in range(200): p3.execute()
    p3.fetch()
    dt = (time.perf_counter()-t0)/200*1e3
    p3.execute(); print(f"packing mode {mode}: plan.execute {dt:.3f} ms; kernel ms {p3.last_kernel_ms()}"); p3.close()
ctx.set_pair_packing(-1)
t0 = time.perf_counter()
for _ in range(50):
    p2 = ctx.plan(batch); p2.close()
print(f"plan create+destroy: {(time.perf_counter()-t0)/50*1e3:.3f} ms")
L = loci[0]
alns = L.raw_alns if L.raw_alns else None
if alns:
    ctx.process_reads(L.blocks(), alns); 
    t0 = time.perf_counter()
    for _ in range(50): ctx.process_reads(L.blocks(), alns)
    print(f"process_reads (trim + plan + DP): {(time.perf_counter()-t0)/50*1e3:.3f} ms per locus, {len(alns)} reads")
