"""Manual GPU probe: the multi-width launches (one-wave 11..20, packed 13..20) on a small batch, against a launch per class (no_multi)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _abi, _lib
rng = np.random.default_rng(1)
ctx = _lib.Context(0)
def seq(n): return bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
loci = []
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for M in (700, 930, 1100, 420, 500, 610, 150, 60):
    core = seq(M); hap = seq(30) + core + seq(30)
    reads = []
    for _ in range(8):
        r = bytearray(core)
        for p in rng.choice(M, size=3, replace=False): r[p] = ord("A") if r[p] != ord("A") else ord("C")
        reads.append(bytes(r))
    for i in range(NP * (8 if M < 650 else 1)): loci.append((reads, [hap]))
batch = _abi.PackedBatch(loci)
out = {}
for nm in (1, 0):
    ctx.set_debug("no_multi", nm)
    plan = ctx.plan(batch)
    t0 = time.perf_counter(); plan.execute(); ll, _ = plan.fetch(); dt = time.perf_counter() - t0
    out[nm] = ll.copy()
    st = [k for k in plan.kernel_stats() if k["pairs"]]
    print("no_multi", nm, f"{dt*1e3:.2f} ms", [(k["family"], k["strip_width"], k["pairs"], k.get("ranges")) for k in st], flush=True)
    plan.close()
print("bits equal", bool(np.array_equal(out[0].view(np.uint64), out[1].view(np.uint64))))
