"""Manual GPU check: ONE launch class, N pairs -- pass time against N (rounds of resident wavefronts, ramp, tail).
Pairs: a random 930-base read against a 990-base haplotype carrying it (class one-wave W = 15), mode 3 (one pair per wave)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _abi, _lib

rng = np.random.default_rng(1)
ctx = _lib.Context(0)
ctx.set_pair_packing(3)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 930
def seq(n): return bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
core = seq(M)
hap = seq(30) + core + seq(30)
reads = []
for _ in range(64):
    r = bytearray(core)
    for p in rng.choice(M, size=3, replace=False): r[p] = ord("A") if r[p] != ord("A") else ord("C")
    reads.append(bytes(r))
slots = ctx.device_info()["n_cu"] * 4 * 3
for mult in (0.25, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 4.5, 5.0, 8.0, 16.0, 40.0):
    N = int(slots * mult)
    loci = [([reads[(i + j) % 64] for j in range(8)], [hap]) for i in range(N // 8)]
    batch = _abi.PackedBatch(loci)
    plan = ctx.plan(batch)
    plan.execute(); plan.wait()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); plan.execute(); plan.wait(); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print(f"read {M}: {N:7d} pairs = {mult:5.2f} x {slots} wave slots: {dt*1e3:7.3f} ms per pass, {dt*1e3/ max(mult,1e-9):6.3f} ms per round-equivalent, {plan.cells/dt:.3e} cells/s", flush=True)
    plan.close()
