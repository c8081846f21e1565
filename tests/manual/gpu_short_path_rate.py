"""Manual GPU check (not a pytest file): throughput of the seeded stutter path (a-7) -- period-1 loci
through ltr_calc_hap_aln_probs, against the CPU restatement on a sample."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from longtr_amd import _abi, _lib, synth
import oracle_lib as ol
import short_util as su

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(5)
prm = _abi.default_params(); prm.use_short_path = 1
ctx = _lib.Context(0, prm)
if len(sys.argv) > 2:
    ctx.set_debug("short_lane_kernel", int(sys.argv[2]))       # 1: the lane-per-pair kernel (A/B)
loci = []
cells = 0
for _ in range(N):
    tr, H, R = int(rng.integers(10, 60)), int(rng.integers(2, 5)), 30
    blocks, alns = su.homopolymer_locus(rng, tr, H, R)
    loci.append((blocks, alns))
    hap_len = sum(len(b["alleles"][0]) for b in blocks)
    cells += sum(len(a["seq"]) for a in alns) * hap_len * H
print(f"{N} loci, ~{cells:.3e} read x haplotype cells")
packed = ctx.pack_loci(loci)                       # the Python-side ctypes image is not part of the measurement
out = ctx.calc_hap_aln_probs_packed(packed)
if len(sys.argv) > 3:
    ctx.set_debug("trace", 1); out = ctx.calc_hap_aln_probs_packed(packed); ctx.set_debug("trace", 0)
t0 = time.perf_counter()
for _ in range(3): out = ctx.calc_hap_aln_probs_packed(packed)
dt = (time.perf_counter() - t0) / 3
print(f"GPU (host prep + kernel + scatter): {dt*1e3:.1f} ms per batch, {cells/dt:.3e} cells/s, {N/dt:.0f} loci/s")
sp = _abi.default_stutter_params()
t0 = time.perf_counter(); c = 0
for (Lb, La) in loci[:10]:
    ol.oracle_process_reads_short(prm, sp, Lb, La)
    c += sum(len(a["seq"]) for a in La) * sum(len(b["alleles"][0]) for b in Lb) * len(Lb[1]["alleles"])
dt2 = time.perf_counter() - t0
print(f"CPU restatement (no pooling): {c/dt2:.3e} cells/s on 10 loci")
