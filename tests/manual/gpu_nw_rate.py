"""Manual GPU check: ltr_haplotype_align_to_ref (NeedlemanWunsch::Align + adjust_indels for every
haplotype of every locus in one call) on config-3 loci: cells/s of the NW neighbour next to the DP."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from longtr_amd import _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
loci, desc = synth.config_loci("config3", n_loci=N)
ctx = _lib.Context(0)
blocks = [L.blocks() for L in loci]
cells = sum(len(L.haplotypes[0]) * sum(len(h) for h in L.haplotypes) for L in loci)
nh = sum(len(L.haplotypes) for L in loci)
packed = ctx.pack_haplotypes(blocks)
ctx.haplotype_align_to_ref_packed(packed, decode=False)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); ctx.haplotype_align_to_ref_packed(packed, decode=False); ts.append(time.perf_counter() - t0)
dt = min(ts)
print(f"{desc}: ltr_haplotype_align_to_ref {nh} haplotypes, {cells:.3e} NW cells, {dt*1e3:.1f} ms per call (the C call: packing, kernels, traceback, adjust_indels), {cells/dt:.3e} cells/s, {N/dt:.0f} loci/s")
if len(sys.argv) > 2:                                            # any second argument: one call with the library's phase prints
    ctx.set_debug("trace", 1); ctx.haplotype_align_to_ref_packed(packed, decode=False); ctx.set_debug("trace", 0)
