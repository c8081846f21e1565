"""Manual GPU check (not a pytest file): the top-level drop-in -- ltr_calc_hap_aln_probs on raw
alignments (pool + trim + pack + upload + DP + scatter) -- for config-3-like loci: end-to-end
loci/s including all host work, next to the resident-kernel rate of bench.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from longtr_amd import _abi, _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
WL = sys.argv[2] if len(sys.argv) > 2 else "config3"
loci, desc = synth.config_loci(WL, n_loci=N, raw=True)
ctx = _lib.Context(0)
items = [(L.blocks(), L.raw_alns) for L in loci]
cells = sum(sum(len(r) for r in L.trimmed_reads) * sum(len(h) - 60 for h in L.haplotypes) for L in loci)
packed = ctx.pack_loci(items)
ctx.calc_hap_aln_probs_packed(packed)
if len(sys.argv) > 3:                                            # any third argument: one call with the library's phase prints
    ctx.set_debug("trace", 1); ctx.calc_hap_aln_probs_packed(packed); ctx.set_debug("trace", 0)
t0 = time.perf_counter()
for _ in range(3): ctx.calc_hap_aln_probs_packed(packed)
dt = (time.perf_counter() - t0) / 3
print(f"{desc}: ltr_calc_hap_aln_probs {dt*1e3:.1f} ms per call, {N/dt:.0f} loci/s, ~{cells/dt:.3e} cells/s (unpooled count)")
