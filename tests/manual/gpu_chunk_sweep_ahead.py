"""Manual GPU check: ltr_calc_hap_aln_probs, chunk count x growth law with the next chunk prepared ahead (round 5: the host side of
a chunk is shorter than its GPU side on the catalogue, so a short first chunk shortens the lead-in).  Visited in two orders.
    python tests/manual/gpu_chunk_sweep_ahead.py <workload> <N>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from longtr_amd import _lib, synth

WL = sys.argv[1] if len(sys.argv) > 1 else "catalogue"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
loci, desc = synth.config_loci(WL, n_loci=N, raw=True)
ctx = _lib.Context(0)
packed = ctx.pack_loci([(L.blocks(), L.raw_alns) for L in loci])
ctx.calc_hap_aln_probs_packed(packed)
combos = [(None, None, 0), (3, 1.0, -1), (3, 1.0, 0), (3, 1.3, 0), (4, 1.0, 0), (4, 1.3, 0), (4, 1.6, 0), (5, 1.0, 0), (5, 1.3, 0), (5, 1.6, 0), (6, 1.3, 0), (6, 1.5, 0), (8, 1.3, 0), (4, 1.3, -1), (5, 1.3, -1)]
for order in (combos, combos[::-1]):
    for ch, gr, ahead in order:
        ctx.set_debug("reset", 0)
        if ch is not None: ctx.set_debug("chunks", ch); ctx.set_debug("chunk_growth", gr)
        ctx.set_debug("prep_ahead", ahead)
        ctx.calc_hap_aln_probs_packed(packed)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.calc_hap_aln_probs_packed(packed); ts.append(time.perf_counter() - t0)
        print(f"{WL} N {N} chunks {ch} growth {gr} prep_ahead {ahead}: best {min(ts)*1e3:.1f} ms (median {sorted(ts)[2]*1e3:.1f}), {N/min(ts):.0f} loci/s", flush=True)
ctx.set_debug("reset", 0)
