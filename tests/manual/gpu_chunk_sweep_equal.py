"""Manual GPU check: ltr_calc_hap_aln_probs with 2 .. 10 equal chunks (a host-bound call ends with the GPU time of its last chunk).
    python tests/manual/gpu_chunk_sweep_equal.py <workload> <N>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from longtr_amd import _lib, synth
WL = sys.argv[1]; N = int(sys.argv[2])
loci_all, desc = synth.config_loci(WL, n_loci=N, raw=True)
ctx = _lib.Context(0)
packed = ctx.pack_loci([(L.blocks(), L.raw_alns) for L in loci_all])
ctx.calc_hap_aln_probs_packed(packed)
combos = [None, 2, 3, 4, 5, 6, 8, 10]
for order in (combos, combos[::-1]):
    for nc in order:
        ctx.set_debug("reset", 0)
        if nc is not None:
            ctx.set_debug("chunks", nc); ctx.set_debug("chunk_growth", 1.0)
        ctx.calc_hap_aln_probs_packed(packed)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.calc_hap_aln_probs_packed(packed); ts.append(time.perf_counter() - t0)
        print(f"{WL} N {N} chunks {nc}: best {min(ts)*1e3:.1f} ms (mean {sum(ts)/len(ts)*1e3:.1f}) = {N/min(ts):.0f} loci/s", flush=True)
