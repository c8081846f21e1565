"""Manual GPU check: ltr_calc_hap_aln_probs with the next chunk of loci pooled + trimmed by a thread of its own while the calling
thread plans and launches the current one (ltr_ctx_set_debug "prep_ahead": -1 = off, n = threads of the second pool), interleaved.
    python tests/manual/gpu_prep_ahead_ab.py [n_loci] [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from longtr_amd import _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
WL = sys.argv[2] if len(sys.argv) > 2 else "catalogue"
loci, desc = synth.config_loci(WL, n_loci=N, raw=True)
ctx = _lib.Context(0)
packed = ctx.pack_loci([(L.blocks(), L.raw_alns) for L in loci])
ctx.set_debug("prep_ahead", -1)
ref = [(a.copy(), b.copy()) for a, b in ctx.calc_hap_aln_probs_packed(packed)]       # (the call writes into the same arrays every time)
def same(a, b):
    return all(np.array_equal(np.asarray(x[0]).view(np.uint64), np.asarray(y[0]).view(np.uint64)) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))
for rep in range(2):
    for name, v in (("one chunk after the other", -1), ("ahead, 4 threads", 4), ("ahead, 8 threads", 8), ("ahead, 12 threads", 12), ("ahead, 16 threads", 16)):
        ctx.set_debug("prep_ahead", v)
        for a, b in packed["outs"]:
            a[:] = np.nan; b[:] = -12345
        out = ctx.calc_hap_aln_probs_packed(packed)
        ok = same(out, ref)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.calc_hap_aln_probs_packed(packed); ts.append(time.perf_counter() - t0)
        print(f"{desc}: {name}: {min(ts)*1e3:.1f} ms per call (median {sorted(ts)[2]*1e3:.1f}), {N/min(ts):.0f} loci/s, rows and seeds equal: {ok}", flush=True)
ctx.set_debug("prep_ahead", 0)
if len(sys.argv) > 3:
    ctx.set_debug("trace", 1); ctx.calc_hap_aln_probs_packed(packed); ctx.set_debug("trace", 0)
