"""Manual GPU check: ltr_calc_hap_aln_probs with the library's staging arrays pinned (rule) against pageable (A/B switch
"pageable_staging"), same box, alternating.    python tests/manual/gpu_staging_ab.py [N] [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from longtr_amd import _lib, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
WL = sys.argv[2] if len(sys.argv) > 2 else "catalogue"
loci, desc = synth.config_loci(WL, n_loci=N, raw=True)
ctx = _lib.Context(0)
packed = ctx.pack_loci([(L.blocks(), L.raw_alns) for L in loci])
for rep in range(3):
    for pageable in (1, 0):
        ctx.set_debug("pageable_staging", pageable)
        ctx.calc_hap_aln_probs_packed(packed); ctx.calc_hap_aln_probs_packed(packed)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.calc_hap_aln_probs_packed(packed); ts.append(time.perf_counter() - t0)
        print(f"{WL} {N} loci, staging {'pageable' if pageable else 'pinned  '}: best {min(ts)*1e3:.2f} ms, mean {sum(ts)/len(ts)*1e3:.2f} ms, {N/min(ts):.0f} loci/s", flush=True)
