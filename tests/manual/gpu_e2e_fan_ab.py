"""Manual GPU A/B: ltr_calc_hap_aln_probs with the plans' launches on 1 / 2 / 4 lanes (ltr_ctx_set_debug fan_lanes).
    python tests/manual/gpu_e2e_fan_ab.py <workload> <N>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from longtr_amd import _lib, synth
WL = sys.argv[1] if len(sys.argv) > 1 else "config3skew"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
loci, desc = synth.config_loci(WL, n_loci=N, raw=True)
ctx = _lib.Context(0)
packed = ctx.pack_loci([(L.blocks(), L.raw_alns) for L in loci])
ctx.calc_hap_aln_probs_packed(packed)
for rep in range(2):
    for fl in (1, 2, 4):
        ctx.set_debug("fan_lanes", fl)
        ctx.calc_hap_aln_probs_packed(packed)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.calc_hap_aln_probs_packed(packed); ts.append(time.perf_counter() - t0)
        print(f"{WL} N {N} fan_lanes {fl}: best {min(ts)*1e3:.1f} ms mean {sum(ts)/len(ts)*1e3:.1f}", flush=True)
