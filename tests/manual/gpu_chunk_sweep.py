"""Manual GPU check: ltr_calc_hap_aln_probs on N raw config-3 loci for several chunk counts /
stream counts / chunk growth laws (the LTR_CHUNKS / LTR_CHUNK_STREAMS / LTR_CHUNK_GROWTH debugging
overrides read by the library per call).  Every combination is visited twice, in two orders."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from longtr_amd import _lib, synth

NS = [int(x) for x in sys.argv[1:]] or [6000]
loci_all, desc = synth.config_loci("config3", n_loci=max(NS), raw=True)
ctx = _lib.Context(0)
for N in NS:
    packed = ctx.pack_loci([(L.blocks(), L.raw_alns) for L in loci_all[:N]])
    for k in ("LTR_CHUNKS", "LTR_CHUNK_STREAMS", "LTR_CHUNK_GROWTH"): os.environ.pop(k, None)
    ctx.calc_hap_aln_probs_packed(packed)
    combos = [(None, None, None)] + [(c, s, g) for c in (3, 4, 6, 8, 10) for s in (2, 3) for g in (0, 1.6)]
    for order in (combos, combos[::-1]):
        for c, s, g in order:
            for k, v in (("LTR_CHUNKS", c), ("LTR_CHUNK_STREAMS", s), ("LTR_CHUNK_GROWTH", g)):
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = str(v)
            ctx.calc_hap_aln_probs_packed(packed)
            ts = []
            for _ in range(4):
                t0 = time.perf_counter(); ctx.calc_hap_aln_probs_packed(packed); ts.append(time.perf_counter() - t0)
            dt = min(ts)
            print(f"N {N} chunks {c} streams {s} growth {g}: best {dt*1e3:.1f} ms (mean {sum(ts)/len(ts)*1e3:.1f}), {N/dt:.0f} loci/s", flush=True)
