"""Manual GPU check: ltr_calc_hap_aln_probs with DECREASING chunk sizes (chunk_growth < 1): a host-bound call ends with the\nGPU time of its last chunk.    python tests/manual/gpu_chunk_sweep_decreasing.py <workload> <N>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from longtr_amd import _lib, synth
WL = sys.argv[1]; N = int(sys.argv[2])
loci_all, desc = synth.config_loci(WL, n_loci=N, raw=True)
ctx = _lib.Context(0)
KEYS = ("chunks", "chunk_streams", "chunk_growth")
packed = ctx.pack_loci([(L.blocks(), L.raw_alns) for L in loci_all])
ctx.calc_hap_aln_probs_packed(packed)
combos = [(None, None, None), (3, 2, 1.0), (3, 2, 0.6), (4, 2, 0.6), (4, 2, 0.7), (5, 2, 0.7), (4, 2, 0.5), (3, 2, 0.4), (2, 2, 0.5), (6, 2, 0.8)]
for order in (combos, combos[::-1]):
    for combo in order:
        ctx.set_debug("reset", 0)
        for k, v in zip(KEYS, combo):
            if v is not None: ctx.set_debug(k, v)
        ctx.calc_hap_aln_probs_packed(packed)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); ctx.calc_hap_aln_probs_packed(packed); ts.append(time.perf_counter() - t0)
        print(f"{WL} N {N} {combo}: best {min(ts)*1e3:.1f} ms (mean {sum(ts)/len(ts)*1e3:.1f})", flush=True)
