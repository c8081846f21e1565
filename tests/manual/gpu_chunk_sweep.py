"""Manual GPU check: ltr_calc_hap_aln_probs on N raw loci of a workload for several chunk counts /
stream counts / chunk growth laws (the ltr_ctx_set_debug "chunks" / "chunk_streams" / "chunk_growth"
overrides).  Every combination is visited twice, in two orders.
    python tests/manual/gpu_chunk_sweep.py <workload> <N> [<N> ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from longtr_amd import _lib, synth

WL = sys.argv[1] if len(sys.argv) > 1 else "config3"
NS = [int(x) for x in sys.argv[2:]] or [6000]
loci_all, desc = synth.config_loci(WL, n_loci=max(NS), raw=True)
ctx = _lib.Context(0)
KEYS = ("chunks", "chunk_streams", "chunk_growth")
for N in NS:
    packed = ctx.pack_loci([(L.blocks(), L.raw_alns) for L in loci_all[:N]])
    ctx.set_debug("reset", 0)
    ctx.calc_hap_aln_probs_packed(packed)
    combos = [(None, None, None), (1, 1, 0), (2, 2, 1.0), (2, 2, 3.0), (3, 2, 1.0), (3, 2, 2.0), (4, 2, 1.0), (4, 2, 1.5), (4, 2, 2.0), (6, 2, 1.0), (6, 2, 1.5), (8, 2, 1.0), (8, 2, 1.3), (12, 2, 1.0), (6, 3, 1.0)]
    for order in (combos, combos[::-1]):
        for combo in order:
            ctx.set_debug("reset", 0)
            for k, v in zip(KEYS, combo):
                if v is not None: ctx.set_debug(k, v)
            ctx.calc_hap_aln_probs_packed(packed)
            ts = []
            for _ in range(4):
                t0 = time.perf_counter(); ctx.calc_hap_aln_probs_packed(packed); ts.append(time.perf_counter() - t0)
            dt = min(ts)
            print(f"{WL} N {N} chunks/streams/growth {combo}: best {dt*1e3:.1f} ms (mean {sum(ts)/len(ts)*1e3:.1f}), {N/dt:.0f} loci/s", flush=True)
