"""Manual GPU sweep: resident-plan cells/s of the packed kernels (64 / LP pairs per wavefront) by repeat length and
lanes per pair, against one pair per wavefront; every mode's scores must be the same bits.
    python tests/manual/gpu_pack_sweep.py [tr,tr,...] [pairs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from longtr_amd import _abi, _lib, synth

TRS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [12, 20, 30, 45, 60, 80, 110, 150, 220, 300]
PAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
MODES = [(0, "one-wave"), (1, "LP32"), (5, "LP16"), (6, "LP8"), (7, "LP4"), (8, "LP2"), (-1, "auto")]
ctx = _lib.Context(0)
info = ctx.device_info()
peak = info["n_cu"] * 64 * info["clock_mhz"] * 1e6 / 11.0
for tr in TRS:
    rng = np.random.default_rng(tr)
    H, R = 6, 12
    n_loci = max(PAIRS // (H * R), 1)
    loci = [synth.synth_locus(rng, tr, max(1, min(6, tr // 4)), H, R, sub_rate=0.002, indel_rate=0.001) for _ in range(n_loci)]
    batch, _ = synth.pack_loci(loci, pooled=False)
    ref = None
    row = []
    for mode, name in MODES:
        ctx.set_pair_packing(mode)
        plan = ctx.plan(batch)
        plan.execute(); ll, _ = plan.fetch()
        if ref is None:
            ref = ll.copy()
        bad = int((ll.view(np.uint64) != ref.view(np.uint64)).sum())
        t0 = time.perf_counter()
        for _ in range(5): plan.execute()
        plan.fetch()
        dt = (time.perf_counter() - t0) / 5
        st = [k for k in plan.kernel_stats() if k["pairs"] and k["family"] != "exact"]
        cls = ",".join(f"{k['lanes_per_pair']}x{k['strip_width']}:{k['pairs']}" for k in st[:4])
        row.append(f"{name} {plan.cells/dt:.2e} ({plan.cells/dt/peak:.2f}){' BAD ' + str(bad) if bad else ''} [{cls}]")
        plan.close()
    print(f"TR {tr} ({batch.ll_size} pairs, {plan.cells:.2e} cells):\n   " + "\n   ".join(row), flush=True)
ctx.set_pair_packing(-1)
