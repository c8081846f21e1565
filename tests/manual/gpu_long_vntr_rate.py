"""Manual GPU check: pairs of very long reads (several kb: the eight-wave workgroup kernels, W = 13 .. 20) --
cells/s of a resident plan.    python tests/manual/gpu_long_vntr_rate.py <tr_len> [n_loci] [pair_packing mode] [wg_first_pass: 1 certificates, 2 thresholds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from longtr_amd import _abi, _lib, synth

TR = int(sys.argv[1]) if len(sys.argv) > 1 else 7400
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 48
rng = np.random.default_rng(5)
loci = [synth.synth_locus(rng, TR, 31, 4, 8, sub_rate=0.001, indel_rate=0.0005) for _ in range(NL)]
batch, _ = synth.pack_loci(loci)
ctx = _lib.Context(0, _abi.make_params(synth.ONT_PARAMS))
if len(sys.argv) > 3:
    ctx.set_pair_packing(int(sys.argv[3]))
if len(sys.argv) > 4:
    ctx.set_debug("wg_first_pass", int(sys.argv[4]))
for kv in sys.argv[5:]:
    ctx.set_debug(kv.split("=")[0], float(kv.split("=")[1]))
plan = ctx.plan(batch)
plan.execute(); plan.fetch()
t0 = time.perf_counter()
for _ in range(5): plan.execute()
ll, _ = plan.fetch()
dt = (time.perf_counter() - t0) / 5
st = [k for k in plan.kernel_stats() if k["pairs"]]
print(f"[first pass {sys.argv[4] if len(sys.argv) > 4 else 'rule'} {' '.join(sys.argv[5:])}] TR {TR}: {batch.ll_size} pairs, {plan.cells:.3e} cells, {dt*1e3:.2f} ms per pass, {plan.cells/dt:.3e} cells/s; classes", [(k["family"], k["lanes_per_pair"], k["strip_width"], k["pairs"]) for k in st], "finished", float((ll > -600).mean()))
