"""Manual GPU check: differential fuzz.  Random batches (locus count, repeat length 5 bp .. 3 kb, reads / haplotypes
per locus, error rates up to 6 %, default / ONT / asymmetric parameters, lower-case and N bases) scored under the
automatic schedule (-1: the plan kernel -- one launch, failed certificates scored in line -- for symmetric models), the automatic
schedule of round 4 ("c": plan kernel off: a launch per class, exact lists), the same with the multi-width launches forced on
whatever the batch size and no per-length floor on the packing ("m": ltr_dp_multi_kernel / ltr_dp_pack_multi_kernel), the
single-stream one-wave schedule (3), the exact kernels only (4) and -- round 6 -- the automatic schedule with the threshold kernels
as the first pass of the workgroup classes and no compact plans ("t"): the six must agree bit for bit on every pair; batches small
enough are also compared with the CPU oracle.  One batch in eight holds repeats of 3 - 9 kb (eight-wave workgroups).
    python tests/manual/gpu_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from longtr_amd import _abi, _lib, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _lib.Context(0)
PARAMS = [None, synth.ONT_PARAMS, (-1.2, -0.3, -0.9, -0.5, -0.0001, -5.0, -4.0), (-0.5, -1.0, -0.5, -1.0, -0.0005, -3.0, -3.0)]
MODES = (-1, 3, 4) if os.environ.get("LTR_FUZZ_NO_MULTI") else (-1, "c", "m", "t", 3, 4)      # (experimental builds without the multi-width kernels)
t0, n_batches, n_pairs, n_oracle = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    prm = PARAMS[int(rng.integers(len(PARAMS)))]
    params = _abi.default_params() if prm is None else _abi.make_params(prm)
    ctx.set_params(params)
    shape = int(rng.integers(4))
    n_loci = int(rng.integers(1, 40)) if shape else int(rng.integers(60, 200))
    loci = []
    long_batch = rng.random() < 0.125
    if long_batch:
        n_loci = int(rng.integers(1, 6))
    for _ in range(n_loci):
        tr = int(rng.choice([rng.integers(5, 80), rng.integers(80, 1000), rng.integers(1000, 3000)], p=[0.5, 0.4, 0.1]))
        if long_batch:
            tr = int(rng.integers(3000, 9000))
        err = float(rng.choice([0.0005, 0.002, 0.02, 0.06]))
        loci.append(synth.synth_locus(rng, tr, int(rng.integers(1, 7)), int(rng.integers(1, 8)), int(rng.integers(1, 12)), sub_rate=err, indel_rate=err / 2))
    batch, _ = synth.pack_loci(loci)
    if rng.random() < 0.3:                                        # bytes outside ACGT: the generic exact kernel
        rb = batch.read_bytes
        for k in rng.integers(0, len(rb), size=max(1, len(rb) // 5000)):
            rb[k] = ord("N") if rng.random() < 0.5 else (rb[k] | 0x20)
    out = {}
    for mode in MODES:
        ctx.set_pair_packing(-1 if mode in ("m", "c", "t") else mode)
        if mode in ("m", "c"):
            ctx.set_debug("plan_kernel", 1)
        if mode == "t":
            ctx.set_debug("wg_first_pass", 2); ctx.set_debug("compact_plan", -1)
        if mode == "m":
            ctx.set_debug("no_multi", -1); ctx.set_debug("pack_rule", 2)
        out[mode], _ = ctx.align_batch(batch)
        ctx.set_debug("reset", 0)
    ctx.set_pair_packing(-1)
    for mode in MODES[1:]:
        bad = np.where(out[-1].view(np.uint64) != out[mode].view(np.uint64))[0]
        if bad.size:
            print(f"MISMATCH batch {n_batches} (seed state lost: rerun with the same seed), mode -1 vs {mode}: {bad.size}/{out[-1].size} pairs, first {bad[:5]}: {out[-1][bad[:5]]} vs {out[mode][bad[:5]]}")
            sys.exit(1)
    cells = float(synth.nominal_cells(batch, params.indel_flank_len))
    if cells < 3e8:
        want, _, _ = ol.oracle_align_batch(batch, params)
        bad = np.where(out[-1].view(np.uint64) != want.view(np.uint64))[0]
        if bad.size:
            print(f"MISMATCH vs oracle, batch {n_batches}: {bad.size} pairs, first {bad[:5]}: {out[-1][bad[:5]]} vs {want[bad[:5]]}")
            sys.exit(1)
        n_oracle += batch.ll_size
    n_batches += 1; n_pairs += batch.ll_size
print(f"fuzz ok: {n_batches} batches, {n_pairs} pairs under {len(MODES)} schedules bit-identical, {n_oracle} of them also against the oracle, {time.time()-t0:.0f} s")
