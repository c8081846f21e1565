"""-m gpu: the workgroup-per-pair kernels (ltr_dp_wg.hpp) against the rolling oracle, bit for bit:
every strip width of the 4-wave (reads of 1026..3585 bases) and 8-wave (3586..10241) classes at
their edges, pairs that abort or sit at the -600 line, the one-wave latency variant, and the
fallbacks (asymmetric models, very long reads, parameters changed under a resident plan)."""
import numpy as np
import pytest

import oracle_lib as ol
from longtr_amd import _abi, _lib, synth

pytestmark = pytest.mark.gpu


def _near_pair(rng, m, nmis=None, where=None):
    """read = haplotype window with substitutions (+ one small deletion): the whole DP runs."""
    core = bytearray(synth._rand_seq(rng, m).tobytes())
    read = bytearray(core)
    k = min(max(1, m // 150), 30) if nmis is None else nmis
    lo, hi = where if where else (0, m)
    for p in rng.choice(np.arange(lo, hi), size=k, replace=False):
        read[p] = ord("A") if read[p] != ord("A") else ord("C")
    return bytes(read), synth._rand_seq(rng, 30).tobytes() + bytes(core) + synth._rand_seq(rng, 30).tobytes()


def _check_pairs(ctx, pairs, params=None, modes=(-1,)):
    """pairs: list of (read, hap).  GPU (one locus per pair) == rolling oracle, bit for bit."""
    if params is not None:
        ctx.set_params(params)
    try:
        want = np.asarray([ol.oracle_align_long(h, r, ctx.params, rolling=True) for r, h in pairs])
        b = _abi.PackedBatch([([r], [h]) for r, h in pairs])
        out = None
        for mode in modes:
            ctx.set_pair_packing(mode)
            ll, _ = ctx.align_batch(b)
            bad = np.where(ll.view(np.uint64) != want.view(np.uint64))[0]
            assert bad.size == 0, (mode, bad[:8], ll[bad[:8]], want[bad[:8]], [len(pairs[i][0]) for i in bad[:8]])
            out = ll
    finally:
        ctx.set_pair_packing(-1)
        if params is not None:
            ctx.set_params(_abi.default_params())
    return out


def _with_first_pass(ctx, v, fn):
    ctx.set_debug("wg_first_pass", v)
    try:
        return fn()
    finally:
        ctx.set_debug("wg_first_pass", 0)


def _classes(ctx, batch):
    plan = ctx.plan(batch)
    plan.execute()
    plan.fetch()
    st = plan.kernel_stats()
    plan.close()
    return st


def test_every_strip_width_of_the_4_and_8_wave_classes(gpu_ctx):
    rng = np.random.default_rng(31)
    ms = []
    for W in range(5, 21):                       # 4 waves: C in (256(W-1), 256W], W = 5..20 (C >= 1025)
        lo = max(256 * (W - 1), 1024)
        ms += [lo + 2, lo + 2 + int(rng.integers(1, 250)), 256 * W + 1]
    for W in range(11, 21):                      # 8 waves: C in (512(W-1), 512W], from C = 5121
        lo = max(512 * (W - 1), 5120)
        ms += [lo + 2, 512 * W + 1] + ([lo + 2 + int(rng.integers(1, 500))] if W % 3 == 0 else [])
    pairs = [_near_pair(rng, m) for m in ms]
    ll = _check_pairs(gpu_ctx, pairs, modes=(-1, 0, 2, 4))      # (2: workgroup kernels everywhere -- the wide four-wave strips W = 15 .. 20 too)
    assert (ll > -600.0).all()
    gpu_ctx.set_pair_packing(0)                  # (the automatic mode folds classes of a few pairs into their wider neighbours)
    try:
        st = _classes(gpu_ctx, _abi.PackedBatch([([r], [h]) for r, h in pairs]))
    finally:
        gpu_ctx.set_pair_packing(-1)
    used4 = {k["strip_width"] for k in st if k["family"] != "exact" and k["lanes_per_pair"] == 256 and k["pairs"]}
    used8 = {k["strip_width"] for k in st if k["family"] != "exact" and k["lanes_per_pair"] == 512 and k["pairs"]}
    # (C <= 1280 fits one wavefront's widest strips, W = 17..20: the 4-wave class W = 5 only on request, mode 2)
    # (a handful of pairs is one round of workgroups either way: reads of 3586 .. 5121 bases stay on eight waves, W = 8 .. 10)
    assert used4 == set(range(6, 15)) and used8 == set(range(8, 21))
    assert _lib.Plan.exact_pairs(st) == 0        # nothing needed the exact kernels
    gpu_ctx.set_pair_packing(2)
    try:
        st2 = _classes(gpu_ctx, _abi.PackedBatch([([r], [h]) for r, h in pairs]))
    finally:
        gpu_ctx.set_pair_packing(-1)
    assert {k["strip_width"] for k in st2 if k["family"] != "exact" and k["lanes_per_pair"] == 256 and k["pairs"]} == set(range(5, 21))


def test_long_pairs_abort_uncertain_and_unequal_lengths(gpu_ctx):
    rng = np.random.default_rng(32)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    pairs = []
    for m, where, nmis in [(1500, (0, 1500), 64), (1500, (0, 1500), 68), (2600, (0, 800), 90), (2600, (1700, 2500), 90),
                           (4200, (1500, 2200), 80), (3300, (3000, 3300), 75), (5200, (100, 400), 70), (6100, (5000, 6100), 66),
                           (9000, (100, 8900), 62), (9000, (8000, 9000), 72)]:
        pairs.append(_near_pair(rng, m, nmis, where))
    pairs += [(rs(2300), rs(2500)), (rs(3000), rs(2700)), (b"AC" * 1200, b"GT" * 1300), (rs(6000), rs(6100))]   # unrelated: abort at once
    # n != m: reads with a long deletion / insertion against the haplotype (|n - m| up to 590)
    for m, d in [(1800, 300), (1800, -300), (4000, 590), (4000, -590), (7000, 450)]:
        r, h = _near_pair(rng, m)
        core = h[30:-30]
        h2 = h[:30] + (core[:m // 2] + rs(d) + core[m // 2:] if d > 0 else core[:m // 2] + core[m // 2 - d:]) + h[-30:]
        pairs.append((r, h2))
    pairs.append((rs(1200), rs(62)))             # |n - m| > 600 shortcut in a workgroup class
    pairs.append((rs(1200), rs(60)))             # haplotype <= 60
    ll = _check_pairs(gpu_ctx, pairs, modes=(-1, 3, 4))
    assert (ll == -700.0).sum() >= 6 and (ll > -600.0).sum() >= 4
    _check_pairs(gpu_ctx, pairs[:8] + pairs[10:], _abi.make_params(synth.ONT_PARAMS))


def test_workgroup_kernels_share_a_batch_with_every_other_class(gpu_ctx):
    """One plan with short (two per wave), medium (one per wave), long (4- and 8-wave) and non-ACGT
    pairs; many pairs per class so that workgroups loop over several pairs (ring / progress reset)."""
    rng = np.random.default_rng(33)
    pairs = []
    for _ in range(3):
        for m in (50, 300, 700, 1000, 1100, 1400, 2300, 3500, 3700, 5200):
            pairs.append(_near_pair(rng, m + int(rng.integers(0, 20))))
    r, h = _near_pair(rng, 1300)
    pairs.append((r[:-1] + b"n", h))
    many = pairs * 40                            # 1240 pairs: the 4-wave classes queue up behind 768 resident workgroups
    gpu_ctx.set_pair_packing(1)
    try:
        b = _abi.PackedBatch([([r], [h]) for r, h in many])
        ll, _ = gpu_ctx.align_batch(b)
    finally:
        gpu_ctx.set_pair_packing(-1)
    want = np.asarray([ol.oracle_align_long(h, r, gpu_ctx.params, rolling=True) for r, h in pairs])
    assert np.array_equal(ll.view(np.uint64), np.tile(want, 40).view(np.uint64))


def test_one_wave_variant_on_request(gpu_ctx):
    """Mode 2 puts every short read on the one-wave variant of the workgroup kernel (haplotype rows and
    first-column table streamed through LDS); the default never does.  Same bits."""
    loci, _ = synth.config_loci("config2")
    small, _ = synth.pack_loci(loci)
    ref, _, _ = ol.oracle_align_batch(small, gpu_ctx.params)
    for mode, want_all in ((2, True), (-1, False)):
        gpu_ctx.set_pair_packing(mode)
        try:
            plan = gpu_ctx.plan(small)
            plan.execute()
            ll, _ = plan.fetch()
            st = plan.kernel_stats()
            plan.close()
        finally:
            gpu_ctx.set_pair_packing(-1)
        assert np.array_equal(ll.view(np.uint64), ref.view(np.uint64))
        wg1 = sum(k["pairs"] for k in st if k["lanes_per_pair"] == 64 and k.get("family") == "workgroup")
        assert wg1 == (small.ll_size if want_all else 0)


def test_fallbacks_asymmetric_model_and_param_change(gpu_ctx):
    rng = np.random.default_rng(35)
    pairs = [_near_pair(rng, m) for m in (1300, 2500, 4000)]
    asym = _abi.make_params((-1.0, -0.458675, -1.0, -0.458675, -0.00005800168, -10.448214728, -6.0))
    _check_pairs(gpu_ctx, pairs, asym)           # no workgroup kernels for f != g: column blocks on one wavefront
    b = _abi.PackedBatch([([r], [h]) for r, h in pairs])
    plan = gpu_ctx.plan(b)                       # binned into workgroup classes under the symmetric defaults
    gpu_ctx.set_params(asym)
    try:
        with pytest.raises(_lib.LtrError) as e:
            plan.execute()
        assert e.value.code == -1 and "create it again" in str(e.value)
    finally:
        gpu_ctx.set_params(_abi.default_params())
    plan.execute()                               # symmetric again: fine
    ll, _ = plan.fetch()
    plan.close()
    want = np.asarray([ol.oracle_align_long(h, r, gpu_ctx.params, rolling=True) for r, h in pairs])
    assert np.array_equal(ll.view(np.uint64), want.view(np.uint64))


def test_two_per_wave_partner_with_a_much_shorter_haplotype_at_the_buffer_end(gpu_ctx):
    """Two pairs of one strip class whose haplotypes differ by several hundred rows share a wavefront
    and run in lock step: the shorter one's row stream runs on past its haplotype -- the LAST one of
    the batch -- into the buffer's tail pad (the pad must cover the longest window)."""
    rng = np.random.default_rng(36)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    read = rs(100)
    long_h = rs(30) + read[:50] + rs(560) + read[50:] + rs(30)      # n = 660, m = 100: |n - m| = 560
    short_h = rs(30) + read[:40] + read[60:] + rs(30)               # n = 80
    gpu_ctx.set_pair_packing(1)
    try:
        b = _abi.PackedBatch([([read], [long_h]), ([read], [short_h])])
        ll, _ = gpu_ctx.align_batch(b)
    finally:
        gpu_ctx.set_pair_packing(-1)
    want = [ol.oracle_align_long(h, read, gpu_ctx.params, rolling=True) for h in (long_h, short_h)]
    assert np.array_equal(ll.view(np.uint64), np.asarray(want).view(np.uint64))


def _long_pairs_of_every_kind(rng):
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    pairs = []
    for m, where, nmis in [(1500, (0, 1500), 64), (1500, (0, 1500), 68), (2600, (0, 800), 90), (2600, (1700, 2500), 90),
                           (4200, (1500, 2200), 80), (3300, (3000, 3300), 75), (5200, (100, 400), 70), (6100, (5000, 6100), 66),
                           (9000, (100, 8900), 62), (9000, (8000, 9000), 72)]:
        pairs.append(_near_pair(rng, m, nmis, where))
    pairs += [(rs(2300), rs(2500)), (rs(3000), rs(2700)), (b"AC" * 1200, b"GT" * 1300), (rs(6000), rs(6100))]
    for m, d in [(1800, 300), (1800, -300), (4000, 590), (4000, -590), (7000, 450)]:
        r, h = _near_pair(rng, m)
        core = h[30:-30]
        h2 = h[:30] + (core[:m // 2] + rs(d) + core[m // 2:] if d > 0 else core[:m // 2] + core[m // 2 - d:]) + h[-30:]
        pairs.append((r, h2))
    return pairs


def test_threshold_kernels_as_the_first_pass_of_every_class(gpu_ctx):
    """ltr_dp_wg_kernel<W, NW, SYM, FULL = true> (round 6): every cell against the exact threshold table, a settled row without a
    passing cell aborts the pair -- exact in one pass.  Every strip width of both workgroup families at its edges (odd classes run
    the next even width; the last lane's slack columns: every Wl), pairs that finish, abort, sit at the -600 line, unequal
    lengths; default and ONT-like parameters.  Nothing may reach the exact lists."""
    rng = np.random.default_rng(41)
    ms = []
    for W in range(5, 21):
        lo = max(256 * (W - 1), 1024)
        ms += [lo + 2, lo + 2 + int(rng.integers(1, 250)), 256 * W + 1, 256 * W]
    for W in range(11, 21):
        lo = max(512 * (W - 1), 5120)
        ms += [lo + 2, 512 * W + 1] + ([lo + 2 + int(rng.integers(1, 500))] if W % 3 == 0 else [])
    ms += [1290 + k for k in range(0, 24)]                         # every remainder of the last lane's strip
    pairs = [_near_pair(rng, m) for m in ms]
    ll = _with_first_pass(gpu_ctx, 2, lambda: _check_pairs(gpu_ctx, pairs, modes=(-1, 0, 2)))
    assert (ll > -600.0).all()
    hard = _long_pairs_of_every_kind(rng)
    ll = _with_first_pass(gpu_ctx, 2, lambda: _check_pairs(gpu_ctx, hard, modes=(-1, 2)))
    assert (ll == -700.0).sum() >= 6 and (ll > -600.0).sum() >= 4
    _with_first_pass(gpu_ctx, 2, lambda: _check_pairs(gpu_ctx, hard[:8] + hard[10:], _abi.make_params(synth.ONT_PARAMS), modes=(-1, 2)))

    def lists_stay_empty():
        gpu_ctx.set_pair_packing(2)
        try:
            st = _classes(gpu_ctx, _abi.PackedBatch([([r], [h]) for r, h in hard]))
        finally:
            gpu_ctx.set_pair_packing(-1)
        assert _lib.Plan.exact_pairs(st) == 0, [k for k in st if k["family"] == "exact" and k["pairs"]]
    _with_first_pass(gpu_ctx, 2, lists_stay_empty)


def test_the_first_pass_is_learnt_from_the_reads(gpu_ctx):
    """ltr_ctx_wg_first_pass: a plan whose long pairs fail their certificates (they abort, like every pair of BASELINE config 5)
    switches the context to the threshold kernels; a plan whose pairs finish switches it back; ltr_ctx_set_params forgets.
    Same bits all along."""
    rng = np.random.default_rng(42)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    aborting = [(rs(2000 + 37 * k), rs(2060 + 37 * k)) for k in range(12)] + [_near_pair(rng, 2400)]
    finishing = [_near_pair(rng, 1500 + 211 * k) for k in range(12)] + [(rs(2000), rs(2100))]
    want_a = np.asarray([ol.oracle_align_long(h, r, gpu_ctx.params, rolling=True) for r, h in aborting])
    want_f = np.asarray([ol.oracle_align_long(h, r, gpu_ctx.params, rolling=True) for r, h in finishing])
    assert (want_a == -700.0).sum() == 12 and (want_f > -600.0).sum() == 12
    ba = _abi.PackedBatch([([r], [h]) for r, h in aborting])
    bf = _abi.PackedBatch([([r], [h]) for r, h in finishing])

    def run(b, want):
        ll, _ = gpu_ctx.align_batch(b)
        assert np.array_equal(ll.view(np.uint64), want.view(np.uint64))
        return gpu_ctx.wg_first_pass()

    gpu_ctx.set_params(_abi.default_params())                      # forgets what earlier tests taught the context
    assert gpu_ctx.wg_first_pass()[0] == 0
    mode, unfinished, scored = run(ba, want_a)
    assert (mode, unfinished, scored) == (1, 12, 13)               # certificates first: 12 of 13 failed -> thresholds from now on
    mode, unfinished, scored = run(ba, want_a)
    assert (mode, unfinished, scored) == (1, 12, 13)               # thresholds first: 12 of 13 aborted -> stays
    plan = gpu_ctx.plan(ba)
    plan.execute()
    plan.fetch()
    st = plan.kernel_stats()
    plan.close()
    assert _lib.Plan.exact_pairs(st) == 0                          # one pass: nothing on the exact lists
    mode, unfinished, scored = run(bf, want_f)
    assert (mode, unfinished, scored) == (0, 1, 13)                # 1 of 13 aborted -> back to the certificates
    mode, _, _ = run(bf, want_f)
    assert mode == 0
    run(ba, want_a)
    assert gpu_ctx.wg_first_pass()[0] == 1
    gpu_ctx.set_params(_abi.default_params())
    assert gpu_ctx.wg_first_pass()[0] == 0
