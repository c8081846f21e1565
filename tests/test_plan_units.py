"""CPU: the planning units of ltr_plan_create (csrc/ltr_plan.cpp) through their ltr_debug_* entry points -- the table of
launch classes, the rule that gives a pair its class and launch-order key, the class sort with folding.  They decide
WHICH kernel scores a pair and in what order, never the score (the GPU tests run every case in ten scheduling modes)."""
import ctypes as C

import numpy as np
import pytest

from longtr_amd import _abi, _lib

L = _lib.lib()
L.ltr_debug_class_info.argtypes = [C.c_int] + [C.POINTER(C.c_int)] * 4
L.ltr_debug_classify.argtypes = [C.POINTER(_abi.AlignParams), C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
L.ltr_debug_sort_by_class.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
NK = L.ltr_debug_num_classes()
N_CU = 256


def class_info(k):
    v = [C.c_int(0) for _ in range(4)]
    assert L.ltr_debug_class_info(k, *[C.byref(x) for x in v]) == 0
    return dict(zip(("family", "W", "waves", "lanes"), (x.value for x in v)))


def classify(n, m, mode=-1, pairs=10 ** 6, long_pairs=0, hfl=None, generic=0, params=None):
    cls, key, xl = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    p = params or _abi.default_params()
    hfl = n + 60 if hfl is None else hfl
    assert L.ltr_debug_classify(C.byref(p), mode, N_CU, pairs, long_pairs, n, m, hfl, generic, C.byref(cls), C.byref(key), C.byref(xl)) == 0
    return cls.value, key.value, xl.value


def test_class_table_is_consistent_with_the_library():
    assert NK == L.ltr_num_kernels()
    fam = {0: 0, 1: 0, 2: 0, 3: 0}
    for k in range(NK):
        ci = class_info(k)
        assert ci["family"] == L.ltr_kernel_family(k) and ci["lanes"] == L.ltr_kernel_lanes_per_pair(k)
        fam[ci["family"]] += 1
        if ci["family"] == 1:
            assert ci["lanes"] in (2, 4, 8, 16, 32) and 1 <= ci["W"] <= 20
    assert fam[0] == 20 and fam[1] == 100 and fam[3] == 6


def test_every_pair_fits_its_class():
    """Whatever the rule picks, the read's columns fit the class: lanes x strip width (one block) for the packed and
    workgroup classes; one-wave classes take any length (column blocks)."""
    rng = np.random.default_rng(3)
    for mode in (-1, 0, 1, 2, 3, 5, 6, 7, 8):
        for _ in range(400):
            m = int(rng.integers(2, 12000)) if rng.random() < 0.3 else int(rng.integers(2, 700))
            n = max(2, m + int(rng.integers(-40, 40)))
            pairs = int(10 ** rng.uniform(1, 7))
            cls, key, xl = classify(n, m, mode=mode, pairs=pairs, long_pairs=int(rng.integers(0, 5000)))
            ci = class_info(cls)
            C_ = m - 1
            assert ci["family"] in (0, 1, 2) and (1 <= key <= 511 or abs(n - m) > 600)
            if ci["family"] == 1:
                assert C_ <= ci["lanes"] * ci["W"] and (ci["lanes"] * (ci["W"] - 1) < C_ or ci["W"] == 1), (m, ci)
            elif ci["family"] == 2:
                assert C_ <= ci["lanes"] * ci["W"]
            else:
                ncb = -(-C_ // 1280)
                assert ci["W"] == -(-C_ // (64 * ncb))
            assert (xl == 1) == (C_ <= 256) or xl != 1


def test_modes_and_batch_size_rules():
    # a one-locus batch keeps one pair per wavefront; a big batch packs short reads; explicit modes force the geometry
    assert class_info(classify(220, 221, pairs=224)[0])["family"] == 0
    assert class_info(classify(40, 40, pairs=3 * 10 ** 6)[0])["family"] == 1
    assert class_info(classify(40, 40, mode=0)[0])["family"] == 0
    for mode, lanes in ((1, 32), (5, 16), (6, 8), (7, 4), (8, 2)):
        ci = class_info(classify(40, 40, mode=mode)[0])
        assert ci["family"] == 1 and ci["lanes"] == lanes and ci["W"] == -(-39 // lanes)
    # a read too long for the forced segment takes the next wider one; beyond 641 bases one pair per wavefront
    assert class_info(classify(200, 200, mode=8)[0])["lanes"] == 16                  # 199 columns: 2 x 20 and 4 x 20 and 8 x 20 too few
    assert class_info(classify(700, 700, mode=1)[0])["family"] == 0
    # long reads: workgroup kernels while there are few of them, column blocks on one wavefront otherwise / in mode 3
    assert class_info(classify(5000, 5000, long_pairs=100)[0])["lanes"] == 512       # one round either way: eight waves finish a pair sooner
    ci = class_info(classify(5000, 5000, long_pairs=600)[0])                         # one round of four-wave workgroups, two of eight-wave ones
    assert ci["lanes"] == 256 and ci["W"] == 20
    assert class_info(classify(5000, 5000, long_pairs=1868)[0])["lanes"] == 256      # 2.4 rounds: two whole ones on four waves; ltr_plan_create moves the rest to eight (config5hifi)
    assert class_info(classify(5000, 5000, long_pairs=800)[0])["lanes"] == 512       # one round and a bit: two rounds of eight waves are as good
    assert class_info(classify(6000, 6000, long_pairs=100)[0])["lanes"] == 512
    # reads of up to two column blocks (2560 columns): four-wave workgroups while the batch cannot fill the GPU's wave slots; in a
    # batch that can they stay with the one-wave classes -- the plan kernel holds every wave slot, a launch beside it would starve
    assert class_info(classify(2000, 2000, pairs=2000, long_pairs=100)[0])["lanes"] == 256
    ci = class_info(classify(2000, 2000, long_pairs=100)[0])
    assert ci["family"] == 0 and ci["W"] == 16                                       # 1999 columns = two blocks of 64 x 16
    assert class_info(classify(2000, 2000, mode=1, long_pairs=100)[0])["lanes"] == 256   # (explicit modes: no plan kernel)
    assert class_info(classify(2700, 2700, long_pairs=100)[0])["lanes"] == 256
    # a length difference no certificate can hold goes straight to its exact list (automatic mode)
    cls, _, xl = classify(300, 880)
    assert class_info(cls)["family"] == 3 and cls == NK - 6 + xl
    assert class_info(classify(300, 780)[0])["family"] != 3 and class_info(classify(300, 880, mode=3)[0])["family"] == 0
    assert class_info(classify(5000, 5000, long_pairs=9216)[0])["lanes"] == 256       # wide four-wave strips: beyond the eight-wave regime too
    assert class_info(classify(5000, 5000, long_pairs=10 ** 5)[0])["family"] == 0     # ... up to 80 long pairs per CU
    assert class_info(classify(3000, 3000, long_pairs=10 ** 5)[0])["family"] == 0
    assert class_info(classify(6000, 6000, long_pairs=10 ** 5)[0])["family"] == 0
    assert class_info(classify(5000, 5000, mode=3, long_pairs=100)[0])["family"] == 0
    # shortcuts keep a one-wave class and the last place in the launch order; non-ACGT pairs start in the generic exact list
    cls, key, _ = classify(0, 50, hfl=50)
    assert class_info(cls)["family"] == 0 and key == 0
    cls, key, _ = classify(1000, 50)
    assert key == 0
    cls, _, xl = classify(100, 100, generic=1)
    assert class_info(cls)["family"] == 3 and xl == 0
    # mode 4: every pair straight to the exact kernel of its length
    for m, want in ((100, 1), (400, 2), (900, 3), (2000, 4), (6000, 5)):
        cls, _, xl = classify(m, m, mode=4)
        assert class_info(cls)["family"] == 3 and xl == want
    # an asymmetric indel model has no LUT exact kernels and no workgroup kernels; under the plan kernel (automatic mode) its failed
    # certificates still go by read length -- to the threshold bodies the plan kernel calls itself (round 6) --, without it to the
    # generic list
    asym = _abi.make_params((-1.0, -0.45, -1.0, -0.5, -0.0001, -10.0, -9.0))
    cls, _, xl = classify(5000, 5000, long_pairs=10, params=asym)
    assert class_info(cls)["family"] == 0 and xl == 5
    cls, _, xl = classify(5000, 5000, long_pairs=10, params=asym, mode=0)
    assert class_info(cls)["family"] == 0 and xl == 0


def test_order_key_is_monotone_in_the_work():
    keys = [classify(n, n, mode=0)[1] for n in (30, 60, 120, 250, 500, 1000, 2000, 4000)]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)


def _sort(cls, key, fold):
    cls = np.ascontiguousarray(cls, dtype=np.int16); key = np.ascontiguousarray(key, dtype=np.int16)
    order = np.zeros(len(cls), dtype=np.int32); first = np.zeros(NK + 1, dtype=np.int32)
    assert L.ltr_debug_sort_by_class(cls.ctypes.data, key.ctypes.data, len(cls), int(fold), N_CU, order.ctypes.data, first.ctypes.data) == 0
    return order, first


def test_sort_by_class_is_a_stable_partition_longest_first():
    rng = np.random.default_rng(11)
    n = 300000                                                          # several 64 k blocks and 32 k segments: the parallel paths
    cls = rng.choice([3, 11, 14, 25, 47, 90, NK - 2], size=n).astype(np.int16)
    key = rng.integers(0, 512, size=n).astype(np.int16)
    order, first = _sort(cls, key, fold=False)
    assert sorted(order.tolist()) == list(range(n)) and first[0] == 0 and first[NK] == n
    for k in range(NK):
        seg = order[first[k]:first[k + 1]]
        assert (cls[seg] == k).all()
        kk = key[seg]
        assert (np.diff(kk.astype(np.int32)) <= 0).all()                # longest first
        same = np.flatnonzero(np.diff(kk.astype(np.int32)) == 0)
        assert (seg[same] < seg[same + 1]).all()                        # input order kept inside a key
    assert _sort(np.zeros(0), np.zeros(0), False)[1][NK] == 0
    bad = np.array([NK], dtype=np.int16)
    o = np.zeros(1, dtype=np.int32); f = np.zeros(NK + 1, dtype=np.int32)
    assert L.ltr_debug_sort_by_class(bad.ctypes.data, np.zeros(1, dtype=np.int16).ctypes.data, 1, 0, N_CU, o.ctypes.data, f.ctypes.data) == _abi.LTR_ERR_INVALID


def test_folding_merges_underfilled_classes_into_wider_strips_only():
    # one-wave classes W = 11 (k = 10) .. 14 with a few pairs each: folded upwards while the strip stays within 4/3
    cls = np.array([10] * 50 + [11] * 60 + [12] * 70 + [13] * 80 + [19] * 5, dtype=np.int16)
    key = np.arange(len(cls), dtype=np.int16) % 500 + 1
    order, first = _sort(cls, key, fold=True)
    sizes = {k: int(first[k + 1] - first[k]) for k in range(NK) if first[k + 1] > first[k]}
    assert sum(sizes.values()) == len(cls)
    assert sizes.get(13) == 260 and sizes.get(19) == 5 and 10 not in sizes and 11 not in sizes and 12 not in sizes
    for k, s in sizes.items():                                          # a pair never lands in a NARROWER class than its own
        assert (cls[order[first[k]:first[k + 1]]] <= k).all()
    # a class that fills the GPU stays where it is, and a lone small class keeps its strip width (nothing wider to join)
    big = np.array([10] * 200000 + [11] * 50, dtype=np.int16)
    _, first = _sort(big, np.ones(len(big), dtype=np.int16), fold=True)
    assert first[11] - first[10] == 200000 and first[12] - first[11] == 50


def test_folding_of_the_workgroup_families():
    # four-wave classes W = 17 and W = 18 with a few hundred pairs each become one launch; a class that fills several rounds
    # of workgroups stays; the eight-wave family folds the same way; never across families
    wg4 = next(k for k in range(NK) if class_info(k)["family"] == 2 and class_info(k)["lanes"] == 256 and class_info(k)["W"] == 17)
    wg8 = next(k for k in range(NK) if class_info(k)["family"] == 2 and class_info(k)["lanes"] == 512 and class_info(k)["W"] == 12)
    cls = np.array([wg4] * 1116 + [wg4 + 1] * 420 + [wg8] * 30 + [wg8 + 2] * 40, dtype=np.int16)
    _, first = _sort(cls, np.ones(len(cls), dtype=np.int16), fold=True)
    sizes = {k: int(first[k + 1] - first[k]) for k in range(NK) if first[k + 1] > first[k]}
    assert sizes == {wg4 + 1: 1536, wg8 + 2: 70}
    cls = np.array([wg4] * 5000 + [wg4 + 1] * 420, dtype=np.int16)
    _, first = _sort(cls, np.ones(len(cls), dtype=np.int16), fold=True)
    assert first[wg4 + 1] - first[wg4] == 5000 and first[wg4 + 2] - first[wg4 + 1] == 420
    last4 = max(k for k in range(NK) if class_info(k)["family"] == 2 and class_info(k)["lanes"] == 256)
    cls = np.array([last4] * 10 + [last4 + 1] * 10, dtype=np.int16)              # the widest four-wave class next to the narrowest eight-wave one
    _, first = _sort(cls, np.ones(len(cls), dtype=np.int16), fold=True)
    assert first[last4 + 1] - first[last4] == 10


def test_shard_costs_are_the_plans_own_model():
    """longtr_amd/shard.py balances shards on ltr_debug_pair_costs = the cost classify_pair derives the launch-order key from
    (no hand copy of the model in Python): for a grid of (window, read columns) -- packed geometries, one wavefront per pair,
    column blocks beyond 1280 columns, workgroup kernels when long pairs are few -- key == clamp(int(16 log2 cost) - 16)."""
    import math
    from longtr_amd import shard
    for pairs, long_pairs in ((1 << 21, 1 << 20), (1 << 21, 100), (3000, 0)):
        for n in (2, 25, 64, 200, 640, 900, 1300, 2600, 5000):
            for Cc in (1, 20, 63, 64, 129, 400, 641, 1280, 1281, 2000, 3585, 5200):
                if abs(n - (Cc + 1)) > 600:
                    continue
                _, key, _ = classify(n, Cc + 1, pairs=pairs, long_pairs=long_pairs)
                win, rl = np.asarray([n], dtype=np.int32), np.asarray([Cc + 1], dtype=np.int32)
                hl, out = win + 60, np.zeros(1)
                L.ltr_debug_pair_costs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 4
                p = _abi.default_params()
                assert L.ltr_debug_pair_costs(C.byref(p), -1, N_CU, pairs, long_pairs, 1, win.ctypes.data, rl.ctypes.data, hl.ctypes.data, out.ctypes.data) == 0
                c = float(out[0])
                assert key == min(511, max(1, (int(math.log2(c) * 16.0) if c > 1.0 else 0) - 16)), (n, Cc, pairs, long_pairs, c, key)
    # the Python entry point: same numbers, vectorised; a long pair costs more than a short one, cost grows with both sides
    c = shard.pair_time_cost(np.asarray([50, 50, 500, 900]), np.asarray([49, 99, 499, 899]))
    assert (np.diff(c) > 0).all() and c[0] > 0


def test_risky_pairs_go_straight_to_the_exact_body_per_gap_direction():
    """A pair whose length difference alone costs ~520 of the 600 the reference allows (HapAligner.cpp:283, :297-306) cannot hold a
    one-cell-per-lane certificate: it starts with the exact body (class family 3).  The cost of a gap depends on its direction
    (:285-295): haplotype longer = match->ins f, ins->ins a, ins->match b; read longer = g, c, d.  Round 6: one threshold per
    direction (a model with a != c let the dearer direction's pairs through on the cheaper direction's threshold)."""
    fam = lambda n, m, **kw: class_info(classify(n, m, **kw)[0])["family"]
    # defaults: a = c = -1, open + close = 10.9: risky from |n - m| = 511
    assert fam(700, 700 - 510) != 3 and fam(700, 700 - 511) == 3
    assert fam(700 - 510, 700) != 3 and fam(700 - 511, 700) == 3
    # a != c: ins->ins -1.2 (haplotype longer: (520 - 5.3) / 1.2 + 1 = 430), del->del -0.9 (read longer: (520 - 4.5) / 0.9 + 1 = 574)
    asym = _abi.make_params((-1.2, -0.3, -0.9, -0.5, -0.0001, -5.0, -4.0))
    assert fam(700, 700 - 429, params=asym) != 3 and fam(700, 700 - 430, params=asym) == 3
    assert fam(700 - 573, 700, params=asym) != 3 and fam(700 - 574, 700, params=asym) == 3
    # not in the explicit modes (mode 0: every pair keeps its certificate class)
    assert fam(700, 100, mode=0) != 3
