"""CPU: the C oracle against the committed golden vectors (outputs of the real reference code)
and, when the reference build is present (dev container), against the reference directly."""
import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from longtr_amd import _abi, synth


def bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("gi", range(len(gu.load("align_long")["groups"])))
def test_align_long_golden(gi):
    g = gu.load("align_long")["groups"][gi]
    b = gu.group_batch(g)
    ll, _, _ = ol.oracle_align_batch(b, gu.params_from(g["params"]))
    assert np.array_equal(bits(ll), bits(gu.unhex(g["ll_hex"]))), g["name"]


def test_survey_known_answer_values():
    g = gu.load("align_long")["groups"][0]
    assert g["name"] == "survey_known_answer_cag"
    # -12.9194140592, -0.0130565529, -18.9184660191
    assert np.allclose(gu.unhex(g["ll_hex"]), [-12.9194140592, -0.0130565529, -18.9184660191], atol=5e-11)


def test_golden_covers_sentinels():
    allv = np.concatenate([gu.unhex(g["ll_hex"]) for g in gu.load("align_long")["groups"]])
    assert (allv == -700.0).sum() >= 5 and (allv == -1e9).sum() >= 5 and ((allv > -600) & (allv < 0)).sum() > 100


def test_process_locus_golden():
    d = gu.load("process_locus")
    p = gu.params_from(d["params"])
    for L in d["loci"]:
        rc, probs, seeds = ol.oracle_process_reads(p, gu.locus_blocks(L), gu.locus_alns(L))
        assert rc == 0
        assert np.array_equal(bits(probs.ravel()), bits(gu.unhex(L["ll_hex"])))
        assert list(seeds) == [len(a["seq"]) - 1 for a in L["alns"]]        # HapAligner.cpp:562-563
        # trimmed lengths as the reference produced them (10 = the empty-trim substitute)
        s1 = L["start"] + len(L["lflank"])
        e1 = s1 + len(L["alleles"][0])
        for a, tl in zip(gu.locus_alns(L), L["trim_len"]):
            rc, lt, rt = ol.oracle_trim(a, s1, e1, p.indel_flank_len)
            assert rc == 0
            n = len(a["seq"]) - lt - rt
            assert (n if n > 0 else 10) == tl
            t, lt2, rt2 = synth.trim_like_reference(a, s1, e1, p.indel_flank_len)
            assert (lt2, rt2) == (lt, rt)
        assert 10 in L["trim_len"]


def test_pooling_golden():
    for s in gu.load("pooling")["sets"]:
        reads = [r.encode() for r in s["reads"]]
        import ctypes as C
        keep = [np.frombuffer(r, dtype=np.uint8).copy() for r in reads]
        ptrs = (C.c_void_p * len(reads))(*[k.ctypes.data for k in keep])
        lens = np.asarray([len(r) for r in reads], dtype=np.int32)
        idx = np.zeros(len(reads), dtype=np.int32)
        n = ol.oracle().ltr_oracle_pool_reads(ptrs, lens.ctypes.data_as(C.c_void_p), len(reads),
                                              idx.ctypes.data_as(C.c_void_p))
        assert n == s["n_pools"] and list(idx) == s["pool_index"]
        assert synth.pool_reads(reads)[1] == s["pool_index"]


def test_posterior_known_answer():
    c = gu.load("posteriors")["cases"][0]
    r = ol.oracle_posteriors(np.asarray(c["ll"]), c["log_p1"], c["log_p2"], c["sample_label"], c["n_samples"])
    assert f"{r['total_ll']:.10f}" == c["total_ll_10dp"]
    assert list(r["gts"][0]) == c["gt"]
    assert r["clamped_ll"][3, 2] == -600.0                     # clamp in place, genotyper.cpp:57-58
    assert abs(np.exp(r["post"][0]).sum() - 1.0) < 1e-12


def test_rolling_matches_materialised():
    rng = np.random.default_rng(3)
    p = _abi.default_params()
    for _ in range(60):
        L = synth.synth_locus(rng, int(rng.integers(1, 200)), int(rng.integers(1, 7)), 3, 2, sub_rate=0.02, indel_rate=0.02)
        for r in L.trimmed_reads:
            for h in L.haplotypes:
                a = ol.oracle_align_long(h, r, p)
                b = ol.oracle_align_long(h, r, p, rolling=True)
                assert bits([a])[0] == bits([b])[0]


@pytest.mark.skipif(not ol.have_ref(), reason="reference build (oracle/_ref) not present")
def test_oracle_vs_reference_random():
    rng = np.random.default_rng(12345)
    for params in (_abi.default_params(), _abi.make_params(synth.ONT_PARAMS)):
        loci = [synth.synth_locus(rng, int(rng.integers(1, 260)), int(rng.integers(1, 7)), int(rng.integers(2, 6)), 4,
                                  sub_rate=0.02, indel_rate=0.02) for _ in range(40)]
        b, _ = synth.pack_loci(loci)
        a, _, _ = ol.oracle_align_batch(b, params)
        r, _ = ol.ref_align_batch(b, params)
        # m <= n+1 everywhere here (alleles differ by a few periods), so the reference is defined
        assert np.array_equal(bits(a), bits(r))


@pytest.mark.skipif(not ol.have_ref(), reason="reference build (oracle/_ref) not present")
def test_oracle_vs_reference_raw_reads():
    rng = np.random.default_rng(777)
    p = _abi.default_params()
    for _ in range(10):
        L = synth.synth_locus(rng, int(rng.integers(5, 150)), int(rng.integers(1, 7)), 4, 6, sub_rate=0.02,
                              indel_rate=0.04, raw=True)
        ll, _, tlen = ol.ref_process_locus(L, p)
        rc, probs, _ = ol.oracle_process_reads(p, L.blocks(), L.raw_alns)
        assert rc == 0 and np.array_equal(bits(probs.ravel()), bits(ll.ravel()))
        assert [len(t) for t in L.trimmed_reads] == list(tlen)
