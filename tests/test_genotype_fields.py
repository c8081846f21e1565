"""Genotype fields: Genotyper::extract_genotypes_and_likelihoods
(genotyper.cpp:132-256) -- product ltr_extract_genotypes (host code in the C-ABI library) against
the CPU restatement, and the restatement's building blocks against the reference's own mathops.cpp
(golden vectors from oracle/_ref, plus a live comparison when the reference build is present)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from longtr_amd import _lib

GOLD = os.path.join(os.path.dirname(__file__), "golden", "mathops.json")
fh = float.fromhex
bits = lambda a: np.asarray(a, dtype=np.float64).view(np.uint64)


def test_mathops_restatement_matches_reference_golden():
    g = json.load(open(GOLD))
    O = ol.oracle()
    assert fh(g["consts"]["LOG_THRESH"]) == np.log(0.001)
    assert fh(g["consts"]["LOG_E_BASE_10"]) == 0.4342944819 and fh(g["consts"]["TOLERANCE"]) == 1e-10
    for c in g["two"]:
        a, b = fh(c["a"]), fh(c["b"])
        assert float(O.ltr_oracle_fast_log_sum_exp2(a, b)).hex() == c["fast"], (a, b)
        assert float(O.ltr_oracle_log_sum_exp2(a, b)).hex() == c["exact"], (a, b)
    for c in g["streams"]:
        v = np.array([fh(x) for x in c["vals"]], dtype=np.float64)
        assert float(O.ltr_oracle_streaming_log_sum_exp(ol._p(v), len(v))).hex() == c["lse"]
    for c in g["int_log"]:
        assert float(O.ltr_oracle_int_log(c["v"])).hex() == c["log"]


@pytest.mark.skipif(not ol.have_ref(), reason="reference build (oracle/_ref) not present")
def test_mathops_restatement_matches_reference_live():
    O, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng(5)
    for _ in range(20000):
        a = float(-rng.exponential(10.0))
        b = a + float(rng.normal(0, rng.choice([1e-9, 0.1, 3.0, 20.0])))
        assert O.ltr_oracle_fast_log_sum_exp2(a, b) == R.ltr_ref_fast_log_sum_exp2(a, b)
        assert O.ltr_oracle_log_sum_exp2(a, b) == R.ltr_ref_log_sum_exp2(a, b)
    for _ in range(300):
        v = np.ascontiguousarray(-rng.exponential(20.0, size=int(rng.integers(1, 200))))
        assert O.ltr_oracle_streaming_log_sum_exp(ol._p(v), len(v)) == R.ltr_ref_streaming_log_sum_exp(ol._p(v), len(v))


def _random_case(rng, S, H, V, haploid):
    """A normalised posterior matrix like calc_log_sample_posteriors leaves it, its argmax, a map."""
    post = -rng.exponential(rng.choice([1.0, 30.0]), size=(S, H, H))
    if haploid:                                   # impossible heterozygotes carry the -DBL_MAX/2 prior
        for s in range(S):
            off = ~np.eye(H, dtype=bool)
            post[s][off] = -8.98846567431158e307
    stl = np.empty(S)
    for s in range(S):
        m = post[s].max()
        t = m + np.log(np.exp(post[s] - m).sum())
        post[s] -= t
        stl[s] = -rng.exponential(50.0)
    best = np.zeros((S, 2), dtype=np.int32)
    for s in range(S):
        k = int(np.argmax(post[s]))               # first maximum, row-major (genotyper.cpp:91-96)
        best[s] = (k // H, k % H)
    h2a = rng.integers(0, V, size=H).astype(np.int32)
    h2a[:min(H, V)] = np.arange(min(H, V))        # every allele reachable when H >= V
    return post, stl, best, h2a


@pytest.mark.parametrize("haploid", [False, True])
def test_extract_genotypes_matches_restatement(haploid):
    rng = np.random.default_rng(11 + haploid)
    for _ in range(60):
        S = int(rng.integers(1, 5)); H = int(rng.integers(1, 9)); V = int(rng.integers(1, H + 1))
        post, stl, best, h2a = _random_case(rng, S, H, V, haploid)
        got = _lib.extract_genotypes(post, stl, best, h2a, V, haploid)
        ref = ol.oracle_extract_genotypes(post, stl, best, h2a, V, haploid)
        assert set(got) == set(ref)
        for k in ref:
            if ref[k].dtype == np.float64:
                assert np.array_equal(bits(got[k]), bits(ref[k])), (k, got[k], ref[k])
            else:
                assert np.array_equal(got[k], ref[k]), (k, got[k], ref[k])


def test_extract_genotypes_properties():
    rng = np.random.default_rng(3)
    S, H = 3, 6
    post, stl, best, _ = _random_case(rng, S, H, H, False)
    ident = np.arange(H, dtype=np.int32)
    r = _lib.extract_genotypes(post, stl, best, ident, H, False)
    # identity map: genotype == haplotype pair; the phased posterior is the matrix entry itself
    assert np.array_equal(r["best_gts"], best)
    for s in range(S):
        a, b = best[s]
        assert r["log_phased_posteriors"][s] == pytest.approx(post[s, a, b], abs=1e-12)
        assert r["hap_log_phased_posteriors"][s] == post[s, a, b]
        assert r["log_unphased_posteriors"][s] >= r["log_phased_posteriors"][s]
        assert r["pls"][s].min() == 0 and r["pls"][s].max() <= 999
        assert r["gl_diffs"][s] == pytest.approx(np.sort(r["gls"][s])[-1] - np.sort(r["gls"][s])[-2], abs=1e-9) or r["gl_diffs"][s] <= 0
    # all haplotypes collapse onto one allele: a single genotype with posterior 1
    one = np.zeros(H, dtype=np.int32)
    r1 = _lib.extract_genotypes(post, stl, best, one, 1, False)
    assert np.all(r1["best_gts"] == 0)
    assert np.allclose(r1["log_phased_posteriors"], 0.0, atol=1e-9)
    assert r1["gls"].shape == (S, 1) and np.all(r1["pls"] == 0)


def test_extract_genotypes_rejects_bad_input():
    post = np.zeros((1, 2, 2)); stl = np.zeros(1)
    with pytest.raises(_lib.LtrError):
        _lib.extract_genotypes(post, stl, [[0, 2]], [0, 1], 2)          # haplotype index out of range
    with pytest.raises(_lib.LtrError):
        _lib.extract_genotypes(post, stl, [[0, 1]], [0, 2], 2)          # allele index out of range
