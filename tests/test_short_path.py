"""CPU: the short (stutter) path restatement, pinned to the COMPILED REFERENCE (oracle/_ref) wherever the reference links
without htslib: its inside -- the stutter-block rows: StutterAlignerClass, RepeatStutterInfo, StutterModel, BaseQuality,
fast_log_sum_exp(vector) (tests/golden/stutter_pieces.json + live) -- and the outer functions compute_aln_logprob,
calc_best_seed_position, calc_seed_base (tests/golden/short_outer.json + live; the product's host calc_seed_base too).
Only the flank rows of align_seq_to_hap_short stay on the SURVEY.md 8c known answers: they call
Haplotype::homopolymer_length (HapAligner.cpp:121-122 -> Haplotype.cpp:280 -> bam_io.h -> htslib)."""
import numpy as np

import oracle_lib as ol
import short_util as su
from longtr_amd import _abi


def test_known_answer_values():
    blocks, alns = su.known_answer_case()
    rc, probs, seeds = ol.oracle_process_reads_short(_abi.default_params(), _abi.default_stutter_params(), blocks, alns)
    assert rc == 0
    assert [f"{x:.10f}" for x in probs[0]] == ["-7.8693081508", "-4.3896419406"]
    assert seeds[0] == 77


def test_seed_base_rules():
    blocks, alns = su.known_answer_case()
    a = alns[0]
    # seed = middle of the longest '=' run outside the repeat, >= 5 bp from its ends (HapAligner.cpp:494-542)
    assert ol.oracle_calc_seed_base(a, blocks) == 77
    short = dict(a, cigar=[("=", 4), ("X", 1)] * 18 + [("=", 5)])        # no '=' run reaches MIN_SEED_DIST
    short["seq"] = a["seq"][:95]
    short["qual"] = a["qual"][:95]
    assert ol.oracle_calc_seed_base(short, blocks) == -1
    bad = dict(a, cigar=[("S", 3)] + list(a["cigar"]))
    assert ol.oracle_calc_seed_base(bad, blocks) == -2


def test_no_seed_gives_zero_row_and_masks_hold():
    rng = np.random.default_rng(4)
    blocks, alns = su.homopolymer_locus(rng, 15, 3, 5)
    alns[1] = dict(alns[1], cigar=[("X", len(alns[1]["seq"]))])          # no '=' run at all -> seed -1
    rr = np.array([1, 1, 0, 1, 1], dtype=np.uint8)
    rh = np.array([1, 0, 1], dtype=np.uint8)
    rc, probs, seeds = ol.oracle_process_reads_short(_abi.default_params(), _abi.default_stutter_params(), blocks, alns,
                                                     realign_hap=rh, realign_read=rr)
    assert rc == 0
    assert seeds[1] == -1 and (probs[1] == 0).all()                      # HapAligner.cpp:570-574
    assert np.isnan(probs[2]).all() and seeds[2] == -12345               # masked read untouched
    assert np.isnan(probs[[0, 3, 4], 1]).all() and np.isfinite(probs[[0, 3, 4]][:, [0, 2]]).all()
    assert (probs[[0, 3, 4]][:, [0, 2]] < 1e-10).all()                   # assert(total_LL < TOLERANCE), :231


# ---- the pieces that compile without htslib, pinned to the COMPILED REFERENCE (oracle/_ref) ---------------------------
# StutterAlignerClass (ctor, load_read, align_pcr_insertion_reverse, align_pcr_deletion_reverse), RepeatStutterInfo,
# StutterModel::log_stutter_pmf, BaseQuality, fast_log_sum_exp(vector): tests/golden/stutter_pieces.json holds the
# reference's outputs (oracle/gen_golden_short.py), bit for bit.
def _golden():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stutter_pieces.json")))


def _sp(g, i):
    return _abi.StutterParams(*g["stutter_params"][i])


def _bits(x):
    return np.asarray(x, dtype=np.float64).view(np.uint64)


def test_stutter_block_rows_equal_the_compiled_reference():
    g = _golden()
    assert len(g["rows"]) >= 100
    seen_ins = seen_del = 0
    for r in g["rows"]:
        prev = np.array([float.fromhex(h) for h in r["prev_row"]])
        want = np.array([float.fromhex(h) for h in r["match"]])
        got, _, _ = ol.stutter_block_row("oracle", _sp(g, r["sp"]), r["block"].encode(), r["period"], r["left_align"],
                                         r["seq"].encode(), bytes.fromhex(r["qual_hex"]), prev)
        assert np.array_equal(_bits(got), _bits(want)), (r["block"], r["seq"], got[:4], want[:4])
        seen_ins += sum(1 for s in r["art_size"] if s > 0)
        seen_del += sum(1 for s in r["art_size"] if -10000 < s < 0)
    assert seen_ins > 10 and seen_del > 100           # positions whose best artifact is an insertion / a deletion both occur


def test_stutter_scalars_equal_the_compiled_reference():
    g = _golden()
    for si in range(len(g["stutter_params"])):
        pmf, art, _, _ = ol.stutter_scalars("oracle", _sp(g, si))
        for s, m, a, b, h in g["pmf"]:
            if s == si:
                assert pmf(m, a, b).hex() == h, (m, a, b)
        for s, p, a, d, h in g["artifact"]:
            if s == si:
                assert art(p, a, d).hex() == h, (p, a, d)
    _, _, bq, lse = ol.stutter_scalars("oracle", _sp(g, 0))
    for q, e, c in g["base_quality"]:
        assert tuple(x.hex() for x in bq(q)) == (e, c), q
    for vals, h in g["fast_lse"]:
        assert lse([float.fromhex(v) for v in vals]).hex() == h


def test_stutter_pieces_live_against_the_reference_build():
    """Fresh random cases against oracle/_ref itself (dev container; the GPU box has the prebuilt file too)."""
    import pytest
    if not ol.have_ref():
        pytest.skip("oracle/_ref/libltr_ref.so not built")
    rng = np.random.default_rng(99)
    sp = _abi.default_stutter_params()
    for it in range(150):
        period = 1 if it % 2 else int(rng.integers(2, 5))
        block = bytes(int(x) for x in rng.choice(list(b"AACT"), size=period)) * int(rng.integers(0, 20))
        seq = bytes(int(x) for x in rng.choice(list(b"AAACT"), size=int(rng.integers(1, 60))))
        qual = bytes(int(q) for q in rng.integers(30, 80, size=len(seq)))
        prev = np.cumsum(-rng.random(len(seq)))
        a, _, _ = ol.stutter_block_row("oracle", sp, block, period, it % 2, seq, qual, prev)
        b, _, _ = ol.stutter_block_row("ref", sp, block, period, it % 2, seq, qual, prev)
        assert np.array_equal(_bits(a), _bits(b)), (block, seq)


# ---- the outer functions that link against the compiled reference: compute_aln_logprob (HapAligner.cpp:165-233),
# calc_best_seed_position (:467-493), calc_seed_base (:494-542).  tests/golden/short_outer.json holds the reference's outputs
# (oracle/gen_golden_seed.py); the C restatement AND the product's host calc_seed_base (ltr_debug_calc_seed_base) must equal them.
def _outer():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "short_outer.json")))


def _bblocks(jb):
    return [dict(b, alleles=[a.encode() for a in b["alleles"]]) for b in jb]


def _cigar(s):
    import re
    return [(t, int(k)) for k, t in re.findall(r"(\d+)([=XID])", s)]


def test_calc_seed_base_equals_the_compiled_reference():
    g = _outer()
    assert len(g["seed_base"]) >= 300
    outcomes = set()
    for c in g["seed_base"]:
        blocks = _bblocks(c["blocks"])
        aln = dict(start=c["start"], stop=c["stop"], seq=b"A" * c["seq_len"], cigar=_cigar(c["cigar"]), qual=b"I" * c["seq_len"])
        assert ol.calc_seed_base("oracle", aln, blocks) == c["seed"], c["cigar"]
        assert ol.calc_seed_base("product", aln, blocks) == c["seed"], c["cigar"]          # ltr_short.hip's host function
        outcomes.add(c["seed"] >= 0)
    assert outcomes == {True, False}


def test_calc_best_seed_position_equals_the_compiled_reference():
    g = _outer()
    none = 0
    for rs, re_, a0, a1, d, q in g["best_seed_position"]:
        assert ol.calc_best_seed_position("oracle", rs, re_, a0, a1) == (d, q), (rs, re_, a0, a1)
        none += d == -1
    assert 0 < none < len(g["best_seed_position"])


def test_compute_aln_logprob_equals_the_compiled_reference():
    g = _outer()
    assert len(g["aln_logprob"]) >= 200
    nblocks = set()
    for c in g["aln_logprob"]:
        c = dict(c, blocks=_bblocks(c["blocks"]))
        lM, rM = su.logprob_matrices(c)
        v, _ = ol.compute_aln_logprob("oracle", c["blocks"], c["counts"], c["base_seq_len"], c["seed_base"], c["seed_char"],
                                      float.fromhex(c["log_seed_wrong"]), float.fromhex(c["log_seed_correct"]), lM, float.fromhex(c["l_prob"]),
                                      rM, float.fromhex(c["r_prob"]))
        assert v.hex() == c["total_LL"], (c["counts"], c["seed_base"])
        nblocks.add(len(c["blocks"]))
    assert nblocks >= {3, 5}                          # one and more repeat blocks per haplotype


def test_short_outer_functions_live_against_the_reference_build():
    """Fresh random cases against oracle/_ref itself: all three functions, the product's calc_seed_base too."""
    import pytest
    if not ol.have_ref():
        pytest.skip("oracle/_ref/libltr_ref.so not built")
    rng = np.random.default_rng(2026)
    for it in range(300):
        blocks, aln = su.seed_case(rng)
        want = ol.calc_seed_base("ref", aln, blocks)
        assert ol.calc_seed_base("oracle", aln, blocks) == want and ol.calc_seed_base("product", aln, blocks) == want
    for it in range(150):
        c = su.logprob_case(rng, 5 + 11 * it)
        lM, rM = su.logprob_matrices(c)
        args = (c["blocks"], c["counts"], c["base_seq_len"], c["seed_base"], c["seed_char"], c["log_seed_wrong"], c["log_seed_correct"], lM, c["l_prob"], rM, c["r_prob"])
        a, _ = ol.compute_aln_logprob("oracle", *args)
        b, _ = ol.compute_aln_logprob("ref", *args)
        assert a.hex() == b.hex()
