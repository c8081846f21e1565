"""CPU: the short (stutter) path restatement against the only reference outputs that exist for it
(SURVEY.md 8c known answers) -- parity is otherwise UNPINNED (Haplotype.cpp cannot be built here)."""
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
