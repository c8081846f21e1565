"""Haplotype -> reference-haplotype alignment: the C restatement of
Haplotype::aln_haps_to_ref (NeedlemanWunsch::Align + adjust_indels + M / I / D string) on hand-checked cases
(CPU), and the HIP kernel behind ltr_haplotype_align_to_ref against it (-m gpu).  Parity with the reference
is UNPINNED (NeedlemanWunsch.h needs htslib)."""
import numpy as np
import pytest

import oracle_lib as ol
from longtr_amd import synth

LF = b"ACGTTGCAAGCTTAGGCTAACGTTAGCCATGGATC"
RF = b"GGATCCTTAGCAATCGGATTACAGGCTTAACCGTA"


def test_restatement_hand_checked():
    pl, pr = b"TTGAC", b"CAGTT"
    ref = LF + pl + b"CAG" * 6 + pr + RF
    s0, s1 = 1000, 1035
    assert ol.oracle_nw_aln_info(ref, ref, s0, s1) == "M" * len(ref)
    # one unit more: the insertion is left-aligned by the DP (ties prefer extending from the match state last) --
    # it must sit inside the repeat block, 3 I's, everything else M
    ins = ol.oracle_nw_aln_info(ref, LF + pl + b"CAG" * 7 + pr + RF, s0, s1)
    assert ins.count("I") == 3 and ins.count("D") == 0 and ins.count("M") == len(ref) and "III" in ins
    assert 35 <= ins.index("III") <= 35 + 5 + 18
    dele = ol.oracle_nw_aln_info(ref, LF + pl + b"CAG" * 4 + pr + RF, s0, s1)
    assert dele.count("D") == 6 and dele.count("I") == 0 and "DDDDDD" in dele and 35 <= dele.index("DDDDDD")
    # a substitution is an M; N matches everything (score table :88-92)
    sub = bytearray(ref); sub[50] = ord("A") if sub[50] != ord("A") else ord("C")
    assert ol.oracle_nw_aln_info(ref, bytes(sub), s0, s1) == "M" * len(ref)
    # adjust_indels: a deletion the DP leaves in the left flank slides right while the bases allow it
    ref2 = b"G" * 20 + b"AAAAAAAAAA" + b"C" * 20
    alt2 = b"G" * 20 + b"AAAAAAAA" + b"C" * 20
    info = ol.oracle_nw_aln_info(ref2, alt2, 0, 25)           # repeat block "starts" at 25: inside the A run
    assert info.count("D") == 2 and info.index("DD") >= 20


def _consistent(info, L1, L2):
    return info.count("M") + info.count("D") == L1 and info.count("M") + info.count("I") == L2 and set(info) <= set("MID")


def test_restatement_lengths_consistent_on_random_pairs():
    rng = np.random.default_rng(81)
    for _ in range(40):
        L = synth.synth_locus(rng, int(rng.integers(5, 80)), int(rng.integers(1, 7)), int(rng.integers(2, 6)), 1)
        haps = L.haplotypes
        for h in haps:
            info = ol.oracle_nw_aln_info(haps[0], h, L.start, L.start + 35)
            assert _consistent(info, len(haps[0]), len(h))


@pytest.mark.gpu
def test_gpu_kernel_equals_restatement(gpu_ctx):
    rng = np.random.default_rng(82)
    loci = []
    for k in range(60):
        tr = int(rng.integers(5, 400)) if k % 10 else int(rng.integers(900, 1200))
        L = synth.synth_locus(rng, tr, int(rng.integers(1, 40)), int(rng.integers(1, 8)), 1)
        if k % 7 == 3:                                             # sequence-level differences too, not only length
            a = bytearray(L.alleles[-1]); a[len(a) // 2] = ord("A") if a[len(a) // 2] != ord("A") else ord("G"); L.alleles[-1] = bytes(a)
        loci.append(L)
    long_l = synth.synth_locus(rng, 2400, 31, 3, 1)               # references > 1280 bases: the workgroup kernel, rolling diagonals in global memory
    loci.append(long_l)
    mid_l = synth.synth_locus(rng, 1400, 17, 3, 1)                # ... in LDS
    loci.append(mid_l)
    for tr in (1, 2, 180, 186, 187, 188, 442, 443, 444, 698, 699, 700, 954, 955, 956, 1208, 1209, 1210, 1211):   # reference lengths around every strip-width edge (64 W - 70)
        loci.append(synth.synth_locus(rng, tr, int(rng.integers(1, 7)), int(rng.integers(2, 5)), 1))
    odd = synth.synth_locus(rng, 90, 4, 4, 1)                     # N and lower-case bases (base_to_int: anything else matches everything)
    a = bytearray(odd.alleles[1]); a[3] = ord("N"); a[7] = ord("c"); odd.alleles[1] = bytes(a)
    a = bytearray(odd.alleles[0]); a[5] = ord("n"); a[11] = ord("g"); odd.alleles[0] = bytes(a)
    loci.append(odd)
    # gaps the DP leaves in the LEFT FLANK (adjust_indels slides them right: the library's string-building path, not its
    # "codes read backwards" shortcut): homopolymer runs that straddle the flank / repeat boundary, deletions and insertions
    flank_gaps = []
    for run, ref_n, alt_ns in ((b"A", 5, (3, 8, 1)), (b"T", 9, (2, 14)), (b"C", 3, (1, 6, 9))):
        lf = bytes(_b for _b in (b"G" if run != b"G" else b"T") * 30) + run * 5
        fl = synth.Locus(start=500, period=1, lflank=lf, rflank=b"ACGTACGTTGCATGCAAGCTTAGGCTAACGTTAGC",
                         alleles=[run * ref_n] + [run * k for k in alt_ns], read_allele=[], trimmed_reads=[])
        flank_gaps.append(fl)
        loci.append(fl)
    got = gpu_ctx.haplotype_align_to_ref([L.blocks() for L in loci])
    n = slid = 0
    for L, infos in zip(loci, got):
        haps = L.haplotypes
        assert len(infos) == len(haps)
        for h, info in zip(haps, infos):
            want = ol.oracle_nw_aln_info(haps[0], h, L.start, L.start + len(L.lflank))
            if L in flank_gaps and ("D" in info or "I" in info):
                first_gap = min(info.index(c) for c in "DI" if c in info)
                slid += first_gap >= 30                              # the gap sits behind the 30 non-run bases of the flank
            assert info == want, (len(haps[0]), len(h))
            assert _consistent(info, len(haps[0]), len(h))
            n += 1
    assert n > 200 and slid >= 6
    tm = gpu_ctx.timers(reset=True)
    assert tm["hap_build_calls"] >= 1 and tm["hap_build_s"] > 0
    assert tm["nw_kernel_ms"] > 0 and tm["nw_kernel_ms"] < tm["hap_build_s"] * 1e3     # device time of the NW kernels, inside the call's wall time
