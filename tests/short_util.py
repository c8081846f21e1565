"""Shared builders for the short (stutter) path tests."""
import numpy as np

from longtr_amd import synth

KA_LF = b"ACGTTGCAAGCTTAGGCTAACGTTAGCCATGGATC"
KA_RF = b"GGATCCTTAGCAATCGGATTACAGGCTTAACCGTA"


def known_answer_case():
    """The probe behind SURVEY.md 8c's short-path values (-7.8693081508, -4.3896419406): a 14-bp
    poly-A locus with two alleles and one read carrying the +1 allele, all qualities 'I'."""
    pl, pr, rep = b"TTGAC", b"CAGTT", b"A" * 14
    s0 = 1000
    blocks = [dict(start=s0, end=s0 + 35, is_repeat=False, period=0, alleles=[KA_LF]),
              dict(start=s0 + 35, end=s0 + 59, is_repeat=True, period=1, alleles=[pl + rep + pr, pl + rep + b"A" + pr]),
              dict(start=s0 + 59, end=s0 + 94, is_repeat=False, period=0, alleles=[KA_RF])]
    rs = KA_LF + pl + rep + b"A" + pr + KA_RF
    aln = dict(start=s0, stop=s0 + 93, seq=rs, cigar=[("=", 54), ("I", 1), ("=", 40)], qual=b"I" * len(rs))
    return blocks, [aln]


def homopolymer_locus(rng, tr_len, n_alleles, n_reads, sub_rate=0.01, indel_rate=0.02):
    """Period-1 locus with raw reads (exact =/X/I/D CIGARs) and random Phred+33 qualities."""
    L = synth.synth_locus(rng, tr_len, 1, n_alleles, n_reads, sub_rate=sub_rate, indel_rate=indel_rate, raw=True)
    alns = []
    for a in L.raw_alns:
        q = rng.integers(ord("!") + 2, ord("J") + 1, size=len(a["seq"])).astype(np.uint8)
        if rng.random() < 0.2:
            q[rng.integers(0, len(q))] = ord("~")          # above 'J': clamped (base_quality.h:49-51)
        alns.append(dict(a, qual=q.tobytes()))
    return L.blocks(), alns
