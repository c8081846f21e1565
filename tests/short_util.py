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


homopolymer_locus = synth.homopolymer_locus       # (the bench's neighbour measurement draws the same loci)
