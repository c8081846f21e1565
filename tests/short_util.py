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


# ---- inputs for the pins of compute_aln_logprob / calc_seed_base / calc_best_seed_position (HapAligner.cpp:165-233, :467-542)
def lcg_matrix(seed, n):
    """n doubles in (-977.2, -0.5], every 17th residue class IMPOSSIBLE-like (-1e9): exact integer arithmetic + one
    correctly rounded division, so that a golden file can name a matrix by its seed."""
    k = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)) % np.uint64(1000003)
    v = -0.5 - k.astype(np.float64) / 1024.0
    v[k % np.uint64(17) == 0] = -1.0e9
    return v


def _rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(int(x) for x in rng.choice(list(alphabet), size=n))


def random_blocks(rng, n_repeats=None, start=None):
    """flank | repeat | flank [| repeat | flank ...]: contiguous coordinates, 1-3 alleles per repeat block."""
    nrep = int(rng.integers(1, 4)) if n_repeats is None else n_repeats
    pos = int(rng.integers(100, 5000)) if start is None else start
    blocks = []
    for b in range(2 * nrep + 1):
        if b % 2 == 0:
            seq = _rand_seq(rng, int(rng.integers(3, 40)))
            blocks.append(dict(start=pos, end=pos + len(seq), is_repeat=False, period=0, alleles=[seq]))
        else:
            period = int(rng.integers(1, 4))
            motif = _rand_seq(rng, period)
            units = int(rng.integers(1, 15))
            ref = motif * units
            alts = [motif * max(units + int(d), 0) for d in rng.choice([-2, -1, 1, 2, 3], size=int(rng.integers(0, 3)), replace=False)]
            alts = [a for a in alts if len(a) > 0]
            seq = ref
            blocks.append(dict(start=pos, end=pos + len(seq), is_repeat=True, period=period, alleles=[ref] + alts))
        pos += len(seq)
    return blocks


def seed_case(rng):
    """(blocks, alignment) for calc_seed_base: a random =/X/I/D CIGAR that starts before, at or inside the first block and
    may run past the last one."""
    blocks = random_blocks(rng)
    first, last = blocks[0]["start"], blocks[-1]["end"]
    start = first + int(rng.integers(-30, 12))
    cigar, pos, nbases = [], start, 0
    stop_at = last + int(rng.integers(-10, 30))
    while pos < stop_at:
        t = "=XID"[int(rng.choice(4, p=[0.55, 0.15, 0.15, 0.15]))]
        k = int(rng.integers(1, 30)) if t == "=" else int(rng.integers(1, 4))
        if cigar and cigar[-1][0] == t:
            continue
        cigar.append((t, k))
        if t in "=XD":
            pos += k
        if t in "=XI":
            nbases += k
    seq = _rand_seq(rng, max(nbases, 1))
    return blocks, dict(start=start, stop=pos - 1, seq=seq, cigar=cigar, qual=b"I" * len(seq))


def logprob_case(rng, mseed):
    """Arguments of compute_aln_logprob for a random haplotype (random allele per block) and seed position."""
    blocks = random_blocks(rng)
    counts = [int(rng.integers(0, len(b["alleles"]))) for b in blocks]
    hapsize = sum(len(b["alleles"][c]) for b, c in zip(blocks, counts))
    base_seq_len = int(rng.integers(3, 40))
    seed_base = int(rng.integers(1, base_seq_len - 1))
    lflank, rflank = seed_base, base_seq_len - seed_base - 1
    q = int(rng.integers(2, 42))
    err = 10.0 ** (-q / 10.0)
    return dict(blocks=blocks, counts=counts, base_seq_len=base_seq_len, seed_base=seed_base, seed_char=int(rng.choice(list(b"ACGT"))),
                log_seed_wrong=float(np.log(err / 3.0)), log_seed_correct=float(np.log1p(-err)), mseed=int(mseed),
                l_prob=-float(rng.random() * 30) - 0.5, r_prob=-float(rng.random() * 30) - 0.5, n_l=lflank * hapsize, n_r=rflank * hapsize)


def logprob_matrices(c):
    return lcg_matrix(c["mseed"], c["n_l"]), lcg_matrix(c["mseed"] + 7919, c["n_r"])
