"""Raw-read preparation and candidate haplotypes: product host code (ltr_prep.cpp through the
C-ABI) against the pure-Python restatement (oracle/ltr_oracle_prep.py) on synthetic raw long reads -- whole reads of
several kb with M/I/D/S/H CIGARs, HP tags, reads that do not span, soft clips, reads with the repeat deleted -- plus
an end-to-end check: raw reads -> left_align_reads -> build_haplotype reproduces the generator's true alleles, and
the prepared reads trimmed by the library equal the generator's own trimmed reads.  CPU only.  Parity with the
reference is UNPINNED (htslib / spoa)."""
import importlib.util
import os

import numpy as np

from longtr_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("ltr_oracle_prep", os.path.join(ROOT, "oracle", "ltr_oracle_prep.py"))
op = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(op)


def _raw_locus(rng, tr, period, n_alleles, n_reads, n_samples=2, read_flank=3000, err=0.002):
    """A chromosome window, a TR at its centre, and whole long reads (flanks of ~read_flank bp) with 'M' CIGARs the
    way an aligner reports them (no =/X), small indels, optional soft / hard clips, HP tags."""
    motif = synth._rand_seq(rng, period)
    rep0 = np.tile(motif, tr // period + 2)[:tr]
    lflank, rflank = synth._rand_seq(rng, read_flank + 300), synth._rand_seq(rng, read_flank + 300)
    chrom = np.concatenate([lflank, rep0, rflank])
    chrom_start = 50000
    region_start, region_stop = chrom_start + len(lflank), chrom_start + len(lflank) + tr
    reps = [rep0]
    for k in range(1, n_alleles):
        d = period * ((k + 1) // 2)
        reps.append(np.concatenate([rep0, np.tile(motif, d // period + 2)[:d]]) if k % 2 else rep0[:max(len(rep0) - d, period)])
    true = rng.choice(n_alleles, size=min(2, n_alleles), replace=False)
    raw, truth = [], []
    for i in range(n_reads):
        k = int(rng.choice(true)) if rng.random() < 0.9 else int(rng.integers(0, n_alleles))
        lo = int(rng.integers(200, 300)) if rng.random() < 0.85 else int(read_flank + 300 + 5)       # some reads start inside / after the TR
        hi = int(rng.integers(200, 300)) if rng.random() < 0.9 else int(read_flank + 300 + 5)
        left = lflank[lo:] if lo < len(lflank) else lflank[:0]
        right = rflank[:len(rflank) - hi] if hi < len(rflank) else rflank[:0]
        common = min(len(reps[k]), len(rep0))
        pieces = [("ref", np.concatenate([left, rep0[:common]]))]
        d = len(reps[k]) - len(rep0)
        if d > 0:
            pieces.append(("ins", reps[k][common:]))
        elif d < 0:
            pieces.append(("del", rep0[common:]))
        pieces.append(("ref", right))
        if lo >= len(lflank):                                       # read starts right at / after the TR start: does not span
            pieces = [("ref", np.concatenate([rep0[len(rep0) // 2:], right]))]
            start = region_start + len(rep0) // 2
        else:
            start = chrom_start + lo
        aln = synth._build_read(rng, pieces, err, err / 2, start)
        cig = []
        for t, n in aln["cigar"]:                                   # aligner style: =/X -> M
            t = "M" if t in "=X" else t
            if cig and cig[-1][0] == t:
                cig[-1] = (t, cig[-1][1] + n)
            else:
                cig.append((t, n))
        bases = aln["seq"]
        if i % 11 == 5:                                             # soft clip at the far end (survives if trimmed away, else the read is dropped)
            cig = [("S", 7)] + cig
            bases = b"ACGTACG" + bases
        if i % 13 == 6:
            cig = cig + [("H", 30)]
        if i % 5 == 0:
            bases = bases.lower()
        raw.append(dict(pos=aln["start"], end_pos=aln["stop"] + 1, bases=bases, cigar=cig, sample=int(rng.integers(0, n_samples)),
                        hp=int(rng.integers(0, 3)), quals=bytes(rng.integers(40, 70, size=len(bases)).astype(np.uint8))))
        truth.append(k)
    return dict(chrom=chrom.tobytes(), chrom_start=chrom_start, region=(region_start, region_stop), raw=raw, truth=truth,
                alleles=[r.tobytes() for r in reps], period=period, n_samples=n_samples)


def _compare_read_sets(rs, want):
    left, p1, p2, fail = want
    assert rs.size == len(left) and rs.fail_count == fail and rs.n_p1s == p1 and rs.n_p2s == p2
    for got, w in zip(rs.reads, left):
        assert got["start"] == w["start"] and got["stop"] == w["stop"] and got["seq"] == w["seq"].encode("latin-1")
        assert got["cigar"] == list(w["cigar"]) and got["aln"] == w["aln"].encode("latin-1")
        assert got["deleted"] == w["deleted"] and got["source"] == w["source"] and got["sample"] == w["sample"]


def test_left_align_reads_and_extract_sequence_equal_restatement():
    rng = np.random.default_rng(91)
    n_reads = 0
    for trial in range(25):
        d = _raw_locus(rng, int(rng.integers(12, 400)), int(rng.integers(1, 7)), int(rng.integers(1, 5)), 24,
                       read_flank=int(rng.integers(400, 4000)))
        rs = _lib.ReadSet(d["raw"], d["n_samples"], d["region"][0], d["region"][1], d["chrom"], d["chrom_start"])
        want = op.left_align_reads(d["raw"], d["n_samples"], d["region"][0], d["region"][1], d["chrom"], d["chrom_start"])
        _compare_read_sets(rs, want)
        n_reads += rs.size
        for i, w in enumerate(want[0]):
            assert w["stop"] - w["start"] <= (d["region"][1] - d["region"][0]) + 2 * 200 + 50      # cut to region -+ 200 bp
            for (a, b) in [(d["region"][0] - 5, d["region"][1] + 5), (d["region"][0], d["region"][1]), (w["start"] + 1, w["stop"] - 1),
                           (w["start"] - 3, w["stop"]), (d["region"][0] + 2, d["region"][0] + 2)]:
                ws = op.extract_sequence(w, a, b)
                gs = rs.extract_sequence(i, a, b)
                assert gs == (None if ws is None else ws.encode("latin-1"))
        rs.close()
    assert n_reads > 300


def test_deleted_repeat_and_clip_handling():
    rng = np.random.default_rng(92)
    d = _raw_locus(rng, 30, 3, 2, 4)
    rs0, re0 = d["region"]
    chrom, cs = d["chrom"], d["chrom_start"]
    span = lambda a, b: chrom[a - cs:b - cs]
    # a read whose CIGAR deletes the whole repeat (and the bases either side within -+200): everything left is trimmed away
    gone = dict(pos=rs0 - 1000, end_pos=re0 + 1000, bases=span(rs0 - 1000, rs0 - 250) + span(re0 + 250, re0 + 1000),
                cigar=[("M", 750), ("D", 500 + (re0 - rs0)), ("M", 750)], sample=0, hp=1)
    # a read with the repeat deleted but flank bases kept: TrimAlignment marks it deleted_, the read survives
    part = dict(pos=rs0 - 600, end_pos=re0 + 600, bases=span(rs0 - 600, rs0) + span(re0, re0 + 600),
                cigar=[("M", 600), ("D", re0 - rs0), ("M", 600)], sample=1, hp=2)
    clip = dict(pos=rs0 - 100, end_pos=re0 + 100, bases=b"ACGT" + span(rs0 - 100, re0 + 100), cigar=[("S", 4), ("M", re0 - rs0 + 200)], sample=0)
    short = dict(pos=rs0 + 1, end_pos=re0 + 500, bases=span(rs0 + 1, re0 + 500), cigar=[("M", re0 + 499 - rs0)], sample=0)
    raw = [gone, part, clip, short]
    rs = _lib.ReadSet(raw, 2, rs0, re0, chrom, cs)
    want = op.left_align_reads(raw, 2, rs0, re0, chrom, cs)
    _compare_read_sets(rs, want)
    assert rs.size == 2 and rs.fail_count == 2                       # soft clip inside the window, read that starts inside the repeat
    assert rs.reads[0]["deleted"] and rs.reads[0]["seq"] == b"" and (rs.reads[0]["start"], rs.reads[0]["stop"]) == (rs0, re0)
    assert rs.reads[1]["deleted"] and rs.reads[1]["cigar"] == [("=", 200), ("D", re0 - rs0), ("=", 200)]
    assert rs.extract_sequence(0, rs0 - 5, re0 + 5) == b"" and rs.n_p1s == [0, 0] and rs.n_p2s == [0, 1]   # (the placeholder read is added before the HP count, :62-71)


def test_build_haplotype_equals_restatement_and_recovers_true_alleles():
    rng = np.random.default_rng(93)
    recovered = 0
    for trial in range(30):
        period = int(rng.integers(2, 7))
        d = _raw_locus(rng, int(rng.integers(20, 300)), period, int(rng.integers(2, 5)), 30, err=0.0005)
        rs0, re0 = d["region"]
        rs = _lib.ReadSet(d["raw"], d["n_samples"], rs0, re0, d["chrom"], d["chrom_start"])
        left = op.left_align_reads(d["raw"], d["n_samples"], rs0, re0, d["chrom"], d["chrom_start"])[0]
        chrom_len = d["chrom_start"] + len(d["chrom"])
        got = rs.build_haplotype(rs0, re0, period, d["chrom_start"], chrom_len)
        want = op.build_haplotype(left, d["n_samples"], rs0, re0, period, d["chrom"], d["chrom_start"], chrom_len)
        assert got == want, (trial, got, want)
        if got["blocks"] is None:
            continue
        b = got["blocks"]
        assert len(b) == 3 and b[0]["end"] == b[1]["start"] and b[1]["end"] == b[2]["start"] and len(b[0]["alleles"][0]) <= 35
        assert b[1]["alleles"][0] == d["chrom"][b[1]["start"] - d["chrom_start"]:b[1]["end"] - d["chrom_start"]]
        # alleles carried by >= 2 error-free reads of a sample are candidates: the generator's true alleles are among them
        core = lambda a: d["chrom"][b[1]["start"] - d["chrom_start"]:rs0 - d["chrom_start"]] + a + d["chrom"][re0 - d["chrom_start"]:b[1]["end"] - d["chrom_start"]]
        used = {k for k in d["truth"]}
        hits = sum(core(d["alleles"][k]) in b[1]["alleles"] for k in used if d["truth"].count(k) >= 8)
        recovered += hits
        # downstream: the library's own trimming of the prepared reads (HapAligner::trim_alignment) accepts them
        alns = [dict(start=r["start"], stop=r["stop"], seq=r["seq"], cigar=r["cigar"]) for r in rs.reads if not r["deleted"]]
        for a in alns[:5]:
            rc, lt, rt = _lib.trim_alignment(a, b[1]["start"], b[1]["end"], 5)
            assert rc == 0 and lt + rt < len(a["seq"])
        rs.close()
    assert recovered >= 30


def test_phasing_priors_equal_the_restatement():
    """ltr_phasing_priors = SNPBamProcessor::process_phased_reads for unpaired reads (snp_bam_processor.cpp:141-226): hand cases
    (a sample with too many untagged reads, the verdict staying for the samples after it, a first sample without reads) and
    random ones against the Python restatement."""
    import pytest
    # sample 0: 2 + 2 tagged -> phased; sample 1: running totals 8 reads, 2 untagged = 0.25 > 0.2 -> not phased
    p1, p2, n = _lib.phasing_priors([0, 0, 0, 0, 1, 1, 1, 1], [1, 1, 2, 2, 1, 2, -1, -1], 2)
    assert n == 4 and list(p1[:4]) == [-0.000001, -0.000001, -1000.0, -1000.0] and list(p2[:4]) == [-1000.0, -1000.0, -0.000001, -0.000001]
    assert not p1[4:].any() and not p2[4:].any()
    # the verdict is sticky: sample 0 fails (one read of haplotype 2), sample 1 alone would pass
    p1, p2, n = _lib.phasing_priors([0, 0, 0, 1, 1, 1, 1], [1, 1, 2, 1, 1, 2, 2], 2)
    assert n == 0 and not p1.any() and not p2.any()
    # a first sample without reads: 0 / 0 compares false, but "at most one read of haplotype 2" holds -> nobody is phased
    p1, p2, n = _lib.phasing_priors([1, 1, 1, 1], [1, 1, 2, 2], 2)
    assert n == 0
    rng = np.random.default_rng(12)
    for _ in range(300):
        S = int(rng.integers(1, 5))
        R = int(rng.integers(0, 40))
        so = rng.integers(0, S, size=R)
        hp = rng.choice([-1, 1, 2], size=R, p=[0.1, 0.45, 0.45])
        want = op.phasing_priors(list(so), list(hp), S)
        got = _lib.phasing_priors(so, hp, S)
        assert list(got[0]) == want[0] and list(got[1]) == want[1] and got[2] == want[2]
    with pytest.raises(_lib.LtrError):
        _lib.phasing_priors([0], [3], 1)                       # assert(haplotype == 1 || haplotype == 2), :132
