"""The genotyper's last steps: unused-allele pruning, haplotype re-mapping, allele
extraction and the VCF record -- product code (longtr_amd/csrc/ltr_vcf.cpp, through the C-ABI) against
the C restatement (oracle/ltr_oracle_vcf.c) on random loci, and against hand-checked records.  CPU only:
alignment probabilities come from the oracle DP, posteriors from the oracle's calc_log_sample_posteriors.
Parity with the reference itself is UNPINNED for these functions (seq_stutter_genotyper.cpp needs htslib)."""
import numpy as np

import oracle_lib as ol
from longtr_amd import _abi, _lib, synth


def _locus(rng, tr, period, H, R, S, haploid=False, deleted_allele=False, phased=True):
    L = synth.synth_locus(rng, tr, period, H, R, sub_rate=0.002, indel_rate=0.002, raw=True, start=5000)
    blocks, alns = L.blocks(), L.raw_alns
    if deleted_allele:                                           # the whole repeat block deleted in one alternate -> "<DEL>"
        blocks[1]["alleles"] = list(blocks[1]["alleles"]) + [b""]
    prm = _abi.default_params()
    rc, ll, _ = ol.oracle_process_reads(prm, blocks, alns)
    assert rc == 0
    Hn = ll.shape[1]
    lab = rng.integers(0, S, size=R).astype(np.int32)
    lab[:S] = np.arange(S)                                       # every sample has a read
    if phased:
        tag = rng.integers(0, 3, size=R)
        p1 = np.where(tag == 1, -0.000001, np.where(tag == 2, -1000.0, 0.0))
        p2 = np.where(tag == 2, -0.000001, np.where(tag == 1, -1000.0, 0.0))
    else:
        p1 = p2 = np.zeros(R)
    post = ol.oracle_posteriors(ll, p1, p2, lab, S, haploid=haploid)
    s0 = blocks[1]["start"]
    chrom = synth._rand_seq(rng, 400).tobytes().lower()          # (lower case on purpose: get_alleles upper-cases the flanks)
    d = dict(chrom="chr7", region_start=s0 + 5 - int(rng.integers(0, 3)), region_stop=blocks[1]["end"] - 5 + int(rng.integers(0, 3)),
             name="TR%d" % tr if rng.random() < 0.7 else "", motif="ACG"[:min(period, 3)], period_str=str(period),
             chrom_seq=chrom, chrom_seq_start=s0 - 150, blocks=blocks, block=1,
             inexact_allele=rng.integers(0, 2, size=len(blocks[1]["alleles"])).astype(np.uint8),
             log_aln_probs=post["clamped_ll"], log_p1=p1, log_p2=p2, sample_label=lab, alns=alns,
             aln_deleted=(rng.random(R) < 0.1).astype(np.uint8) if deleted_allele else None,
             log_sample_posteriors=post["post"], sample_total_ll=post["sample_total_ll"], best_haplotypes=post["gts"],
             n_p1s=rng.integers(0, 9, size=S), n_p2s=rng.integers(0, 9, size=S),
             sample_names=["S%d" % s for s in range(S)], haploid=haploid)
    return d, Hn


def test_vcf_record_product_equals_restatement_on_random_loci():
    rng = np.random.default_rng(71)
    n = 0
    for trial in range(60):
        haploid = trial % 5 == 4
        d, H = _locus(rng, int(rng.integers(8, 90)), int(rng.integers(1, 7)), int(rng.integers(1, 6)), int(rng.integers(6, 25)),
                      int(rng.integers(1, 4)), haploid=haploid, deleted_allele=(trial % 7 == 3), phased=(trial % 3 != 0))
        S = len(d["sample_names"])
        if trial % 4 == 1:
            d["sample_filter"] = ["" if s else "LOW_QUAL" for s in range(S)]
        if trial % 6 == 2:
            d["out_sample_names"] = ["S0", "ABSENT"] + ["S%d" % s for s in range(1, S)]
        pv = _abi.PackedVcfLocus(d)
        for opt in (None, _abi.vcf_options(output_gls=1, output_pls=1, output_phased_gls=1, output_filters=1, output_haplotype_data=1),
                    _abi.vcf_options(output_allreads=0, output_mallreads=0)):
            got, pos = _lib.vcf_record(pv, opt)
            want, wpos = ol.oracle_vcf_record(pv, opt)
            assert got == want and pos == wpos, (trial, got, want)
            n += 1
        assert _lib.get_alleles(pv) == ol.oracle_get_alleles(pv)
        cols = got.split("\t")
        assert cols[0] == "chr7" and int(cols[1]) == pos and len(cols) == 9 + (len(d.get("out_sample_names") or d["sample_names"]))
    assert n == 180


def test_vcf_record_hand_checked():
    """One diploid sample, two alleles (CAG x 5 / CAG x 6), reads split 3 / 3, no phasing: every field of the
    record worked out by hand from the reference's statements."""
    lf, rf = b"ACGTTGCAAGCTTAGGCTAACGTTAGCCATGGATC", b"GGATCCTTAGCAATCGGATTACAGGCTTAACCGTA"
    pl, pr = b"TTGAC", b"CAGTT"
    a0, a1 = pl + b"CAG" * 5 + pr, pl + b"CAG" * 6 + pr
    s0 = 1000
    blocks = [dict(start=s0, end=s0 + 35, is_repeat=False, period=0, alleles=[lf]),
              dict(start=s0 + 35, end=s0 + 35 + len(a0), is_repeat=True, period=3, alleles=[a0, a1]),
              dict(start=s0 + 35 + len(a0), end=s0 + 70 + len(a0), is_repeat=False, period=0, alleles=[rf])]
    chrom_seq = b"N" * 50 + lf + a0 + rf + b"N" * 50
    ll = np.array([[-0.01, -12.0]] * 3 + [[-12.0, -0.01]] * 3)
    lab = np.zeros(6, dtype=np.int32)
    post = ol.oracle_posteriors(ll, np.zeros(6), np.zeros(6), lab, 1)
    assert post["gts"].tolist() == [[0, 1]]
    reads = [dict(start=s0, stop=s0 + 69 + len(a0), seq=lf + a0 + rf, cigar=[("=", 70 + len(a0))])] * 3
    ins = [dict(start=s0, stop=s0 + 69 + len(a0), seq=lf + a1 + rf, cigar=[("=", 55), ("I", 3), ("=", 15 + len(a0))])] * 3
    d = dict(chrom="chr1", region_start=s0 + 40, region_stop=s0 + 40 + 15, name="CAGtest", motif="CAG", period_str="3",
             chrom_seq=chrom_seq, chrom_seq_start=s0 - 50, blocks=blocks, block=1, log_aln_probs=ll, log_p1=np.zeros(6),
             log_p2=np.zeros(6), sample_label=lab, alns=reads + ins, log_sample_posteriors=post["post"],
             sample_total_ll=post["sample_total_ll"], best_haplotypes=post["gts"], n_p1s=[0], n_p2s=[0], sample_names=["NA1"])
    pv = _abi.PackedVcfLocus(d)
    pos, alleles = _lib.get_alleles(pv)
    # block = pad + repeat + pad; the pads are trimmed back to the region boundary, no padding base needed:
    # both alleles start with 'C'.  POS is 1-based region start.
    assert alleles == ["CAG" * 5, "CAG" * 6] and pos == s0 + 41
    line, _ = _lib.vcf_record(pv)
    cols = line.split("\t")
    assert cols[:7] == ["chr1", str(s0 + 41), "CAGtest", "CAG" * 5, "CAG" * 6, ".", "."]
    info = dict(kv.split("=") for kv in cols[7].split(";"))
    assert info == {"START": str(s0 + 41), "END": str(s0 + 55), "MOTIF": "CAG", "PERIOD": "3", "NSKIP": "0", "NFILT": "0",
                    "INEXACT_ALLELE": "0", "BPDIFFS": "3", "DP": "6", "DSNP": "0", "DFLANKINDEL": "0", "AN": "2", "REFAC": "1", "AC": "1"}
    assert cols[8] == "GT:GB:Q:PQ:DP:DSNP:DFLANKINDEL:PDP:PSNP:GLDIFF:ALLREADS:MALLREADS"
    f = cols[9].split(":")
    # GT 0|1, GB 0|3, posterior of the unphased genotype ~1 (two phasings of 0.5 each), DP 6, no SNP reads,
    # ALLREADS from the CIGARs over region +- 5 (three reads 0 bp, three +3 bp), MALLREADS from the ML haplotype
    assert f[0] == "0|1" and f[1] == "0|3" and f[2] == "1.00" and f[3] == "0.50" and f[4:9] == ["6", "0", "0", "0|0", "0|0"]
    assert f[10] == "0|3;3|3" and f[11] == "0|3;3|3" and float(f[9]) > 10.0
    assert line == ol.oracle_vcf_record(pv)[0]


def test_unused_alleles_and_haplotype_remap():
    rng = np.random.default_rng(72)
    for trial in range(40):
        L = synth.synth_locus(rng, int(rng.integers(8, 60)), 3, int(rng.integers(2, 7)), 3, raw=False)
        blocks = L.blocks()
        if trial % 3 == 0:                                       # a second variable block: haplotypes follow the Gray walk
            blocks[2]["alleles"] = [blocks[2]["alleles"][0], blocks[2]["alleles"][0][:-1] + b"A"]
        H = int(np.prod([len(b["alleles"]) for b in blocks]))
        h2a = _lib.haps_to_alleles(blocks, 1)
        assert np.array_equal(h2a, ol.oracle_haps_to_alleles(blocks, 1)) and len(h2a) == H
        S = int(rng.integers(1, 5))
        best = rng.integers(0, H, size=(S, 2))
        aligned = rng.integers(0, 2, size=S).astype(np.uint8)
        filt = rng.integers(0, 2, size=S).astype(np.uint8)
        nopt = len(blocks[1]["alleles"])
        un = _lib.unused_alleles(best, h2a, nopt, aligned, filt)
        assert un == ol.oracle_unused_alleles(best, h2a, nopt, aligned, filt)
        called = {int(h2a[h]) for s in range(S) if aligned[s] and not filt[s] for h in best[s]}
        assert un == [a for a in range(1, nopt) if a not in called]
        # remove them (HapBlock::remove_alleles) and, sometimes, add a new candidate: old -> new haplotype mapping
        new_blocks = [dict(b, alleles=list(b["alleles"])) for b in blocks]
        new_blocks[1]["alleles"] = [a for i, a in enumerate(blocks[1]["alleles"]) if i not in un]
        if trial % 2:
            new_blocks[1]["alleles"].append(blocks[1]["alleles"][0] + b"CAGCAG")
        m, realign = _lib.remap_haplotypes(blocks, new_blocks)
        mo, ro = ol.oracle_remap_haplotypes(blocks, new_blocks)
        assert np.array_equal(m, mo) and np.array_equal(realign, ro)
        Hn = len(realign)
        old_seqs, new_seqs = _lib.haplotype_seqs(blocks), _lib.haplotype_seqs(new_blocks)
        for j in range(H):
            if m[j] == -1:                                       # gone -- or a LATER old haplotype has the same sequence (std::map assignment, :328)
                assert old_seqs[j] not in new_seqs or old_seqs[j] in old_seqs[j + 1:]
            else:
                assert new_seqs[m[j]] == old_seqs[j]
        assert [int(x) for x in realign] == [int(s not in old_seqs) for s in new_seqs]
        old_ll = rng.normal(size=(4, H))
        new_ll = _lib.remap_aln_probs(old_ll, m, Hn)
        for j in range(H):
            if m[j] >= 0 and list(m).count(m[j]) == 1:
                assert np.array_equal(new_ll[:, m[j]], old_ll[:, j])
        assert (new_ll[:, realign.astype(bool)] == -100000.0).all()


def test_vcf_header_hand_checked_and_equal_to_the_restatement():
    """ltr_vcf_header = Genotyper::get_vcf_header (genotyper.cpp:258-336): product (table-driven) against the C
    restatement (a chain of prints like the reference) for every combination of the optional FORMAT switches, and the
    default header against a hand-written copy of what the reference prints (its texts, typo included)."""
    import ctypes as C
    import itertools
    O = ol.oracle()
    O.ltr_oracle_vcf_header.restype = C.c_int64
    O.ltr_oracle_vcf_header.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.c_char_p, C.c_int64]
    contigs = "##contig=<ID=chr1,length=248956422>\n##contig=<ID=chrX,length=156040895>\n"
    names = ["HG002", "HG003", "HG004"]

    def restated(opt):
        arr = (C.c_char_p * 3)(*[n.encode() for n in names])
        buf = C.create_string_buffer(1 << 16)
        n = O.ltr_oracle_vcf_header(b"hg38.fa", b"LongTR --bams a.bam", contigs.encode(), C.byref(opt), arr, 3, buf, 1 << 16)
        assert n > 0
        return buf.raw[:n].decode()

    fields = ["output_gls", "output_pls", "output_phased_gls", "output_allreads", "output_mallreads", "output_filters", "output_haplotype_data"]
    for bits in itertools.product((0, 1), repeat=len(fields)):
        opt = _abi.vcf_options()
        for f, b in zip(fields, bits):
            setattr(opt, f, b)
        got = _lib.vcf_header("hg38.fa", "LongTR --bams a.bam", contigs, names, opt)
        assert got == restated(opt)
        assert got.count("##FORMAT=<ID=") == 9 + sum(bits) + bits[6]              # HQ and PHQ come together
    hdr = _lib.vcf_header("hg38.fa", "LongTR --bams a.bam", contigs, names, _abi.vcf_options())    # defaults: ALLREADS and MALLREADS on (genotyper.cpp:342-343)
    lines = hdr.split("\n")
    assert lines[:3] == ["##fileformat=VCFv4.1", "##command=LongTR --bams a.bam", "##reference=hg38.fa"]
    assert lines[3:5] == contigs.strip().split("\n")
    assert lines[5] == '##INFO=<ID=START,Number=1,Type=Integer,Description="Inclusive start coodinate for the repetitive portion of the reference allele">'
    assert [l.split(",")[0] for l in lines[5:19]] == ["##INFO=<ID=" + k for k in
        ("START", "END", "MOTIF", "PERIOD", "NSKIP", "NFILT", "INEXACT_ALLELE", "BPDIFFS", "DP", "DSNP", "DFLANKINDEL", "AN", "REFAC", "AC")]
    assert [l.split(",")[0] for l in lines[19:28]] == ["##FORMAT=<ID=" + k for k in ("GT", "GB", "Q", "PQ", "DP", "DSNP", "PSNP", "PDP", "GLDIFF")]
    assert lines[27] == '##FORMAT=<ID=GLDIFF,Number=1,Type=Float,Description="Difference in likelihood between the reported and next best genotypes">'
    assert lines[28] == '##FORMAT=<ID=ALLREADS,Number=1,Type=String,Description="Base pair difference observed in each read\'s Needleman-Wunsch alignment">'
    assert lines[29].startswith("##FORMAT=<ID=MALLREADS,Number=1,Type=String,Description=\"Maximum likelihood bp diff in each read")
    assert lines[30] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tHG002\tHG003\tHG004" and lines[31] == "" and len(lines) == 32
    # every FORMAT key a record carries is defined in the header of the same options
    rng = np.random.default_rng(5)
    d, _ = _locus(rng, 30, 3, 3, 12, 2)
    for opt in (_abi.vcf_options(), _abi.vcf_options(output_gls=1, output_pls=1, output_phased_gls=1, output_filters=1, output_haplotype_data=1)):
        line = _lib.vcf_record(_abi.PackedVcfLocus(d), opt)[0]
        keys = line.split("\t")[8].split(":")
        hdr2 = _lib.vcf_header("x.fa", "cmd", None, d["sample_names"], opt)
        # (DFLANKINDEL: the reference writes the per-sample field, seq_stutter_genotyper.cpp:1217, but its FORMAT definition is
        # commented out in get_vcf_header, genotyper.cpp:303 -- only the INFO one exists; reproduced as it is)
        assert all(f"##FORMAT=<ID={k}," in hdr2 for k in keys if k != "DFLANKINDEL"), keys
        info_keys = [kv.split("=")[0] for kv in line.split("\t")[7].split(";")]
        assert all(f"##INFO=<ID={k}," in hdr2 for k in info_keys), info_keys
