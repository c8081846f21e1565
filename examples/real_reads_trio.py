"""The whole chain on REAL reads: the reference's bundled HiFi BAMs of the HG002 / HG003 / HG004 trio
(test_data/*.bam) at the loci of its bundled BED, through this library only:

  BED text           -> ltr_read_regions            (the bundled file puts the motif in column 7: converted first)
  BAM + BAI          -> ltr_bam_*                   (no htslib)
  reference sequence -> rebuilt from the reads      (hg38 is not bundled: HiFi CIGARs are =/X/I/D, so every
                                                     base some read matches ('=') is known; see rebuild_reference)
  left_align_reads   -> ltr_left_align_reads
  candidate alleles  -> ltr_build_haplotype         (exact alleles, no POA)
  read x haplotype   -> ltr_calc_hap_aln_probs      (GPU; every locus in one call)
  phasing priors     -> ltr_phasing_priors          (the HP tags, process_phased_reads' rule: --phased-bam)
  posteriors, GT     -> ltr_posteriors
  VCF                -> ltr_vcf_header, ltr_vcf_record, ltr_vcf_writer_*

    python examples/real_reads_trio.py [out.vcf.gz]

run(...) returns the per-locus results (used by tests/test_gpu_real_reads.py, which also bit-compares the
LL matrices with the CPU oracle and checks the trio for Mendelian consistency)."""
import math, os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longtr_amd import _abi, _lib  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "bam")
SAMPLES = ["HG002", "HG003", "HG004"]
PAD = 400                     # reference window either side of a region (left_align_reads cuts reads to region -+ 200)


def convert_bed(src, dst):
    """CHROM START STOP PERIOD COPIES NAME MOTIF (the bundled file) -> CHROM START STOP MOTIF NAME (readRegions)."""
    with open(dst, "w") as out:
        for line in open(src):
            c = line.split()
            if len(c) >= 7:
                out.write("\t".join([c[0], c[1], c[2], c[6].replace("/", ","), c[5]]) + "\n")      # (alternative motifs: "/" -> the reader's ",")


def rebuild_reference(reads, lo, hi):
    """Reference bases of [lo, hi): known wherever some read matches ('=' runs).  A position every read
    mismatches ('X': the trio is homozygous for a substitution against hg38) takes the reads' majority base --
    NOT the hg38 base, which nothing here can know; it only ever sits in a flank.  'N' where the reads say nothing
    (a base all of them delete)."""
    ref = bytearray(b"N" * (hi - lo))
    votes = {}
    for r in reads:
        rp, qp = r["pos"], 0
        for t, n in r["cigar"]:
            if t in "=X":
                a, b = max(rp, lo), min(rp + n, hi)
                if a < b and t == "=":
                    ref[a - lo:b - lo] = r["seq"][qp + (a - rp):qp + (b - rp)].encode()
                elif a < b:
                    for k in range(a, b):
                        votes.setdefault(k, []).append(r["seq"][qp + (k - rp)])
            if t in "M=X":
                rp += n; qp += n
            elif t in "DN":
                rp += n
            elif t in "IS":
                qp += n
    for k, v in votes.items():
        if ref[k - lo] == ord("N"):
            ref[k - lo] = ord(max(sorted(set(v)), key=v.count))
    return bytes(ref)


def phasing_priors(sample, hp):
    """--phased-bam: SNPBamProcessor::process_phased_reads (snp_bam_processor.cpp:141-226) = ltr_phasing_priors: a read with an
    HP tag gets FROM_HAP_LL / OTHER_HAP_LL unless too few reads are phased (running totals over the samples, sticky verdict)."""
    p1, p2, _ = _lib.phasing_priors(sample, [h if h in (1, 2) else -1 for h in hp], len(SAMPLES))
    return p1, p2


def run(ctx, vcf_path=None, max_loci=None, tmp_dir="/tmp"):
    bed = os.path.join(tmp_dir, f"ltr_regions_{os.getpid()}.bed")
    convert_bed(os.path.join(DATA, "test_regions_hg38.bed"), bed)
    regions, _ = _lib.read_regions(bed, order=True)
    os.remove(bed)
    if max_loci:
        regions = regions[:max_loci]
    bam = _lib.Bam([os.path.join(DATA, f"{s}_sample_reads.bam") for s in SAMPLES])
    sample_of_file = {rg["file"]: SAMPLES.index(rg["sample"]) for rg in bam.read_groups()}
    chrom_len = dict(bam.refs())
    loci = []
    for reg in regions:
        if reg["period"] < 1:
            continue
        recs = bam.fetch(reg["chrom"], reg["start"], reg["stop"], tags=("HP",))
        recs = [r for r in recs if r["mapq"] >= 20 and not (r["flag"] & 0x704)]            # mapped, primary, not QC-fail / duplicate
        lo, hi = max(reg["start"] - PAD, 0), min(reg["stop"] + PAD, chrom_len[reg["chrom"]])
        ref = rebuild_reference(recs, lo, hi)
        raw = [dict(pos=r["pos"], end_pos=r["end_pos"], bases=r["seq"].encode(), cigar=r["cigar"], sample=sample_of_file[r["file"]],
                    hp=int(r.get("HP", 0)), quals=r["qual"].encode("latin-1"), reverse=int(bool(r["flag"] & 16))) for r in recs]
        loc = dict(region=reg, n_raw=len(raw), ref_unknown=ref.count(b"N"), status="ok")
        loci.append(loc)
        if b"N" in ref[PAD - 250:len(ref) - PAD + 250]:
            loc["status"] = "reference not covered by matching reads"; continue
        rs = _lib.ReadSet(raw, len(SAMPLES), reg["start"], reg["stop"], ref, lo)
        hb = rs.build_haplotype(reg["start"], reg["stop"], reg["period"], lo, chrom_len[reg["chrom"]])
        reads = [r for r in rs.reads if not r["deleted"]]
        n_p1s, n_p2s = rs.n_p1s, rs.n_p2s
        rs.close()
        if hb["blocks"] is None or len(reads) < 5:
            loc["status"] = hb["failure"] or "too few reads"; continue
        loc.update(blocks=hb["blocks"], alns=[dict(start=r["start"], stop=r["stop"], seq=r["seq"], cigar=r["cigar"]) for r in reads],
                   sample=[r["sample"] for r in reads], hp=[raw[r["source"]]["hp"] for r in reads], ref=ref, ref_start=lo, n_p1s=n_p1s, n_p2s=n_p2s)
    bam.close()
    todo = [l for l in loci if l["status"] == "ok"]
    res = ctx.calc_hap_aln_probs([(l["blocks"], l["alns"], None) for l in todo])              # one GPU pass for every locus
    writer = _lib.VcfWriter(vcf_path) if vcf_path else None
    if writer:
        # Genotyper::get_vcf_header: field definitions a downstream tool can type the records with (no FASTA here: the
        # reference windows are rebuilt from the reads, so there are no ##contig lines)
        writer.header(_lib.vcf_header("(no hg38 FASTA bundled: windows rebuilt from the reads' = runs)", "examples/real_reads_trio.py", None, SAMPLES))
    for l, (ll, seeds) in zip(todo, res):
        R, H = ll.shape
        log_p1, log_p2 = phasing_priors(l["sample"], l["hp"])
        lab = np.asarray(l["sample"], dtype=np.int32)
        post = ctx.posteriors(ll, log_p1, log_p2, lab, len(SAMPLES))
        alleles = l["blocks"][1]["alleles"]
        l.update(ll=ll, seeds=seeds, gts=post["gts"], allele_lens=[len(a) for a in alleles], log_p1=log_p1,
                 gt_lens=[tuple(sorted(len(alleles[int(g)]) for g in gt)) for gt in post["gts"]])
        if writer:                                                  # SeqStutterGenotyper::write_vcf_record: GT:GB:Q:PQ:DP:...
            reg = l["region"]
            pv = _abi.PackedVcfLocus(dict(chrom=reg["chrom"], region_start=reg["start"], region_stop=reg["stop"], name=reg["name"], motif=reg["motif"],
                                          period_str=reg["period_str"], chrom_seq=l["ref"], chrom_seq_start=l["ref_start"], blocks=l["blocks"], block=1,
                                          inexact_allele=np.zeros(len(alleles), dtype=np.uint8), log_aln_probs=post["clamped_ll"], log_p1=log_p1, log_p2=log_p2,
                                          sample_label=lab, alns=l["alns"], log_sample_posteriors=post["post"], sample_total_ll=post["sample_total_ll"],
                                          best_haplotypes=post["gts"], n_p1s=l["n_p1s"], n_p2s=l["n_p2s"], sample_names=SAMPLES))
            line, pos = _lib.vcf_record(pv)
            l["vcf_line"] = line
            writer.add_record(reg["chrom"], pos, line)
    if writer:
        writer.close()
    return loci


def main():
    ctx = _lib.Context(0)
    out = sys.argv[1] if len(sys.argv) > 1 else None
    loci = run(ctx, out)
    for l in loci:
        if l["status"] != "ok":
            print(f"{l['region']['name']:>16} {l['region']['chrom']}:{l['region']['start']}-{l['region']['stop']}  skipped: {l['status']}")
            continue
        child, pa, ma = l["gt_lens"]
        ok = any((child[0] in p1 and child[1] in p2) for p1, p2 in ((pa, ma), (ma, pa)))
        print(f"{l['region']['name']:>16} {l['region']['chrom']}:{l['region']['start']}-{l['region']['stop']} motif {l['region']['motif']:<8} "
              f"reads {len(l['alns']):3d} alleles(bp) {l['allele_lens']}  HG002 {child} HG003 {pa} HG004 {ma}  {'mendelian' if ok else 'MENDELIAN VIOLATION'}")


if __name__ == "__main__":
    main()
