"""TEST INFRASTRUCTURE (see oracle/ltr_oracle.h): pure-Python restatement of the raw-read preparation and the
exact-allele candidate generation, statement by statement, for small cases:

  BamAlignment::TrimAlignment               src/bam_io.cpp:267-372
  GenotyperBamProcessor::left_align_reads   src/genotyper_bam_processor.cpp:38-168
  HaplotypeGenerator::extract_sequence      src/SeqAlignment/HaplotypeGenerator.cpp:84-165
  HaplotypeGenerator::gen_candidate_seqs    :295-373, :474-480 (without the POA clustering branch :376-472)
  HaplotypeGenerator::trim                  :14-82
  HaplotypeGenerator::add_haplotype_block   :530-578
  HaplotypeGenerator::fuse_haplotype_blocks :580-607

PARITY UNPINNED: bam_io.cpp / genotyper_bam_processor.cpp need htslib, HaplotypeGenerator.cpp needs spoa; none can be
compiled in the dev container.  Only tests import this module."""

FLANK_SIZE = 200            # bam_io.h:28
INT_MAX = 2 ** 31 - 1


def trim_alignment(pos, end_pos, bases, quals, cigar, min_read_start, max_read_stop):
    """bam_io.cpp:267-372.  cigar: list of [type, length] (consumed in place).  Returns (pos, end_pos, bases, quals, deleted)."""
    ltrim = 0
    start_pos = pos
    while start_pos < min_read_start and len(cigar) > 0:             # :274-299
        t = cigar[0][0]
        if t in "M=X":
            ltrim += 1
            start_pos += 1
        elif t == "D":
            start_pos += 1
        elif t in "IS":
            ltrim += 1
        elif t == "H":
            pass
        else:
            raise ValueError("Invalid CIGAR option encountered in TrimAlignment")
        if cigar[0][1] == 1:
            cigar.pop(0)
        else:
            cigar[0][1] -= 1
    repeat_pointer = start_pos                                       # :302-305
    repeat_start = min_read_start + FLANK_SIZE
    repeat_end = max_read_stop - FLANK_SIZE
    deletion_size = 0
    tmp = [list(c) for c in cigar]
    while repeat_pointer >= min_read_start and repeat_pointer < repeat_end and len(tmp) > 0:     # :311-336
        t = tmp[0][0]
        if t in "M=X":
            repeat_pointer += 1
        elif t == "D":
            if repeat_pointer >= repeat_start:
                deletion_size += 1
            repeat_pointer += 1
        elif t in "ISH":
            pass
        else:
            raise ValueError("Invalid CIGAR option encountered in TrimAlignment")
        if tmp[0][1] == 1:
            tmp.pop(0)
        else:
            tmp[0][1] -= 1
    deleted = deletion_size >= (repeat_end - repeat_start)           # :337-339
    rtrim = 0
    e = end_pos
    while e > max_read_stop and len(cigar) > 0:                      # :344-366
        t = cigar[-1][0]
        if t in "M=X":
            rtrim += 1
            e -= 1
        elif t == "D":
            e -= 1
        elif t in "IS":
            rtrim += 1
        elif t == "H":
            pass
        else:
            raise ValueError("Invalid CIGAR option encountered in trimAlignment")
        if cigar[-1][1] == 1:
            cigar.pop()
        else:
            cigar[-1][1] -= 1
    assert ltrim + rtrim <= len(bases)
    bases = bases[ltrim:len(bases) - rtrim]                          # :369-370
    if quals is not None:
        quals = quals[ltrim:len(quals) - rtrim]
    return start_pos, e, bases, quals, deleted


def left_align_reads(raw, n_samples, region_start, region_stop, chrom_seq, chrom_seq_start):
    """genotyper_bam_processor.cpp:38-168.  raw: list of dict(pos, end_pos, bases(bytes), quals, cigar, sample, hp).
    Returns (left_alns, n_p1s, n_p2s, align_fail_count); a left_aln is a dict(start, stop, seq, cigar, aln, deleted, source, sample)."""
    def chrom_at(p):
        k = p - chrom_seq_start
        return chr(chrom_seq[k]).upper() if 0 <= k < len(chrom_seq) else "\0"
    left, fail = [], 0
    n_p1s, n_p2s = [0] * n_samples, [0] * n_samples
    for idx, r in enumerate(raw):
        if r["pos"] > region_start or r["end_pos"] < region_stop:    # :56-59
            fail += 1
            continue
        cigar = [[t, k] for t, k in r["cigar"]]
        pos, end_pos, bases, quals, deleted = trim_alignment(r["pos"], r["end_pos"], r["bases"].decode("latin-1"),
                                                             None if r.get("quals") is None else r["quals"].decode("latin-1"), cigar,
                                                             region_start - FLANK_SIZE if region_start > FLANK_SIZE else 1, region_stop + FLANK_SIZE)   # :61
        if len(bases) == 0:                                          # :62-71
            left.append(dict(start=region_start, stop=region_stop, seq="", cigar=[], aln="", deleted=True, source=idx, sample=r.get("sample", 0)))
            continue
        read_sequence = bases.upper()                                # :76
        new = dict(start=pos, stop=end_pos - 1, seq=read_sequence, cigar=[], aln="", deleted=deleted, source=idx, sample=r.get("sample", 0), qual=quals)
        seq_index, ref_index = 0, pos
        soft_clipped = False
        aln = []
        for t, length in cigar:                                      # :80-136
            cigar_index, prev_type, prev_num = 0, "=", 0
            if t == "H":
                pass
            elif t == "S":
                new["cigar"].append(("S", length))
                seq_index += length
                soft_clipped = True
            elif t == "I":
                new["cigar"].append(("I", length))
                aln.append(read_sequence[seq_index:seq_index + length])
                seq_index += length
            elif t == "D":
                new["cigar"].append(("D", length))
                aln.append("-" * length)
                ref_index += length
            elif t in "M=X":
                while cigar_index < length:
                    b = read_sequence[seq_index] if seq_index < len(read_sequence) else "\0"
                    if b == chrom_at(ref_index):
                        if prev_type == "=":
                            prev_num += 1
                        else:
                            if prev_num != 0:
                                new["cigar"].append((prev_type, prev_num))
                            prev_type, prev_num = "=", 1
                    else:
                        if prev_type == "X":
                            prev_num += 1
                        else:
                            if prev_num != 0:
                                new["cigar"].append((prev_type, prev_num))
                            prev_type, prev_num = "X", 1
                    aln.append(b)
                    cigar_index += 1
                    ref_index += 1
                    seq_index += 1
                if prev_num != 0:
                    new["cigar"].append((prev_type, prev_num))
            else:
                raise ValueError("Invalid CIGAR option encountered in convertAlignment")
        new["aln"] = "".join(aln)
        if soft_clipped:                                             # :137-140
            fail += 1
            continue
        left.append(new)
        if r.get("hp", 0) == 1:                                      # :145-150
            n_p1s[r.get("sample", 0)] += 1
        if r.get("hp", 0) == 2:
            n_p2s[r.get("sample", 0)] += 1
    return left, n_p1s, n_p2s, fail


def extract_sequence(aln, region_start, region_end):
    """HaplotypeGenerator.cpp:84-165.  Returns the sequence or None (read does not span)."""
    if aln["deleted"]:
        return ""
    if aln["start"] >= region_start:
        return None
    if aln["stop"] <= region_end:
        return None
    align_index = char_index = 0
    pos = aln["start"]
    it = 0
    cig = aln["cigar"]
    reg = []
    while it < len(cig):
        t, num = cig[it]
        if char_index == num:
            it += 1
            char_index = 0
        elif pos > region_end:
            return "".join(reg).upper()
        elif pos == region_end:
            if t == "I":
                reg.append(aln["aln"][align_index:align_index + num])
                align_index += num
                char_index = 0
                it += 1
            else:
                return "".join(reg).upper()
        elif pos >= region_start:
            num_bases = min(region_end - pos, num - char_index)
            if t == "I":
                num_bases = num
                reg.append(aln["aln"][align_index:align_index + num_bases])
            elif t in "=XM":
                reg.append(aln["aln"][align_index:align_index + num_bases])
                pos += num_bases
            elif t == "D":
                pos += num_bases
            else:
                raise ValueError("Invalid CIGAR char in extractRegionSequences()")
            align_index += num_bases
            char_index += num_bases
        else:
            if t == "I":
                num_bases = num - char_index
            else:
                num_bases = min(region_start - pos, num - char_index)
                pos += num_bases
            align_index += num_bases
            char_index += num_bases
    raise ValueError("Logical error in extract_sequence")


def _order(s):
    return (len(s), s)                                               # orderByLengthAndSequence, stringops.cpp:35-39


def trim(ideal_min_length, left_pad, right_pad, region_start, region_end, sequences):
    """HaplotypeGenerator.cpp:14-82.  Returns (region_start, region_end, sequences)."""
    min_len = min(len(s) for s in sequences)
    if min_len <= ideal_min_length:
        return region_start, region_end, sequences
    max_left = max_right = 0
    while max_left < min_len - ideal_min_length:
        j = 1
        while j < len(sequences):
            if sequences[j][max_left] != sequences[j - 1][max_left]:
                break
            j += 1
        if j != len(sequences):
            break
        max_left += 1
    while max_right < min_len - ideal_min_length:
        c = sequences[0][len(sequences[0]) - 1 - max_right]
        j = 1
        while j < len(sequences):
            if sequences[j][len(sequences[j]) - 1 - max_right] != c:
                break
            j += 1
        if j != len(sequences):
            break
        max_right += 1
    max_left = min(left_pad, max_left)                               # :47-48
    max_right = min(right_pad, max_right)
    max_left = max(0, min(min_len - right_pad, max_left))            # :51-52
    max_right = max(0, min(min_len - left_pad, max_right))
    if min_len - 2 * min(max_left, max_right) <= ideal_min_length:   # :57-65
        left_trim = right_trim = min(max_left, max_right)
        while min_len - left_trim - right_trim < ideal_min_length:
            if left_trim > right_trim:
                left_trim -= 1
            else:
                right_trim -= 1
    else:
        if max_left > max_right:
            right_trim = max_right
            left_trim = min(max_left, min_len - ideal_min_length - max_right)
        else:
            left_trim = max_left
            right_trim = min(max_right, min_len - ideal_min_length - max_left)
    sequences = [s[left_trim:len(s) - right_trim] for s in sequences]
    return region_start + left_trim, region_end - right_trim, sequences


def build_haplotype(left_alns, n_samples, region_start, region_stop, period, chrom_seq, chrom_seq_start, chrom_len, indel_flank_len=5):
    """build_haplotype (seq_stutter_genotyper.cpp:416-482) -> add_haplotype_block + fuse_haplotype_blocks, one region,
    no VCF alleles, no POA branch.  Returns dict(blocks or None, failure, unplaced_reads, samples_needing_clustering)."""
    MIN_FRAC_READS, MIN_FRAC_SAMPLES, MIN_FRAC_STRONG_SAMPLE, MIN_READS_STRONG_SAMPLE, MIN_STRONG_SAMPLES = 0.05, 0.05, 0.2, 2, 1
    LEFT_PAD = RIGHT_PAD = indel_flank_len
    REF_FLANK_LEN = 35

    def sub(p, n):
        k = p - chrom_seq_start
        return chrom_seq[max(k, 0):max(k + n, 0)].decode("latin-1").upper() if n > 0 else ""
    out = dict(blocks=None, failure="", unplaced_reads=0, samples_needing_clustering=0)
    min_aln_start = min([a["start"] for a in left_alns], default=INT_MAX)     # :421-426
    max_aln_stop = max([a["stop"] for a in left_alns], default=-INT_MAX - 1)
    if region_start < REF_FLANK_LEN + LEFT_PAD or region_stop + REF_FLANK_LEN + RIGHT_PAD > chrom_len:    # :536-539
        out["failure"] = "Haplotype blocks are too near to the chromosome ends"
        return out
    gen = [a for a in left_alns if a.get("use_for_hap_generation", True)]
    gmin = min([a["start"] for a in gen], default=INT_MAX)
    gmax = max([a["stop"] for a in gen], default=-INT_MAX - 1)
    rs, re = region_start - LEFT_PAD, region_stop + RIGHT_PAD        # :546-547
    ref_seq = sub(rs, re - rs)
    if gmin + 5 >= rs or gmax - 5 <= re:                             # :549-552
        out["failure"] = "No spanning alignments"
        return out
    ideal_min_length = 3 * period                                    # :566
    # gen_candidate_seqs :295-373
    sample_counts, read_counts, must_inc = {}, {}, {}
    tot_reads = tot_samples = 0
    per_sample = [[] for _ in range(n_samples)]
    for a in gen:
        s = extract_sequence(a, rs, re)
        if s is not None:
            per_sample[a["sample"]].append(s)
    for i in range(n_samples):
        counts = {}
        samp_reads = len(per_sample[i])
        for s in per_sample[i]:
            read_counts[s] = read_counts.get(s, 0) + 1
            counts[s] = counts.get(s, 0) + 1
            tot_reads += 1
        for s in sorted(counts):                                     # std::map order
            if counts[s] >= MIN_READS_STRONG_SAMPLE and counts[s] >= MIN_FRAC_STRONG_SAMPLE * samp_reads:
                must_inc[s] = must_inc.get(s, 0) + 1
            sample_counts[s] = sample_counts.get(s, 0.0) + counts[s] * 1.0 / samp_reads
        if samp_reads > 0:
            tot_samples += 1
    sequences = []
    ref_index = -1
    for s in sorted(must_inc):                                       # :345-356
        if must_inc[s] >= MIN_STRONG_SAMPLES:
            del sample_counts[s]
            del read_counts[s]
            sequences.append(s)
            if s == ref_seq:
                ref_index = len(sequences) - 1
    for s in sorted(sample_counts):                                  # :359-365
        if sample_counts[s] > MIN_FRAC_SAMPLES * tot_samples * 2 or read_counts.get(s, 0) > MIN_FRAC_READS * tot_reads * 2:
            sequences.append(s)
            if ref_index == -1 and s == ref_seq:
                ref_index = len(sequences) - 1
    if ref_index == -1:                                              # :368-373
        sequences.insert(0, ref_seq)
    else:
        sequences[ref_index] = sequences[0]
        sequences[0] = ref_seq
    for i in range(n_samples):                                       # :376-398 (counted only)
        ignored = sum(1 for s in per_sample[i] if s not in sequences)
        out["unplaced_reads"] += ignored
        if ignored > len(per_sample[i]) * 0.25:
            out["samples_needing_clustering"] += 1
    sequences = [sequences[0]] + sorted(sequences[1:], key=_order)   # :475
    rs, re, sequences = trim(ideal_min_length, LEFT_PAD, RIGHT_PAD, rs, re, sequences)   # :480
    # fuse_haplotype_blocks :580-607
    if rs < REF_FLANK_LEN or re + REF_FLANK_LEN > chrom_len:
        out["failure"] = "Haplotype blocks are too near to the chromosome ends"
        return out
    min_start = min(rs - 10, max(rs - REF_FLANK_LEN, min_aln_start))                      # :590-591
    max_stop = max(re + 10, min(re + REF_FLANK_LEN, max_aln_stop))
    out["blocks"] = [dict(start=min_start, end=rs, is_repeat=False, period=0, alleles=[sub(min_start, rs - min_start).encode("latin-1")]),
                     dict(start=rs, end=re, is_repeat=True, period=period, alleles=[s.encode("latin-1") for s in sequences]),
                     dict(start=re, end=max_stop, is_repeat=False, period=0, alleles=[sub(re, max_stop - re).encode("latin-1")])]
    return out


def phasing_priors(sample_of_read, haplotype, n_samples):
    """SNPBamProcessor::process_phased_reads (snp_bam_processor.cpp:141-226) for unpaired reads, restated: read groups in
    order, running totals (:158-183), the verdict of :187-193 (0 / 0 is NaN there: it compares false), the priors of :216-227.
    haplotype: 1 / 2 = HP tag, -1 = none (get_haplotype, :126-134).  Returns (log_p1, log_p2, phased reads).  UNPINNED (htslib)."""
    FROM_HAP_LL, OTHER_HAP_LL = -0.000001, -1000.0
    n = len(sample_of_read)
    p1, p2 = [0.0] * n, [0.0] * n
    phased = total = h1 = h2 = 0
    not_enough = False
    for s in range(n_samples):
        idx = [r for r in range(n) if sample_of_read[r] == s]
        for r in idx:
            total += 1
            if haplotype[r] == 1:
                h1 += 1
            elif haplotype[r] == 2:
                h2 += 1
        unphased_frac = (total - (h1 + h2)) / total if total else float("nan")
        if unphased_frac > 0.2 or h2 <= 1 or h1 <= 1:
            not_enough = True
        for r in idx:
            if haplotype[r] != -1 and not not_enough:
                phased += 1
                p1[r] = FROM_HAP_LL if haplotype[r] == 1 else OTHER_HAP_LL
                p2[r] = FROM_HAP_LL if haplotype[r] == 2 else OTHER_HAP_LL
    return p1, p2, phased
