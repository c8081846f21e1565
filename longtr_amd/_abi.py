"""ctypes mirror of include/ltr_gpu.h (structs + flattening helpers).

Pure data-layout code: no compute lives here.  The product binding
(longtr_amd/_lib.py) uses these structs; the test-side checker reuses them so
that the same flattened inputs go to both sides.
"""
import ctypes as C

import numpy as np

LTR_OK = 0
LTR_ERR_INVALID = -1
LTR_ERR_NO_DEVICE = -2
LTR_ERR_HIP = -3
LTR_ERR_NOMEM = -4
LTR_ERR_CIGAR = -5
LTR_ERR_UNSUPPORTED = -6

LTR_IMPOSSIBLE = -1000000000.0
LTR_ABORT_SCORE = -700.0


class AlignParams(C.Structure):
    """struct ltr_align_params (reference: AlignmentModel, HapAligner.h:12-37)."""

    _fields_ = [
        ("log_ins_to_ins", C.c_float),
        ("log_ins_to_match", C.c_float),
        ("log_del_to_del", C.c_float),
        ("log_del_to_match", C.c_float),
        ("log_match_to_match", C.c_float),
        ("log_match_to_ins", C.c_float),
        ("log_match_to_del", C.c_float),
        ("indel_flank_len", C.c_int32),
        ("use_short_path", C.c_int32),
    ]

    def as_tuple(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


def default_params():
    """Defaults of HapAligner.h:118 (double literals narrowed to float) + INDEL_FLANK_LEN 5."""
    p = AlignParams()
    vals = (-1.0, -0.458675, -1.0, -0.458675, -0.00005800168, -10.448214728, -10.448214728)
    for (name, _), v in zip(AlignParams._fields_[:7], vals):
        setattr(p, name, float(np.float32(v)))
    p.indel_flank_len = 5
    p.use_short_path = 0
    return p


def make_params(values7, indel_flank_len=5, use_short_path=0):
    """--alignment-params a,b,c,d,e,f,g (strings parsed like std::stof -> float32)."""
    p = AlignParams()
    for (name, _), v in zip(AlignParams._fields_[:7], values7):
        setattr(p, name, float(np.float32(v)))
    p.indel_flank_len = indel_flank_len
    p.use_short_path = use_short_path
    return p


class LocusBatch(C.Structure):
    """struct ltr_locus_batch."""

    _fields_ = [
        ("n_loci", C.c_int64),
        ("locus_read_off", C.POINTER(C.c_int64)),
        ("locus_hap_off", C.POINTER(C.c_int64)),
        ("n_reads", C.c_int64),
        ("read_bytes", C.POINTER(C.c_uint8)),
        ("read_off", C.POINTER(C.c_int64)),
        ("n_haps", C.c_int64),
        ("hap_bytes", C.POINTER(C.c_uint8)),
        ("hap_off", C.POINTER(C.c_int64)),
        ("realign_read", C.POINTER(C.c_uint8)),
        ("realign_hap", C.POINTER(C.c_uint8)),
    ]


class GenotypeFields(C.Structure):
    """struct ltr_genotype_fields (every pointer optional)."""

    _fields_ = [
        ("best_gts", C.c_void_p),
        ("log_phased_posteriors", C.c_void_p),
        ("log_unphased_posteriors", C.c_void_p),
        ("hap_log_phased_posteriors", C.c_void_p),
        ("hap_log_unphased_posteriors", C.c_void_p),
        ("gls", C.c_void_p),
        ("gl_diffs", C.c_void_p),
        ("pls", C.c_void_p),
        ("phased_gls", C.c_void_p),
    ]


def genotype_field_buffers(n_samples, n_variants, haploid, want=("gls", "gl_diffs", "pls", "phased_gls")):
    """Allocate numpy outputs for ltr_extract_genotypes; returns (GenotypeFields, dict of arrays)."""
    import numpy as np
    S, V = n_samples, n_variants
    n_gl = V if haploid else V * (V + 1) // 2
    n_pgl = V if haploid else V * V
    arrs = {
        "best_gts": np.full((S, 2), -1, dtype=np.int32),
        "log_phased_posteriors": np.full(S, np.nan), "log_unphased_posteriors": np.full(S, np.nan),
        "hap_log_phased_posteriors": np.full(S, np.nan), "hap_log_unphased_posteriors": np.full(S, np.nan),
    }
    if "gls" in want:
        arrs["gls"] = np.full((S, n_gl), np.nan)
    if "gl_diffs" in want:
        arrs["gl_diffs"] = np.full(S, np.nan)
    if "pls" in want:
        arrs["pls"] = np.full((S, n_gl), -1, dtype=np.int32)
    if "phased_gls" in want:
        arrs["phased_gls"] = np.full((S, n_pgl), np.nan)
    f = GenotypeFields()
    for k, a in arrs.items():
        setattr(f, k, a.ctypes.data_as(C.c_void_p))
    return f, arrs


class PosteriorBatch(C.Structure):
    """struct ltr_posterior_batch."""

    _fields_ = [
        ("n_loci", C.c_int64),
        ("locus_read_off", C.POINTER(C.c_int64)),
        ("n_reads", C.c_int64),
        ("pool_index", C.POINTER(C.c_int32)),
        ("log_p1", C.POINTER(C.c_double)),
        ("log_p2", C.POINTER(C.c_double)),
        ("sample_label", C.POINTER(C.c_int32)),
        ("n_samples", C.POINTER(C.c_int32)),
        ("haploid", C.c_int32),
    ]


class HaplotypeBlocks(C.Structure):
    """struct ltr_haplotype_blocks."""

    _fields_ = [
        ("n_blocks", C.c_int32),
        ("block_start", C.POINTER(C.c_int32)),
        ("block_end", C.POINTER(C.c_int32)),
        ("is_repeat", C.POINTER(C.c_uint8)),
        ("period", C.POINTER(C.c_int32)),
        ("n_alleles", C.POINTER(C.c_int32)),
        ("allele_bytes", C.POINTER(C.c_uint8)),
        ("allele_off", C.POINTER(C.c_int64)),
    ]


class Alignment(C.Structure):
    """struct ltr_alignment (reference: class Alignment, AlignmentData.h:28-140)."""

    _fields_ = [
        ("start", C.c_int32),
        ("stop", C.c_int32),
        ("seq", C.POINTER(C.c_uint8)),
        ("seq_len", C.c_int32),
        ("n_cigar", C.c_int32),
        ("cigar_type", C.c_char_p),
        ("cigar_num", C.POINTER(C.c_int32)),
        ("qual", C.POINTER(C.c_uint8)),
    ]


class StutterParams(C.Structure):
    """struct ltr_stutter_params (reference: StutterModel, stutter_model.h:35-62)."""

    _fields_ = [("in_geom", C.c_double), ("in_up", C.c_double), ("in_down", C.c_double),
                ("out_geom", C.c_double), ("out_up", C.c_double), ("out_down", C.c_double)]


def default_stutter_params():
    """hipstr_main.cpp:140,362-363: the CLI always installs this fixed model."""
    return StutterParams(0.95, 0.05, 0.05, 0.95, 0.01, 0.01)


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def _concat(seqs):
    """list of bytes -> (uint8 array, int64 offsets[n+1])."""
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    if seqs:
        off[1:] = np.cumsum([len(s) for s in seqs])
    buf = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if off[-1] > 0 else np.zeros(1, dtype=np.uint8)
    return buf, off


class PackedBatch:
    """Flattened loci: keeps the numpy buffers alive next to the ctypes struct."""

    def __init__(self, loci, realign_read=None, realign_hap=None):
        """loci: list of (reads: list[bytes], haps: list[bytes])."""
        reads, haps = [], []
        lro, lho = [0], [0]
        for rs, hs in loci:
            reads.extend(rs)
            haps.extend(hs)
            lro.append(len(reads))
            lho.append(len(haps))
        self.locus_read_off = np.asarray(lro, dtype=np.int64)
        self.locus_hap_off = np.asarray(lho, dtype=np.int64)
        self.read_bytes, self.read_off = _concat(reads)
        self.hap_bytes, self.hap_off = _concat(haps)
        self.realign_read = None if realign_read is None else np.ascontiguousarray(realign_read, dtype=np.uint8)
        self.realign_hap = None if realign_hap is None else np.ascontiguousarray(realign_hap, dtype=np.uint8)
        self.n_loci = len(loci)
        self.n_reads = len(reads)
        self.n_haps = len(haps)
        P = np.diff(self.locus_read_off)
        H = np.diff(self.locus_hap_off)
        self.ll_off = np.zeros(self.n_loci + 1, dtype=np.int64)
        self.ll_off[1:] = np.cumsum(P * H)
        self.ll_size = int(self.ll_off[-1])
        b = LocusBatch()
        b.n_loci = self.n_loci
        b.locus_read_off = _ptr(self.locus_read_off, C.c_int64)
        b.locus_hap_off = _ptr(self.locus_hap_off, C.c_int64)
        b.n_reads = self.n_reads
        b.read_bytes = _ptr(self.read_bytes, C.c_uint8)
        b.read_off = _ptr(self.read_off, C.c_int64)
        b.n_haps = self.n_haps
        b.hap_bytes = _ptr(self.hap_bytes, C.c_uint8)
        b.hap_off = _ptr(self.hap_off, C.c_int64)
        b.realign_read = _ptr(self.realign_read, C.c_uint8) if self.realign_read is not None else None
        b.realign_hap = _ptr(self.realign_hap, C.c_uint8) if self.realign_hap is not None else None
        self.struct = b

    def locus_matrix(self, ll, l):
        """View of locus l's [P x H] matrix inside a flat LL buffer."""
        P = int(self.locus_read_off[l + 1] - self.locus_read_off[l])
        H = int(self.locus_hap_off[l + 1] - self.locus_hap_off[l])
        return ll[self.ll_off[l]:self.ll_off[l + 1]].reshape(P, H)


class PackedHaplotype:
    """blocks: list of dict(start, end, is_repeat, period, alleles=[bytes, ...])."""

    def __init__(self, blocks):
        self.blocks = blocks
        nb = len(blocks)
        self.block_start = np.asarray([b["start"] for b in blocks], dtype=np.int32)
        self.block_end = np.asarray([b["end"] for b in blocks], dtype=np.int32)
        self.is_repeat = np.asarray([1 if b.get("is_repeat") else 0 for b in blocks], dtype=np.uint8)
        self.period = np.asarray([b.get("period", 0) for b in blocks], dtype=np.int32)
        self.n_alleles = np.asarray([len(b["alleles"]) for b in blocks], dtype=np.int32)
        seqs = [a for b in blocks for a in b["alleles"]]
        self.allele_bytes, self.allele_off = _concat(seqs)
        h = HaplotypeBlocks()
        h.n_blocks = nb
        h.block_start = _ptr(self.block_start, C.c_int32)
        h.block_end = _ptr(self.block_end, C.c_int32)
        h.is_repeat = _ptr(self.is_repeat, C.c_uint8)
        h.period = _ptr(self.period, C.c_int32)
        h.n_alleles = _ptr(self.n_alleles, C.c_int32)
        h.allele_bytes = _ptr(self.allele_bytes, C.c_uint8)
        h.allele_off = _ptr(self.allele_off, C.c_int64)
        self.struct = h

    @property
    def num_combs(self):
        return int(np.prod(self.n_alleles.astype(np.int64)))


class PackedAlignments:
    """alns: list of dict(start, stop, seq=bytes, cigar=[(type_char, num), ...])."""

    def __init__(self, alns):
        self.alns = alns
        n = len(alns)
        self._keep = []
        arr = (Alignment * max(n, 1))()
        for i, a in enumerate(alns):
            seq = np.frombuffer(a["seq"], dtype=np.uint8).copy() if len(a["seq"]) else np.zeros(1, dtype=np.uint8)
            ctype = bytes(ord(t) if isinstance(t, str) else t for t, _ in a["cigar"])
            cnum = np.asarray([k for _, k in a["cigar"]], dtype=np.int32) if a["cigar"] else np.zeros(1, dtype=np.int32)
            qual = None
            if a.get("qual") is not None:
                qual = np.frombuffer(a["qual"], dtype=np.uint8).copy() if len(a["qual"]) else np.zeros(1, dtype=np.uint8)
            self._keep.append((seq, ctype, cnum, qual))
            arr[i].start = a["start"]
            arr[i].stop = a["stop"]
            arr[i].seq = _ptr(seq, C.c_uint8)
            arr[i].seq_len = len(a["seq"])
            arr[i].n_cigar = len(a["cigar"])
            arr[i].cigar_type = ctype
            arr[i].cigar_num = _ptr(cnum, C.c_int32)
            arr[i].qual = _ptr(qual, C.c_uint8) if qual is not None else None
        self.array = arr
        self.n = n


class Locus(C.Structure):
    """struct ltr_locus."""

    _fields_ = [("hap", C.POINTER(HaplotypeBlocks)), ("alns", C.POINTER(Alignment)), ("n_alns", C.c_int32),
                ("second_mate", C.POINTER(C.c_uint8)), ("realign_to_hap", C.POINTER(C.c_uint8)),
                ("realign_pool", C.POINTER(C.c_uint8)), ("copy_read", C.POINTER(C.c_uint8))]


class Timers(C.Structure):
    """struct ltr_timers."""

    _fields_ = [("hap_build_s", C.c_double), ("hap_aln_s", C.c_double), ("posterior_s", C.c_double), ("dp_kernel_ms", C.c_double),
                ("hap_build_calls", C.c_int64), ("hap_aln_calls", C.c_int64), ("posterior_calls", C.c_int64),
                ("nw_kernel_ms", C.c_double), ("short_kernel_ms", C.c_double), ("dp_cells", C.c_double), ("dp_pairs", C.c_int64)]


class VcfOptions(C.Structure):
    """struct ltr_vcf_options (Genotyper's output switches, genotyper.cpp:339-346)."""

    _fields_ = [("output_gls", C.c_int32), ("output_pls", C.c_int32), ("output_phased_gls", C.c_int32), ("output_allreads", C.c_int32),
                ("output_mallreads", C.c_int32), ("output_filters", C.c_int32), ("output_haplotype_data", C.c_int32),
                ("max_flank_indel_frac", C.c_float)]


def vcf_options(**kw):
    o = VcfOptions(0, 0, 0, 1, 1, 0, 0, 0.15)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class VcfLocus(C.Structure):
    """struct ltr_vcf_locus."""

    _fields_ = [("chrom", C.c_char_p), ("region_start", C.c_int32), ("region_stop", C.c_int32), ("name", C.c_char_p),
                ("motif", C.c_char_p), ("period_str", C.c_char_p), ("chrom_seq", C.POINTER(C.c_uint8)),
                ("chrom_seq_start", C.c_int64), ("chrom_seq_len", C.c_int64), ("hap", C.POINTER(HaplotypeBlocks)),
                ("block", C.c_int32), ("inexact_allele", C.POINTER(C.c_uint8)), ("n_reads", C.c_int32), ("n_samples", C.c_int32),
                ("haploid", C.c_int32), ("log_aln_probs", C.POINTER(C.c_double)), ("log_p1", C.POINTER(C.c_double)),
                ("log_p2", C.POINTER(C.c_double)), ("sample_label", C.POINTER(C.c_int32)), ("alns", C.POINTER(Alignment)),
                ("aln_deleted", C.POINTER(C.c_uint8)), ("log_sample_posteriors", C.POINTER(C.c_double)),
                ("sample_total_ll", C.POINTER(C.c_double)), ("best_haplotypes", C.POINTER(C.c_int32)),
                ("n_p1s", C.POINTER(C.c_int32)), ("n_p2s", C.POINTER(C.c_int32)), ("sample_names", C.POINTER(C.c_char_p)),
                ("sample_filter", C.POINTER(C.c_char_p)), ("n_out_samples", C.c_int32), ("out_sample_names", C.POINTER(C.c_char_p))]


class PackedVcfLocus:
    """Keeps every buffer of a ltr_vcf_locus alive.  d: dict with the struct's fields as python / numpy values
    (blocks = haplotype block dicts, alns = alignment dicts, strings as str)."""

    def __init__(self, d):
        self.d = d
        self.ph = PackedHaplotype(d["blocks"])
        self.pa = PackedAlignments(d["alns"]) if d.get("alns") is not None else None
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
        self.keep = dict(ll=f64(d["log_aln_probs"]), p1=f64(d["log_p1"]), p2=f64(d["log_p2"]), lab=i32(d["sample_label"]),
                         post=f64(d["log_sample_posteriors"]), stl=f64(d["sample_total_ll"]), best=i32(d["best_haplotypes"]),
                         chrom_seq=np.frombuffer(d["chrom_seq"], dtype=np.uint8).copy())
        S = len(d["sample_names"])
        names = (C.c_char_p * S)(*[n.encode() for n in d["sample_names"]])
        v = VcfLocus()
        v.chrom = d["chrom"].encode()
        v.region_start, v.region_stop = d["region_start"], d["region_stop"]
        v.name = d.get("name", "").encode()
        v.motif = d.get("motif", "").encode()
        v.period_str = d.get("period_str", "").encode()
        v.chrom_seq = _ptr(self.keep["chrom_seq"], C.c_uint8)
        v.chrom_seq_start, v.chrom_seq_len = d.get("chrom_seq_start", 0), len(d["chrom_seq"])
        v.hap = C.pointer(self.ph.struct)
        v.block = d["block"]
        if d.get("inexact_allele") is not None:
            self.keep["inexact"] = u8(d["inexact_allele"])
            v.inexact_allele = _ptr(self.keep["inexact"], C.c_uint8)
        v.n_reads, v.n_samples, v.haploid = len(self.keep["p1"]), S, int(bool(d.get("haploid", False)))
        v.log_aln_probs = _ptr(self.keep["ll"], C.c_double)
        v.log_p1, v.log_p2 = _ptr(self.keep["p1"], C.c_double), _ptr(self.keep["p2"], C.c_double)
        v.sample_label = _ptr(self.keep["lab"], C.c_int32)
        if self.pa is not None:
            v.alns = self.pa.array
        if d.get("aln_deleted") is not None:
            self.keep["del"] = u8(d["aln_deleted"])
            v.aln_deleted = _ptr(self.keep["del"], C.c_uint8)
        v.log_sample_posteriors = _ptr(self.keep["post"], C.c_double)
        v.sample_total_ll = _ptr(self.keep["stl"], C.c_double)
        v.best_haplotypes = _ptr(self.keep["best"], C.c_int32)
        for k in ("n_p1s", "n_p2s"):
            if d.get(k) is not None:
                self.keep[k] = i32(d[k])
                setattr(v, k, _ptr(self.keep[k], C.c_int32))
        self.keep["names"] = names
        v.sample_names = names
        if d.get("sample_filter") is not None:
            self.keep["filt"] = (C.c_char_p * S)(*[(x or "").encode() for x in d["sample_filter"]])
            v.sample_filter = self.keep["filt"]
        if d.get("out_sample_names"):
            n = len(d["out_sample_names"])
            self.keep["out"] = (C.c_char_p * n)(*[x.encode() for x in d["out_sample_names"]])
            v.n_out_samples, v.out_sample_names = n, self.keep["out"]
        self.struct = v


class RawAlignment(C.Structure):
    """struct ltr_raw_alignment (BamAlignment fields the path reads)."""

    _fields_ = [("pos", C.c_int32), ("end_pos", C.c_int32), ("bases", C.POINTER(C.c_uint8)), ("quals", C.POINTER(C.c_uint8)),
                ("length", C.c_int32), ("n_cigar", C.c_int32), ("cigar_type", C.c_char_p), ("cigar_num", C.POINTER(C.c_int32)),
                ("sample", C.c_int32), ("haplotype_tag", C.c_int32), ("reverse", C.c_uint8), ("use_for_hap_generation", C.c_uint8)]
