"""Seeded synthetic tandem-repeat loci: the ONE generator shared by
the GPU runs, the CPU baseline and the parity tests.

A locus mirrors what SeqStutterGenotyper hands to HapAligner::process_reads
(reference: src/seq_stutter_genotyper.cpp:416-482 builds three blocks
[35-bp left flank][repeat block with N alleles][35-bp right flank]; the repeat
block carries 5-bp pads either side of the repeat proper):

  haplotype k   = lflank + pad_l + allele_k + pad_r + rflank        (what Haplotype::get_seq() returns)
  raw read      = ext_l + (haplotype of its allele, with sequencing errors) + ext_r,
                  with an exact =/X/I/D CIGAR against the reference allele
  trimmed read  = what HapAligner::trim_alignment leaves: repeat block +- INDEL_FLANK_LEN bp

No reference code is involved; numpy's PCG64 is the RNG; every locus has a
generator of its own keyed by (seed, configuration, locus index), so CPU and GPU runs, every rank of a sharded
run and every worker process of a parallel generation see byte-identical loci.
"""
from dataclasses import dataclass, field

import numpy as np

from . import _abi

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
REF_FLANK_LEN = 35          # HaplotypeGenerator.h REF_FLANK_LEN, HapAligner.cpp:245
PAD_LEN = 5
EXT_LEN = 165               # reads are cut to region +-200 bp at load (bam_io.h:28): 165 + 35 = 200


def _rand_seq(rng, n):
    return BASES[rng.integers(0, 4, size=n)]


@dataclass
class Locus:
    start: int                      # reference coordinate of the left flank block
    period: int
    lflank: bytes
    rflank: bytes
    alleles: list                   # repeat-block alleles (pads included), allele 0 = reference
    read_allele: list               # true allele index of every read
    trimmed_reads: list             # bytes per read, what the DP sees
    raw_alns: list = field(default_factory=list)   # dict(start, stop, seq, cigar) per read (optional)
    ext_l: bytes = b""
    ext_r: bytes = b""

    @property
    def haplotypes(self):
        return [self.lflank + a + self.rflank for a in self.alleles]

    def blocks(self):
        """ltr_haplotype_blocks-style description."""
        s0 = self.start
        s1 = s0 + len(self.lflank)
        e1 = s1 + len(self.alleles[0])
        return [
            dict(start=s0, end=s1, is_repeat=False, period=0, alleles=[self.lflank]),
            dict(start=s1, end=e1, is_repeat=True, period=self.period, alleles=list(self.alleles)),
            dict(start=e1, end=e1 + len(self.rflank), is_repeat=False, period=0, alleles=[self.rflank]),
        ]


def _mutate(rng, seq, sub_rate, indel_rate):
    """Apply substitutions / 1-bp indels to a uint8 array.

    Returns (new_seq, ops) where ops is a per-reference-base list encoded as an
    array: 0 '=', 1 'X', 2 'D' (base deleted); ins_after[i] = bases inserted
    after reference base i (array of arrays is avoided: insertions are 1 bp).
    """
    n = len(seq)
    out = seq.copy()
    code = np.zeros(n, dtype=np.uint8)
    if sub_rate > 0:
        m = rng.random(n) < sub_rate
        k = int(m.sum())
        if k:
            out[m] = BASES[(np.searchsorted(BASES, out[m]) + rng.integers(1, 4, size=k)) % 4]
            code[m] = 1
    ins = np.zeros(n, dtype=bool)
    if indel_rate > 0:
        r = rng.random(n)
        dele = r < indel_rate / 2
        ins = (r >= indel_rate / 2) & (r < indel_rate)
        code[dele] = 2
        ins &= ~dele
    return out, code, ins


def _build_read(rng, pieces, sub_rate, indel_rate, start):
    """pieces: list of (kind, ref_seq, read_seq) with kind in {'ref','ins','del'}.

    Walks the reference left to right producing the read bytes and an exact
    =/X/I/D CIGAR (adjacent equal ops merged)."""
    read = []
    cig = []

    def push(t, k=1):
        if k <= 0:
            return
        if cig and cig[-1][0] == t:
            cig[-1][1] += k
        else:
            cig.append([t, k])

    ref_len = 0
    for kind, seq in pieces:
        if kind == "ins":
            read.append(seq)
            push("I", len(seq))
        elif kind == "del":
            push("D", len(seq))
            ref_len += len(seq)
        else:
            out, code, ins = _mutate(rng, seq, sub_rate, indel_rate)
            ref_len += len(seq)
            if not code.any() and not ins.any():
                read.append(out)
                push("=", len(seq))
                continue
            # per reference base: '=' / 'X' / 'D', then an 'I' after the bases that carry an insertion --
            # built as one op array and run-length encoded (no per-base Python loop)
            n = len(seq)
            n_ins = int(ins.sum())
            ins_bases = _rand_seq(rng, n_ins) if n_ins else np.zeros(0, dtype=np.uint8)
            pos = np.arange(n) + np.concatenate([[0], np.cumsum(ins)[:-1]])      # slot of base i in the op array
            ops = np.empty(n + n_ins, dtype=np.uint8)
            ops[pos] = np.where(code == 2, ord("D"), np.where(code == 1, ord("X"), ord("=")))
            ipos = pos[ins] + 1
            ops[ipos] = ord("I")
            rd = np.empty(n + n_ins, dtype=np.uint8)
            rd[pos] = out
            rd[ipos] = ins_bases
            read.append(rd[ops != ord("D")])
            cut = np.flatnonzero(np.diff(ops)) + 1
            starts = np.concatenate([[0], cut])
            lens = np.diff(np.concatenate([starts, [len(ops)]]))
            for st, ln in zip(starts, lens):
                push(chr(ops[st]), int(ln))
    seq = np.concatenate(read) if read else np.zeros(0, dtype=np.uint8)
    return dict(start=start, stop=start + ref_len - 1, seq=seq.tobytes(), cigar=[(t, k) for t, k in cig])


def trim_like_reference(aln, repeat_start, repeat_end, padding):
    """Pure-python HapAligner::trim_alignment (HapAligner.cpp:346-465) used ONLY by the
    generator to derive trimmed reads from raw reads; the parity tests check it
    against the oracle and the reference build."""
    cig = [[t, k] for t, k in aln["cigar"]]
    lo, hi = repeat_start - padding, repeat_end + padding
    sp, ep = aln["start"] + 1, aln["stop"] + 1
    lt = rt = 0

    def pop(front):
        e = cig[0] if front else cig[-1]
        if e[1] == 1:
            cig.pop(0 if front else -1)
        else:
            e[1] -= 1

    while sp <= lo and cig:
        t = cig[0][0]
        if t in "M=X":
            lt += 1
            sp += 1
        elif t == "D":
            sp += 1
        elif t in "IS":
            lt += 1
        pop(True)
    mid = sp
    while lo < mid <= lo + padding and cig:
        t = cig[0][0]
        if t in "M=X":
            mid += 1
        elif t == "D":
            lt -= 1
            mid += 1
        pop(True)
    while ep > hi and cig:
        t = cig[-1][0]
        if t in "M=X":
            rt += 1
            ep -= 1
        elif t == "D":
            ep -= 1
        elif t in "IS":
            rt += 1
        pop(False)
    mid = ep
    while hi - padding < mid <= hi and cig:
        t = cig[-1][0]
        if t in "M=X":
            mid -= 1
        elif t == "D":
            rt -= 1
            mid -= 1
        pop(False)
    lt, rt = max(lt, 0), max(rt, 0)
    n = len(aln["seq"])
    return aln["seq"][lt:n - rt], lt, rt


def synth_locus(rng, tr_len, period, n_alleles, n_reads, sub_rate=0.0015, indel_rate=0.0005,
                raw=False, start=100000, allele_step=None, true_alleles=None, indel_flank_len=5):
    """One locus.  Alleles = reference repeat +- k*period (k = 1,1,2,2,...: alternating sign)."""
    motif = _rand_seq(rng, period)
    rep0 = np.tile(motif, tr_len // period + 2)[:tr_len]
    lflank, rflank = _rand_seq(rng, REF_FLANK_LEN), _rand_seq(rng, REF_FLANK_LEN)
    pad_l, pad_r = _rand_seq(rng, PAD_LEN), _rand_seq(rng, PAD_LEN)
    step = allele_step or period
    reps = [rep0]
    kpos = kneg = 0
    for h in range(1, n_alleles):
        if h % 2 == 0 and len(rep0) - step * (kneg + 1) >= max(period, 1):
            kneg += 1
            reps.append(rep0[:len(rep0) - step * kneg])
        else:
            kpos += 1
            d = step * kpos
            reps.append(np.concatenate([rep0, np.tile(motif, d // period + 2)[:d]]))
    alleles = [np.concatenate([pad_l, r, pad_r]) for r in reps]
    if true_alleles is None:
        true_alleles = rng.choice(n_alleles, size=min(2, n_alleles), replace=False)
    ext_l = _rand_seq(rng, EXT_LEN) if raw else np.zeros(0, dtype=np.uint8)
    ext_r = _rand_seq(rng, EXT_LEN) if raw else np.zeros(0, dtype=np.uint8)

    s1 = start + REF_FLANK_LEN                      # repeat block start (pads are part of the block)
    e1 = s1 + len(alleles[0])
    read_allele, trimmed, raw_alns = [], [], []
    for _ in range(n_reads):
        # 90 % of reads come from the (up to) two true alleles, the rest from any candidate
        k = int(rng.choice(true_alleles)) if rng.random() < 0.9 else int(rng.integers(0, n_alleles))
        read_allele.append(k)
        if raw:
            d = len(reps[k]) - len(rep0)
            common = min(len(reps[k]), len(rep0))
            pieces = [("ref", np.concatenate([ext_l, lflank, pad_l, rep0[:common]]))]
            if d > 0:
                pieces.append(("ins", reps[k][common:]))
            elif d < 0:
                pieces.append(("del", rep0[common:]))
            pieces.append(("ref", np.concatenate([pad_r, rflank, ext_r])))
            aln = _build_read(rng, pieces, sub_rate, indel_rate, start - EXT_LEN)
            raw_alns.append(aln)
            t, _, _ = trim_like_reference(aln, s1, e1, indel_flank_len)
            if len(t) == 0:                          # HapAligner.cpp:820-823
                t = lflank.tobytes()[-5:] + rflank.tobytes()[:5]
            trimmed.append(t)
        else:
            core = np.concatenate([lflank[-indel_flank_len:] if indel_flank_len else lflank[:0], alleles[k],
                                   rflank[:indel_flank_len]])
            out, code, ins = _mutate(rng, core, sub_rate, indel_rate)
            if code.max(initial=0) == 2 or ins.any():
                parts = []
                for i in range(len(core)):
                    if code[i] != 2:
                        parts.append(out[i:i + 1])
                    if ins[i]:
                        parts.append(_rand_seq(rng, 1))
                out = np.concatenate(parts) if parts else out[:1]
            trimmed.append(out.tobytes())
    return Locus(start=start, period=period, lflank=lflank.tobytes(), rflank=rflank.tobytes(),
                 alleles=[a.tobytes() for a in alleles], read_allele=read_allele,
                 trimmed_reads=trimmed, raw_alns=raw_alns, ext_l=ext_l.tobytes(), ext_r=ext_r.tobytes())


def pool_reads(reads):
    """ReadPooler semantics (read_pooler.cpp:3-20): exact-sequence dedupe, first occurrence order."""
    seen, pools, idx = {}, [], []
    for r in reads:
        k = seen.get(r)
        if k is None:
            k = seen[r] = len(pools)
            pools.append(r)
        idx.append(k)
    return pools, idx


def _period_for(rng, tr_len):
    return int(rng.integers(1, 7)) if tr_len < 100 else int(rng.integers(10, 61))


CONFIG_SEED = 20250225
CONFIGS = ["config2", "config3", "config5", "config5hifi", "config3skew", "catalogue"]


def _catalogue_tr(rng):
    """Repeat lengths like the reference's bundled catalogue (test_data/test_regions_hg38.bed: 40 loci, 38 of them
    11-77 bp with a median of 17, one of 380 bp, one of 2.9 kb): 96 % short (10 + an exponential tail, < 100 bp),
    4 % log-uniform in 100-1000 bp (the reference's --max-tr-len default)."""
    if rng.random() < 0.96:
        return min(10 + int(rng.exponential(12.0)), 99)
    return int(np.exp(rng.uniform(np.log(100.0), np.log(1000.0))))


def _locus_header(name, rng):
    """First draws of a locus' own generator: (tr_len, period, n_alleles, n_reads, sub_rate, indel_rate)."""
    if name == "config2":       # 1 locus, 64 HiFi-like reads x 8 haplotypes, 200-bp STR
        return 200, int(rng.integers(3, 7)), 8, 64, 0.0015, 0.0005
    if name in ("config3", "config3skew", "catalogue"):
        if name == "config3":   # 30x, TR 20..1000, H 2..12
            tr = int(rng.integers(20, 1001))
        elif name == "config3skew":   # 80 % of the loci are short TRs (< 100 bp)
            tr = int(rng.integers(20, 100)) if rng.random() < 0.8 else int(rng.integers(100, 1001))
        else:
            tr = _catalogue_tr(rng)
        period = _period_for(rng, tr)
        period = max(1, min(period, tr // 2))
        # candidate alleles: short repeats have few (the reference keeps alleles with >= 2 supporting reads)
        h = int(rng.integers(2, 13)) if name != "catalogue" else int(rng.integers(2, 4 + min(tr // 8, 9)))
        return tr, period, h, 30, 0.0015, 0.0005
    if name == "config5":       # ONT stress: 5-kb VNTR, 3-5 % error, f=g=-4.6
        return 5000, int(rng.integers(30, 61)), 4, 8, 0.025, 0.015
    if name == "config5hifi":   # the same 5-kb VNTR geometry with HiFi-like reads: pairs finish instead of aborting
        return 5000, int(rng.integers(30, 61)), 4, 8, 0.0015, 0.0005
    raise ValueError(name)


_DEFAULT_N = {"config2": 1, "config3": 10000, "config3skew": 10000, "config5": 8, "config5hifi": 64, "catalogue": 100000}
_DESC = {
    "config2": "config2: 1 locus, 64 reads x 8 haplotypes, TR 200 bp",
    "config3": "config3: {n} loci, 30x coverage, TR 20-1000 bp, H 2-12, default alignment params",
    "config3skew": "config3skew: {n} loci, 30x coverage, 80 % TR 20-99 bp / 20 % TR 100-1000 bp, H 2-12, default alignment params",
    "config5": "config5: {n} loci, TR 5 kb VNTR, ONT-like 4 % error, f=g=-4.6",
    "config5hifi": "config5hifi: {n} loci, TR 5 kb VNTR, HiFi-like 0.2 % error, 8 reads x 4 haplotypes, f=g=-4.6",
    "catalogue": "catalogue: {n} loci, 30x coverage, TR lengths like test_regions_hg38.bed (96 % 10-99 bp, median 18; 4 % 100-1000 bp), H 2-12, default alignment params",
}


def _locus_rng(name, seed, i):
    """Every locus of a configuration has a generator of its own, keyed by (seed, configuration, locus index):
    any subset of the loci can be generated without the others (a rank generates its shard only) and in any
    number of processes, byte-identically."""
    return np.random.default_rng([int(seed), CONFIGS.index(name), int(i)])


def config_headers(name, seed=CONFIG_SEED, n_loci=None):
    """(tr_len, n_alleles, n_reads) arrays of every locus of a configuration WITHOUT generating it."""
    n = _DEFAULT_N[name] if n_loci is None else n_loci
    out = np.zeros((n, 3), dtype=np.int64)
    for i in range(n):
        h = _locus_header(name, _locus_rng(name, seed, i))
        out[i] = (h[0], h[2], h[3])
    return out


def header_costs(headers, indel_flank_len=5):
    """Modelled DP cost per locus from its header alone (reads x alleles x (TR + pads + flanks)^2): what a rank needs
    to shard the catalogue before any locus exists."""
    side = headers[:, 0].astype(np.float64) + 2 * PAD_LEN + 2 * indel_flank_len
    return headers[:, 2] * headers[:, 1] * side * side


def _gen_loci(job):
    name, seed, ids, raw = job
    out = []
    for i in ids:
        rng = _locus_rng(name, seed, i)
        tr, period, h, r, sub, indel = _locus_header(name, rng)
        out.append(synth_locus(rng, tr, period, h, r, sub_rate=sub, indel_rate=indel, raw=raw))
    return out


def config_loci(name, seed=CONFIG_SEED, n_loci=None, raw=False, ids=None, workers=None):
    """BASELINE.json configs + the catalogue-shaped workload.  Returns (list[Locus], description).
    ids: generate these loci of the configuration only.  workers: processes to generate with (default: by size;
    worker processes are fresh interpreters that only import numpy: safe after the GPU was initialised)."""
    n = _DEFAULT_N[name] if n_loci is None else n_loci
    ids = list(range(n)) if ids is None else [int(i) for i in ids]
    desc = _DESC[name].format(n=n)
    per_locus_ms = (9.0 if raw else 1.0) * (3.0 if name in ("config3", "config5", "config5hifi") else 1.0)
    if workers is None:
        import os
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        workers = min(ncpu, 32) if len(ids) * per_locus_ms > 8000.0 else 1
    if workers <= 1 or len(ids) < 64:
        return _gen_loci((name, seed, ids, raw)), desc
    # worker processes: fresh interpreters running `python -m longtr_amd._synth_worker` (numpy only, no GPU, nothing of the
    # caller's main module), jobs and results pickled over their pipes
    import pickle
    import subprocess
    import sys
    import threading
    workers = min(workers, (len(ids) + 31) // 32)
    cuts = [len(ids) * k // workers for k in range(workers + 1)]
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-m", "longtr_amd._synth_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, cwd=root)
             for _ in range(workers)]
    parts = [None] * workers

    def talk(k):
        out, _ = procs[k].communicate(pickle.dumps((name, seed, ids[cuts[k]:cuts[k + 1]], raw)))
        parts[k] = pickle.loads(out) if procs[k].returncode == 0 else None

    th = [threading.Thread(target=talk, args=(k,)) for k in range(workers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if any(p is None for p in parts):
        raise RuntimeError("a generator worker process failed")
    return [L for part in parts for L in part], desc


ONT_PARAMS = (-1.0, -0.458675, -1.0, -0.458675, -0.00005800168, -4.6, -4.6)


def pack_loci(loci, pooled=True):
    """Flatten loci into a PackedBatch of (pooled, trimmed reads) x (haplotype strings).

    Returns (PackedBatch, pool_index list per locus)."""
    flat, pidx = [], []
    for L in loci:
        if pooled:
            pools, idx = pool_reads(L.trimmed_reads)
        else:
            pools, idx = list(L.trimmed_reads), list(range(len(L.trimmed_reads)))
        flat.append((pools, L.haplotypes))
        pidx.append(idx)
    return _abi.PackedBatch(flat), pidx


def nominal_cells(batch, indel_flank_len=5):
    """BASELINE.md cell count: n*m per pair, 0 for pairs that take a shortcut."""
    total = 0
    cut = 2 * (REF_FLANK_LEN - indel_flank_len)
    rl = np.diff(batch.read_off)
    hl = np.diff(batch.hap_off)
    for l in range(batch.n_loci):
        m = rl[batch.locus_read_off[l]:batch.locus_read_off[l + 1]].astype(np.int64)
        hf = hl[batch.locus_hap_off[l]:batch.locus_hap_off[l + 1]].astype(np.int64)
        n = np.where(hf - cut >= 0, hf - cut, hf - (REF_FLANK_LEN - indel_flank_len))
        ok = (hf > 60)[None, :] & (np.abs(n[None, :] - m[:, None]) <= 600)
        total += int((m[:, None] * n[None, :] * ok).sum())
    return total



def homopolymer_locus(rng, tr_len, n_alleles, n_reads, sub_rate=0.01, indel_rate=0.02):
    """Period-1 locus for the seeded stutter path (HapAligner.cpp:545-581, --stutter-align-len): raw reads
    (exact =/X/I/D CIGARs, +-200 bp of flank) with random Phred+33 base qualities.  Returns (blocks, alignments)."""
    L = synth_locus(rng, tr_len, 1, n_alleles, n_reads, sub_rate=sub_rate, indel_rate=indel_rate, raw=True)
    alns = []
    for a in L.raw_alns:
        q = rng.integers(ord("!") + 2, ord("J") + 1, size=len(a["seq"])).astype(np.uint8)
        if rng.random() < 0.2:
            q[rng.integers(0, len(q))] = ord("~")          # above 'J': clamped (base_quality.h:49-51)
        alns.append(dict(a, qual=q.tobytes()))
    return L.blocks(), alns
