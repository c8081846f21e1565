"""ltr_bam (indexed BAM input without htslib) on the reference's own bundled BAM files
(tests/golden/bam/: copies of test_data/HG002_sample_reads.bam, HG004_sample_reads.bam and their .bai --
data files the reference ships for its tests).

Parity is UNPINNED: the reference reads BAM through htslib, which is not in this tree.  The checker here
is an independent decoder written in Python from the SAM specification (whole file through gzip, records
through struct), a linear scan instead of the index: the two implementations must agree on every field of
every record of every region asked for."""
import gzip, os, struct

import numpy as np
import pytest

from longtr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAMS = [os.path.join(ROOT, "tests", "golden", "bam", n) for n in ("HG002_sample_reads.bam", "HG004_sample_reads.bam")]


def decode_bam(path):
    """Header + every record, straight from the specification (section 4.2)."""
    raw = gzip.decompress(open(path, "rb").read())
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].rstrip(b"\x00").decode()
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, at)[0]; at += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, at)[0]
        name = raw[at + 4:at + 4 + l_name - 1].decode()
        refs.append((name, struct.unpack_from("<i", raw, at + 4 + l_name)[0]))
        at += 8 + l_name
    recs = []
    while at < len(raw):
        size = struct.unpack_from("<i", raw, at)[0]
        ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, mref, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", raw, at + 4)
        p = at + 36
        name = raw[p:p + l_name - 1].decode(); p += l_name
        cig = [("MIDNSHP=X"[c & 15], c >> 4) for c in struct.unpack_from(f"<{n_cig}I", raw, p)]; p += 4 * n_cig
        packed = raw[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(packed[i >> 1] >> (0 if i & 1 else 4)) & 15] for i in range(l_seq))
        qual = "".join(chr(q + 33) for q in raw[p:p + l_seq]); p += l_seq
        aux = raw[p:at + 4 + size]
        rlen = sum(n for o, n in cig if o in "MDN=X")
        if (flag & 4) or not cig or rlen == 0:
            rlen = 1
        recs.append(dict(name=name, ref_id=ref_id, pos=pos, end_pos=pos + rlen, mapq=mapq, flag=flag, mate_ref_id=mref, mate_pos=mpos, tlen=tlen,
                         seq=seq, qual=qual, cigar=cig, aux=aux))
        at += 4 + size
    return text, refs, recs


def aux_tags(aux):
    """tag -> python value for the scalar / string types."""
    out, p = {}, 0
    while p + 3 <= len(aux):
        tag, t = aux[p:p + 2].decode(), chr(aux[p + 2]); p += 3
        if t in "AcCsSiIf":
            fmt = {"A": "c", "c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[t]
            v = struct.unpack_from("<" + fmt, aux, p)[0]; p += struct.calcsize(fmt)
            out[tag] = v.decode() if t == "A" else v
        elif t in "ZH":
            e = aux.index(b"\x00", p); out[tag] = aux[p:e].decode(); p = e + 1
        elif t == "B":
            st, n = chr(aux[p]), struct.unpack_from("<I", aux, p + 1)[0]
            p += 5 + n * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[st]
        else:
            raise ValueError(t)
    return out


@pytest.fixture(scope="module")
def decoded():
    return [decode_bam(p) for p in BAMS]


def _same(got, want, tags):
    for k in ("name", "ref_id", "pos", "end_pos", "mapq", "flag", "mate_ref_id", "mate_pos", "tlen", "seq", "qual"):
        assert got[k] == want[k], (k, got["name"])
    assert got["cigar"] == want["cigar"]
    wt = aux_tags(want["aux"])
    for t in tags:
        assert got.get(t) == wt.get(t), (t, got["name"])


def test_header_and_whole_chromosomes(decoded):
    for path, (text, refs, recs) in zip(BAMS, decoded):
        b = _lib.Bam([path])
        assert b.refs() == refs
        rgs = [l for l in text.splitlines() if l.startswith("@RG")]
        assert len(b.read_groups()) == len(rgs)
        for rg, line in zip(b.read_groups(), rgs):
            f = dict(t.split(":", 1) for t in line.split("\t")[1:])
            assert (rg["id"], rg["sample"], rg["library"]) == (f.get("ID", ""), f.get("SM", ""), f.get("LB", ""))
        assert len(recs) > 100 and recs == sorted(recs, key=lambda r: (r["ref_id"] if r["ref_id"] >= 0 else 1 << 30, r["pos"]))
        tags = sorted({t for r in recs[:200] for t in aux_tags(r["aux"]) if isinstance(aux_tags(r["aux"])[t], (int, str))})[:6]
        n = 0
        for tid, (name, length) in enumerate(refs):
            want = [r for r in recs if r["ref_id"] == tid]
            if not want:
                continue
            got = b.fetch(name, 0, length, tags=tags)
            assert len(got) == len(want), name
            for g, w in zip(got, want):
                _same(g, w, tags)
            n += len(got)
        assert n == sum(r["ref_id"] >= 0 for r in recs)
        b.close()


def test_region_queries_equal_a_linear_scan(decoded):
    rng = np.random.default_rng(91)
    for path, (text, refs, recs) in zip(BAMS, decoded):
        b = _lib.Bam([path])
        mapped = [r for r in recs if r["ref_id"] >= 0]
        for _ in range(300):
            anchor = mapped[int(rng.integers(len(mapped)))]
            start = max(0, anchor["pos"] + int(rng.integers(-3000, 3000)))
            end = start + int(rng.choice([1, 50, 500, 5000, 20000, 300000]))
            chrom = refs[anchor["ref_id"]][0]
            want = [r for r in recs if r["ref_id"] == anchor["ref_id"] and r["pos"] < end and r["end_pos"] > start]
            got = b.fetch(chrom, start, end)
            assert [g["name"] for g in got] == [w["name"] for w in want], (chrom, start, end)
            for g, w in zip(got, want):
                _same(g, w, ())
        assert b.fetch(refs[0][0], 10, 10) == [] and b.fetch(refs[0][0], 0, 1) == [r for r in [] ]
        with pytest.raises(_lib.LtrError):
            b.fetch("no_such_chromosome", 0, 100)
        b.close()


def test_two_files_as_one_stream(decoded):
    (_, refs, ra), (_, refs_b, rb) = decoded
    assert refs == refs_b
    chrom_id = max(range(len(refs)), key=lambda t: sum(r["ref_id"] == t for r in ra))
    chrom, length = refs[chrom_id]
    b = _lib.Bam(BAMS, merge_by_position=True)
    got = b.fetch(chrom, 0, length)
    wa = [r for r in ra if r["ref_id"] == chrom_id]; wb = [r for r in rb if r["ref_id"] == chrom_id]
    assert len(got) == len(wa) + len(wb) and [g["pos"] for g in got] == sorted(g["pos"] for g in got)
    assert [g["name"] for g in got if g["file"] == 0] == [r["name"] for r in wa]            # each file's own order survives the merge
    assert [g["name"] for g in got if g["file"] == 1] == [r["name"] for r in wb]
    assert len(b.read_groups()) >= 2 and {rg["file"] for rg in b.read_groups()} == {0, 1}
    b.close()
    f = _lib.Bam(BAMS, merge_by_position=False)                                             # ORDER_ALNS_BY_FILE
    got = f.fetch(chrom, 0, length)
    assert [g["file"] for g in got] == [0] * len(wa) + [1] * len(wb)
    f.close()


def test_damaged_inputs(tmp_path):
    raw = open(BAMS[0], "rb").read()
    (tmp_path / "t.bam").write_bytes(raw[:len(raw) // 2])                                   # truncated file, intact index
    (tmp_path / "t.bam.bai").write_bytes(open(BAMS[0] + ".bai", "rb").read())
    b = _lib.Bam([str(tmp_path / "t.bam")])
    name, length = b.refs()[0]
    with pytest.raises(_lib.LtrError):
        for nm, ln in b.refs():
            b.fetch(nm, 0, ln)
    b.close()
    (tmp_path / "n.bam").write_bytes(b"not a bam file at all")
    with pytest.raises(_lib.LtrError):
        _lib.Bam([str(tmp_path / "n.bam")])
    (tmp_path / "i.bam").write_bytes(raw)                                                   # no index next to it
    with pytest.raises(_lib.LtrError) as e:
        _lib.Bam([str(tmp_path / "i.bam")])
    assert "index" in str(e.value)


def test_real_reads_chain_up_to_the_gpu_call(tmp_path):
    """examples/real_reads_trio.py without a GPU: BED -> regions, BAM -> reads, reference rebuilt from the '=' runs,
    ltr_left_align_reads, ltr_build_haplotype -- everything before ltr_calc_hap_aln_probs (the GPU test runs the rest
    and compares with the oracle)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("real_reads_trio", os.path.join(ROOT, "examples", "real_reads_trio.py"))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)

    class Stop(Exception):
        pass

    class FakeCtx:
        def calc_hap_aln_probs(self, items):
            self.items = items
            raise Stop()

    ctx = FakeCtx()
    with pytest.raises(Stop):
        rt.run(ctx, tmp_dir=str(tmp_path))
    assert len(ctx.items) >= 35
    n_poly = 0
    for blocks, alns, _ in ctx.items:
        assert len(blocks) == 3 and blocks[1]["is_repeat"] and len(alns) >= 5
        ref_allele = blocks[1]["alleles"][0]
        assert set(ref_allele) <= set(b"ACGT") and all(a["cigar"] for a in alns)
        n_poly += len(blocks[1]["alleles"]) > 1
    assert n_poly >= 10
