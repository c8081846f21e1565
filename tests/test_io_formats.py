"""On-disk formats that need no htslib: region (BED) reader, indexed-FASTA reader,
position-ordered BGZF VCF writer.  Host code: no GPU needed.

The region reader is pinned to the REFERENCE: tests/golden/regions.json holds what the reference's own
readRegions / orderRegions (src/region.cpp, compiled from where it lies: oracle/Makefile `regions`,
oracle/gen_golden_regions.py) returned or died with; where the compiled command is present the cases are
also replayed against it live.  The FASTA reader and the VCF writer sit on htslib in the reference
(faidx, bgzf): unpinned -- checked against the published formats (.fai arithmetic, gzip members) and a
Python model of the record heap."""
import gzip, json, os, struct, subprocess, sys

import numpy as np
import pytest

from longtr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "regions.json")))["cases"]
REF = os.path.join(ROOT, "oracle", "_ref", "ref_regions")


def _run_ours(case, tmp_path, k):
    path = tmp_path / f"r{k}.bed"
    path.write_text(case["text"])
    try:
        regs, lines = _lib.read_regions(str(path), case["max_regions"], case["chrom_limit"] or None, bool(case["order"]))
        return dict(regions=regs, lines=lines), str(path)
    except _lib.LtrError as e:
        assert e.code == -1
        return dict(error=str(e).split(": ", 1)[1]), str(path)


def test_regions_equal_the_reference_golden(tmp_path):
    assert len(GOLDEN) >= 25 and sum("error" in c["expect"] for c in GOLDEN) >= 10
    for k, case in enumerate(GOLDEN):
        got, path = _run_ours(case, tmp_path, k)
        want = case["expect"]
        if "error" in want:
            assert "error" in got, (k, got)
            assert got["error"] == want["error"].replace("<PATH>", path), k
        else:
            assert got == want, k


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/ref_regions not built (needs /root/reference)")
def test_regions_equal_the_compiled_reference_live(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden_regions as g
    rng = np.random.default_rng(7)
    cases = g.cases()
    for k in range(30):                                        # + fresh random files the golden set has never seen
        n = int(rng.integers(1, 60))
        lines = []
        for i in range(n):
            start = int(rng.integers(1, 5000))
            lines.append("\t".join([f"chr{int(rng.integers(1, 3))}", str(start), str(start + int(rng.integers(1, 90))),
                                    ",".join("ACGT"[int(x)] * int(rng.integers(1, 4)) for x in rng.integers(0, 4, int(rng.integers(1, 3))))]
                                   + ([f"n{i}"] if rng.random() < 0.3 else [])))
        cases.append(dict(text="\n".join(lines) + "\n", max_regions=int(rng.integers(1, 80)), chrom_limit=("chr1" if k % 3 == 0 else ""), order=1))
    for k, case in enumerate(cases):
        got, path = _run_ours(case, tmp_path, k)
        want = g.run_reference(case)
        if "error" in want:
            assert got.get("error") == want["error"].replace("<PATH>", path), k
        else:
            assert got == want, k


def _write_fasta(path, seqs, width):
    fai = []
    with open(path, "wb") as f:
        for name, s in seqs:
            f.write(f">{name} some description\n".encode())
            off = f.tell()
            for i in range(0, len(s), width):
                f.write(s[i:i + width].encode() + b"\n")
            fai.append(f"{name}\t{len(s)}\t{off}\t{width}\t{width + 1}")
    with open(str(path) + ".fai", "w") as f:
        f.write("\n".join(fai) + "\n")


def test_fasta_reader(tmp_path):
    rng = np.random.default_rng(11)
    seqs = [(f"chr{k}", "".join(rng.choice(list("ACGTacgtN"), size=int(n)))) for k, n in enumerate([1, 59, 60, 61, 1234, 7001])]
    p = tmp_path / "ref.fa"
    _write_fasta(p, seqs, 60)
    fa = _lib.Fasta(str(p))
    assert fa.names() == [n for n, _ in seqs]
    for name, s in seqs:
        assert fa.seq_len(name) == len(s)
        assert fa.fetch(name, 0, len(s) - 1) == s
        for _ in range(40):
            a = int(rng.integers(-5, len(s) + 5)); b = int(rng.integers(-5, len(s) + 70))
            want = s[max(a, 0):min(b, len(s) - 1) + 1] if max(a, 0) <= min(b, len(s) - 1) else ""     # faidx_fetch_seq clamps
            assert fa.fetch(name, a, b) == want, (name, a, b)
    assert fa.seq_len("chrZ") == -1
    with pytest.raises(_lib.LtrError) as e:
        fa.fetch("chrZ", 0, 10)
    assert "No entry for chromosome chrZ found in FASTA files" in str(e.value)
    assert fa.contig_lines() == "".join(f"##contig=<ID={n},length={len(s)}>\n" for n, s in seqs)
    fa.close()
    # a directory of *.fa files; a file without an index; a name in two files
    d = tmp_path / "dir"; d.mkdir()
    _write_fasta(d / "a.fa", seqs[:2], 50); _write_fasta(d / "b.fa", seqs[2:4], 70); (d / "notes.txt").write_text("x")
    fd = _lib.Fasta(str(d))
    assert sorted(fd.names()) == sorted(n for n, _ in seqs[:4]) and fd.fetch("chr3", 10, 20) == seqs[3][1][10:21]
    fd.close()
    (d / "c.fa").write_text(">q\nACGT\n")
    with pytest.raises(_lib.LtrError) as e:
        _lib.Fasta(str(d))
    assert "No FASTA index file exists for" in str(e.value) and "samtools faidx" in str(e.value)
    os.remove(d / "c.fa")
    _write_fasta(d / "c.fa", seqs[:1], 60)
    with pytest.raises(_lib.LtrError) as e:
        _lib.Fasta(str(d))
    assert "Multiple entries for chromosome chr0 exist in FASTA files" in str(e.value)
    with pytest.raises(_lib.LtrError) as e:
        _lib.Fasta(str(tmp_path / "empty_dir_that_is_missing"))
    assert "Failed to access directory" in str(e.value)


def _model_writer(events):
    """VCFWriter::add_vcf_record (vcf_writer.cpp:7-36) + close, for records with distinct positions per chromosome."""
    out, heap, chrom = [], [], ""
    for c, pos, text in events:
        if c != chrom:
            out += [t for _, t in sorted(heap)]; heap = []; chrom = c
        else:
            heap.sort()
            while heap and heap[0][0] < pos - 50:
                out.append(heap.pop(0)[1])
        heap.append((pos, text))
    out += [t for _, t in sorted(heap)]
    return out


def test_vcf_writer_orders_records_and_writes_bgzf(tmp_path):
    rng = np.random.default_rng(12)
    events = []
    for chrom in ("chr1", "chr2", "chr10"):
        pos = sorted(set(int(x) for x in rng.integers(1, 200000, 3000)))
        jit = [p + int(rng.integers(-25, 26)) for p in pos]                    # arrival order: sorted region starts, record positions up to 50 bp apart
        seen = set()
        for p, q in zip(pos, jit):
            if q in seen or q < 1:
                continue
            seen.add(q)
            events.append((chrom, q, f"{chrom}\t{q}\t.\tA\tAT\t.\t.\tSTART={p};" + "X" * int(rng.integers(0, 120))))
    header = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    want = header + "".join(t + "\n" for t in _model_writer(events))
    for name in ("out.vcf", "out.vcf.gz"):
        w = _lib.VcfWriter(str(tmp_path / name))
        w.header(header)
        for c, p, t in events:
            w.add_record(c, p, t)
        w.close()
    assert (tmp_path / "out.vcf").read_text() == want
    raw = (tmp_path / "out.vcf.gz").read_bytes()
    assert gzip.decompress(raw).decode() == want                               # a series of gzip members
    assert raw.endswith(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))   # BGZF end-of-file block
    at, blocks, total = 0, 0, 0
    while at < len(raw):                                                        # every member: BC extra field, BSIZE, <= 64 KB, ISIZE
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 14] == b"BC"
        bsize = struct.unpack("<H", raw[at + 16:at + 18])[0] + 1
        isize = struct.unpack("<I", raw[at + bsize - 4:at + bsize])[0]
        assert bsize <= 65536 and isize <= 65536
        total += isize; blocks += 1; at += bsize
    assert at == len(raw) and total == len(want.encode()) and blocks >= 3
    # records at the same position: all of them come out, in some heap order
    w = _lib.VcfWriter(str(tmp_path / "ties.vcf"))
    for k in range(20):
        w.add_record("chr1", 100 + (k % 3), f"r{k}")
    w.close()
    got = (tmp_path / "ties.vcf").read_text().split()
    assert sorted(got) == sorted(f"r{k}" for k in range(20)) and [int(g[1:]) % 3 for g in got] == sorted(int(g[1:]) % 3 for g in got)
