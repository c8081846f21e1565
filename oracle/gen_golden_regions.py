"""TEST INFRASTRUCTURE: golden vectors for ltr_read_regions from the reference's own reader.

    python oracle/gen_golden_regions.py        # needs oracle/_ref/ref_regions (make -C oracle regions)

Writes tests/golden/regions.json: region files (text), the arguments, and what the reference's
readRegions / orderRegions (src/region.cpp, compiled from where it lies) printed -- the parsed regions and
its log -- or, for a malformed file, the message it died with."""
import json, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref", "ref_regions")


def cases():
    rng = np.random.default_rng(20250226)
    out = []
    motifs = ["A", "AC", "AAT", "AAAG", "GATA", "AAT,AAG", "A,AC", "AC,", ",AC", "aag", "ACGTN", "AAAAC"]
    def line(chrom=None, start=None, stop=None, motif=None, name=None, sep="\t"):
        chrom = chrom or f"chr{int(rng.integers(1, 4))}"
        start = int(rng.integers(1, 100000)) if start is None else start
        stop = start + int(rng.integers(1, 2000)) if stop is None else stop
        motif = motifs[int(rng.integers(len(motifs)))] if motif is None else motif
        cols = [chrom, str(start), str(stop), motif] + ([name] if name else [])
        return sep.join(cols)
    for k in range(12):                                   # well-formed files
        n = int(rng.integers(1, 40))
        lines = [line(name=(f"STR_{i}" if rng.random() < 0.5 else None), sep=("\t" if k % 3 else " ")) for i in range(n)]
        if k == 5:
            lines = [l + "\textra\tcolumns 7" for l in lines]
        text = "\n".join(lines) + ("\n" if k % 2 else "")
        out.append(dict(text=text, max_regions=int(rng.integers(1, 50)) if k % 4 == 0 else 1000000, chrom_limit=("chr2" if k % 5 == 1 else ""), order=int(k % 2)))
    bad = ["chr1\t0\t120\tAC", "chr1\t100\t100\tAC", "chr1\t100\t90\tAC", "chr1\t100\t200\tA1C", "chr1\t100\t200\t5\t5.8\tHuman_STR_211\tAAAAC",
           "chr1\t100\t200", "chr1\tx\t200\tAC", "chr1\t100\t2e2\tAC", "", "chr1\t100\t200\tAC-T", "chr1\t3000000000\t3000000100\tAC"]
    for b in bad:
        good = [line() for _ in range(3)]
        out.append(dict(text="\n".join(good[:2] + [b] + good[2:]) + "\n", max_regions=1000000, chrom_limit="", order=0))
    out.append(dict(text="\n".join(line(chrom="chr1") for _ in range(5)) + "\n", max_regions=1000000, chrom_limit="chrX", order=0))       # nothing on the requested chromosome
    out.append(dict(text="\n".join(line(chrom="chr1") for _ in range(5)) + "\nchr2\t0\t5\tA\n", max_regions=5, chrom_limit="", order=0))  # the bad line is read but never parsed
    out.append(dict(text="\n".join(line(chrom="chr1") for _ in range(5)) + "\nchr2\t0\t5\tA\n", max_regions=6, chrom_limit="", order=0))
    return out


def run_reference(case):
    with tempfile.NamedTemporaryFile("w", suffix=".bed", delete=False) as f:
        f.write(case["text"]); path = f.name
    try:
        r = subprocess.run([REF, path, str(case["max_regions"]), case["chrom_limit"] or "-", str(case["order"])], capture_output=True, text=True)
    finally:
        os.unlink(path)
    if r.returncode != 0:
        msg = r.stderr
        assert msg.startswith("ERROR: ") and msg.endswith("\nExiting...\n"), msg
        return dict(error=msg[len("ERROR: "):-len("\nExiting...\n")].replace(path, "<PATH>"))
    body, log = r.stdout.split("--log--\n")
    regions = []
    for l in body.splitlines():
        chrom, start, stop, motif, name, period, pstr = l.split("\t")
        regions.append(dict(chrom=chrom, start=int(start), stop=int(stop), motif=motif, name=name, period=int(period), period_str=pstr))
    lines = int(log.split("Region file contains ")[1].split(" regions")[0])
    return dict(regions=regions, lines=lines)


def main():
    if not os.path.exists(REF):
        sys.exit("build oracle/_ref/ref_regions first: make -C oracle regions")
    out = []
    for c in cases():
        out.append(dict(c, expect=run_reference(c)))
    dst = os.path.join(HERE, "..", "tests", "golden", "regions.json")
    json.dump(dict(generator="oracle/gen_golden_regions.py (reference src/region.cpp compiled from source)", cases=out), open(dst, "w"), indent=0)
    print(f"{len(out)} cases -> {dst}; {sum('error' in c['expect'] for c in out)} of them end in printErrorAndDie")


if __name__ == "__main__":
    main()
