"""-m gpu: REAL reads through the whole chain (examples/real_reads_trio.py): the reference's bundled HiFi BAMs of
the HG002 / HG003 / HG004 trio at the loci of its bundled BED -- BAM and BED input, reference rebuilt from the
reads' '=' runs, left_align_reads, exact-allele candidates, the GPU DP for every locus in one call, posteriors,
genotypes, VCF.  Checked: the LL matrix of every locus bit for bit against the CPU oracle (pool -> process_reads
-> scatter, the reference's own sequence of steps), the trio's genotypes for Mendelian consistency, and the
BGZF VCF for being readable."""
import gzip, importlib.util, os

import numpy as np
import pytest

from longtr_amd import _abi
from test_gpu_host_path import _expected_calc_hap_aln_probs, bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("real_reads_trio", os.path.join(ROOT, "examples", "real_reads_trio.py"))
rt = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(rt)


@pytest.mark.gpu
def test_trio_real_reads_end_to_end(gpu_ctx, tmp_path):
    vcf = tmp_path / "trio.vcf.gz"
    loci = rt.run(gpu_ctx, str(vcf), tmp_dir=str(tmp_path))
    ok = [l for l in loci if l["status"] == "ok"]
    assert len(loci) >= 35 and len(ok) >= 35
    prm, sp = _abi.default_params(), _abi.default_stutter_params()
    pairs = 0
    for l in ok:
        want, ws = _expected_calc_hap_aln_probs(prm, sp, l["blocks"], l["alns"], None)
        assert np.array_equal(bits(l["ll"]), bits(want)) and np.array_equal(l["seeds"], ws), l["region"]["name"]
        assert (l["ll"].max(axis=1) > -600.0).mean() > 0.9               # real reads: (nearly) every one aligns to some candidate of its locus
        pairs += l["ll"].size
    assert pairs > 5000
    big = [l for l in ok if max(l["allele_lens"]) > 2000]                   # Human_STR_219: a 2.9 kb VNTR, every sample homozygous for a 5.3 kb allele
    assert big and all(l["gt_lens"][0] == l["gt_lens"][1] == l["gt_lens"][2] for l in big)
    # Mendelian consistency of the called allele lengths (HG002 = child of HG003, HG004)
    poly = [l for l in ok if len(l["allele_lens"]) > 1]
    assert len(poly) >= 5
    viol = []
    for l in ok:
        child, pa, ma = l["gt_lens"]
        if not any(child[0] in x and child[1] in y for x, y in ((pa, ma), (ma, pa))):
            viol.append((l["region"]["name"], l["region"]["period"]))
    # every locus with a motif of 2+ bases is consistent; homopolymers (no stutter model in this chain) may miss by a base
    assert all(p == 1 for _, p in viol) and len(viol) <= 3, viol
    text = gzip.decompress(vcf.read_bytes()).decode().splitlines()
    recs = [t.split("\t") for t in text if not t.startswith("#")]
    assert len(recs) == len(ok) and [int(r[1]) for r in recs] == sorted(int(r[1]) for r in recs)
    for r in recs:                                                       # the reference's record layout (write_vcf_record)
        assert len(r) == 9 + 3 and r[8].startswith("GT:GB:Q:PQ:DP") and "PERIOD=" in r[7] and "END=" in r[7]
        gt = r[9].split(":")[0]
        assert gt == "." or all(x.isdigit() for x in gt.replace("|", "/").split("/"))
