"""The drop-in boundary, compiled: integration/GpuHapAligner.h (the adapter a LongTR maintainer adds,
INTEGRATION.md) is built against the reference's own headers and objects + libltr_gpu.so by
`make -C oracle adapter` (dev container; the GPU box receives the prebuilt binary like oracle/_ref's
library).  CPU: the adapter's flattening of reference Haplotype objects; GPU: reference objects ->
adapter -> C-ABI -> bits equal to the golden vectors of the reference's own align_seq_to_hap."""
import os
import subprocess

import numpy as np
import pytest

import adapter_util as au
import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "adapter"], check=True)
    return os.path.exists(au.BIN)


def test_adapter_compiles_against_reference_headers_and_flattens():
    if not _built():
        pytest.skip("oracle/_ref/adapter_check is built in the dev container (needs the reference headers)")
    d = gu.load("process_locus")
    for L in d["loci"]:
        r = au.run("flatten", d["params"], L)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.splitlines()
        assert len(lines) == len(L["alleles"]) + 1
        for k, ln in enumerate(lines[:-1]):
            mine, theirs = ln.split()
            # haplotype k through the adapter's flattened blocks + ltr_haplotype_seq == Haplotype::get_seq()
            assert mine == theirs == L["lflank"] + L["alleles"][k] + L["rflank"]
        assert lines[-1] == f"blocks 3 repeat_block_period {L['period']}"


@pytest.mark.gpu
def test_reference_objects_through_adapter_equal_golden():
    assert os.path.exists(au.BIN), "oracle/_ref/adapter_check missing: run build() in the dev container"
    d = gu.load("process_locus")
    for L in d["loci"]:
        r = au.run("run", d["params"], L)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.splitlines()
        nv = len(L["alns"]) * len(L["alleles"])
        got = np.asarray([float.fromhex(x) for x in lines[:nv]])
        assert np.array_equal(got.view(np.uint64), gu.unhex(L["ll_hex"]).view(np.uint64))
        assert [int(x.split()[1]) for x in lines[nv:]] == [len(a["seq"]) - 1 for a in L["alns"]]


def _check_lines(lines, loci):
    at = 0
    for L in loci:
        nv = len(L["alns"]) * len(L["alleles"])
        got = np.asarray([float.fromhex(x) for x in lines[at:at + nv]])
        assert np.array_equal(got.view(np.uint64), gu.unhex(L["ll_hex"]).view(np.uint64))
        at += nv
        assert [int(x.split()[1]) for x in lines[at:at + len(L["alns"])]] == [len(a["seq"]) - 1 for a in L["alns"]]
        at += len(L["alns"])
    assert at == len(lines)


@pytest.mark.gpu
def test_reference_objects_through_the_batch_adapter_equal_golden():
    """integration/GpuHapAlignerBatch.h: every golden locus staged as reference Haplotype / Alignment objects, ONE
    ltr_calc_hap_aln_probs call (the throughput entry point: pooling, trimming, scoring, fan-out inside), per-read rows
    equal to the golden bits of the reference's own align_seq_to_hap -- and the same loci one GpuHapAligner each."""
    assert os.path.exists(au.BIN), "oracle/_ref/adapter_check missing: run build() in the dev container"
    d = gu.load("process_locus")
    loci = list(d["loci"])
    r = au.run("batch", d["params"], loci)
    assert r.returncode == 0, r.stderr
    _check_lines(r.stdout.splitlines(), loci)
    r = au.run("run", d["params"], loci)
    assert r.returncode == 0, r.stderr
    _check_lines(r.stdout.splitlines(), loci)


@pytest.mark.gpu
def test_per_locus_latency_with_one_context_per_process():
    """The reference constructs a HapAligner per locus: with the process-wide context of GpuContext a GpuHapAligner is
    cheap to construct -- the per-locus latency including construction stays well below a millisecond."""
    assert os.path.exists(au.BIN)
    d = gu.load("process_locus")
    r = au.run("latency", d["params"], list(d["loci"]), 40)
    assert r.returncode == 0, r.stderr
    ms = float(r.stdout.split()[1])
    print(r.stdout.strip())
    assert ms < 1.0, r.stdout
