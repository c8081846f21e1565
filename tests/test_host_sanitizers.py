"""Host code of the C-ABI library under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not
available on this pool): tests/host_sanitize/harness.cpp fuzzes every host-only entry point."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "longtr_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_host_entry_points_under_sanitizers(tmp_path, sanitizer):
    """ASan + UBSan, then ThreadSanitizer (ltr_calc_hap_aln_probs prepares its loci on several threads)."""
    exe = str(tmp_path / "harness")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={sanitizer}",
           "-fno-sanitize-recover=all", "-ffp-contract=off", "-pthread",
           os.path.join(ROOT, "tests", "host_sanitize", "harness.cpp"), os.path.join(CSRC, "ltr_host.cpp"),
           os.path.join(CSRC, "ltr_genotype.cpp"), os.path.join(CSRC, "ltr_vcf.cpp"), os.path.join(CSRC, "ltr_prep.cpp"), os.path.join(CSRC, "ltr_io.cpp"), os.path.join(CSRC, "ltr_bam.cpp"), "-lz", "-o", exe]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "bam", "HG002_sample_reads.bam")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host sanitizer harness" in r.stdout
