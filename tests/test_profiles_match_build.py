"""The committed counter summary belongs to the library in the tree: bench.py attaches `roofline.traffic` / `valu_issue_frac` to its line
only when `profiles/<round>/pmc_traffic.json` names the `source_id` of the sources that are built (longtr_amd/_lib.py::source_id: flags +
every file of longtr_amd/csrc + include/ltr_gpu.h).  Mid-round, after a kernel change and before the profiles are collected again, the
ids differ: that is reported as a skip, not a failure -- the round's last commit is expected to make this test pass."""
import glob
import json
import os

import pytest

from longtr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_latest_counter_summary_is_of_this_build():
    rounds = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_traffic.json")))
    assert rounds, "no profiles/rNN/pmc_traffic.json"
    d = json.load(open(rounds[-1]))
    lib = d.get("library", {})
    assert "source_id" in lib and "dominant_kernel" in d and d["dominant_kernel"] in d["per_kernel"]
    k = d["per_kernel"][d["dominant_kernel"]]
    assert k["fetch_bytes"] > 0 and k["write_bytes"] > 0 and 0.0 < k["valu_issue_frac"] <= 1.0
    if lib["source_id"] != _lib.source_id():
        pytest.skip(f"{os.path.relpath(rounds[-1], ROOT)} is of source_id {lib['source_id']}, the tree is {_lib.source_id()}: re-collect (profiles/collect.sh)")
    # the bench line of the same round was produced with these counters attached
    line = json.loads(open(os.path.join(os.path.dirname(rounds[-1]), "bench_config3.json")).read())
    assert line["library"]["source_id"] == lib["source_id"]
    assert line["roofline"]["counters_from"] and abs(line["roofline"]["traffic"] - (k["fetch_bytes"] + k["write_bytes"])) <= 0.01 * line["roofline"]["traffic"]
