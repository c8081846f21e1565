#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (profiles/collect.sh) to one small JSON: per-kernel average
duration from the kernel trace and per-kernel, per-launch counter sums from the PMC passes."""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(name):
    m = re.search(r"(ltr_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][:60]

def lib_id():
    """Build id of the library the profiled runs loaded: ltr_version() + _lib.source_id() (hash of sources + flags, independent
    of the build directory) + the .so's own SHA-256.  bench.py attaches the counters only when source_id matches."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "longtr_amd", "csrc", "libltr_gpu.so")
    try:
        sys.path.insert(0, root)
        from longtr_amd import _lib
        ver, sid = _lib.lib().ltr_version().decode(), _lib.source_id()
    except Exception:
        ver, sid = "unknown", None
    return {"version": ver, "source_id": sid, "so_sha256_16": hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] if os.path.exists(path) else None}


def main(out, workload="config3"):
    res = {"source": f"rocprofv3 (profiles/collect.sh): python3 bench.py --workload {workload} --no-cpu-baseline --no-end-to-end --steps 2 --warmup 1 --debug fan_lanes=1 (one stream: every launch of a kernel is the launch bench.py times)",
           "library": lib_id(), "kernels": {}, "counters": {}}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            res["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(int))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
        for k in acc:
            for c in acc[k]:
                res["counters"].setdefault(k, {})[c] = {"per_launch": acc[k][c] / max(calls[k][c], 1), "launches": calls[k][c]}
    json.dump(res, sys.stdout, indent=1, sort_keys=True)

if __name__ == "__main__":
    main(*sys.argv[1:3])
