#!/usr/bin/env python3
"""profiles/make_traffic.py <summary.json from profiles/collect.sh> <round dir>

Reduces the rocprofv3 summary of one collection to the files committed under profiles/<round>/:
  kernel_stats.csv     per kernel: calls, average / total ms, % of GPU time (kernel trace)
  pmc_traffic.json     per kernel and launch: HBM bytes (FETCH_SIZE + WRITE_SIZE, KiB x 1024; separate
                       --pmc passes) and the share of VALU issue slots in use
                       (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)) -- read by bench.py.
Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE under-reports a WIDE coalesced
streaming read by half on gfx950; these kernels' global traffic is 1-2-byte haplotype-row loads, 8-16-byte table /
strip loads and stores and 40-byte descriptors, not 16-byte-per-lane streams, so no x2 is applied (an upper bound of
the read side is 2x the figure given)."""
import csv, json, os, sys

def main(summary, out_dir):
    d = json.load(open(summary))
    os.makedirs(out_dir, exist_ok=True)
    ks = d["kernels"]
    with open(os.path.join(out_dir, "kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "avg_ms", "total_ms", "pct"])
        for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"]):
            w.writerow([k, v["calls"], "%.6f" % v["avg_ms"], "%.3f" % v["total_ms"], v["pct"]])
    per = {}
    for k, c in d["counters"].items():
        e = {}
        if "FETCH_SIZE" in c:
            e["fetch_bytes"] = c["FETCH_SIZE"]["per_launch"] * 1024.0
        if "WRITE_SIZE" in c:
            e["write_bytes"] = c["WRITE_SIZE"]["per_launch"] * 1024.0
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]["per_launch"] > 0:
            e["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]["per_launch"]
            e["valu_issue_frac"] = c["SQ_INSTS_VALU"]["per_launch"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"]["per_launch"] / 8.0)
        if "SQ_INSTS_SALU" in c:
            e["salu_insts_per_launch"] = c["SQ_INSTS_SALU"]["per_launch"]
        if k in ks:
            e["avg_ms_rocprof"] = ks[k]["avg_ms"]
        if e:
            per[k] = e
    dom = max((k for k in ks if k.startswith("ltr_dp")), key=lambda k: ks[k]["total_ms"], default=None)
    out = {"source": d["source"] + "; profiles/make_traffic.py",
           "library": d.get("library"),
           "units": "bytes per launch = counter (KiB) x 1024, FETCH_SIZE and WRITE_SIZE from separate --pmc passes; no x2 read correction "
                    "(narrow loads, see the script's header); valu_issue_frac = SQ_INSTS_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)",
           "dominant_kernel": dom,
           "dominant_kernel_hbm_bytes_per_launch": (per.get(dom, {}).get("fetch_bytes", 0) + per.get(dom, {}).get("write_bytes", 0)) if dom else None,
           "dominant_kernel_valu_issue_frac": per.get(dom, {}).get("valu_issue_frac") if dom else None,
           "per_kernel": per}
    json.dump(out, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print("dominant", dom, per.get(dom))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
