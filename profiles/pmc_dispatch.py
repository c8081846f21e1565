#!/usr/bin/env python3
"""profiles/pmc_dispatch.py <dir of profiles/pmc_dispatch.sh> <kernel filter regex> <bench args>: per-dispatch table."""
import collections, csv, glob, hashlib, os, re, sys

def short(name):
    m = re.search(r"(ltr_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][:40]

def main(out, flt, args):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "longtr_amd", "csrc", "libltr_gpu.so")
    seq = collections.defaultdict(lambda: collections.defaultdict(dict))
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        cnt = collections.Counter()
        for r in rows:
            k = short(r["Kernel_Name"]); i = cnt[k]; cnt[k] += 1
            seq[k][i]["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            seq[k][i]["wgs"] = int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 1) or 1), 1)
    for d in glob.glob(os.path.join(out, "pmc_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            ids = collections.defaultdict(set)
            for r in rows:
                ids[short(r["Kernel_Name"])].add(int(r["Dispatch_Id"]))
            pos = {k: {did: i for i, did in enumerate(sorted(v))} for k, v in ids.items()}
            for r in rows:
                k = short(r["Kernel_Name"]); i = pos[k][int(r["Dispatch_Id"])]
                seq[k][i][r["Counter_Name"]] = seq[k][i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print(f"# rocprofv3 per dispatch: python3 bench.py {args} --no-cpu-baseline --no-end-to-end --no-verify --steps 1 --warmup 0 --debug fan_lanes=1")
    print(f"# library sha256_16 {hashlib.sha256(open(so, 'rb').read()).hexdigest()[:16] if os.path.exists(so) else None}; kernel filter /{flt}/")
    print("# valu_issue = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; bytes = FETCH_SIZE / WRITE_SIZE (KiB x 1024, no x2)")
    print("kernel dispatch workgroups us VALU SALU LDS waves valu_issue wait fetch_bytes write_bytes")
    for k in sorted(seq):
        if not re.search(flt, k):
            continue
        for i in sorted(seq[k]):
            v = seq[k][i]
            gui = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
            print(k.replace(" ", ""), i, v.get("wgs"), "%.1f" % v.get("us", 0), "%.4g" % v.get("SQ_INSTS_VALU", 0), "%.4g" % v.get("SQ_INSTS_SALU", 0),
                  "%.4g" % v.get("SQ_INSTS_LDS", 0), "%.0f" % v.get("SQ_WAVES", 0), "%.3f" % (v.get("SQ_INSTS_VALU", 0) * 4 / (1024 * gui) if gui else 0),
                  "%.3f" % (v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)), "%.4g" % (v.get("FETCH_SIZE", 0) * 1024), "%.4g" % (v.get("WRITE_SIZE", 0) * 1024))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
