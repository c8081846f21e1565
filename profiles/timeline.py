"""GPU occupancy timeline from a rocprofv3 --kernel-trace CSV: per call of the profiled program's
hot section (separated by idle gaps > --gap ms) the span, the time at least one kernel was running,
and the largest idle gaps inside the span.

    python profiles/timeline.py <..._kernel_trace.csv> [--gap 20]
"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--gap", type=float, default=20.0, help="idle ms that separates two calls")
    ap.add_argument("--top", type=int, default=5)
    a = ap.parse_args()
    iv = []
    with open(a.trace) as f:
        for row in csv.DictReader(f):
            iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
    iv.sort()
    if not iv:
        print("no kernels"); return
    groups, cur = [], [iv[0]]
    end = iv[0][1]
    for s, e, n in iv[1:]:
        if s - end > a.gap * 1e6:
            groups.append(cur); cur = []
        cur.append((s, e, n)); end = max(end, e)
    groups.append(cur)
    for gi, g in enumerate(groups):
        t0 = g[0][0]; t1 = max(e for _, e, _ in g)
        busy, gaps = 0, []
        cs, ce = g[0][0], g[0][1]
        for s, e, _ in g[1:]:
            if s > ce:
                busy += ce - cs; gaps.append((s - ce, ce - t0)); cs, ce = s, e
            else:
                ce = max(ce, e)
        busy += ce - cs
        ksum = sum(e - s for s, e, _ in g)
        gaps.sort(reverse=True)
        print(f"call {gi}: {len(g)} kernels, span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, "
              f"kernel-time sum {ksum / 1e6:.2f} ms; largest idle gaps (ms @ offset): "
              + ", ".join(f"{d / 1e6:.2f}@{o / 1e6:.1f}" for d, o in gaps[:a.top]))


if __name__ == "__main__":
    main()
