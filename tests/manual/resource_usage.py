#!/usr/bin/env python3
"""Per-function register / scratch figures of one kernel translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tests/manual/resource_usage.py ltr_k_plan.hip [extra hipcc flags]"""
import os, re, subprocess, sys
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "longtr_amd", "csrc")
def usage(tu, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-honor-nans", "-std=c++17", "-fPIC", "-pthread",
           "-Rpass-analysis=kernel-resource-usage", *extra, "-c", os.path.join(CSRC, tu), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    cur, rows = None, {}
    for l in err.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = m.group(1); rows[cur] = {}; continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+?): (\S+) \[-Rpass", l)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"\(anonymous namespace\)::", "", n): v for n, v in zip(names, rows.values())}
if __name__ == "__main__":
    for n, v in usage(sys.argv[1], sys.argv[2:]).items():
        n = re.sub(r"\(.*", "", n)
        print(f"{n[:70]:70s} V={v.get('VGPRs')} S={v.get('TotalSGPRs')} scratch={v.get('ScratchSize [bytes/lane]')} sgprSpill={v.get('SGPRs Spill')} vgprSpill={v.get('VGPRs Spill')} occ={v.get('Occupancy [waves/SIMD]')} lds={v.get('LDS Size [bytes/block]')}")
