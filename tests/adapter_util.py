"""Driver for oracle/_ref/adapter_check (integration/GpuHapAligner.h compiled against the reference headers)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "adapter_check")


def locus_text(params, L):
    toks = [*params["values7_hex"], str(params["indel_flank_len"]), str(L["start"]), L["lflank"], str(len(L["alleles"])),
            *L["alleles"], L["rflank"], str(L["period"]), str(len(L["alns"]))]
    for a in L["alns"]:
        toks += [str(a["start"]), str(a["stop"]), a["seq"], str(len(a["cigar"]))]
        for t, k in a["cigar"]:
            toks += [t, str(k)]
    return " ".join(toks) + "\n"


def run(mode, params, L):
    return subprocess.run([BIN, mode], input=locus_text(params, L), capture_output=True, text=True, timeout=300)
