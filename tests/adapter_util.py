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


def run(mode, params, L, *args):
    """L: one golden locus or a list of them (any number of loci on the harness' stdin)."""
    loci = L if isinstance(L, list) else [L]
    return subprocess.run([BIN, mode, *[str(a) for a in args]], input="".join(locus_text(params, x) for x in loci),
                          capture_output=True, text=True, timeout=600)
