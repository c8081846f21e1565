"""Manual GPU check (not a pytest file): config 2 mismatches in detail, fast vs exact kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from longtr_amd import _abi, _lib, synth
import oracle_lib as ol
ctx = _lib.Context(0)
p = ctx.params
loci, _ = synth.config_loci("config2")
def run(tag, flat):
    loci = flat
    batch = _abi.PackedBatch(flat)
    ll, _ = ctx.align_batch(batch)
    ref, _, _ = ol.oracle_align_batch(batch, p)
    bad = np.where(ll.view(np.uint64) != ref.view(np.uint64))[0]
    print(tag, "mismatches", bad.size, "of", ll.size)
    off = 0
    for li, (reads, haps) in enumerate(loci):
        H = len(haps)
        for r, rd in enumerate(reads):
            for h, hp in enumerate(haps):
                k = off + r * H + h
                if k in bad[:12]:
                    print(f"  locus {li} read {r} (m={len(rd)}) hap {h} (len={len(hp)}): gpu {ll[k]!r} ref {ref[k]!r}")
        off += len(reads) * H
flat = []
for L in loci:
    pools, idx = synth.pool_reads(L.trimmed_reads)
    flat.append((pools, L.haplotypes))
run("fast", flat)
# force the exact (byte-compare) kernel: one lower-case base per read makes every pair "generic"
lo = []
for reads, haps in flat:
    lo.append(([bytes(rd[:-1]) + bytes([rd[-1] | 0x20]) for rd in reads], haps))
run("exact", lo)
