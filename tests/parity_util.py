"""Checker-side helpers shared by the -m gpu scale tests and bench.py's post-timing check (test
infrastructure: uses the CPU oracle; nothing under longtr_amd/ imports this)."""
import numpy as np

import oracle_lib as ol
from longtr_amd import _abi


def sub_batch(batch, loci_ids, reads_per_locus=None):
    """PackedBatch of the given loci of `batch` (optionally only their first reads_per_locus pooled reads)."""
    sub = []
    for l in loci_ids:
        r0, r1 = int(batch.locus_read_off[l]), int(batch.locus_read_off[l + 1])
        if reads_per_locus is not None:
            r1 = min(r1, r0 + reads_per_locus)
        reads = [batch.read_bytes[batch.read_off[r]:batch.read_off[r + 1]].tobytes() for r in range(r0, r1)]
        haps = [batch.hap_bytes[batch.hap_off[h]:batch.hap_off[h + 1]].tobytes()
                for h in range(batch.locus_hap_off[l], batch.locus_hap_off[l + 1])]
        sub.append((reads, haps))
    return _abi.PackedBatch(sub)


def read_class(m):
    """Launch class of a read of length m as the plan bins it (ltr_gpu.hip): (lanes per pair, strip width)."""
    C = max(m - 1, 1)
    if C <= 16 * 8:
        return (16, (C + 15) // 16)
    if C <= 32 * 20:
        return (32, (C + 31) // 32)
    ncb = (C + 1023) // 1024
    return (64, (C + 64 * ncb - 1) // (64 * ncb))


def stratified_oracle_check(batch, ll, params, n_loci_target=240, reads_per_locus=3, seed=7, extra_loci=()):
    """Bit-compare a sample of a full pass with the CPU oracle.  The sample is stratified over the
    launch classes (strip widths) of the loci's reads, so that every kernel instantiation the pass
    used is checked, plus `extra_loci` (e.g. the most expensive ones)."""
    rng = np.random.default_rng(seed)
    rl = np.diff(batch.read_off)
    by_class = {}
    for l in range(batch.n_loci):
        r0 = int(batch.locus_read_off[l])
        if int(batch.locus_read_off[l + 1]) == r0:
            continue
        by_class.setdefault(read_class(int(rl[r0])), []).append(l)
    per = max(2, -(-n_loci_target // max(len(by_class), 1)))
    pick = [int(x) for x in extra_loci]
    for cls in sorted(by_class):
        ids = by_class[cls]
        pick.extend(int(x) for x in rng.choice(ids, size=min(per, len(ids)), replace=False))
    pick = sorted(set(pick))
    sub = sub_batch(batch, pick, reads_per_locus)
    ref, _, cells = ol.oracle_align_batch(sub, params)
    checked = mism = 0
    for k, l in enumerate(pick):
        want = sub.locus_matrix(ref, k)
        got = batch.locus_matrix(ll, l)[:want.shape[0]]
        checked += want.size
        mism += int((want.view(np.uint64) != got.view(np.uint64)).sum())
    return {"loci": len(pick), "checked_pairs": int(checked), "mismatches": int(mism), "classes_covered": len(by_class),
            "cells": cells, "reads_per_locus": reads_per_locus, "checker": "oracle/libltr_oracle.so (bit-exact compare)"}
