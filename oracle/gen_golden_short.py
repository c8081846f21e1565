#!/usr/bin/env python3
"""Generate tests/golden/stutter_pieces.json: the pieces of the SHORT (stutter) path that compile without htslib, run
through the COMPILED REFERENCE (oracle/_ref/libltr_ref.so: StutterAlignerClass, RepeatStutterInfo, StutterModel,
BaseQuality, fast_log_sum_exp(vector); harness oracle/ref_driver.cpp) on seeded inputs.  TEST INFRASTRUCTURE.

Run in the dev container only (needs the reference build):   python oracle/gen_golden_short.py
The fixture is data (inputs + the reference's outputs as hex doubles); no reference source travels with it.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
from longtr_amd import _abi  # noqa: E402

SEED = 20250225
hexd = lambda x: [float(v).hex() for v in np.asarray(x, dtype=np.float64).ravel()]


def cases(rng):
    """Stutter-block rows: homopolymers (the only blocks the CLI sends down this path), plus period 2-3 blocks, reads
    that carry the block with -3 .. +3 repeat units and a few substitutions, flat and mixed qualities, both alignment
    directions (left_align), a prev_row like a flank row (cumulative log-probabilities)."""
    out = []
    for it in range(60):
        period = 1 if it % 3 else int(rng.integers(2, 4))
        units = int(rng.integers(1, 26)) if it % 7 else 0
        motif = bytes(int(x) for x in rng.choice(list(b"ACGT"), size=period))
        block = (motif * units)[:units * period]
        if it % 11 == 5 and len(block) > 3:                       # an interrupted repeat
            b = bytearray(block); b[len(b) // 2] = ord("G") if b[len(b) // 2] != ord("G") else ord("T"); block = bytes(b)
        d = int(rng.integers(-3, 4)) * period
        core = (motif * (units + 8))[:max(len(block) + d, 0)]
        rb = lambda k: bytes(int(x) for x in rng.choice(list(b"ACGT"), size=k))
        lf, rf = rb(int(rng.integers(1, 12))), rb(int(rng.integers(0, 9)))
        seq = bytearray(lf + core + rf)
        for _ in range(int(rng.integers(0, 3))):
            if seq: seq[int(rng.integers(0, len(seq)))] = int(rng.choice(list(b"ACGT")))
        seq = bytes(seq) or b"A"
        if it % 4 == 0:
            qual = bytes([ord("I")] * len(seq))
        else:
            qual = bytes(int(q) for q in rng.integers(ord("!") - 1, ord("J") + 3, size=len(seq)))       # incl. out-of-range characters
        prev = np.cumsum(-rng.random(len(seq)) * 0.7) - (rng.random(len(seq)) < 0.05) * 9.0
        out.append(dict(block=block.decode(), period=period, left_align=int(it % 2), seq=seq.decode(), qual_hex=qual.hex(), prev_row=hexd(prev)))
    return out


def main():
    rng = np.random.default_rng(SEED)
    sps = [_abi.default_stutter_params(), _abi.StutterParams(0.9, 0.02, 0.08, 0.8, 0.003, 0.007)]
    doc = {"source": "oracle/_ref/libltr_ref.so (the reference's StutterAlignerClass.cpp, stutter_model.cpp, base_quality.h, mathops.cpp, "
                     "RepeatStutterInfo.h compiled from where they lie); harness oracle/ref_driver.cpp; generator oracle/gen_golden_short.py",
           "seed": SEED, "stutter_params": [[sp.in_geom, sp.in_up, sp.in_down, sp.out_geom, sp.out_up, sp.out_down] for sp in sps],
           "rows": [], "pmf": [], "artifact": [], "base_quality": [], "fast_lse": []}
    cs = cases(rng)
    for si, sp in enumerate(sps):
        pmf, art, bq, lse = ol.stutter_scalars("ref", sp)
        for c in cs:
            prev = np.array([float.fromhex(h) for h in c["prev_row"]])
            m, size, pos = ol.stutter_block_row("ref", sp, c["block"].encode(), c["period"], c["left_align"], c["seq"].encode(), bytes.fromhex(c["qual_hex"]), prev)
            doc["rows"].append(dict(c, sp=si, match=hexd(m), art_size=[int(x) for x in size], art_pos=[int(x) for x in pos]))
        for motif_len in (1, 2, 3, 5):
            for sample in (0, 4, 12, 30):
                for read in range(max(sample - 8 * motif_len, 0), sample + 8 * motif_len + 1, max(1, motif_len // 2)):
                    doc["pmf"].append([si, motif_len, sample, read, pmf(motif_len, sample, read).hex()])
        for period in (1, 2, 4):
            for allele in (0, 3, 10, 24):
                for d in range(-8 * period, 8 * period + 1):
                    doc["artifact"].append([si, period, allele, d, art(period, allele, d).hex()])
    _, _, bq, lse = ol.stutter_scalars("ref", sps[0])
    for q in range(0, 128):
        e, c = bq(q)
        doc["base_quality"].append([q, e.hex(), c.hex()])
    for it in range(200):
        n = int(rng.integers(1, 40))
        v = -rng.random(n) * (10.0 ** rng.integers(-2, 3)) - (rng.random(n) < 0.2) * 1e9 * (rng.random(n) < 0.3)
        doc["fast_lse"].append([hexd(v), lse(v).hex()])
    path = os.path.join(ROOT, "tests", "golden", "stutter_pieces.json")
    json.dump(doc, open(path, "w"), separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes;", len(doc["rows"]), "rows,", len(doc["pmf"]), "pmf,", len(doc["artifact"]), "artifact points")


if __name__ == "__main__":
    main()
