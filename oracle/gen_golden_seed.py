#!/usr/bin/env python3
"""Generate tests/golden/short_outer.json: the OUTER functions of the short (stutter) path that link against the COMPILED
REFERENCE without Haplotype.cpp -- HapAligner::compute_aln_logprob (HapAligner.cpp:165-233), calc_best_seed_position
(:467-493), calc_seed_base (:494-542) -- run through oracle/_ref/libltr_ref.so (harness oracle/ref_driver.cpp:
ltr_ref_compute_aln_logprob, ltr_ref_calc_best_seed_position, ltr_ref_calc_seed_base) on seeded inputs.  TEST INFRASTRUCTURE.

Run in the dev container only (needs the reference build):   python oracle/gen_golden_seed.py
The fixture is data (inputs + the reference's outputs; doubles as hex); the match matrices of compute_aln_logprob are named
by a seed (tests/short_util.py::lcg_matrix: exact integer arithmetic).  No reference source travels with it.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
import short_util as su  # noqa: E402

SEED = 20250226


def jblocks(blocks):
    return [dict(b, alleles=[a.decode() for a in b["alleles"]]) for b in blocks]


def main():
    rng = np.random.default_rng(SEED)
    doc = {"source": "oracle/_ref/libltr_ref.so (the reference's HapAligner.cpp, mathops.cpp compiled from where they lie; Haplotype / HapAligner "
                     "objects filled by hand, cur_size_ included); harness oracle/ref_driver.cpp; generator oracle/gen_golden_seed.py",
           "seed": SEED, "seed_base": [], "best_seed_position": [], "aln_logprob": []}
    for _ in range(400):
        blocks, aln = su.seed_case(rng)
        doc["seed_base"].append(dict(blocks=jblocks(blocks), start=aln["start"], stop=aln["stop"], seq_len=len(aln["seq"]),
                                     cigar="".join(f"{k}{t}" for t, k in aln["cigar"]), seed=ol.calc_seed_base("ref", aln, blocks)))
    for _ in range(600):
        nrep = int(rng.integers(0, 5))
        rs, re, p = [], [], int(rng.integers(0, 50))
        for _k in range(nrep):
            p += int(rng.integers(0, 20)); rs.append(p)
            p += int(rng.integers(0, 15)); re.append(p)
        a0 = int(rng.integers(-5, p + 30))
        a1 = a0 + int(rng.integers(-2, 60))
        d, q = ol.calc_best_seed_position("ref", rs, re, a0, a1)
        doc["best_seed_position"].append([rs, re, a0, a1, d, q])
    for it in range(300):
        c = su.logprob_case(rng, 1000 + 37 * it)
        lM, rM = su.logprob_matrices(c)
        v, mi = ol.compute_aln_logprob("ref", c["blocks"], c["counts"], c["base_seq_len"], c["seed_base"], c["seed_char"], c["log_seed_wrong"],
                                       c["log_seed_correct"], lM, c["l_prob"], rM, c["r_prob"])
        doc["aln_logprob"].append(dict(c, blocks=jblocks(c["blocks"]), log_seed_wrong=c["log_seed_wrong"].hex(), log_seed_correct=c["log_seed_correct"].hex(),
                                       l_prob=c["l_prob"].hex(), r_prob=c["r_prob"].hex(), total_LL=v.hex(), max_index=mi))
    path = os.path.join(ROOT, "tests", "golden", "short_outer.json")
    json.dump(doc, open(path, "w"), separators=(",", ":"))
    seeds = [c["seed"] for c in doc["seed_base"]]
    print(path, os.path.getsize(path), "bytes;", len(seeds), "seed cases,", sum(s >= 0 for s in seeds), "with a seed;",
          len(doc["best_seed_position"]), "seed positions;", len(doc["aln_logprob"]), "compute_aln_logprob cases")


if __name__ == "__main__":
    main()
