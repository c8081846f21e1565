#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REAL reference hot path (oracle/_ref/libltr_ref.so,
built by `make -C oracle ref` from /root/reference) on seeded inputs.  TEST INFRASTRUCTURE.

Run in the dev container only (needs the reference build):   python oracle/gen_golden.py
The fixtures are data (inputs + the reference's outputs as hex doubles); no reference source
travels with them.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
from longtr_amd import _abi, synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 20250225


def hexd(x):
    return [float(v).hex() for v in np.asarray(x, dtype=np.float64).ravel()]


def pdict(p):
    t = p.as_tuple()
    return {"values7_hex": [float(v).hex() for v in t[:7]], "indel_flank_len": t[7]}


def win_len(hap_len, F):
    pos = 35 - F
    cnt = hap_len - 2 * pos
    rest = hap_len - pos
    return rest if (cnt < 0 or cnt > rest) else cnt


def gen_align_long():
    rng = np.random.default_rng(SEED)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    groups = []

    def add(name, params, loci):
        """loci: list of (reads, haps).  Where m > n+1 the reference reads past the end of the
        haplotype string in its first-row loop (HapAligner.cpp:268): its output then depends on
        heap contents.  Such loci are kept only when the run reproduced the defined behaviour
        ("past the end never matches") bit for bit; `ub_pairs` counts the pairs concerned."""
        F = params.indel_flank_len
        loci = [(r, h) for r, h in loci if all(len(x) <= 60 or win_len(len(x), F) >= 1 for x in h)]
        b = _abi.PackedBatch(loci)
        ll, _ = ol.ref_align_batch(b, params)
        lo, _, _ = ol.oracle_align_batch(b, params)
        keep, ub_pairs = [], 0
        for li, (reads, haps) in enumerate(loci):
            ub = sum(1 for r in reads for h in haps if len(h) > 60 and len(r) > win_len(len(h), F) + 1)
            a = ll[b.ll_off[li]:b.ll_off[li + 1]]
            o = lo[b.ll_off[li]:b.ll_off[li + 1]]
            same = np.array_equal(a.view(np.uint64), o.view(np.uint64))
            assert ub > 0 or same, "restatement differs from the reference on a defined pair"
            if same:
                keep.append(li)
                ub_pairs += ub
        groups.append({"name": name, "params": pdict(params), "ub_pairs": ub_pairs, "dropped_loci": len(loci) - len(keep),
                       "loci": [{"reads": [r.decode() for r in loci[li][0]], "haps": [h.decode() for h in loci[li][1]]}
                                for li in keep],
                       "ll_hex": hexd(np.concatenate([ll[b.ll_off[li]:b.ll_off[li + 1]] for li in keep]))})
        print(f"  {name}: {len(keep)}/{len(loci)} loci kept, {ub_pairs} m>n+1 pairs among them")

    dflt = _abi.default_params()
    ont = _abi.make_params(synth.ONT_PARAMS)
    # SURVEY.md 8c known-answer locus
    lf = b"ACGTTGCAAGCTTAGGCTAACGTTAGCCATGGATC"; rf = b"GGATCCTTAGCAATCGGATTACAGGCTTAACCGTA"
    pl, pr, rep = b"TTGAC", b"CAGTT", b"CAG" * 20
    al = [pl + rep + pr, pl + rep + b"CAG" + pr, pl + rep[6:] + pr]
    add("survey_known_answer_cag", dflt, [([lf[-5:] + al[1] + rf[:5]], [lf + a + rf for a in al])])
    # synthetic loci, default params (HiFi-like) -- m<n, m>n, indels, all three strip classes
    loci = [synth.synth_locus(rng, int(tr), int(rng.integers(1, 7)), int(rng.integers(2, 7)), 5, sub_rate=0.01,
                              indel_rate=0.01) for tr in [1, 7, 20, 33, 64, 100, 150, 200, 236, 260]]
    add("synthetic_default", dflt, [(L.trimmed_reads, L.haplotypes) for L in loci])
    loci = [synth.synth_locus(rng, int(tr), 12, 2, 2, sub_rate=0.005, indel_rate=0.002) for tr in [493, 520, 1010]]
    add("synthetic_default_wide", dflt, [(L.trimmed_reads, L.haplotypes) for L in loci])
    loci = [synth.synth_locus(rng, int(tr), int(rng.integers(5, 40)), 4, 4, sub_rate=0.03, indel_rate=0.03)
            for tr in [60, 150, 300]]
    add("synthetic_ont_params", ont, [(L.trimmed_reads, L.haplotypes) for L in loci])
    # shortcuts and degenerate shapes
    cases = []
    for hl in [10, 60, 61, 62, 63, 75, 100]:
        for m in [1, 2, 3, 5, 16, 17, 40]:
            cases.append(([rs(m)], [rs(hl)]))
    for hl, m in [(760, 100), (761, 100), (700, 41), (1300, 650), (1261, 601), (1262, 601)]:
        cases.append(([rs(m)], [rs(hl)]))
    add("shortcuts_and_tiny", dflt, cases)
    # row abort (-700): nothing aligns
    add("row_abort", dflt, [([b"A" * 700], [b"C" * 800]), ([b"A" * 300], [b"C" * 900]), ([b"ACGT" * 100], [b"ACGT" * 130]),
                            ([b"AC" * 150], [b"GT" * 460])])
    for F in [0, 3, 10, 35]:
        pf = _abi.make_params(dflt.as_tuple()[:7], indel_flank_len=F)
        cs = [([rs(int(rng.integers(1, 90)))], [rs(int(rng.integers(71, 200)))]) for _ in range(30)]
        add(f"indel_flank_len_{F}", pf, cs)
    return {"provenance": "reference HapAligner::align_seq_to_hap (src/SeqAlignment/HapAligner.cpp:236-343) "
                          "compiled from /root/reference by oracle/Makefile; generator oracle/gen_golden.py",
            "groups": groups}


def gen_process_locus():
    rng = np.random.default_rng(SEED + 1)
    dflt = _abi.default_params()
    out = []
    specs = [(30, 3, 4, 10, 0.01, 0.01), (90, 5, 5, 10, 0.02, 0.03), (200, 4, 3, 8, 0.002, 0.001),
             (12, 2, 3, 8, 0.05, 0.08), (40, 1, 4, 8, 0.01, 0.02)]
    for tr, period, H, R, sub, ind in specs:
        L = synth.synth_locus(rng, tr, period, H, R, sub_rate=sub, indel_rate=ind, raw=True)
        alns = list(L.raw_alns)
        s1 = L.start + 35
        e1 = s1 + len(L.alleles[0])
        # hand-made reads: whole repeat block deleted (empty trim -> 10-bp substitute), soft clips,
        # 'M' ops, insertion inside the 5-bp flank, deletion spanning the flank boundary
        ref_full = L.ext_l + L.lflank + L.alleles[0] + L.rflank + L.ext_r
        a0 = L.start - synth.EXT_LEN
        left_len = synth.EXT_LEN + 35
        blk = len(L.alleles[0])
        right_len = 35 + synth.EXT_LEN
        alns.append(dict(start=a0, stop=a0 + len(ref_full) - 1, seq=ref_full[:left_len - 6] + ref_full[left_len + blk + 6:],
                         cigar=[("=", left_len - 6), ("D", blk + 12), ("=", right_len - 6)]))
        alns.append(dict(start=a0 + 10, stop=a0 + len(ref_full) - 1 - 7, seq=b"GGGG" + ref_full[10:len(ref_full) - 7] + b"TT",
                         cigar=[("S", 4), ("M", len(ref_full) - 17), ("S", 2)]))
        ins_at = left_len - 3
        alns.append(dict(start=a0, stop=a0 + len(ref_full) - 1, seq=ref_full[:ins_at] + b"ACG" + ref_full[ins_at:],
                         cigar=[("=", ins_at), ("I", 3), ("=", len(ref_full) - ins_at)]))
        del_at = left_len - 8
        alns.append(dict(start=a0, stop=a0 + len(ref_full) - 1, seq=ref_full[:del_at] + ref_full[del_at + 6:],
                         cigar=[("=", del_at), ("D", 6), ("=", len(ref_full) - del_at - 6)]))
        ll, toff, tlen = ol.ref_process_locus(L, dflt, alns)
        out.append({"start": L.start, "period": L.period, "lflank": L.lflank.decode(), "rflank": L.rflank.decode(),
                    "alleles": [a.decode() for a in L.alleles],
                    "alns": [{"start": a["start"], "stop": a["stop"], "seq": a["seq"].decode(),
                              "cigar": [[t, k] for t, k in a["cigar"]]} for a in alns],
                    "ll_hex": hexd(ll), "trim_len": [int(x) for x in tlen], "trim_off": [int(x) for x in toff]})
    return {"provenance": "reference HapAligner::trim_alignment (:346-465) + empty-trim substitute (:820-823) + "
                          "align_seq_to_hap per allele, via oracle/ref_driver.cpp ltr_ref_process_locus",
            "params": pdict(dflt), "loci": out}


def gen_pooling():
    rng = np.random.default_rng(SEED + 2)
    sets = []
    for _ in range(6):
        base = [synth._rand_seq(rng, int(rng.integers(5, 40))).tobytes() for _ in range(int(rng.integers(2, 8)))]
        reads = [base[int(rng.integers(0, len(base)))] for _ in range(int(rng.integers(5, 40)))]
        reads.insert(int(rng.integers(0, len(reads))), base[0][:-1])      # prefix of another read: distinct pool
        n, idx = ol.ref_pool_reads(reads)
        sets.append({"reads": [r.decode() for r in reads], "n_pools": int(n), "pool_index": [int(x) for x in idx]})
    return {"provenance": "reference ReadPooler::add_alignment (src/read_pooler.cpp:3-20)", "sets": sets}


def gen_posteriors():
    # SURVEY.md 8c known-answer point, produced during the survey by the reference's
    # Genotyper::calc_log_sample_posteriors (genotyper.cpp:45-83).  genotyper.cpp cannot be built
    # in this image (it includes fasta_reader.h -> htslib), so this is the ONLY reference output
    # pinning the posterior restatement: 10 printed digits of total LL and the argmax genotype.
    return {"provenance": "SURVEY.md section 8c known-answer point (survey-time run of the reference Genotyper); "
                          "posterior parity is otherwise UNPINNED by a reference build",
            "cases": [{"n_samples": 1, "haploid": 0,
                       "ll": [[-0.01, -12.9, -18.9], [-0.02, -12.9, -18.9], [-12.9, -0.01, -25.0], [-12.9, -0.03, -800.0]],
                       "log_p1": [0, 0, 0, 0], "log_p2": [0, 0, 0, 0], "sample_label": [0, 0, 0, 0],
                       "total_ll_10dp": "-4.6343380223", "gt": [0, 1]}]}


def gen_mathops():
    """The reference's own mathops.cpp helpers (compiled into oracle/_ref) that
    Genotyper::extract_genotypes_and_likelihoods (genotyper.cpp:132-256) is made of."""
    import ctypes as C
    R = ol.ref()
    rng = np.random.default_rng(SEED + 7)
    consts = np.zeros(3)
    R.ltr_ref_math_consts(consts.ctypes.data_as(C.c_void_p))
    pairs = []
    for _ in range(400):
        a = float(-rng.exponential(5.0)) if rng.random() < 0.8 else float(rng.normal(0, 50))
        gap = float(rng.choice([0.0, 1e-12, rng.exponential(0.5), rng.exponential(3.0), 6.9, 6.91, 40.0, 800.0]))
        b = a - gap if rng.random() < 0.5 else a + gap
        pairs.append((a, b))
    pairs += [(0.0, 0.0), (-1e-300, -1e-300), (-700.0, -0.1), (-0.1, -700.0), (3.5, 3.5)]
    two = [{"a": float(a).hex(), "b": float(b).hex(),
            "fast": float(R.ltr_ref_fast_log_sum_exp2(a, b)).hex(), "exact": float(R.ltr_ref_log_sum_exp2(a, b)).hex()}
           for a, b in pairs]
    streams = []
    for _ in range(60):
        n = int(rng.integers(1, 40))
        v = -rng.exponential(float(rng.choice([0.5, 5.0, 100.0])), size=n)
        if rng.random() < 0.3:
            v = np.sort(v)
        v = np.ascontiguousarray(v, dtype=np.float64)
        streams.append({"vals": hexd(v), "lse": float(R.ltr_ref_streaming_log_sum_exp(v.ctypes.data_as(C.c_void_p), n)).hex()})
    ints = [0, 1, 2, 3, 4, 5, 7, 10, 12, 13, 100, 1000, 999999]
    return {"provenance": "reference mathops.cpp (fast_log_sum_exp(double,double) :87-96, log_sum_exp(double,double) :55-60, "
                          "update/finish_streaming_log_sum_exp :70-85 from the genotyper's initial state, int_log :16-22) "
                          "compiled from /root/reference into oracle/_ref/libltr_ref.so",
            "consts": {"LOG_THRESH": float(consts[0]).hex(), "LOG_E_BASE_10": float(consts[1]).hex(), "TOLERANCE": float(consts[2]).hex()},
            "two": two, "streams": streams,
            "int_log": [{"v": v, "log": float(R.ltr_ref_int_log(v)).hex()} for v in ints]}


def main():
    assert ol.have_ref(), "build the reference harness first: make -C oracle ref"
    os.makedirs(OUT, exist_ok=True)
    for name, fn in [("align_long", gen_align_long), ("process_locus", gen_process_locus),
                     ("pooling", gen_pooling), ("posteriors", gen_posteriors), ("mathops", gen_mathops)]:
        d = fn()
        d["generator"] = "oracle/gen_golden.py"
        d["seed"] = SEED
        path = os.path.join(OUT, name + ".json")
        with open(path, "w") as f:
            json.dump(d, f, separators=(",", ":"))
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
