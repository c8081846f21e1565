"""-m gpu: the host mirror (process_reads) and the posterior consumer through the C-ABI, against
the golden vectors from the reference build and against the oracle."""
import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from longtr_amd import _abi, _lib, synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("gi", range(len(gu.load("align_long")["groups"])))
def test_align_long_golden_on_gpu(gpu_ctx, gi):
    g = gu.load("align_long")["groups"][gi]
    gpu_ctx.set_params(gu.params_from(g["params"]))
    try:
        ll, _ = gpu_ctx.align_batch(gu.group_batch(g))
    finally:
        gpu_ctx.set_params(_abi.default_params())
    assert np.array_equal(bits(ll), bits(gu.unhex(g["ll_hex"]))), g["name"]


def test_process_reads_golden_on_gpu(gpu_ctx):
    d = gu.load("process_locus")
    for L in d["loci"]:
        probs, seeds = gpu_ctx.process_reads(gu.locus_blocks(L), gu.locus_alns(L))
        assert np.array_equal(bits(probs.ravel()), bits(gu.unhex(L["ll_hex"])))
        assert list(seeds) == [len(a["seq"]) - 1 for a in L["alns"]]


def test_process_reads_masks_and_init_index(gpu_ctx):
    rng = np.random.default_rng(21)
    L = synth.synth_locus(rng, 80, 4, 5, 9, sub_rate=0.01, indel_rate=0.02, raw=True)
    rh = np.array([1, 0, 1, 1, 0], dtype=np.uint8)
    rr = rng.integers(0, 2, size=9).astype(np.uint8)
    rr[0] = 1
    got, gs = gpu_ctx.process_reads(L.blocks(), L.raw_alns, realign_hap=rh, realign_read=rr, init_read_index=2)
    rc, want, ws = ol.oracle_process_reads(gpu_ctx.params, L.blocks(), L.raw_alns, realign_hap=rh, realign_read=rr,
                                           init_read_index=2)
    assert rc == 0
    # untouched cells keep their initial NaN fill on both sides
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    assert np.array_equal(bits(got[m]), bits(want[m])) and np.array_equal(gs, ws)
    assert np.isnan(got[:2]).all()


def test_process_reads_multi_block_haplotype(gpu_ctx):
    # two variable blocks: haplotype k follows Haplotype::next()'s Gray order, not allele order
    rng = np.random.default_rng(22)
    L = synth.synth_locus(rng, 40, 3, 3, 6, raw=True)
    blocks = L.blocks()
    alt = bytearray(blocks[0]["alleles"][0])
    alt[10] = ord("A") if alt[10] != ord("A") else ord("C")
    blocks[0]["alleles"].append(bytes(alt))
    got, _ = gpu_ctx.process_reads(blocks, L.raw_alns)
    rc, want, _ = ol.oracle_process_reads(gpu_ctx.params, blocks, L.raw_alns)
    assert rc == 0 and got.shape == (6, 6) and np.array_equal(bits(got), bits(want))


def test_bad_cigar_is_an_error_code(gpu_ctx):
    rng = np.random.default_rng(23)
    L = synth.synth_locus(rng, 30, 3, 2, 2, raw=True)
    alns = [dict(a) for a in L.raw_alns]
    alns[1]["cigar"] = [("N", 5)] + list(alns[1]["cigar"])
    with pytest.raises(_lib.LtrError) as e:
        gpu_ctx.process_reads(L.blocks(), alns)
    assert e.value.code == _abi.LTR_ERR_CIGAR


def _short_params():
    return _abi.make_params(_abi.default_params().as_tuple()[:7], use_short_path=1)


def test_short_path_known_answer(gpu_ctx):
    import short_util as su
    blocks, alns = su.known_answer_case()
    gpu_ctx.set_params(_short_params())
    try:
        probs, seeds = gpu_ctx.process_reads(blocks, alns)
    finally:
        gpu_ctx.set_params(_abi.default_params())
    # the only reference outputs for this path
    assert [f"{x:.10f}" for x in probs[0]] == ["-7.8693081508", "-4.3896419406"] and seeds[0] == 77


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_short_path_matches_oracle_bit_for_bit(gpu_ctx, seed):
    import short_util as su
    rng = np.random.default_rng(seed)
    sp = _abi.default_stutter_params()
    gpu_ctx.set_params(_short_params())
    try:
        for tr, H, R in [(8, 2, 4), (14, 3, 6), (25, 4, 6), (40, 3, 5), (3, 2, 3)]:
            blocks, alns = su.homopolymer_locus(rng, tr, H, R)
            alns[0] = dict(alns[0], cigar=[("X", len(alns[0]["seq"]))])     # no seed -> all-zero row
            rr = np.ones(R, dtype=np.uint8)
            rr[-1] = 0
            rh = np.ones(H, dtype=np.uint8)
            if H > 2:
                rh[1] = 0
            got, gs = gpu_ctx.process_reads(blocks, alns, realign_hap=rh, realign_read=rr, init_read_index=1)
            rc, want, ws = ol.oracle_process_reads_short(_short_params(), sp, blocks, alns, realign_hap=rh, realign_read=rr,
                                                         init_read_index=1)
            assert rc == 0
            assert np.array_equal(np.isnan(got), np.isnan(want))
            m = ~np.isnan(want)
            assert np.array_equal(bits(got[m]), bits(want[m])), (tr, H, R)
            assert np.array_equal(gs, ws)
            assert (want[1] == 0).all()
    finally:
        gpu_ctx.set_params(_abi.default_params())


def test_short_path_only_for_period_one(gpu_ctx):
    # --stutter-align-len set but period 3: the reference takes the LONG path (HapAligner.cpp:552)
    rng = np.random.default_rng(24)
    L = synth.synth_locus(rng, 30, 3, 3, 3, raw=True)
    gpu_ctx.set_params(_short_params())
    try:
        got, _ = gpu_ctx.process_reads(L.blocks(), L.raw_alns)
    finally:
        gpu_ctx.set_params(_abi.default_params())
    rc, want, _ = ol.oracle_process_reads(_abi.default_params(), L.blocks(), L.raw_alns)
    assert rc == 0 and np.array_equal(bits(got), bits(want))


def test_short_path_other_stutter_model(gpu_ctx):
    import short_util as su
    rng = np.random.default_rng(35)
    blocks, alns = su.homopolymer_locus(rng, 18, 3, 5)
    sp = _abi.StutterParams(0.8, 0.1, 0.07, 0.9, 0.02, 0.03)
    gpu_ctx.set_params(_short_params())
    gpu_ctx.set_stutter_params(sp)
    gpu_ctx.timers(reset=True)
    try:
        got, _ = gpu_ctx.process_reads(blocks, alns)
        tm = gpu_ctx.timers(reset=True)
        assert tm["short_kernel_ms"] > 0 and tm["dp_kernel_ms"] == 0           # the seeded path's own kernels, not the long path's
    finally:
        gpu_ctx.set_params(_abi.default_params())
        gpu_ctx.set_stutter_params(_abi.default_stutter_params())
    rc, want, _ = ol.oracle_process_reads_short(_short_params(), sp, blocks, alns)
    assert rc == 0 and np.array_equal(bits(got), bits(want))


def test_posteriors_known_answer_and_oracle(gpu_ctx):
    c = gu.load("posteriors")["cases"][0]
    r = gpu_ctx.posteriors(np.asarray(c["ll"]), c["log_p1"], c["log_p2"], c["sample_label"], c["n_samples"])
    assert f"{r['total_ll']:.10f}" == c["total_ll_10dp"] and list(r["gts"][0]) == c["gt"]
    assert r["clamped_ll"][3, 2] == -600.0
    rng = np.random.default_rng(25)
    for haploid in (False, True):
        S, R, H = 3, 40, 7
        ll = -rng.random((R, H)) * 30
        ll[rng.random((R, H)) < 0.05] = -700.0
        lab = rng.integers(0, S, size=R)
        hp = rng.integers(0, 3, size=R)            # none / hap1 / hap2 (snp_bam_processor.h:17-18)
        p1 = np.where(hp == 1, -1e-6, np.where(hp == 2, -1000.0, 0.0))
        p2 = np.where(hp == 2, -1e-6, np.where(hp == 1, -1000.0, 0.0))
        g = gpu_ctx.posteriors(ll, p1, p2, lab, S, haploid=haploid)
        o = ol.oracle_posteriors(ll, p1, p2, lab, S, haploid=haploid)
        # north star: posteriors within 1e-4 (device exp/log differ from glibc in the last ulps); GT identical
        assert np.allclose(g["post"], o["post"], rtol=0, atol=1e-9)
        assert np.allclose(g["sample_total_ll"], o["sample_total_ll"], rtol=0, atol=1e-9)
        assert np.array_equal(g["gts"], o["gts"])
        assert np.array_equal(g["clamped_ll"], o["clamped_ll"])


def test_genotype_fields_end_to_end(gpu_ctx):
    """DP -> posteriors on the GPU -> ltr_extract_genotypes, against the all-CPU restatement chain:
    identical GT / PL-argmax, Q = exp(log_unphased) and PQ within 1e-4 (north-star tolerance)."""
    rng = np.random.default_rng(77)
    for haploid in (False, True):
        L = synth.synth_locus(rng, 90, 3, 5, 24, sub_rate=0.002, indel_rate=0.001)
        batch, pidx = synth.pack_loci([L])
        ll, _ = gpu_ctx.align_batch(batch)
        M = batch.locus_matrix(ll, 0)[np.asarray(pidx[0])]                  # pool rows -> read rows
        R, H = M.shape
        S = 2
        lab = rng.integers(0, S, size=R)
        z = np.zeros(R)
        g = gpu_ctx.posteriors(M.copy(), z, z, lab, S, haploid=haploid)
        o = ol.oracle_posteriors(M.copy(), z, z, lab, S, haploid=haploid)
        assert np.array_equal(g["gts"], o["gts"])
        h2a = np.arange(H, dtype=np.int32)                                  # one multi-allele block: hap k == allele k
        gf = _lib.extract_genotypes(g["post"], g["sample_total_ll"], g["gts"], h2a, H, haploid)
        of = ol.oracle_extract_genotypes(o["post"], o["sample_total_ll"], o["gts"], h2a, H, haploid)
        assert np.array_equal(gf["best_gts"], of["best_gts"])
        for k in ("log_phased_posteriors", "log_unphased_posteriors", "hap_log_phased_posteriors",
                  "hap_log_unphased_posteriors"):
            assert np.allclose(np.exp(gf[k]), np.exp(of[k]), rtol=0, atol=1e-4), k      # Q / PQ
        # fast_log_sum_exp is a float approximation with kinks: equal inputs up to 1e-9 can land
        # on different float mantissas, so GLs are compared at their printed scale
        assert np.allclose(gf["gls"], of["gls"], rtol=0, atol=1e-3)
        assert np.array_equal(np.argmin(gf["pls"], axis=1), np.argmin(of["pls"], axis=1))
        assert np.all(np.abs(gf["pls"] - of["pls"]) <= 1)


def test_posterior_ties_resolve_like_reference(gpu_ctx):
    # unphased reads: post[a][b] == post[b][a] exactly, so argmax must pick the first (a<b) like
    # the reference's strict '>' scan (genotyper.cpp:91-96)
    ll = np.array([[-0.01, -12.0, -30.0], [-12.0, -0.01, -30.0]] * 5)
    z = np.zeros(10)
    g = gpu_ctx.posteriors(ll, z, z, np.zeros(10, dtype=np.int32), 1)
    assert np.array_equal(bits(g["post"][0]), bits(g["post"][0].T))
    assert list(g["gts"][0]) == [0, 1]


def test_plan_posteriors_all_loci_on_device(gpu_ctx):
    """Batch path end to end: pooled DP for many loci, then posteriors straight from the resident LL
    matrix (pool rows fanned out to reads on the device), against scatter + per-locus oracle."""
    rng = np.random.default_rng(41)
    loci = [synth.synth_locus(rng, int(rng.integers(10, 160)), int(rng.integers(1, 7)), int(rng.integers(2, 7)), 12,
                              sub_rate=0.002, indel_rate=0.001) for _ in range(25)]
    batch, pidx = synth.pack_loci(loci)
    plan = gpu_ctx.plan(batch)
    plan.execute()
    ll, _ = plan.fetch()
    lro, pool_index, p1, p2, lab, ns = [0], [], [], [], [], []
    for l, L in enumerate(loci):
        R = len(L.trimmed_reads)
        S = int(rng.integers(1, 4))
        hp = rng.integers(0, 3, size=R)
        pool_index += list(pidx[l])
        p1 += list(np.where(hp == 1, -1e-6, np.where(hp == 2, -1000.0, 0.0)))
        p2 += list(np.where(hp == 2, -1e-6, np.where(hp == 1, -1000.0, 0.0)))
        lab += list(rng.integers(0, S, size=R))
        ns.append(S)
        lro.append(lro[-1] + R)
    post, off, stl, gts = plan.posteriors(lro, pool_index, p1, p2, lab, ns)
    u = 0
    for l, L in enumerate(loci):
        H = len(L.haplotypes)
        r0, r1 = lro[l], lro[l + 1]
        M = batch.locus_matrix(ll, l)
        per_read = M[np.asarray(pool_index[r0:r1])]                       # the scatter of seq_stutter_genotyper.cpp:526-538
        o = ol.oracle_posteriors(per_read, p1[r0:r1], p2[r0:r1], lab[r0:r1], ns[l])
        for s in range(ns[l]):
            got = post[off[u]:off[u + 1]].reshape(H, H)
            assert np.allclose(got, o["post"][s], rtol=0, atol=1e-9)
            assert abs(stl[u] - o["sample_total_ll"][s]) < 1e-9
            assert list(gts[u]) == list(o["gts"][s])
            u += 1
    assert u == len(stl)
    plan.close()


def _expected_calc_hap_aln_probs(params, stutter, blocks, alns, second_mate=None):
    """pool -> (oracle) process_reads on the pools -> scatter, the way SeqStutterGenotyper does it."""
    pools, idx = synth.pool_reads([a["seq"] for a in alns])
    first = [idx.index(q) for q in range(len(pools))]
    pooled = []
    for q, f in enumerate(first):
        a = dict(alns[f])
        if a.get("qual") is not None:                                   # ReadPooler::pool: per-position upper median
            members = np.array([np.frombuffer(alns[i]["qual"], dtype=np.int8) for i in range(len(alns)) if idx[i] == q])
            a["qual"] = np.sort(members, axis=0)[members.shape[0] // 2].astype(np.int8).tobytes()
        pooled.append(a)
    short = params.use_short_path and blocks[1]["period"] == 1
    if short:
        rc, pp, ps = ol.oracle_process_reads_short(params, stutter, blocks, pooled)
    else:
        rc, pp, ps = ol.oracle_process_reads(params, blocks, pooled)
    assert rc == 0
    H = pp.shape[1]
    out = np.full(len(alns) * H, np.nan)
    seeds = np.full(len(alns), -12345, dtype=np.int32)
    import ctypes as C
    p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    sm = None if second_mate is None else np.ascontiguousarray(second_mate, dtype=np.uint8)
    idxa = np.asarray(idx, dtype=np.int32)
    rc = ol.oracle().ltr_oracle_scatter_pool_probs(p(np.ascontiguousarray(pp)), p(ps), p(idxa), len(alns), H, None, None, p(sm),
                                                   p(out), p(seeds))
    assert rc == 0
    return out.reshape(len(alns), H), seeds


def test_calc_hap_aln_probs_many_loci(gpu_ctx):
    import short_util as su
    rng = np.random.default_rng(51)
    prm = _short_params()
    sp = _abi.default_stutter_params()
    loci = []
    for k in range(12):
        if k % 4 == 3:
            blocks, alns = su.homopolymer_locus(rng, int(rng.integers(6, 30)), 3, 8, sub_rate=0.0, indel_rate=0.0)
            alns[3] = dict(alns[3], qual=bytes(reversed(alns[3]["qual"])))
        else:
            L = synth.synth_locus(rng, int(rng.integers(10, 400)), int(rng.integers(2, 7)), int(rng.integers(2, 6)), 10,
                                  sub_rate=0.001, indel_rate=0.0005, raw=True)
            blocks, alns = L.blocks(), L.raw_alns
        sm = None
        if k == 1:                                                     # a mate pair: rows i-1 and i are summed
            sm = np.zeros(len(alns), dtype=np.uint8)
            sm[4] = 1
        loci.append((blocks, alns, sm))
    gpu_ctx.set_params(prm)
    try:
        got = gpu_ctx.calc_hap_aln_probs(loci)
    finally:
        gpu_ctx.set_params(_abi.default_params())
    pooled_some = False
    for (blocks, alns, sm), (probs, seeds) in zip(loci, got):
        want, ws = _expected_calc_hap_aln_probs(prm, sp, blocks, alns, sm)
        assert np.array_equal(bits(probs), bits(want)) and np.array_equal(seeds, ws)
        pooled_some |= len(set(a["seq"] for a in alns)) < len(alns)
    assert pooled_some


def test_calc_hap_aln_probs_chunked_pipeline(gpu_ctx):
    """>= 1500 long-path loci: the call scores them in two chunks, 1 : 3 (chunk 1 is prepared on the host cores
    and its plan built while the GPU scores chunk 0); every locus must still equal the one-locus
    path, and an error in one locus must surface as that locus' error.  Then the same loci in five chunks
    on three streams (the per-call override the tuning sweep uses)."""
    rng = np.random.default_rng(52)
    prm = _abi.default_params()
    sp = _abi.default_stutter_params()
    loci = []
    for k in range(1600):
        L = synth.synth_locus(rng, int(rng.integers(5, 60)), int(rng.integers(2, 5)), int(rng.integers(2, 4)), 5,
                              sub_rate=0.002, indel_rate=0.001, raw=True)
        loci.append((L.blocks(), L.raw_alns, None))
    got = gpu_ctx.calc_hap_aln_probs(loci)
    probe = list(range(0, 1600, 41)) + [105, 106, 107, 319, 320, 321, 398, 399, 400, 401, 639, 640, 641, 1066, 1067, 1599]   # incl. the seam (400)
    expect = {}
    for idx in probe:
        blocks, alns, sm = loci[idx]
        expect[idx] = _expected_calc_hap_aln_probs(prm, sp, blocks, alns, sm)
        want, ws = expect[idx]
        assert np.array_equal(bits(got[idx][0]), bits(want)) and np.array_equal(got[idx][1], ws), idx
    gpu_ctx.set_debug("chunks", 5); gpu_ctx.set_debug("chunk_streams", 3); gpu_ctx.set_debug("chunk_growth", 0)    # seams at 106, 320, 640, 1066
    try:
        got5 = gpu_ctx.calc_hap_aln_probs(loci)
    finally:
        gpu_ctx.set_debug("reset", 0)
    for idx in probe:
        want, ws = expect[idx]
        assert np.array_equal(bits(got5[idx][0]), bits(want)) and np.array_equal(got5[idx][1], ws), idx
    # the same call with chunk c + 1 staged only after chunk c is launched (round 4's order; since round 5 a thread of its own
    # stages it ahead): every row of every locus the same
    gpu_ctx.set_debug("prep_ahead", -1)
    try:
        got_serial = gpu_ctx.calc_hap_aln_probs(loci)
    finally:
        gpu_ctx.set_debug("reset", 0)
    for idx in range(len(loci)):
        assert np.array_equal(bits(got_serial[idx][0]), bits(got[idx][0])) and np.array_equal(got_serial[idx][1], got[idx][1]), idx
    bad = list(loci)
    blocks, alns, _ = bad[700]
    alns = [dict(a) for a in alns]
    alns[1]["cigar"] = [("Q", len(alns[1]["seq"]))]                    # invalid CIGAR operation in one read of one locus
    bad[700] = (blocks, alns, None)
    with pytest.raises(_lib.LtrError) as e:
        gpu_ctx.calc_hap_aln_probs(bad)
    assert e.value.code == -5 and "CIGAR" in str(e.value)


def test_calc_hap_aln_probs_chunks_with_short_path_loci(gpu_ctx):
    """Seeded-path (period-1) loci scattered over the chunks of one call, the chunks staged ahead by the helper thread: the
    short-path loci of a chunk are strung onto the call's batch by the calling thread, in locus order; everything equals the
    oracle and the call without the helper thread; a bad alignment record in a late chunk is that call's error."""
    import short_util as su
    rng = np.random.default_rng(53)
    prm = _short_params()
    sp = _abi.default_stutter_params()
    loci = []
    for k in range(1700):
        if k % 211 == 5:
            blocks, alns = su.homopolymer_locus(rng, int(rng.integers(6, 24)), 2, 6, sub_rate=0.0, indel_rate=0.0)
            loci.append((blocks, alns, None))
        else:
            L = synth.synth_locus(rng, int(rng.integers(5, 50)), int(rng.integers(2, 5)), int(rng.integers(2, 4)), 5,
                                  sub_rate=0.002, indel_rate=0.001, raw=True)
            loci.append((L.blocks(), L.raw_alns, None))
    gpu_ctx.set_params(prm)
    try:
        gpu_ctx.set_debug("chunks", 6); gpu_ctx.set_debug("chunk_growth", 1.3)
        got = gpu_ctx.calc_hap_aln_probs(loci)
        gpu_ctx.set_debug("prep_ahead", -1)
        got_serial = gpu_ctx.calc_hap_aln_probs(loci)
        for idx in range(len(loci)):
            assert np.array_equal(bits(got_serial[idx][0]), bits(got[idx][0])) and np.array_equal(got_serial[idx][1], got[idx][1]), idx
        for idx in [5, 216, 427, 638, 1693, 0, 4, 6, 900, 1699]:
            blocks, alns, sm = loci[idx]
            want, ws = _expected_calc_hap_aln_probs(prm, sp, blocks, alns, sm)
            assert np.array_equal(bits(got[idx][0]), bits(want)) and np.array_equal(got[idx][1], ws), idx
        gpu_ctx.set_debug("prep_ahead", 0)
        bad = list(loci)
        blocks, alns, _ = bad[1500]
        alns = [dict(a) for a in alns]
        alns[0]["cigar"] = [("Q", len(alns[0]["seq"]))]
        bad[1500] = (blocks, alns, None)
        with pytest.raises(_lib.LtrError) as e:
            gpu_ctx.calc_hap_aln_probs(bad)
        assert e.value.code == -5 and "CIGAR" in str(e.value)
    finally:
        gpu_ctx.set_debug("reset", 0)
        gpu_ctx.set_params(_abi.default_params())


def test_compact_plans_one_block_one_upload_scores_in_pinned_memory(gpu_ctx):
    """A small plan (BASELINE config 2: one locus, 224 pairs -- what HapAligner::process_reads is called with, once per locus,
    seq_stutter_genotyper.cpp:517-523) is ONE device block filled by ONE upload, its control words arrive with it, its scores are
    written into pinned host memory (round 6).  Same bits as the separate allocations / copies / fills of the large plans
    (compact_plan = -1), over repeated executes (the first one skips the control-word reset, the next ones must not), into the
    plan's own buffer and into a caller's device buffer, with masks, and for plans created back to back (the one pinned image)."""
    import torch
    loci, _ = synth.config_loci("config2")
    small, _ = synth.pack_loci(loci)
    ref, _, _ = ol.oracle_align_batch(small, gpu_ctx.params)
    rng = np.random.default_rng(55)
    others = []
    for k in range(6):
        L = synth.synth_locus(rng, int(rng.integers(8, 300)), int(rng.integers(3, 12)), int(rng.integers(2, 6)), 5, sub_rate=0.004, indel_rate=0.002)
        others.append(synth.pack_loci([L])[0])
    other_ref = [ol.oracle_align_batch(b, gpu_ctx.params)[0] for b in others]
    try:
        for knob in (0, -1):
            gpu_ctx.set_debug("compact_plan", knob)
            plan = gpu_ctx.plan(small)
            for _ in range(3):
                plan.execute()
                ll, _ = plan.fetch()
                assert np.array_equal(bits(ll), bits(ref)), knob
            # ... into a caller's device buffer (a torch tensor: on some boxes torch's own HIP runtime does not come up once this
            # library's has -- "No HIP GPUs are available" --; bench.py, which initialises torch first, drives this path in every run)
            try:
                out = torch.full((small.ll_size,), float("nan"), dtype=torch.float64, device="cuda:0")
            except RuntimeError:
                out = None
            if out is not None:
                plan.execute(out.data_ptr(), torch.cuda.current_stream().cuda_stream)
                plan.wait()
                assert np.array_equal(bits(out.cpu().numpy()), bits(ref)), knob
            plan.execute()
            ll, _ = plan.fetch()
            assert np.array_equal(bits(ll), bits(ref)), knob
            # plans created back to back, alive together: each keeps its own block and its own pinned scores
            plans = [gpu_ctx.plan(b) for b in others]
            for p in plans:
                p.execute()
            for p, want in zip(plans, other_ref):
                got, _ = p.fetch()
                assert np.array_equal(bits(got), bits(want)), knob
            for p in plans:
                p.close()
            plan.execute()
            ll, _ = plan.fetch()
            assert np.array_equal(bits(ll), bits(ref)), knob
            plan.close()
            for b, want in zip(others, other_ref):           # the host-in / host-out call, per locus, as the adapter issues it
                got, _ = gpu_ctx.align_batch(b)
                assert np.array_equal(bits(got), bits(want)), knob
            # every pair pre-seeded into an exact list (mode 4): the list heads arrive with the plan's ONE upload, on the upload
            # stream -- the execute must wait for it before it copies them (a race the round-6 fuzz found: rows of zeros)
            gpu_ctx.set_pair_packing(4)
            try:
                for rep in range(40):
                    for b, want in zip(others, other_ref):
                        got, _ = gpu_ctx.align_batch(b)
                        assert np.array_equal(bits(got), bits(want)), (knob, rep)
            finally:
                gpu_ctx.set_pair_packing(-1)
    finally:
        gpu_ctx.set_debug("reset", 0)


def test_calc_hap_aln_probs_under_host_thread_budgets(gpu_ctx):
    """ltr_ctx_set_host_threads (round 6; the reference: one thread per process, N processes per node, README.md:78-82): the raw-
    alignment call with a budget of 2, 4 and 16 host threads -- bit-identical rows, the helper thread that stages the next chunk
    only from 12 threads up, and the rule's choice never slower than the call with the helper off (what eight ranks on one host
    would otherwise run into: two thread teams on four cores, profiles/r05/prep_ahead_ab.log)."""
    import time
    rng = np.random.default_rng(54)
    loci = []
    for k in range(6000):
        L = synth.synth_locus(rng, int(rng.integers(5, 40)), int(rng.integers(2, 5)), int(rng.integers(2, 4)), 5,
                              sub_rate=0.002, indel_rate=0.001, raw=True)
        loci.append((L.blocks(), L.raw_alns))
    packed = gpu_ctx.pack_loci(loci)
    held = gpu_ctx.host_threads()
    assert 1 <= held <= 16 and held == _lib.lib().ltr_host_threads_rule(0)

    def call(reps=3):
        best, out = 1e9, None
        for _ in range(reps):
            t0 = time.perf_counter()
            out = gpu_ctx.calc_hap_aln_probs_packed(packed)
            best = min(best, time.perf_counter() - t0)
        return out, best

    def flat(res):
        return np.concatenate([np.asarray(m, dtype=np.float64).ravel() for m, _ in res]), np.concatenate([np.asarray(sd).ravel() for _, sd in res])

    try:
        ref, _ = call(1)
        ref_ll, ref_seed = flat(ref)
        times = {}
        for budget in (2, 4, 16):
            gpu_ctx.set_host_threads(budget)
            assert gpu_ctx.host_threads() == budget
            assert _lib.lib().ltr_debug_prep_ahead_rule(0) == (1 if budget >= 12 else 0)
            got, t_rule = call()
            ll, seed = flat(got)
            assert np.array_equal(bits(ll), bits(ref_ll)) and np.array_equal(seed, ref_seed), budget
            gpu_ctx.set_debug("prep_ahead", -1)                     # the helper thread off whatever the budget
            got, t_serial = call()
            gpu_ctx.set_debug("prep_ahead", 0)
            ll, seed = flat(got)
            assert np.array_equal(bits(ll), bits(ref_ll)) and np.array_equal(seed, ref_seed), budget
            times[budget] = (t_rule, t_serial)
            # never slower than serial: at budgets below 12 the rule IS the serial path (same code, ratio 1 up to noise), at 16 the
            # helper thread wins (profiles/r06: 235 against 252 ms); the margin is for a shared host's noise, best of three each
            assert t_rule <= 1.6 * t_serial, (budget, times)
        print("host-thread budgets (rule, helper off) seconds per call:", times)
    finally:
        gpu_ctx.set_debug("reset", 0)
        gpu_ctx.set_host_threads(0)
    assert gpu_ctx.host_threads() == held


def test_plans_survive_their_context_and_buffers_are_recycled():
    """Handles stay valid in any destroy order (a plan whose context is gone reports an error and
    can still be destroyed), and a context's device buffers are reused across per-locus calls."""
    ctx = _lib.Context(0)
    loci, _ = synth.config_loci("config2")
    batch, _ = synth.pack_loci(loci)
    ref, _ = ctx.align_batch(batch)
    for _ in range(30):                                       # same sizes every time: served from the pool
        ll, _ = ctx.align_batch(batch)
        assert np.array_equal(bits(ll), bits(ref))
    plan = ctx.plan(batch)
    plan.execute()
    ctx.close()                                               # context first ...
    with pytest.raises(_lib.LtrError):
        plan.execute()
    plan.close()                                              # ... then the plan: legal


def test_calc_hap_aln_probs_three_argument_form_and_timers(gpu_ctx):
    """calc_hap_aln_probs(realign_to_haplotype, realign_pool, copy_read) (seq_stutter_genotyper.cpp:514-563),
    the form add_and_remove_alleles uses after new haplotypes were added (:390-391): only flagged columns /
    pools are scored, only flagged reads' rows are rewritten, mate rows are summed over flagged columns
    only; every other cell keeps what the caller's matrix held.  Plus the reference's clocks."""
    import ctypes as C
    rng = np.random.default_rng(61)
    prm = gpu_ctx.params
    loci, inits, exps = [], [], []
    p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    for k in range(9):
        L = synth.synth_locus(rng, int(rng.integers(20, 300)), int(rng.integers(2, 7)), int(rng.integers(3, 7)), 12,
                              sub_rate=0.002, indel_rate=0.001, raw=True)
        blocks, alns = L.blocks(), L.raw_alns
        H, R = len(L.alleles), len(alns)
        pools, idx = synth.pool_reads([a["seq"] for a in alns])
        first = [idx.index(q) for q in range(len(pools))]
        rh = rng.integers(0, 2, size=H).astype(np.uint8)
        rh[int(rng.integers(0, H))] = 1
        # (k == 7 passes copy_read = NULL = "every read": then every pool has to be realigned, or the reference
        # copies rows of its pool matrix it never wrote)
        rp = rng.integers(0, 2, size=len(pools)).astype(np.uint8) if (k % 3 and k != 7) else None
        cr = rng.integers(0, 2, size=R).astype(np.uint8) if k % 2 else np.ones(R, dtype=np.uint8)
        if rp is not None:
            cr = cr & rp[np.asarray(idx)]                              # a copied read's pool was realigned (else the reference copies an unset row)
        sm = None
        if k in (1, 4):
            sm = np.zeros(R, dtype=np.uint8)
            sm[5] = 1
        masks = dict(realign_to_hap=rh if k != 8 else None, realign_pool=rp, copy_read=cr if k != 7 else None)
        init = rng.normal(size=(R, H)) - 50.0
        rc, pp, ps = ol.oracle_process_reads(prm, blocks, [alns[f] for f in first])
        assert rc == 0
        want = init.copy().ravel()
        seeds = np.full(R, -12345, dtype=np.int32)
        rc = ol.oracle().ltr_oracle_scatter_pool_probs(p(np.ascontiguousarray(pp)), p(ps), p(np.asarray(idx, dtype=np.int32)), R, H,
                                                       p(masks["realign_to_hap"]), p(masks["copy_read"]), p(sm), p(want), p(seeds))
        assert rc == 0
        loci.append((blocks, alns, sm, masks))
        inits.append(init)
        exps.append((want.reshape(R, H), seeds, masks, idx))
    gpu_ctx.timers(reset=True)
    packed = gpu_ctx.pack_loci(loci, out_init=inits)
    got = gpu_ctx.calc_hap_aln_probs_packed(packed)
    for (probs, seeds), (want, ws, masks, idx) in zip(got, exps):
        assert np.array_equal(bits(probs), bits(want))
        cr = masks["copy_read"]
        keep = np.ones(len(ws), dtype=bool) if cr is None else cr.astype(bool)
        assert np.array_equal(seeds[keep], ws[keep]) and (seeds[~keep] == -12345).all()
    tm = gpu_ctx.timers()
    assert tm["hap_aln_calls"] == 1 and tm["hap_aln_s"] > 0 and tm["dp_kernel_ms"] > 0 and tm["posterior_calls"] == 0
    ll = np.asarray(got[0][0], dtype=np.float64)
    gpu_ctx.posteriors(np.nan_to_num(ll, nan=-1.0), np.zeros(ll.shape[0]), np.zeros(ll.shape[0]), np.zeros(ll.shape[0], dtype=np.int32), 1)
    tm = gpu_ctx.timers(reset=True)
    assert tm["posterior_calls"] == 1 and tm["posterior_s"] > 0
    assert gpu_ctx.timers()["hap_aln_calls"] == 0
