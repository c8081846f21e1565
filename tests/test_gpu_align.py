"""-m gpu: the HIP alignment path (through the C-ABI) against the oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib as ol
from longtr_amd import _abi, synth

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


def _check(ctx, batch, params=None):
    """GPU == oracle bit for bit under every scheduling mode: automatic (-1: under-filled classes folded into the next
    wider one, launches on two streams from 16 pairs per CU on), one pair per wavefront (0), two pairs per
    wavefront wherever the read fits (1; small test batches would never take the packed kernels under
    the default size rule), the latency variant of the workgroup kernels for every short read (2), and
    no workgroup kernels at all (3: long reads walk their column blocks on one wavefront; in modes
    0-2 reads over 1025 bases go to the 4- / 8-wave workgroup kernels), and no certificate kernels at
    all (4: every pair straight to the exact kernel of its length class -- the reference's cell-by-cell
    row maximum), and the packed kernels with 16 / 8 / 4 / 2 lanes per pair wherever the read fits (5 .. 8)."""
    if params is not None:
        ctx.set_params(params)
    try:
        ref, rseed, _ = ol.oracle_align_batch(batch, ctx.params)
        for mode in (-1, 0, 1, 2, 3, 4, 5, 6, 7, 8):
            ctx.set_pair_packing(mode)
            ll, seed = ctx.align_batch(batch)
            bad = np.where(ll.view(np.uint64) != ref.view(np.uint64))[0]
            assert bad.size == 0, (f"packing {mode}: {bad.size}/{ll.size} differ; first {bad[:5]} gpu {ll[bad[:5]]} "
                                   f"oracle {ref[bad[:5]]}")
            assert np.array_equal(seed, rseed)
    finally:
        ctx.set_pair_packing(-1)
        if params is not None:
            ctx.set_params(_abi.default_params())
    return ll


def test_known_answer_cag(gpu_ctx):
    # SURVEY.md 8c known-answer point (reference output): read == allele 1 of a 3-allele CAG locus
    lf = b"ACGTTGCAAGCTTAGGCTAACGTTAGCCATGGATC"; rf = b"GGATCCTTAGCAATCGGATTACAGGCTTAACCGTA"
    pl, pr, rep = b"TTGAC", b"CAGTT", b"CAG" * 20
    al = [pl + rep + pr, pl + rep + b"CAG" + pr, pl + rep[6:] + pr]
    read = lf[-5:] + al[1] + rf[:5]
    b = _abi.PackedBatch([([read], [lf + a + rf for a in al])])
    ll = _check(gpu_ctx, b)
    assert np.allclose(ll, [-12.9194140592, -0.0130565529, -18.9184660191], atol=1e-9)


def test_config2_bit_exact(gpu_ctx):
    loci, _ = synth.config_loci("config2")
    batch, _ = synth.pack_loci(loci)
    _check(gpu_ctx, batch)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_small_loci(gpu_ctx, seed):
    rng = np.random.default_rng(seed)
    loci = [synth.synth_locus(rng, int(rng.integers(1, 300)), int(rng.integers(1, 7)), int(rng.integers(2, 9)), 6,
                              sub_rate=0.01, indel_rate=0.01) for _ in range(40)]
    batch, _ = synth.pack_loci(loci)
    _check(gpu_ctx, batch)


def test_all_strip_widths_and_column_blocks(gpu_ctx):
    # read lengths around every strip-width edge (C = 64*W, W = 1..8), then 2, 3 and 5 column blocks
    rng = np.random.default_rng(5)
    loci = []
    for tr in [1, 20, 43, 44, 45, 46, 107, 108, 109, 110, 171, 172, 173, 174, 235, 236, 237, 238, 299, 300, 301, 302,
               363, 364, 365, 366, 427, 428, 429, 430, 491, 492, 493, 494, 640, 1003, 1004, 1005, 1006, 1300, 1517, 2100]:
        loci.append(synth.synth_locus(rng, tr, 12, 3, 3, sub_rate=0.01, indel_rate=0.005))
    batch, _ = synth.pack_loci(loci)
    _check(gpu_ctx, batch)


def test_exact_kernels_strip_width_edges(gpu_ctx):
    """The LUT exact kernels pick their strip width per pair (6 / 8 / 10 and 12 / 14 / 16 columns per lane): read
    lengths either side of every edge C = 64 W, near-abort and ordinary pairs, through all modes (4 = exact only)."""
    rng = np.random.default_rng(12)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    cases = []
    for W in (4, 6, 8, 10, 12, 14, 16):
        for C in (64 * W - 1, 64 * W, 64 * W + 1):
            m = C + 1
            base = rs(m)
            cases.append(([base], [b"T" * 30 + base + b"T" * 30, b"G" * 30 + base[:m // 2] + rs(7) + base[m // 2:] + b"G" * 30]))
            r = bytearray(base)
            for pos in rng.choice(m, size=62, replace=False):                    # ~62 mismatches: the -600 line
                r[pos] = ord("A") if r[pos] != ord("A") else ord("C")
            cases.append(([bytes(r)], [b"T" * 30 + base + b"T" * 30]))
    _check(gpu_ctx, _abi.PackedBatch(cases))


def test_every_read_length_1_to_140(gpu_ctx):
    # every slack configuration of lane 0 (W0 = 1..W) for the narrow strips, haplotype windows 1..3
    rng = np.random.default_rng(6)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    hap = rs(61 + 75)
    cases = [([hap[30:30 + m] if m <= 76 else hap[30:106] + rs(m - 76)], [hap]) for m in range(1, 141)]
    cases += [([rs(m)], [rs(61 + k)]) for m in (1, 2, 3, 9, 65, 66) for k in (0, 1, 2)]
    _check(gpu_ctx, _abi.PackedBatch(cases))


def test_certificate_and_exact_redo(gpu_ctx):
    """Pairs that abort, or come close to the -600 line, must go through the exact kernel and still
    match the oracle; ordinary pairs must not need it."""
    rng = np.random.default_rng(8)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    hard = [([b"A" * 700], [b"C" * 800]), ([rs(400)], [rs(460)]), ([rs(900)], [rs(1000)]),
            ([b"AC" * 150], [b"GT" * 460]), ([rs(50)], [rs(650)]), ([rs(640)], [rs(110)]),
            # long shared prefix then garbage: rows near the end dip towards the abort line
            ([(lambda p: p + rs(70))(rs(500))], [b"G" * 30 + rs(570) + b"G" * 30])]
    # a read that matches the haplotype except for ~64 mismatches (score ~ -580 .. -620)
    base = bytearray(rs(900))
    for nmis in (60, 64, 66, 68, 72):
        r = bytearray(base)
        for p in rng.choice(900, size=nmis, replace=False):
            r[p] = ord("A") if r[p] != ord("A") else ord("C")
        hard.append(([bytes(r)], [b"T" * 30 + bytes(base) + b"T" * 30]))
    b = _abi.PackedBatch(hard)
    ll = _check(gpu_ctx, b)
    assert (ll == -700.0).sum() >= 4
    plan = gpu_ctx.plan(b)
    plan.set_timing(True)
    plan.execute()
    plan.fetch()
    st = plan.kernel_stats()
    assert plan.exact_pairs(st) >= 4                    # the exact kernel took the uncertain pairs
    plan.close()
    loci, _ = synth.config_loci("config2")
    plan = gpu_ctx.plan(synth.pack_loci(loci)[0])
    plan.execute()
    plan.fetch()
    assert plan.exact_pairs(plan.kernel_stats()) == 0   # nothing near the abort line: certificate clears all
    plan.close()


def test_edge_shapes(gpu_ctx):
    rng = np.random.default_rng(7)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    cases = []
    for hl in [10, 60, 61, 62, 63, 75, 100]:
        for m in [1, 2, 3, 5, 16, 17, 40, 100]:
            cases.append(([rs(m)], [rs(hl)]))
    for hl, m in [(760, 100), (761, 100), (700, 41), (100, 640), (100, 641), (100, 700), (1300, 650),
                  (1261, 601), (1262, 601)]:
        cases.append(([rs(m)], [rs(hl)]))
    cases.append(([b"A" * 700], [b"C" * 800]))      # every row aborts
    cases.append(([b"A" * 300], [b"C" * 900]))
    cases.append(([b"ACGT" * 100], [b"ACGT" * 130]))
    _check(gpu_ctx, _abi.PackedBatch(cases))


def test_ont_params_and_flank_lengths(gpu_ctx):
    rng = np.random.default_rng(9)
    loci = [synth.synth_locus(rng, int(rng.integers(50, 700)), int(rng.integers(2, 40)), 4, 4, sub_rate=0.03,
                              indel_rate=0.03) for _ in range(12)]
    batch, _ = synth.pack_loci(loci)
    _check(gpu_ctx, batch, _abi.make_params(synth.ONT_PARAMS))
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    for F in [0, 3, 10, 35]:
        cs = [([rs(int(rng.integers(1, 160)))], [rs(int(rng.integers(71, 260)))]) for _ in range(40)]
        _check(gpu_ctx, _abi.PackedBatch(cs), _abi.make_params(_abi.default_params().as_tuple()[:7], indel_flank_len=F))


def test_non_acgt_bytes_take_the_byte_compare_path(gpu_ctx):
    # N, lower case and arbitrary bytes: the LUT kernels only see pure ACGT pairs, everything else
    # must be scored by the byte-compare (exact) kernel with identical results
    rng = np.random.default_rng(12)
    loci = [synth.synth_locus(rng, int(tr), 4, 3, 4, sub_rate=0.01, indel_rate=0.01) for tr in [30, 120, 300, 700]]
    flat = []
    for k, L in enumerate(loci):
        reads = [bytearray(r) for r in L.trimmed_reads]
        haps = [bytearray(h) for h in L.haplotypes]
        reads[0][len(reads[0]) // 2] = ord("N")
        reads[1][3] = ord("a")                       # 'a' != 'A' for the reference (byte compare)
        haps[0][40] = ord("N")
        haps[1][45] = ord("c")
        if k == 0:
            haps[2][50] = 0xC3                       # arbitrary byte, also present in a read
            reads[2][20] = 0xC3
        flat.append(([bytes(r) for r in reads], [bytes(h) for h in haps]))
    b = _abi.PackedBatch(flat)
    _check(gpu_ctx, b)
    plan = gpu_ctx.plan(b)
    plan.execute()
    plan.fetch()
    assert plan.exact_pairs(plan.kernel_stats()) >= 20    # they all went through the exact kernel
    plan.close()


def test_asymmetric_transition_params(gpu_ctx):
    # ins != del transitions take the general 13-op cell body (the defaults take the 11-op one)
    rng = np.random.default_rng(10)
    loci = [synth.synth_locus(rng, int(tr), int(rng.integers(2, 20)), 3, 3, sub_rate=0.02, indel_rate=0.02)
            for tr in [15, 70, 130, 260, 420, 700, 1100]]
    batch, _ = synth.pack_loci(loci)
    for vals in [(-1.5, -0.3, -0.8, -0.6, -0.0001, -7.5, -9.25), (-0.7, -0.9, -2.0, -0.2, -0.01, -3.0, -12.0),
                 (-1.0, -0.458675, -1.0, -0.458675, -0.00005800168, -10.448214728, -6.0)]:
        _check(gpu_ctx, batch, _abi.make_params(vals))


def test_masks_leave_cells_untouched(gpu_ctx):
    rng = np.random.default_rng(11)
    loci = [synth.synth_locus(rng, 60, 3, 5, 8) for _ in range(3)]
    flat = [(L.trimmed_reads, L.haplotypes) for L in loci]
    rr = rng.integers(0, 2, size=sum(len(r) for r, _ in flat)).astype(np.uint8)
    rh = rng.integers(0, 2, size=sum(len(h) for _, h in flat)).astype(np.uint8)
    b = _abi.PackedBatch(flat, realign_read=rr, realign_hap=rh)
    sentinel = np.full(b.ll_size, 12345.0)
    ll, _ = gpu_ctx.align_batch(b, out_ll=sentinel.copy())
    ref = sentinel.copy()
    import ctypes as C
    rc = ol.oracle().ltr_oracle_align_batch(C.byref(gpu_ctx.params), C.byref(b.struct), ref.ctypes.data_as(C.c_void_p),
                                            None, None)
    assert rc == 0
    assert _bits_equal(ll, ref)
    assert (ll == 12345.0).any() and (ll != 12345.0).any()


def _near_pairs(rng, ms, lower=False):
    """(read, haplotype) pairs that really align: the read is the haplotype window with a few
    substitutions and one small indel, so the whole DP runs (no abort, no shortcut)."""
    cases = []
    for m in ms:
        core = bytearray(synth._rand_seq(rng, m).tobytes())
        read = bytearray(core)
        for p in rng.choice(m, size=min(max(1, m // 150), 30), replace=False):      # (> 66 mismatches would abort)
            read[p] = ord("A") if read[p] != ord("A") else ord("C")
        if m > 20:
            cut = int(rng.integers(5, m - 5))
            read = read[:cut] + read[cut + int(rng.integers(1, 3)):]
        if lower:
            read[-1] |= 0x20                       # not ACGT -> "generic" -> byte-compare (exact) kernel
        hap = synth._rand_seq(rng, 30).tobytes() + bytes(core) + synth._rand_seq(rng, 30).tobytes()
        cases.append(([bytes(read)], [hap]))
    return cases


def test_tail_lane_geometry_every_width(gpu_ctx):
    """Slack columns live in the last lane of the last block: for every strip width W = 1..20 walk
    the read length through the block's corner cases (last lane holds 1, 2, W/2, W-1, W columns;
    1 lane, 63 and 64 lanes), certificate kernels and -- same shapes with a non-ACGT byte -- the
    exact kernel (W = 8: up to three column blocks here)."""
    rng = np.random.default_rng(16)
    ms = set()
    for W in range(1, 21):
        lo = 64 * (W - 1)
        for C in (lo + 1, lo + 2, lo + W // 2 + 1, lo + W, lo + W + 1, 64 * W - W, 64 * W - W + 1, 64 * W - 1, 64 * W):
            if C >= 1:
                ms.add(C + 3)                      # the generator deletes 1-2 bases again: lengths scatter around the edges
    ms = sorted(ms)
    _check(gpu_ctx, _abi.PackedBatch(_near_pairs(rng, ms)))
    ex = _abi.PackedBatch(_near_pairs(rng, ms[::3] + [8 * 64 * k + d for k in (1, 2, 3) for d in (-1, 0, 1, 2, 9)], lower=True))
    _check(gpu_ctx, ex)
    plan = gpu_ctx.plan(ex)
    plan.execute()
    plan.fetch()
    assert plan.exact_pairs(plan.kernel_stats()) == ex.ll_size
    plan.close()


def test_many_column_blocks_and_very_long_read(gpu_ctx):
    """2..6 column blocks through the certificate kernel (W <= 20) and the exact kernel, then one
    33 kb read (26 blocks of 20-column strips / 65 blocks in the exact kernel): geometry at lengths
    where the block count is recomputed from the balanced lane count."""
    rng = np.random.default_rng(17)
    ms = [2050, 3100, 4097, 5200, 7000]
    _check(gpu_ctx, _abi.PackedBatch(_near_pairs(rng, ms)))
    _check(gpu_ctx, _abi.PackedBatch(_near_pairs(rng, [1100, 2300, 4200], lower=True)))
    for lower in (False, True):
        (reads, haps), = _near_pairs(rng, [33001], lower=lower)
        b = _abi.PackedBatch([(reads, haps)])
        ll, _ = gpu_ctx.align_batch(b)
        ref = ol.oracle_align_long(haps[0], reads[0], gpu_ctx.params, rolling=True)
        assert _bits_equal([ll[0]], [ref]) and ll[0] > -600.0


def test_pair_packing_rule(gpu_ctx):
    """Default: a single locus (config 2, 224 pairs) keeps one pair per wavefront -- latency -- and a
    batch that fills the GPU packs several short reads per wavefront -- throughput -- except the read lengths that
    are too rare in the batch to put two wavefronts on every SIMD that way.  Same scores."""
    def classes(batch):
        plan = gpu_ctx.plan(batch)
        plan.execute()
        ll, _ = plan.fetch()
        st = plan.kernel_stats()
        plan.close()
        return ll, sum(k["pairs"] for k in st if k["family"] == "packed"), sum(k["pairs"] for k in st if k["family"] != "exact" and k["lanes_per_pair"] == 64)
    loci, _ = synth.config_loci("config2")
    small, _ = synth.pack_loci(loci)
    ll_s, two, one = classes(small)
    assert two == 0 and one == small.ll_size
    rng = np.random.default_rng(21)
    many = [synth.synth_locus(rng, int(rng.integers(20, 200)), 3, 6, 20, sub_rate=0.002, indel_rate=0.001) for _ in range(220)]
    big, _ = synth.pack_loci(many)
    assert big.ll_size >= 32 * gpu_ctx.device_info()["n_cu"]
    ll_b, two, one = classes(big)
    assert two > 0.6 * big.ll_size
    gpu_ctx.set_pair_packing(0)
    try:
        ll_b0, two0, _ = classes(big)
    finally:
        gpu_ctx.set_pair_packing(-1)
    assert two0 == 0 and _bits_equal(ll_b, ll_b0)


def test_long_pairs_that_abort_leave_early_with_the_same_score(gpu_ctx):
    """Reads that need several column blocks and cross the -600 line early, late, or barely: the
    blocks that already hold a row's whole +-600 band settle it (abort / uncertain) without waiting
    for the last block; scores must stay the reference's."""
    rng = np.random.default_rng(23)
    rs = lambda n: synth._rand_seq(rng, n).tobytes()
    cases = []
    for m, where, nmis in [(2600, (0, 800), 90), (2600, (1700, 2500), 90), (4200, (1500, 2200), 80), (2100, (0, 2100), 64),
                           (2100, (0, 2100), 66), (2100, (0, 2100), 68), (2100, (0, 2100), 72), (3300, (3000, 3300), 75),
                           (5200, (100, 400), 70)]:
        core = bytearray(rs(m))
        read = bytearray(core)
        for p in rng.choice(np.arange(where[0], where[1]), size=nmis, replace=False):
            read[p] = ord("A") if read[p] != ord("A") else ord("C")
        cases.append(([bytes(read)], [rs(30) + bytes(core) + rs(30)]))
    cases += [([rs(2300)], [rs(2500)]), ([rs(3000)], [rs(2700)]), ([b"AC" * 1200], [b"GT" * 1300])]      # unrelated: abort within a few rows
    b = _abi.PackedBatch(cases)
    ll = _check(gpu_ctx, b)
    assert (ll == -700.0).sum() >= 8 and (ll > -700.0).sum() >= 1
    lower = [([r[0][:-1] + bytes([r[0][-1] | 0x20])], h) for r, h in cases]      # same shapes through the exact kernel only
    _check(gpu_ctx, _abi.PackedBatch(lower))
