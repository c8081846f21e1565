"""-m gpu: BASELINE-sized batches checked through size-independent properties (the oracle would
need hours there) plus an oracle spot check on a random subset of loci."""
import numpy as np
import pytest

import oracle_lib as ol
from longtr_amd import _abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    loci, _ = synth.config_loci("config3", n_loci=400)
    batch, pidx = synth.pack_loci(loci)
    return loci, batch, pidx


def test_config3_properties(gpu_ctx, big):
    loci, batch, _ = big
    plan = gpu_ctx.plan(batch)
    plan.execute()
    ll, seed = plan.fetch()
    assert plan.cells == synth.nominal_cells(batch)
    assert np.isfinite(ll).all() and (ll < 0).all()
    # idempotence: a second execute of the resident plan gives the same bits
    plan.execute()
    ll2, _ = plan.fetch()
    assert np.array_equal(ll.view(np.uint64), ll2.view(np.uint64))
    # every value is a sentinel or a genuine log-likelihood above the abort line
    sent = (ll == -700.0) | (ll == -1e9)
    assert ((ll[~sent] > -600.0 - 1e-9)).mean() > 0.999
    # an error-free read scores best against its own allele; two identical reads score identically
    hits = tot = 0
    for l, L in enumerate(loci[:200]):
        M = batch.locus_matrix(ll, l)
        pools, idx = synth.pool_reads(L.trimmed_reads)
        windows = [h[30:len(h) - 30] for h in L.haplotypes]
        for p, r in enumerate(pools):
            if r in windows:
                tot += 1
                hits += int(np.argmax(M[p]) == windows.index(r))
    assert tot > 100 and hits == tot
    plan.close()


def test_config3_permutation_invariance(gpu_ctx, big):
    # scores do not depend on where a locus sits in the batch (scheduling / binning independence)
    loci, batch, _ = big
    ll, _ = gpu_ctx.align_batch(batch)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(loci))
    b2, _ = synth.pack_loci([loci[i] for i in perm])
    ll2, _ = gpu_ctx.align_batch(b2)
    for new, old in enumerate(perm[:150]):
        assert np.array_equal(b2.locus_matrix(ll2, new).view(np.uint64), batch.locus_matrix(ll, old).view(np.uint64))


def test_config3_oracle_spot_check(gpu_ctx, big):
    loci, batch, _ = big
    ll, _ = gpu_ctx.align_batch(batch)
    rng = np.random.default_rng(1)
    # the cheapest 40 loci of a random 120 (keeps the oracle at a few seconds)
    cand = rng.choice(len(loci), size=120, replace=False)
    cost = [sum(len(r) for r in set(loci[i].trimmed_reads)) * sum(len(h) for h in loci[i].haplotypes) for i in cand]
    pick = [int(cand[k]) for k in np.argsort(cost)[:40]]
    sub, _ = synth.pack_loci([loci[i] for i in pick])
    ref, _, _ = ol.oracle_align_batch(sub, gpu_ctx.params)
    for k, i in enumerate(pick):
        assert np.array_equal(sub.locus_matrix(ref, k).view(np.uint64), batch.locus_matrix(ll, i).view(np.uint64))


def test_config5_ont_long_vntr(gpu_ctx):
    # 5-kb VNTR, ONT error profile, f=g=-4.6: 5 column blocks at W=16; oracle on one small locus
    loci, _ = synth.config_loci("config5", n_loci=1)
    L = loci[0]
    L.trimmed_reads = L.trimmed_reads[:2]
    L.alleles = L.alleles[:2]
    batch, _ = synth.pack_loci([L])
    p = _abi.make_params(synth.ONT_PARAMS)
    gpu_ctx.set_params(p)
    try:
        ll, _ = gpu_ctx.align_batch(batch)
    finally:
        gpu_ctx.set_params(_abi.default_params())
    ref, _, _ = ol.oracle_align_batch(batch, p)
    assert np.array_equal(ll.view(np.uint64), ref.view(np.uint64))
