"""-m gpu: BASELINE-sized batches checked through size-independent properties (the oracle would
need hours there) plus an oracle spot check on a random subset of loci."""
import os
import subprocess
import sys
import numpy as np
import pytest

import oracle_lib as ol
import parity_util
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


# ---- the BASELINE configurations at their stated size --------------------------------------------
def test_config3_stated_size_10k_loci(gpu_ctx):
    """BASELINE config 3 as bench.py runs it: 10 000 loci, 30x, TR 20-1000 bp -- one resident plan,
    the pass that produces the headline number, checked: nominal cell count, idempotence, value
    ranges, and bit-exactness against the oracle on a sample stratified over every launch class
    plus the 12 most expensive loci (not the cheapest ones)."""
    loci, _ = synth.config_loci("config3")
    assert len(loci) == 10000
    batch, _ = synth.pack_loci(loci)
    plan = gpu_ctx.plan(batch)
    assert plan.cells == synth.nominal_cells(batch) and plan.cells > 5e11
    plan.execute()
    ll, _ = plan.fetch()
    plan.execute()
    ll2, _ = plan.fetch()
    assert np.array_equal(ll.view(np.uint64), ll2.view(np.uint64))
    assert np.isfinite(ll).all() and (ll < 0).all()
    sent = (ll == -700.0) | (ll == -1e9)
    assert (ll[~sent] > -600.0 - 1e-9).mean() > 0.999          # (a pair can finish below -600 without any row aborting)
    rl, hl = np.diff(batch.read_off).astype(np.float64), np.diff(batch.hap_off).astype(np.float64)
    cost = np.array([rl[batch.locus_read_off[l]:batch.locus_read_off[l + 1]].sum() * hl[batch.locus_hap_off[l]:batch.locus_hap_off[l + 1]].sum()
                     for l in range(batch.n_loci)])
    res = parity_util.stratified_oracle_check(batch, ll, gpu_ctx.params, n_loci_target=220, reads_per_locus=2,
                                              extra_loci=np.argsort(-cost)[:12])
    assert res["loci"] >= 200 and res["classes_covered"] >= 20 and res["mismatches"] == 0, res
    plan.close()


def _rolling(L, params, r, h):
    return ol.oracle_align_long(L.haplotypes[h], L.trimmed_reads[r], params, rolling=True)


def test_config5_stated_size_long_and_short_path_in_one_call(gpu_ctx):
    """BASELINE config 5 on one GPU: 64 loci of 5-kb VNTRs with ONT-like 4 % error and f=g=-4.6
    (raw alignments, 8 reads x 4 alleles) AND a period-1 sub-batch through the seeded stutter path
    (--stutter-align-len), all in ONE ltr_calc_hap_aln_probs call.  Long loci: sampled pairs against
    the rolling oracle (most abort -> -700); every other cell must be a legal value.  Short loci:
    every cell against the short-path restatement."""
    import short_util as su
    import test_gpu_host_path as hp
    prm = _abi.make_params(synth.ONT_PARAMS, use_short_path=1)
    sp = _abi.default_stutter_params()
    long_loci, _ = synth.config_loci("config5", n_loci=64, raw=True)
    rng = np.random.default_rng(55)
    items, kinds = [], []
    for k, L in enumerate(long_loci):
        items.append((L.blocks(), L.raw_alns, None))
        kinds.append(("long", L))
        if k % 4 == 3:                                        # 16 homopolymer loci interleaved with the VNTRs
            blocks, alns = su.homopolymer_locus(rng, int(rng.integers(8, 40)), 3, 8, sub_rate=0.002, indel_rate=0.004)
            items.append((blocks, alns, None))
            kinds.append(("short", (blocks, alns)))
    gpu_ctx.set_params(prm)
    try:
        got = gpu_ctx.calc_hap_aln_probs(items)
    finally:
        gpu_ctx.set_params(_abi.default_params())
    n_short = n_long_checked = 0
    for (kind, obj), (probs, seeds) in zip(kinds, got):
        if kind == "short":
            blocks, alns = obj
            want, ws = hp._expected_calc_hap_aln_probs(prm, sp, blocks, alns, None)
            assert np.array_equal(probs.view(np.uint64), want.view(np.uint64)) and np.array_equal(seeds, ws)
            n_short += 1
        else:
            L = obj
            assert probs.shape == (8, 4) and np.isfinite(probs).all()
            assert ((probs == -700.0) | ((probs > -2000.0) & (probs < 0))).all()
            assert np.array_equal(seeds, [len(a["seq"]) - 1 for a in L.raw_alns])
            for r, h in [(int(rng.integers(0, 8)), int(rng.integers(0, 4)))]:   # one random pair per locus against the oracle
                assert probs[r, h] == _rolling(L, prm, r, h)
                n_long_checked += 1
    assert n_short == 16 and n_long_checked == 64


def test_config5_long_vntr_pairs_that_finish(gpu_ctx):
    """The non-degenerate variant of config 5: the same 5-kb geometry (5 column blocks of 64 lanes,
    or the workgroup-per-pair kernel) with HiFi-like 0.2 % error so that pairs FINISH instead of
    aborting -- 64 loci x 8 reads x 4 alleles, 5e10 cells.  Sampled pairs against the rolling oracle,
    bit for bit; an error-free read scores best against its own allele."""
    loci, _ = synth.config_loci("config5hifi")
    assert len(loci) == 64
    batch, pidx = synth.pack_loci(loci)
    prm = _abi.make_params(synth.ONT_PARAMS)
    gpu_ctx.set_params(prm)
    try:
        plan = gpu_ctx.plan(batch)
        assert plan.cells > 4e10
        plan.execute()
        ll, _ = plan.fetch()
        plan.close()
    finally:
        gpu_ctx.set_params(_abi.default_params())
    assert np.isfinite(ll).all() and (ll < 0).all()
    finished = (ll > -600.0)
    assert finished.mean() > 0.5                               # pairs with the right allele (and its neighbours) finish
    rng = np.random.default_rng(56)
    checked = 0
    for l in rng.choice(len(loci), size=24, replace=False):
        L = loci[int(l)]
        pools, idx = synth.pool_reads(L.trimmed_reads)
        M = batch.locus_matrix(ll, int(l))
        for _ in range(2):
            p, h = int(rng.integers(0, len(pools))), int(rng.integers(0, len(L.haplotypes)))
            want = ol.oracle_align_long(L.haplotypes[h], pools[p], prm, rolling=True)
            assert M[p, h] == want, (int(l), p, h, M[p, h], want)
            checked += 1
    assert checked == 48
    hits = tot = 0
    for l, L in enumerate(loci):
        M = batch.locus_matrix(ll, l)
        pools, _ = synth.pool_reads(L.trimmed_reads)
        windows = [h[30:len(h) - 30] for h in L.haplotypes]
        for p, r in enumerate(pools):
            if r in windows:
                tot += 1
                hits += int(np.argmax(M[p]) == windows.index(r))
    assert hits == tot


def _run_bench(args, timeout=900):
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args + ["--no-cpu-baseline", "--no-end-to-end", "--no-neighbours"],
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                        # one JSON line, from rank 0
    assert len(lines[0]) < 4096 and r.stdout.rstrip().splitlines()[-1] == lines[0]     # compact, and the last line of stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """BASELINE config 4's code path with the real kernels on a one-GPU box: `bench.py --gpus 2 --one-gpu`
    starts two ranks (both on cuda:0, exchange over gloo), cost-shards the same loci from the generator's locus
    headers, every rank generates and scores its shard only with its own resident plan, the ordered gather puts
    rank 0 in possession of every locus in global order; rank 0 then bit-compares every 8th locus with its own
    single-GPU recomputation and with the oracle.  (The 8-GPU run itself belongs to the driver.)"""
    d = _run_bench(["--gpus", "2", "--one-gpu", "--loci", "600", "--steps", "1", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["debug_one_gpu"] is True and d["scaling"] == "strong"
    assert d["config"]["total_loci"] == 600
    sg = d["single_gpu_check"]
    assert sg["order_ok"] and sg["mismatches"] == 0 and sg["checked_pairs"] > 1000 and sg["ranks_covered"] == 2
    assert d["oracle_check"]["mismatches"] == 0 and d["oracle_check"]["checked_pairs"] > 100
    assert d["weak_scaling"]["total_loci"] == 1200


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu():
    """The driver's N = 8 shape on the one-GPU box: eight ranks' resident plans side by side on cuda:0 (exchange over
    gloo), 2000 loci cost-sharded eight ways: shard sizes, the ordered gather, the single-GPU recomputation and the
    oracle at N = 8."""
    d = _run_bench(["--gpus", "8", "--one-gpu", "--loci", "2000", "--steps", "1", "--warmup", "1", "--no-weak"], timeout=1500)
    assert d["n_gpus"] == 8 and d["debug_one_gpu"] is True and d["scaling"] == "strong"
    assert d["config"]["total_loci"] == 2000
    sizes = d["config"]["shard_loci"]
    assert len(sizes) == 8 and sum(sizes) == 2000 and min(sizes) >= 150          # cost-balanced: no rank starves
    sg = d["single_gpu_check"]
    assert sg["order_ok"] and sg["mismatches"] == 0 and sg["loci"] == 250 and sg["ranks_covered"] == 8
    assert d["oracle_check"]["mismatches"] == 0 and d["oracle_check"]["checked_pairs"] > 100


@pytest.mark.gpu
def test_bench_two_gpus_over_rccl():
    """Where the box has two GPUs: the same path on the `nccl` backend (RCCL), one rank per GPU, device tensors in
    the ordered gather.  Skipped on one-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    d = _run_bench(["--gpus", "2", "--loci", "600", "--steps", "2", "--warmup", "1", "--no-weak"])
    assert d["n_gpus"] == 2 and "debug_one_gpu" not in d and d["scaling"] == "strong"
    sg = d["single_gpu_check"]
    assert sg["order_ok"] and sg["mismatches"] == 0 and sg["ranks_covered"] == 2
    assert d["oracle_check"]["mismatches"] == 0 and d["value"] > 0


_RCCL_ONE_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from longtr_amd import shard
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rng = np.random.default_rng(3)
sizes = rng.integers(1, 40, size=300)
ids = rng.permutation(300)                                   # local order != global order
og = shard.OrderedGather(sizes, ids, dev)
ll = torch.from_numpy(rng.standard_normal(int(sizes.sum()))).to(dev)
out = og(ll)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
# expected: locus ids[k] (size sizes[k]) sits at global_off[ids[k]]
loff = np.concatenate([[0], np.cumsum(sizes)]); exp = np.empty(int(sizes.sum()))
for k in range(300):
    exp[og.global_off[ids[k]]:og.global_off[ids[k]] + sizes[k]] = ll[loff[k]:loff[k + 1]].cpu().numpy()
assert np.array_equal(out.cpu().numpy(), exp) and float(t) == 1.5
try:
    shard.OrderedGather(sizes, np.zeros(300, dtype=np.int64), dev)        # ids that do not partition 0..n-1
    raise SystemExit("bad partition accepted")
except ValueError:
    pass
# bench.py's independent derivation of the global layout (tensor collectives on the exchange device) on this backend
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(sys.argv[1], "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
class _B: pass
b = _B(); b.ll_off = loff
gi, goff = bench.expected_global_offsets(b, ids, 1, dev)
assert np.array_equal(gi, np.arange(300)) and np.array_equal(goff, og.global_off)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.destroy_process_group()
print("rccl one-rank exchange ok")
"""


_EXCHANGE_ONE_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
import importlib.util
import numpy as np, torch, torch.distributed as dist
from longtr_amd import shard
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(sys.argv[1], "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
# bench.py's own set-up of an N > 1 run, with one rank: gloo control group, the RCCL probe in a CHILD process (this process has
# not touched the GPU yet), then the RCCL group of the exchange next to the gloo default group
dev, xdev, xgroup, backend, devices = bench.setup_exchange(0, 0, 1, "nccl", False, False, 60.0)
assert backend == "nccl" and xdev == dev and xgroup is not None and dist.get_backend() == "gloo", (backend, xdev)
assert len(devices) == 1 and len(devices[0].split()[0].split(":")) == 3, devices
rng = np.random.default_rng(5)
sizes = rng.integers(1, 40, size=200); ids = rng.permutation(200)
og = shard.OrderedGather(sizes, ids, xdev, group=xgroup)
ll = torch.from_numpy(rng.standard_normal(int(sizes.sum()))).to(dev)
out = og(ll)
dist.barrier()                                                  # control plane: gloo
loff = np.concatenate([[0], np.cumsum(sizes)]); exp = np.empty(int(sizes.sum()))
for k in range(200):
    exp[og.global_off[ids[k]]:og.global_off[ids[k]] + sizes[k]] = ll[loff[k]:loff[k + 1]].cpu().numpy()
assert np.array_equal(out.cpu().numpy(), exp)
class _B: pass
b = _B(); b.ll_off = loff
gi, goff = bench.expected_global_offsets(b, ids, 1, xdev, xgroup)
assert np.array_equal(gi, np.arange(200)) and np.array_equal(goff, og.global_off)
dist.destroy_process_group()
print("exchange set-up ok:", backend, devices[0])
"""


@pytest.mark.gpu
def test_bench_exchange_setup_gloo_control_rccl_data_one_rank():
    """bench.py's N > 1 set-up as the driver's 8-GPU run will execute it, with the one rank this box has: default group gloo
    with a time-out, the RCCL probe as a child process, the verdict agreed over gloo, an RCCL group for the data exchange, the
    ordered gather and the layout check through THAT group with device tensors, the device's PCI address in the line."""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _EXCHANGE_ONE_RANK, root, str(port)], capture_output=True, text=True, timeout=420, env=env)
    assert r.returncode == 0 and "exchange set-up ok: nccl" in r.stdout, (r.stdout[-800:], r.stderr[-2000:])


@pytest.mark.gpu
def test_ordered_gather_on_the_nccl_backend_one_rank():
    """RCCL itself on the one-GPU box: a one-rank `nccl` process group with DEVICE tensors through every collective the
    N > 1 path issues (all_gather of the sizes, gather of ids and payload, all_reduce, barrier) and the global
    re-ordering.  What it cannot show is a second GPU; test_bench_two_gpus_over_rccl does where there is one."""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, root, str(port)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl one-rank exchange ok" in r.stdout, (r.stdout[-800:], r.stderr[-2000:])


@pytest.mark.gpu
def test_two_stream_plan_with_many_multi_block_pairs(gpu_ctx):
    """The automatic mode deals a plan's launches over two streams, each with its own region of the boundary-strip
    scratch.  Enough long reads (> 1280 columns: several column blocks on one wavefront, strips parked in scratch) to
    stay on the one-wave kernels -- >= 16 long pairs per CU -- next to short loci: every pair must equal the
    single-stream, one-class-per-width schedule (mode 3) bit for bit, and the oracle on a sample."""
    rng = np.random.default_rng(77)
    n_cu = gpu_ctx.device_info()["n_cu"]
    loci = []
    while sum(len(L.trimmed_reads) * len(L.haplotypes) for L in loci) < 16 * n_cu + 600:
        loci.append(synth.synth_locus(rng, int(rng.integers(1300, 2600)), int(rng.integers(2, 7)), 5, 6, sub_rate=0.001, indel_rate=0.0005))
    for _ in range(150):
        loci.append(synth.synth_locus(rng, int(rng.integers(20, 900)), int(rng.integers(2, 7)), int(rng.integers(2, 6)), 8))
    batch, _ = synth.pack_loci(loci)
    plan = gpu_ctx.plan(batch)
    plan.execute(); ll_auto, _ = plan.fetch()
    st = plan.kernel_stats()
    plan.close()
    assert sum(k["pairs"] for k in st if k["family"] == "one-wave" and k["strip_width"] >= 11) >= 16 * n_cu   # the long pairs stayed on one wavefront each
    gpu_ctx.set_pair_packing(3)
    try:
        ll_ref, _ = gpu_ctx.align_batch(batch)
    finally:
        gpu_ctx.set_pair_packing(-1)
    assert np.array_equal(ll_auto.view(np.uint64), ll_ref.view(np.uint64))
    assert (ll_auto > -600.0).mean() > 0.95
    pick = [int(x) for x in rng.choice(len(loci), size=10, replace=False)] + [0, 1]
    sub, _ = synth.pack_loci([loci[i] for i in pick])
    want, _, _ = ol.oracle_align_batch(sub, gpu_ctx.params)
    at = 0
    for k, i in enumerate(pick):
        n = sub.ll_off[k + 1] - sub.ll_off[k]
        got = ll_auto[batch.ll_off[i]:batch.ll_off[i + 1]]
        assert np.array_equal(got.view(np.uint64), want[sub.ll_off[k]:sub.ll_off[k + 1]].view(np.uint64)), i


@pytest.mark.gpu
def test_multi_width_launches_equal_a_launch_per_class(gpu_ctx):
    """Automatic mode, large plans: the one-wave classes of strip widths 11 .. 20 run as ONE persistent launch (ltr_dp_multi_kernel:
    a call per pair into the class's own body) and so do the packed widths 13 .. 20 (ltr_dp_pack_multi_kernel).  Forced on for a
    small mixed batch (no_multi = -1) and compared, bit for bit, with a launch per class (no_multi = 1) and with the oracle:
    reads of 60 .. 1100 bases, three one-wave classes, three packed widths, single- and multi-range tables."""
    import oracle_lib as ol
    rng = np.random.default_rng(77)
    def seq(n): return bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
    loci = []
    for M in (700, 930, 1100, 1290, 420, 500, 610, 150, 60):
        core = seq(M)
        haps = [seq(30) + core + seq(30), seq(30) + core[:M // 2] + seq(7) + core[M // 2:] + seq(30)]
        reads = []
        for _ in range(6):
            r = bytearray(core)
            for p in rng.choice(M, size=3, replace=False):
                r[p] = ord("A") if r[p] != ord("A") else ord("C")
            reads.append(bytes(r))
        reads.append(seq(M))                                  # a read that matches nothing: certificate fails, exact kernels
        for _ in range(6 if M >= 650 else 200):               # (enough short pairs for the automatic mode to pack them)
            loci.append((reads, haps))
    batch = _abi.PackedBatch(loci)
    out = {}
    try:
        gpu_ctx.set_debug("pack_rule", 2)                     # no per-length floor on the lanes per pair: every short read is packed
        gpu_ctx.set_debug("plan_kernel", 1)                   # (round 4's launches: the plan kernel has a test of its own below)
        for nm in (1, -1):
            gpu_ctx.set_debug("no_multi", nm)
            plan = gpu_ctx.plan(batch)
            plan.execute()
            out[nm], _ = plan.fetch()
            st = [k for k in plan.kernel_stats() if k["pairs"]]
            plan.close()
            merged = [k for k in st if k.get("ranges") and len({w for _, w, _ in k["ranges"]}) > 1]
            assert (len(merged) == 2) == (nm == -1), st        # one multi-width launch per family, only when asked for
    finally:
        gpu_ctx.set_debug("reset", 0)
    assert np.array_equal(out[1].view(np.uint64), out[-1].view(np.uint64))
    ref, _, _ = ol.oracle_align_batch(batch, gpu_ctx.params)
    assert np.array_equal(out[-1].view(np.uint64), ref.view(np.uint64))


@pytest.mark.gpu
def test_small_and_mid_size_plans_take_the_plan_kernel_by_rule(gpu_ctx):
    """One GPU's share of config 4 at N = 8 (a cost shard of 1250 config-3 loci, ~184 k pairs = ~700 per CU): the automatic mode
    runs EVERY one-wave class and packed strip width as ONE launch, the plan kernel (rule: every automatic-mode plan, either
    indel model), which also scores the pairs whose certificate fails itself.  Same bits as round 4's launches (a launch per
    class with exact lists; the multi-width launches)."""
    from longtr_amd import shard
    n_cu = gpu_ctx.device_info()["n_cu"]
    hdr = synth.config_headers("config3", n_loci=10000)
    ids = shard.shard_by_cost(shard.header_time_costs(hdr), 8)[3]
    loci, _ = synth.config_loci("config3", n_loci=10000, ids=ids)
    batch, _ = synth.pack_loci(loci)

    def run(b, **knobs):
        for k, v in knobs.items():
            gpu_ctx.set_debug(k, v)
        try:
            plan = gpu_ctx.plan(b)
            plan.set_timing(True)
            plan.execute()
            ll, _ = plan.fetch()
            st = [k for k in plan.kernel_stats() if k["pairs"]]
            n_pairs = plan.num_pairs
            plan.close()
        finally:
            gpu_ctx.set_debug("reset", 0)
        merged = [k for k in st if k.get("ranges") and len({w for _, w, _ in k["ranges"]}) > 1]
        return ll, merged, n_pairs, st

    ll_rule, merged, n_pairs, st = run(batch)
    assert n_pairs < 4096 * n_cu
    assert len(merged) == 1 and {lp for lp, _, _ in merged[0]["ranges"]} >= {64, 32}, merged     # one launch: one-wave AND packed ranges
    fast = [k for k in st if k["family"] in ("one-wave", "packed")]
    assert len(fast) == 1 and fast[0]["pairs"] == sum(n for _, _, n in merged[0]["ranges"])
    assert sum(k["pairs"] for k in st if k["family"] == "exact") > 0                             # failed certificates: scored (in line or from a list)
    ll_multi, merged_multi, _, _ = run(batch, plan_kernel=1)
    if 512 * n_cu <= n_pairs:
        assert {k["family"] for k in merged_multi} == {"one-wave", "packed"}, merged_multi       # round 4's rule, still there behind the knob
    ll_per_class, merged_off, _, _ = run(batch, plan_kernel=1, no_multi=1)
    assert not merged_off
    assert np.array_equal(ll_rule.view(np.uint64), ll_per_class.view(np.uint64))
    assert np.array_equal(ll_rule.view(np.uint64), ll_multi.view(np.uint64))
    ll_shares, _, _, _ = run(batch, plan_share=1)                                               # every wavefront starting at the top of the table
    assert np.array_equal(ll_rule.view(np.uint64), ll_shares.view(np.uint64))
    ll_chain, _, _, _ = run(batch, chain=1)                                                     # the chained walk (ltr_dp_chain.hpp; off by default: slower)
    assert np.array_equal(ll_rule.view(np.uint64), ll_chain.view(np.uint64))
    res = parity_util.stratified_oracle_check(batch, ll_rule, gpu_ctx.params, n_loci_target=60)
    assert res["mismatches"] == 0 and res["checked_pairs"] > 500
    # a one-locus batch and a 300-locus one: the plan kernel as well
    for sub in (loci[:1], loci[:300]):
        small, _ = synth.pack_loci(sub)
        ll_s, merged_s, _, st_s = run(small)
        ll_c, _, _, _ = run(small, plan_kernel=1)
        assert np.array_equal(ll_s.view(np.uint64), ll_c.view(np.uint64))
        assert len([k for k in st_s if k["family"] in ("one-wave", "packed")]) == 1
    # asymmetric indel model (any seven negative transitions, HapAligner.h:111-119): the plan kernel's general-model instance
    # (round 6) -- one launch again, the same bits as a launch per class and as the oracle; failed certificates by the generic body
    held = gpu_ctx.params
    try:
        gpu_ctx.set_params(_abi.make_params((-1.2, -0.3, -0.9, -0.5, -0.0001, -5.0, -4.0)))
        for sub in (loci[:40], loci[::4]):
            small, _ = synth.pack_loci(sub)
            ll_a, _, _, st_a = run(small)
            assert len([k for k in st_a if k["family"] in ("one-wave", "packed")]) == 1, st_a
            ll_b, _, _, st_b = run(small, plan_kernel=1)
            assert len([k for k in st_b if k["family"] in ("one-wave", "packed")]) > 1
            assert np.array_equal(ll_a.view(np.uint64), ll_b.view(np.uint64))
        ref, _, _ = ol.oracle_align_batch(small, gpu_ctx.params) if small.ll_size < 30000 else (None, None, None)
        if ref is not None:
            assert np.array_equal(ll_a.view(np.uint64), ref.view(np.uint64))
        else:
            res = parity_util.stratified_oracle_check(small, ll_a, gpu_ctx.params, n_loci_target=60)
            assert res["mismatches"] == 0 and res["checked_pairs"] > 500
        # ONT-like asymmetric transitions: many certificates fail -> the generic exact body inside the launch
        gpu_ctx.set_params(_abi.make_params((-1.0, -0.458675, -1.0, -0.458675, -0.00005800168, -4.6, -5.2)))
        small, _ = synth.pack_loci(loci[:60])
        ll_a, _, _, st_a = run(small)
        ref, _, _ = ol.oracle_align_batch(small, gpu_ctx.params)
        assert np.array_equal(ll_a.view(np.uint64), ref.view(np.uint64))
    finally:
        gpu_ctx.set_params(held)


@pytest.mark.gpu
def test_plan_kernel_next_to_workgroup_classes_and_exact_lists(gpu_ctx):
    """One plan with everything the plan kernel has to live with: one-wave and packed classes (its own entries), pairs that START OUT
    with the exact body -- length differences no certificate can hold, of every list's read length, and bytes outside ACGT --,
    reads beyond 1281 bases (workgroup classes, launches of their own) of which some fail their certificate (lists fed from the
    device, exact launches of their own: the list of 1026 .. 3585-base reads then holds starters AND device-fed pairs).  Bits:
    the oracle's and round 4's launches'."""
    rng = np.random.default_rng(77)
    seq = lambda n: synth._rand_seq(rng, n).tobytes()

    def mutate(s, k):
        r = bytearray(s)
        for p in rng.choice(len(r), size=k, replace=False):
            r[p] = ord("A") if r[p] != ord("A") else ord("C")
        return bytes(r)

    loci = []
    for M in (90, 300, 700, 1100, 1250):                       # starters of the short / mid / long / four-wave lists
        core = seq(M)
        haps = [seq(30) + core + seq(30), seq(30) + core[:max(M - 560, 20)] + seq(30), seq(30) + core + seq(540) + seq(30)]
        reads = [mutate(core, 2) for _ in range(4)] + [mutate(core[:max(M - 560, 20)], 1)]
        loci.append((reads, haps))
    for M in (2700, 3100):                                     # workgroup classes (beyond two column blocks of one wavefront); the garbage read fails its certificate on the device
        core = seq(M)
        haps = [seq(30) + core + seq(30), seq(30) + mutate(core, 5) + seq(30)]
        loci.append(([mutate(core, 3), seq(M), mutate(core, 9)], haps))
    core = seq(400)
    loci.append(([core[:100] + b"N" + core[101:], mutate(core, 2).lower(), core], [seq(30) + core + seq(30), seq(30) + core[:200] + b"n" + core[201:] + seq(30)]))
    for _ in range(40):                                        # bulk: enough pairs for the side streams of the exact launches (32 per CU)
        core = seq(int(rng.integers(60, 640)))
        loci.append(([mutate(core, int(rng.integers(0, 4))) for _ in range(30)], [seq(30) + mutate(core, j) + seq(30) for j in range(8)]))
    batch = _abi.PackedBatch(loci)
    ref, _, _ = ol.oracle_align_batch(batch, gpu_ctx.params)
    out = {}
    try:
        for knob in (0, 1):
            gpu_ctx.set_debug("plan_kernel", knob)
            plan = gpu_ctx.plan(batch)
            plan.set_timing(True)
            for _ in range(2):                                 # (twice: the control words are reset by every execute)
                plan.execute()
            out[knob], _ = plan.fetch()
            st = [k for k in plan.kernel_stats() if k["pairs"]]
            plan.close()
            fams = {k["family"] for k in st}
            assert "workgroup" in fams and "exact" in fams, st
            if knob == 0:
                assert len([k for k in st if k["family"] in ("one-wave", "packed")]) == 1, st
    finally:
        gpu_ctx.set_debug("reset", 0)
    assert np.array_equal(out[0].view(np.uint64), ref.view(np.uint64))
    assert np.array_equal(out[1].view(np.uint64), ref.view(np.uint64))
