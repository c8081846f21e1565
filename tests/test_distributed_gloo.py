"""CPU, world_size 2, gloo: locus sharding + the gather of per-locus results (the N>1 path of
bench.py), with a stand-in compute so no GPU is needed."""
import os
import socket

import pytest
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from longtr_amd import shard, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_ll(locus_id, size):
    return torch.arange(size, dtype=torch.float64) * 1e-3 - float(locus_id)


def _worker(rank, world, port, sizes, shards, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shards[rank]
    ll = torch.cat([_fake_ll(l, sizes[l]) for l in mine]) if mine else torch.zeros(0, dtype=torch.float64)
    res = shard.gather_ll(ll, torch.tensor([sizes[l] for l in mine], dtype=torch.int64),
                          torch.tensor(mine, dtype=torch.int64))
    # the form bench.py uses: sizes exchanged once, then the raw collective every step
    metas = shard.exchange_meta(ll.numel(), len(mine), ll.device)
    for _ in range(2):
        raw = shard.gather_ll_raw(ll, torch.tensor([sizes[l] for l in mine], dtype=torch.int64),
                                  torch.tensor(mine, dtype=torch.int64), metas)
        assert (raw is None) == (rank != 0)
        if rank == 0:
            m, recv_ll, recv_ix = raw
            assert [x[1] for x in m] == [len(s) for s in shards]
            for r in range(world):
                assert torch.equal(recv_ix[r][1, :len(shards[r])], torch.tensor(shards[r], dtype=torch.int64))
    if rank == 0:
        q.put({k: v.numpy().copy() for k, v in res.items()})
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    loci, _ = synth.config_loci("config3", n_loci=24)
    batch, _ = synth.pack_loci(loci)
    costs = shard.locus_costs(batch)
    shards = shard.shard_by_cost(costs, 2)
    assert sorted(shards[0] + shards[1]) == list(range(24))
    loads = [costs[s].sum() for s in shards]
    assert max(loads) / min(loads) < 1.25                      # greedy LPT balances cells
    sizes = [int(x) for x in np.diff(batch.ll_off)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, shards, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert list(got.keys()) == list(range(24))                 # rank 0 holds every locus, in locus order
    for l in range(24):
        assert np.array_equal(got[l], _fake_ll(l, sizes[l]).numpy())


def test_shard_by_cost_edge_cases():
    assert shard.shard_by_cost([5.0], 4) == [[0], [], [], []]
    s = shard.shard_by_cost([1, 1, 1, 1, 10], 2)
    assert sorted(map(tuple, s)) == sorted([(4,), (0, 1, 2, 3)])


def _og_worker(rank, world, port, sizes, shards, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shards[rank]
    ll = torch.cat([_fake_ll(l, sizes[l]) for l in mine] + [torch.zeros(0, dtype=torch.float64)])
    og = shard.OrderedGather([sizes[l] for l in mine], mine, torch.device("cpu"))
    for _ in range(2):                                   # resident: metadata once, payload every step
        glob = og(ll)
        assert (glob is None) == (rank != 0)
    if rank == 0:
        q.put((glob.numpy().copy(), og.global_off.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ordered_gather_world2_product_sharding():
    """config 4's exchange step: the SAME catalogue cost-sharded with the product code, every rank's
    blocks gathered to rank 0 into GLOBAL locus order (stand-in LL values, no GPU)."""
    loci, _ = synth.config_loci("config3", n_loci=30)
    batch, _ = synth.pack_loci(loci)
    costs = shard.locus_time_costs(batch)
    assert (costs > 0).all()
    shards = shard.shard_by_cost(costs, 2)
    loads = [costs[s].sum() for s in shards]
    assert max(loads) / min(loads) < 1.25
    sizes = [int(x) for x in np.diff(batch.ll_off)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_og_worker, args=(r, 2, port, sizes, shards, q)) for r in range(2)]
    for p in procs:
        p.start()
    glob, goff = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(goff, batch.ll_off)                  # global layout == the single-GPU plan's layout
    want = np.concatenate([_fake_ll(l, sizes[l]).numpy() for l in range(30)])
    assert np.array_equal(glob, want)


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself; the dry run drives the
    bench's own sharding + ordered-gather code on gloo with a stand-in execute and prints ONE line."""
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--loci", "36", "--steps", "2", "--warmup", "1"])
    assert rc == 0, err
    assert len(lines) == 1
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["scaling"] == "strong" and ln["value"] is None
    assert ln["total_loci"] == 36 and ln["gathered_loci"] == 36 and ln["misplaced_loci"] == 0 and ln["order_ok"] is True
    assert sum(ln["shard_sizes"]) == 36 and min(ln["shard_sizes"]) > 0
    assert ln["other"] == {"scaling": "weak", "total_loci": 72}


def test_bench_end_to_end_ranks_dry_run():
    """`bench.py --gpus N --end-to-end` (round 6): one step = every rank's drop-in call on the raw alignments of its shard +
    the ordered gather of the per-read rows.  Dry run: the shard's raw loci are generated and packed into ONE contiguous output
    array per rank, a stand-in fills it, the gather must put every locus at its global place; the line names every rank's
    host-thread budget = the library's rule under the launcher's LOCAL_WORLD_SIZE (8 ranks share the host's cores)."""
    from longtr_amd import _lib
    full = _lib.lib().ltr_host_threads_rule(1)
    for n in (2, 8):
        rc, lines, err = _run_bench(["--gpus", str(n), "--dry-run", "--end-to-end", "--loci", "48", "--steps", "2", "--warmup", "1"], timeout=600)
        assert rc == 0, err[-2000:]
        assert len(lines) == 1
        ln = lines[0]
        assert ln["n_gpus"] == n and ln["end_to_end"] is True and ln["dry_run"] is True and ln["value"] is None
        assert ln["gather_check"] == {"gathered_loci": 48, "misplaced_loci": 0, "order_ok": True}
        assert ln["config"]["total_loci"] == 48
        assert ln["host_threads_per_rank"] == [max(1, full // n)] * n, (ln["host_threads_per_rank"], full)
        assert len(ln["rank_ms_per_step"]) == n


def test_bench_under_torch_distributed_run_dry_run():
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- with the dry-run stand-in: bench.py is then ONE rank per
    process (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) and rank 0 alone prints the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--dry-run", "--loci", "36"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["scaling"] == "strong" and ln["dry_run"] is True
    assert ln["gathered_loci"] == 36 and ln["misplaced_loci"] == 0 and ln["order_ok"] is True


@pytest.mark.parametrize("n", [4, 8])
def test_bench_driver_launch_line_at_4_and_8_ranks_dry_run(n):
    """The driver's N = 4 and N = 8 runs (one rank per GPU of an 8-GPU node) have never met hardware here: the same launch line
    with the dry-run stand-in, so that the sharding of BASELINE config 4 (10 000 loci cost-balanced over the ranks -- here a
    small catalogue), the ordered gather and the line's layout check run at the world sizes the driver uses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    loci = 12 * n
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--dry-run", "--loci", str(loci)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    raw = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(raw) == 1 and len(raw[0]) < 4096
    ln = json.loads(raw[0])
    assert ln["n_gpus"] == n and ln["world_size"] == n and ln["backend"] == "gloo" and ln["scaling"] == "strong"
    assert ln["gathered_loci"] == loci and ln["misplaced_loci"] == 0 and ln["order_ok"] is True
    assert ln["config"]["parallelism"] == f"loci-shard x{n}"


LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
             "data", "config", "roofline", "loci_per_s", "library", "detail"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "kernel_cells", "ops_per_cell"}


@pytest.mark.parametrize("n", [1, 2])
def test_bench_line_is_one_compact_json_line(n):
    """The driver keeps a bounded tail of stdout and parses the bench line from it: ONE line starting with `{`, under 4 KB, with
    the contract's keys (+ roofline; + cpu_baseline at N = 1; + backend / world_size as torch.distributed reports them at N > 1).
    The dry run assembles the line exactly as a GPU run does (stand-in per-launch table of 45 classes, a real CPU baseline sample);
    the per-class table and the other long blocks go to the detail file named in the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--dry-run", "--loci", "40", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    raw = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(raw) == 1 and p.stdout.rstrip().splitlines()[-1] == raw[0]        # and it is the LAST line of stdout
    assert len(raw[0]) < 4096
    ln = json.loads(raw[0])
    assert LINE_KEYS <= set(ln) and ROOFLINE_KEYS <= set(ln["roofline"])
    assert ln["steps"] == 2 and ln["warmup"] == 1 and ln["n_gpus"] == n
    if n == 1:
        assert {"value", "unit", "cores", "kind", "sample"} <= set(ln["cpu_baseline"]) and ln["cpu_baseline"]["cores"] == 1
    else:
        assert ln["backend"] == "gloo" and ln["world_size"] == 2 and "cpu_baseline" not in ln
    detail = json.load(open(os.path.join(root, ln["detail"]) if not os.path.isabs(ln["detail"]) else ln["detail"]))
    assert len(detail["kernels"]) == 45 and detail["value"] == ln["value"]


def test_fit_line_drops_optional_keys_only():
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    line = {k: 1.23456789012 for k in b.REQUIRED_KEYS}
    line["config"] = {"workload": "x" * 1000, "v": 0.123456789012}
    for k in b.OPTIONAL_ORDER:
        line[k] = {"blob": "y" * 200, "n": list(range(100))}
    out = b.fit_line(line)
    assert len(out) <= b.LINE_LIMIT
    d = json.loads(out)
    assert set(b.REQUIRED_KEYS) <= set(d)
    assert d["value"] == 1.23456789012 and d["config"]["v"] == 0.123457 and len(d["config"]["workload"]) == 240
    assert "weak_scaling" not in d                      # the first to go


def test_bench_rejects_world_size_mismatch():
    rc, lines, _ = _run_bench(["--gpus", "4", "--dry-run", "--loci", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and not lines


# ---- the N > 1 harness must not hang and must say what it ran on (round 5) --------------------------------------------------
def test_bench_parent_deadline_kills_hung_ranks():
    """Ranks that never come up (here: they sleep before anything else) are killed at --deadline-s and the parent exits 124
    instead of waiting for ever: the first real 8-GPU run is also the first test of RCCL with more than one rank."""
    import time
    t0 = time.monotonic()
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--loci", "12", "--steps", "1", "--warmup", "0", "--deadline-s", "4"],
                                env_extra={"LTR_BENCH_TEST_HANG": "1"}, timeout=120)
    assert rc == 124, err[-1500:]
    assert lines == [] and "deadline reached" in err
    assert time.monotonic() - t0 < 60


def test_bench_parent_retries_with_gloo_when_ranks_die_before_the_first_step():
    """A rank exits non-zero before any rank finished a step while the exchange was allowed to use RCCL: the parent (which never
    touched a GPU) starts FRESH ranks with --exchange gloo, and the line says so.  A failure after the first step is not retried."""
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--loci", "24", "--steps", "1", "--warmup", "0"],
                                env_extra={"LTR_BENCH_TEST_FAIL": "before_first_step"}, timeout=600)
    assert rc == 0, err[-1500:]
    assert "starting fresh ranks with --exchange gloo" in err
    assert len(lines) == 1
    ln = lines[0]
    assert ln["backend"].startswith("gloo (fallback: launcher retry") and ln["world_size"] == 2
    assert ln["gathered_loci"] == 24 and ln["misplaced_loci"] == 0 and ln["order_ok"] is True
    # with the host exchange asked for from the start there is nothing to fall back to: the failure is final
    rc2, lines2, err2 = _run_bench(["--gpus", "2", "--dry-run", "--loci", "24", "--steps", "1", "--warmup", "0", "--exchange", "gloo"],
                                   env_extra={"LTR_BENCH_TEST_FAIL": "before_first_step", "LTR_BENCH_TEST_FAIL_ALWAYS": "1"}, timeout=600)
    assert rc2 == 7 and lines2 == [] and "fresh ranks" not in err2


def test_bench_rank_falls_back_to_gloo_when_the_rccl_probe_fails():
    """Under the driver's launch line there is no parent of ours: every rank asks a CHILD process whether RCCL comes up between
    the ranks (here it cannot: no GPU), the ranks agree on the answer over the gloo control group, and the exchange runs on gloo
    with `backend` saying why.  --exchange nccl makes the same answer an error (exit 5) on every rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["LTR_BENCH_TEST_PROBE"] = "run"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--dry-run", "--loci", "24"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    ln = lines[0]
    assert ln["backend"].startswith("gloo (fallback: RCCL probe: rank 0:") and "no GPU visible" in ln["backend"]
    assert ln["gathered_loci"] == 24 and ln["misplaced_loci"] == 0 and ln["order_ok"] is True
    assert ln["devices"] == ["", ""] and ln["distinct_devices"] == 0       # (a dry run holds no GPU)
    # a probe that hangs costs its time-out, not the run
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--loci", "24", "--steps", "1", "--warmup", "0"],
                                env_extra={"LTR_BENCH_TEST_PROBE": "hang", "LTR_BENCH_PROBE_TIMEOUT_S": "3"}, timeout=600)
    assert rc == 0 and len(lines) == 1, err[-1500:]
    assert "RCCL probe did not finish within 3 s" in lines[0]["backend"]
    # RCCL demanded: every rank leaves with 5 before the first step, and the parent's one retry is the host exchange
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--loci", "24", "--steps", "1", "--warmup", "0", "--exchange", "nccl"], timeout=600)
    assert "--exchange nccl but RCCL probe" in err and "starting fresh ranks with --exchange gloo" in err
    assert rc == 0 and len(lines) == 1 and lines[0]["backend"].startswith("gloo (fallback: launcher retry")
