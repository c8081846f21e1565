#!/usr/bin/env python3
"""bench.py -- read x haplotype DP cells/s (+ loci/s) on BASELINE configs 3 / 4.

  python bench.py --gpus N --steps K --warmup W
      N == 1: BASELINE config 3 (10k synthetic loci, 30x, TR 20-1000 bp) on one MI355X.
      N  > 1: BASELINE config 4 -- the SAME 10k loci (seed 20250225) cost-sharded over N GPUs, one
              process per GPU.  Without WORLD_SIZE in the environment this script starts the N rank
              processes itself (children are spawned before anything touches the GPU); under
              `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one rank.

A "step" = one pass of the hot path over the rank's loci (every pooled read x candidate haplotype
DP, inputs already resident in HBM) + the one exchange step of the path: the gather of the per-locus
log-likelihood blocks to rank 0 in GLOBAL locus order (longtr_amd/shard.py::OrderedGather; N > 1
only).  Loci shard with no data-path collective.  The headline line is strong scaling (total work
fixed = config 4); the weak-scaling figure (every rank its own 10k loci) is measured in the same run
and reported under "weak_scaling".

Rank 0 prints ONE compact JSON line (< 4 KB: fit_line) as the LAST line of stdout and writes everything else to
bench_detail.json next to this script (path under "detail").  `roofline` prices the dominant DP launch against the FP64
vector-ALU peak at the FP64 add/max the recurrence needs (11 per cell for a symmetric indel model, 13 otherwise); this path
is a scalar max-plus recurrence, VALU-bound by ~3 orders of magnitude over its HBM traffic, and the HBM
figure sits next to it (`roofline.hbm`).  `cpu_baseline` times the reference's own align_seq_to_hap (oracle/_ref, built
from the reference sources in the dev container; the C port when that build is absent) on a bounded sample of the same
workload, single thread; `cpu_baseline_ncores` the same on every host core (loci sharded over processes, the reference's own
scale-out model, README.md:78-82).  After the timed region the LL buffer of the full pass is bit-compared with the oracle
on a strip-class-stratified sample (`oracle_check`) and, for N > 1, with a single-GPU recomputation on rank 0
(`single_gpu_check`); for N > 1 the line carries `backend` / `world_size` as torch.distributed reports them.

In the detail file (N == 1): `end_to_end` = ltr_calc_hap_aln_probs on raw alignments, host to host; `kernels` = every launch
class of the pass with its own time; `roofline_most_pairs` = the class that holds most pairs (the packed kernels on
catalogue-shaped workloads); `neighbours` = the NW and seeded-stutter-path kernels on bounded batches of their own, each with
a roofline block.  `library.source_id` = hash of the sources + flags libltr_gpu.so is built from, matched against
profiles/<round>/pmc_traffic*.json before its counters are attached (`roofline.counters_from`).  Other workloads:
--workload catalogue | config3skew | config5 | config5hifi | config2.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = ["config2", "config3", "config3skew", "config5", "config5hifi", "catalogue"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--loci", type=int, default=None, help="loci of the workload (BASELINE config 3/4 = 10000)")
    ap.add_argument("--workload", default="config3", choices=WORKLOADS)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = config 4 (the same loci sharded); weak = every rank its own batch")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the additional weak-scaling measurement")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0, help="CPU baseline sample budget per core (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-timing oracle / single-GPU checks")
    ap.add_argument("--no-neighbours", action="store_true", help="skip the NW / seeded-stutter-path kernel measurements (N = 1 only)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the ltr_calc_hap_aln_probs (raw alignments) measurement")
    ap.add_argument("--end-to-end", action="store_true",
                    help="the timed step is the DROP-IN call itself: every rank runs ltr_calc_hap_aln_probs on the raw alignments of its shard "
                         "(pooling, trimming, planning, upload, DP, download, fan-out: seq_stutter_genotyper.cpp:514-563), then the ordered gather "
                         "of the per-read rows; the line carries every rank's host-thread budget")
    ap.add_argument("--host-threads", type=int, default=0, help="host-thread budget of every rank (0: the library's rule -- affinity mask, cgroup quota, LOCAL_WORLD_SIZE)")
    ap.add_argument("--e2e-loci", type=int, default=None, help="loci of the ltr_calc_hap_aln_probs measurement (default 6000; catalogue: 30000)")
    ap.add_argument("--pair-packing", type=int, default=-1,
                    help="ltr_ctx_set_pair_packing scheduling mode for A/B runs (-1 default; 3 no workgroup kernels; 4 exact kernels only)")
    ap.add_argument("--params", default=None, metavar="a,b,c,d,e,f,g",
                    help="the seven transitions of --alignment-params (HapAligner.h:111-119) instead of the workload's own; e.g. an asymmetric model")
    ap.add_argument("--debug", action="append", default=[], metavar="KEY=VALUE",
                    help="ltr_ctx_set_debug measurement switch, e.g. fan_lanes=1 (profiles/collect.sh: every launch on one stream)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)   # internal: one process of the N-core CPU baseline
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: gloo + CPU tensors and a stand-in for plan.execute (LL = f(locus id)); exercises the launcher, "
                         "the sharding and the ordered gather only -- prints no throughput (CPU test of the N > 1 path)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "nccl", "gloo"],
                    help="N > 1: backend of the gather of the per-locus results.  auto = RCCL ('nccl') when a probe group of all ranks "
                         "(a child process per rank, bounded in time) comes up and moves data, else gloo with host tensors -- the line says "
                         "which (SURVEY.md 8e: 1.5 MB per rank and step, the two are equivalent in cost)")
    ap.add_argument("--deadline-s", type=float, default=1500.0,
                    help="N > 1, started by this script: seconds after which the parent kills its rank processes and exits 124")
    ap.add_argument("--collective-timeout-s", type=float, default=120.0, help="N > 1: time-out of every torch.distributed group")
    ap.add_argument("--nccl-probe", default=None, metavar="PORT", help=argparse.SUPPRESS)   # internal: the RCCL probe child of one rank
    ap.add_argument("--one-gpu", action="store_true",
                    help="N > 1 on a single-GPU box (verification, not a measurement): every rank scores its shard on cuda:0 with the "
                         "real kernels, the exchange runs over gloo with host tensors; the line carries debug_one_gpu = true")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no WORLD_SIZE starts the N ranks itself
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(n, argv, extra_env, t_end):
    """One attempt: N children, one per GPU, rendezvous on 127.0.0.1.  Returns (exit code, reason); reason is "ok", "deadline"
    (children killed at t_end), "before_first_step" (a child failed and no rank had finished a step: nothing of the path ran, a
    different exchange backend may still work) or "failed" (a child failed later: a real error, never retried)."""
    stage_dir = tempfile.mkdtemp(prefix="ltr_ranks_")
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LTR_BENCH_STAGE_DIR=stage_dir, **extra_env)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, start_new_session=True))
    rc, reason = 0, "ok"

    def stop(ps, sig_kill=False):
        for q in ps:                           # exactly the processes started here (each leads a session of its own: its probe child goes with it)
            if q.poll() is None:
                try:
                    os.killpg(q.pid, 9 if sig_kill else 15)
                except OSError:
                    pass

    try:
        pending = list(procs)
        while pending:
            if time.monotonic() > t_end:
                rc, reason = 124, "deadline"
                print(f"bench.py: deadline reached with {len(pending)} of {n} ranks still running -- killing them", file=sys.stderr, flush=True)
                stop(pending)
                time.sleep(2.0)
                break
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    stepped = any(f.startswith("step_") for f in os.listdir(stage_dir))
                    reason = "failed" if stepped else "before_first_step"
                    print(f"bench.py: rank {procs.index(p)} exited {code} ({reason.replace('_', ' ')}) -- stopping the others", file=sys.stderr, flush=True)
                    stop(pending)              # one rank failed: the others would wait in a collective until its time-out
            time.sleep(0.05)
    finally:
        stop(procs, sig_kill=True)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        for f in os.listdir(stage_dir):
            os.unlink(os.path.join(stage_dir, f))
        os.rmdir(stage_dir)
    return rc, reason


def launch_ranks(args):
    """Runs in a parent that never touches the GPU.  Bounded: at --deadline-s the children are killed and the exit code is 124.
    If the ranks die before any of them finished a step while the exchange was allowed to use RCCL, FRESH children are started
    once with --exchange gloo (host gather; the parent holds no GPU state that could be stale) and their line says so."""
    t_end = time.monotonic() + args.deadline_s
    argv = sys.argv[1:]
    rc, reason = _run_ranks(args.gpus, argv, {}, t_end)
    if reason == "before_first_step" and args.exchange != "gloo" and not args.one_gpu:
        print(f"bench.py: starting fresh ranks with --exchange gloo (first attempt: exit {rc} before the first step)", file=sys.stderr, flush=True)
        rc, reason = _run_ranks(args.gpus, argv + ["--exchange", "gloo"],
                                {"LTR_BENCH_RETRY_REASON": f"launcher retry: a rank exited {rc} before the first step with --exchange {args.exchange}"}, t_end)
    return rc


def mark_stage(name):
    """Progress marker for the launching parent (launch_ranks): an empty file per rank and stage."""
    d = os.environ.get("LTR_BENCH_STAGE_DIR")
    if d:
        try:
            open(os.path.join(d, f"{name}_{os.environ.get('RANK', '0')}"), "w").close()
        except OSError:
            pass


def device_identity(index):
    """What tells two GPUs of a node apart, for the line: PCI address + UUID as the runtime reports them."""
    import torch
    p = torch.cuda.get_device_properties(index)
    pci = f"{getattr(p, 'pci_domain_id', 0):04x}:{getattr(p, 'pci_bus_id', 0):02x}:{getattr(p, 'pci_device_id', 0):02x}"
    return {"index": int(index), "pci": pci, "uuid": str(getattr(p, "uuid", "")), "name": p.name}


def nccl_probe(port):
    """Child process of ONE rank (--nccl-probe PORT): joins an RCCL group of all ranks' probe children on a port of its own
    and moves data through the two collectives the path issues.  Prints one JSON line; exit 0 = RCCL works between these ranks.
    A hang here costs the rank its probe's time-out, never the run: the rank kills this child."""
    import datetime
    import torch
    import torch.distributed as dist
    rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    out = {"ok": False, "rank": rank}               # ("ok" first: run_probe looks for the line that starts with it)
    try:
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible")
        if torch.cuda.device_count() <= lr:
            raise RuntimeError(f"LOCAL_RANK {lr} but {torch.cuda.device_count()} visible GPUs")
        dev = torch.device("cuda", lr)
        torch.cuda.set_device(dev)
        out["device"] = device_identity(lr)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=60), device_id=dev)
        x = torch.full((1 << 17,), float(rank + 1), dtype=torch.float64, device=dev)
        dist.all_reduce(x)
        recv = [torch.empty_like(x) for _ in range(world)] if rank == 0 else None
        dist.gather(x, recv, dst=0)
        torch.cuda.synchronize(dev)
        want = world * (world + 1) / 2.0
        good = float(x[0].item()) == want and (rank != 0 or all(float(r[-1].item()) == want for r in recv))
        if not good:
            raise RuntimeError("RCCL all_reduce / gather returned wrong values")
        out["ok"] = True
        dist.destroy_process_group()
    except Exception as e:                       # the answer is the exit code; the text goes into the line
        out["why"] = repr(e)[:200]
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 3


def run_probe(port, timeout_s):
    """This rank's probe child; (ok, reason, device identity)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("LTR_BENCH_STAGE_DIR", None)
    if os.environ.get("LTR_BENCH_TEST_PROBE") == "hang":         # (CPU test of the time-out)
        cmd = [sys.executable, "-c", "import time; time.sleep(3600)"]
    else:
        cmd = [sys.executable, os.path.abspath(__file__), "--nccl-probe", str(port)]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        txt, _ = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill()
        p.communicate()
        return False, f"RCCL probe did not finish within {timeout_s:.0f} s", None
    j = None
    for ln in reversed(txt.splitlines()):                   # (the runtime may print below the answer)
        if ln.startswith('{"ok"'):
            try:
                j = json.loads(ln)
                break
            except ValueError:
                pass
    if j is None:
        return False, f"RCCL probe exited {p.returncode} without an answer: {txt[-120:]!r}", None
    return bool(j.get("ok")) and p.returncode == 0, j.get("why", ""), j.get("device")


# ------------------------------------------------------------------------------------------------
# CPU baselines (reporting only; the reference / oracle is never the product path)
# ------------------------------------------------------------------------------------------------
def _sub_batch(batch, loci_ids, reads_per_locus=None):
    from longtr_amd import _abi
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


def _locus_cells(batch):
    rl, hl = np.diff(batch.read_off).astype(np.float64), np.diff(batch.hap_off).astype(np.float64)
    out = np.zeros(batch.n_loci)
    for l in range(batch.n_loci):
        m = rl[batch.locus_read_off[l]:batch.locus_read_off[l + 1]]
        n = hl[batch.locus_hap_off[l]:batch.locus_hap_off[l + 1]] - 60
        out[l] = float(m.sum() * n.sum())
    return out


def _time_cpu(sb, params):
    """(seconds, kind) of align_seq_to_hap over a packed sub-batch on one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    if ol.have_ref():
        _, secs = ol.ref_align_batch(sb, params)        # std::chrono around align_seq_to_hap only
        return secs, "reference"
    t0 = time.perf_counter()
    ol.oracle_align_batch(sb, params)
    return time.perf_counter() - t0, "port"


def cpu_worker(path):
    """One process of the N-core baseline: scores the loci of its shard one by one until it runs
    out of loci or of its time budget, prints one JSON line (seconds of align_seq_to_hap, nominal cells)."""
    from longtr_amd import _abi, synth
    z = np.load(path, allow_pickle=True)
    params = _abi.make_params([float(x) for x in z["params7"]], int(z["flank"]))
    deadline = float(z["deadline_s"])
    secs, cells, done, kind = 0.0, 0, 0, "port"
    for r, h in zip(z["reads"], z["haps"]):
        sb = _abi.PackedBatch([(list(r), list(h))])
        s1, kind = _time_cpu(sb, params)
        secs += s1
        cells += synth.nominal_cells(sb, params.indel_flank_len)
        done += 1
        if secs >= deadline:
            break
    print(json.dumps({"secs": secs, "cells": cells, "loci": done, "kind": kind}))


def host_cores():
    """Cores this process may use: the affinity mask, cut to the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baselines(batch, params, budget_s):
    from longtr_amd import synth
    cells = _locus_cells(batch)
    cum = np.cumsum(cells)
    # calibrate on a small prefix (~1e8 cells) so that the samples below last ~budget_s on THIS host
    k0 = max(1, int(np.searchsorted(cum, 1.0e8)))
    sb0 = _sub_batch(batch, range(min(k0, batch.n_loci)))
    s0, _ = _time_cpu(sb0, params)
    est_rate = max(synth.nominal_cells(sb0, params.indel_flank_len), 1) / max(s0, 1e-6)
    # ---- one thread: a prefix of the workload worth ~budget_s ----
    k1 = max(1, int(np.searchsorted(cum, est_rate * budget_s)))
    sb = _sub_batch(batch, range(min(k1, batch.n_loci)))
    nominal = synth.nominal_cells(sb, params.indel_flank_len)
    secs, kind = _time_cpu(sb, params)
    one = {"value": nominal / secs, "unit": "cells/s", "cores": 1, "kind": kind, "cpu": cpu_model(),
           "sample": f"first {min(k1, batch.n_loci)} loci of the workload ({nominal:.3e} nominal cells, {secs:.1f} s, "
                     f"align_seq_to_hap only, 1 thread)"}
    # ---- every host core (at most 64 processes): loci sharded over processes, the reference's own
    # scale-out model; every process stops at 1.5 x budget whatever the contention does to its rate ----
    ncores = min(host_cores(), 64)
    many = None
    try:
        from longtr_amd import shard
        kn = max(ncores, int(np.searchsorted(cum, est_rate * budget_s * ncores)))
        kn = min(kn, batch.n_loci)
        shards = shard.shard_by_cost(cells[:kn], ncores)
        tmp = tempfile.mkdtemp(prefix="ltr_cpu_")
        paths = []
        p7 = np.asarray(params.as_tuple()[:7], dtype=np.float64)
        for w, ids in enumerate(shards):
            if not ids:
                continue
            reads = np.empty(len(ids), dtype=object)
            haps = np.empty(len(ids), dtype=object)
            for i, l in enumerate(ids):
                r0, r1 = int(batch.locus_read_off[l]), int(batch.locus_read_off[l + 1])
                reads[i] = [batch.read_bytes[batch.read_off[r]:batch.read_off[r + 1]].tobytes() for r in range(r0, r1)]
                haps[i] = [batch.hap_bytes[batch.hap_off[h]:batch.hap_off[h + 1]].tobytes()
                           for h in range(batch.locus_hap_off[l], batch.locus_hap_off[l + 1])]
            path = os.path.join(tmp, f"shard{w}.npz")
            np.savez(path, reads=reads, haps=haps, params7=p7, flank=params.indel_flank_len, deadline_s=1.5 * budget_s)
            paths.append(path)
        t0 = time.perf_counter()
        running = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path],
                                    stdout=subprocess.PIPE, text=True) for path in paths]
        outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in running]
        wall = time.perf_counter() - t0
        for path in paths:
            os.unlink(path)
        os.rmdir(tmp)
        secs_n = max(o["secs"] for o in outs)
        tot = sum(o["cells"] for o in outs)
        many = {"value": tot / secs_n, "unit": "cells/s", "cores": len(running), "kind": kind, "cpu": cpu_model(),
                "host_cores": host_cores(),
                "sample": f"{sum(o['loci'] for o in outs)} of the first {kn} loci of the workload, cost-sharded over {len(running)} processes "
                          f"({tot:.3e} nominal cells, slowest process {secs_n:.1f} s of align_seq_to_hap, {wall:.1f} s wall incl. start-up)"}
    except Exception as e:                               # reporting only
        many = {"error": repr(e)}
    return one, many


# ------------------------------------------------------------------------------------------------
# post-timing checks (outside the timed region; the oracle is the checker, never the product)
# ------------------------------------------------------------------------------------------------
def oracle_check(batch, ll, params):
    """Bit-compare a strip-class-stratified sample (>= 200 loci, 3 pooled reads x all haplotypes each)
    of the full pass with the CPU oracle (tests/parity_util.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_util
    return parity_util.stratified_oracle_check(batch, ll, params)


LINE_LIMIT = 4000          # bytes of the one stdout line (the driver keeps a bounded tail of stdout)
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")
# dropped first when a line would not fit (least needed first); the required keys never are
OPTIONAL_ORDER = ("weak_scaling", "strong_scaling", "speedup_vs_cpu_ncores", "cpu_baseline_ncores", "end_to_end_frac_of_resident",
                  "single_gpu_check", "oracle_check", "library", "loci_per_s_end_to_end", "speedup_vs_cpu_1thread")


def fit_line(line, limit=LINE_LIMIT):
    """json.dumps(line) within `limit` bytes: floats to 6 significant digits in nested blocks, long strings cut, optional
    keys dropped in OPTIONAL_ORDER.  Everything dropped is in the detail file."""
    def rnd(o, depth=0):
        if isinstance(o, float):
            return float(f"{o:.6g}") if depth > 1 else o
        if isinstance(o, dict):
            return {k: rnd(v, depth + 1) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v, depth + 1) for v in o]
        if isinstance(o, str) and len(o) > 240:
            return o[:237] + "..."
        return o
    out = rnd(line)
    for k in OPTIONAL_ORDER:
        if len(json.dumps(out)) <= limit:
            break
        out.pop(k, None)
    return json.dumps(out)


def write_detail(obj, args):
    """Everything the stdout line leaves out (per-class kernel table, neighbours, end-to-end timers, sample descriptions)
    goes to bench_detail.json next to this script (the temp dir when the tree is read-only).  Returns the path written."""
    name = "bench_detail.json" if args.workload == "config3" and args.pair_packing == -1 else f"bench_detail_{args.workload}.json"
    if args.dry_run:
        name = f"bench_detail_dryrun{args.gpus}.json"
    for d in (ROOT, tempfile.gettempdir()):
        path = os.path.join(d, name)
        try:
            with open(path, "w") as f:
                json.dump(obj, f, indent=1)
            return os.path.relpath(path, ROOT) if d == ROOT else path
        except OSError:
            continue
    return None


class ExchangeUnavailable(RuntimeError):
    """--exchange nccl and RCCL does not come up between the ranks (every rank raises it: the verdict is agreed on first)."""


def setup_exchange(rank, local_rank, world, exchange, one_gpu, dry, timeout_s):
    """The process groups of an N > 1 run.  Returns (compute device, exchange device, exchange group, backend text, devices).

    Control plane (barriers, sizes, verdicts): ALWAYS the default gloo group, with a time-out -- it comes up without the GPU.
    The data exchange (the gather of the per-locus results) runs on an RCCL group of its own when RCCL works between these
    ranks, which a CHILD process per rank finds out first (run_probe): a hang or an abort inside RCCL then costs that child,
    not the run.  The ranks agree on every verdict over gloo, so they all take the same branch.  exchange: "auto" falls back
    to gloo with host tensors and says why in the backend text; "nccl" raises ExchangeUnavailable; "gloo" asks for the host path."""
    import datetime
    import torch
    import torch.distributed as dist
    tmo = datetime.timedelta(seconds=timeout_s)
    fallback_why = os.environ.get("LTR_BENCH_RETRY_REASON")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)

    def agreed(ok):
        v = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        return bool(int(v[0]))

    use_rccl = False
    if exchange != "gloo" and not one_gpu and (not dry or exchange == "nccl" or bool(os.environ.get("LTR_BENCH_TEST_PROBE"))):
        port = torch.tensor([_free_port() if rank == 0 else 0], dtype=torch.int64)
        dist.broadcast(port, src=0)
        ok, why, _ = run_probe(int(port[0]), float(os.environ.get("LTR_BENCH_PROBE_TIMEOUT_S", "150")))
        whys = [None] * world
        dist.all_gather_object(whys, None if ok else (why or "failed"))
        use_rccl = agreed(ok)
        if not use_rccl:
            fallback_why = "RCCL probe: " + "; ".join(f"rank {r}: {w}" for r, w in enumerate(whys) if w)[:300]
            if exchange == "nccl":
                raise ExchangeUnavailable(fallback_why)
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    xgroup = None
    if use_rccl:
        mine_ok = False
        try:
            xgroup = dist.new_group(backend="nccl", timeout=tmo, device_id=dev)
            x = torch.ones(8, dtype=torch.float64, device=dev)
            dist.all_reduce(x, group=xgroup)
            torch.cuda.synchronize(dev)
            mine_ok = float(x[0].item()) == float(world)
        except Exception as e:
            fallback_why = f"RCCL group of the ranks: {e!r}"[:300]
        if not agreed(mine_ok):
            xgroup, use_rccl = None, False
            fallback_why = fallback_why or "RCCL group failed on another rank"
            if exchange == "nccl":
                raise ExchangeUnavailable(fallback_why)
    backend_txt = "nccl" if use_rccl else ("gloo" if not fallback_why else f"gloo (fallback: {fallback_why})")
    ident = None if dry else device_identity(local_rank)
    devices = [None] * world
    dist.all_gather_object(devices, f"{ident['pci']} {ident['uuid']}" if ident else "")
    return dev, (dev if use_rccl else torch.device("cpu")), xgroup, backend_txt, devices


def expected_global_offsets(batch, gids, world, xdev, group=None):
    """The global LL layout derived WITHOUT OrderedGather: every rank contributes (global id, LL size) of its loci, taken from
    its own packed batch; sorted by id, running sum.  Compared with OrderedGather.global_off.  (Tensor collectives on the
    exchange device only -- no pickled objects: the same calls work on gloo and on RCCL.)"""
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(np.stack([np.asarray(gids, dtype=np.int64), np.diff(batch.ll_off).astype(np.int64)])).to(xdev)
    n_mine = torch.tensor([mine.shape[1]], dtype=torch.int64, device=xdev)
    counts = [torch.zeros(1, dtype=torch.int64, device=xdev) for _ in range(world)]
    dist.all_gather(counts, n_mine, group=group)
    n_max = max(int(c.item()) for c in counts)
    padded = torch.full((2, max(n_max, 1)), -1, dtype=torch.int64, device=xdev)
    padded[:, :mine.shape[1]] = mine
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    allp = np.concatenate([p[:, :int(c.item())].cpu().numpy() for p, c in zip(parts, counts)], axis=1)
    order = np.argsort(allp[0], kind="stable")
    off = np.zeros(allp.shape[1] + 1, dtype=np.int64)
    off[1:] = np.cumsum(allp[1][order])
    return allp[0][order], off


# ------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker)
        return 0
    if args.nccl_probe:
        return nccl_probe(int(args.nccl_probe))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    import datetime
    import torch
    import torch.distributed as dist
    dry = args.dry_run
    if os.environ.get("LTR_BENCH_TEST_HANG"):                      # (CPU test of the launcher's deadline)
        time.sleep(3600)
    if args.one_gpu:
        local_rank = 0
    xgroup, backend_txt, devices, xdev = None, None, None, torch.device("cpu")
    if world > 1:
        try:
            dev, xdev, xgroup, backend_txt, devices = setup_exchange(rank, local_rank, world, args.exchange, args.one_gpu, dry, args.collective_timeout_s)
        except ExchangeUnavailable as e:
            print(f"bench.py: --exchange nccl but {e}", file=sys.stderr, flush=True)
            return 5
    elif dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    mark_stage("init")
    if os.environ.get("LTR_BENCH_TEST_FAIL") == "before_first_step" and (args.exchange != "gloo" or os.environ.get("LTR_BENCH_TEST_FAIL_ALWAYS")):   # (CPU test of the launcher's retry)
        return 7

    def dev_sync():
        if not dry:
            torch.cuda.synchronize(dev)

    from longtr_amd import _abi, _lib, shard, synth
    ont = args.workload in ("config5", "config5hifi")
    params = _abi.make_params(synth.ONT_PARAMS) if ont else _abi.default_params()
    if args.params:
        params = _abi.make_params(tuple(float(x) for x in args.params.split(",")))
    # Every rank generates ITS loci only (every locus has a generator of its own, synth._locus_rng): N = 1 the whole
    # configuration; N > 1, strong scaling: the catalogue is cost-sharded from the generator's locus headers
    # (repeat length, alleles, reads -- no locus is generated for that), then each rank draws its shard.
    strong = (args.scaling == "strong") or world == 1
    n_total = args.loci if args.loci is not None else synth._DEFAULT_N[args.workload]
    gen_workers = None if world == 1 else max(1, host_cores() // world)
    t_gen = time.perf_counter()
    shards = None
    if world > 1 and strong:
        headers = synth.config_headers(args.workload, seed=synth.CONFIG_SEED, n_loci=n_total)
        shards = shard.shard_by_cost(shard.header_time_costs(headers, params.indel_flank_len, world=world), world)
        my_ids = shards[rank]
        loci, desc = synth.config_loci(args.workload, seed=synth.CONFIG_SEED, n_loci=n_total, ids=my_ids, workers=gen_workers)
        id_base = 0
    else:
        # N = 1, or weak scaling: rank r scores a whole configuration of its own (seed + r)
        my_ids = list(range(n_total))
        loci, desc = synth.config_loci(args.workload, seed=synth.CONFIG_SEED + rank, n_loci=n_total, workers=gen_workers)
        id_base = rank * n_total
    full, _ = synth.pack_loci(loci)                     # this rank's loci (host side)
    t_gen = time.perf_counter() - t_gen
    ctx = None if dry else _lib.Context(local_rank, params)
    if ctx is not None and args.pair_packing != -1:
        ctx.set_pair_packing(args.pair_packing)
    for kv in (args.debug if ctx is not None else []):
        k, v = kv.split("=", 1)
        ctx.set_debug(k, float(v))
    info = {"arch": "dry-run", "n_cu": 0, "clock_mhz": 0} if dry else ctx.device_info()
    stream = None if dry else torch.cuda.current_stream(dev).cuda_stream

    if ctx is not None and args.host_threads > 0:
        ctx.set_host_threads(args.host_threads)
    if args.end_to_end:
        return end_to_end_ranks(args, ctx, dry, rank, world, strong, my_ids, id_base, n_total, gen_workers, params, desc, dev, xdev, xgroup,
                                backend_txt, devices, info, dev_sync)

    class DryPlan:
        """--dry-run stand-in for a resident plan: LL element k of global locus g is -(g + 1) - k/1024."""
        def __init__(self, batch, gids):
            self.ll_size, self.num_pairs, self.input_bytes = batch.ll_size, batch.ll_size, 0.0
            self.cells = float(synth.nominal_cells(batch, params.indel_flank_len))
            sizes = np.diff(batch.ll_off)
            self.fake = torch.from_numpy(np.concatenate([-(float(g) + 1.0) - np.arange(int(sz)) / 1024.0 for g, sz in zip(gids, sizes)]
                                                        + [np.zeros(0)]))

        def close(self):
            pass

    def make_run(batch, gids):
        """Rank-local resident plan over `batch` (global locus ids `gids`) + the ordered gather of its results."""
        t0 = time.perf_counter()
        # pack + H2D: inputs resident in HBM before timing
        plan = DryPlan(batch, np.asarray(gids, dtype=np.int64)) if dry else ctx.plan(batch)
        t_plan = time.perf_counter() - t0
        out = torch.empty(max(plan.ll_size, 1), dtype=torch.float64, device=dev)
        og = None
        if world > 1:
            og = shard.OrderedGather(np.diff(batch.ll_off), np.asarray(gids, dtype=np.int64), xdev, group=xgroup)
        return dict(batch=batch, plan=plan, out=out, og=og, t_plan=t_plan, ids=list(gids), glob=None)

    def step(run):
        if dry:
            run["out"][:run["plan"].ll_size] = run["plan"].fake
        else:
            run["plan"].execute(run["out"].data_ptr(), stream)
        if run["og"] is not None:
            run["glob"] = run["og"](run["out"] if xdev == dev else run["out"].to(xdev))   # rank 0 now holds every locus, in global locus order

    def timed(run):
        for _ in range(args.warmup):
            step(run)
        dev_sync()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(run)
        dev_sync()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        mark_stage("step")
        c = torch.tensor([run["plan"].cells, float(run["batch"].n_loci), float(run["plan"].num_pairs), el], dtype=torch.float64)
        if world > 1:
            tmax = c[3:].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            c[3] = tmax[0]
        cells, nl, npairs, el = (float(x) for x in c.tolist())
        return dict(elapsed=el, cells=cells, loci=nl, pairs=npairs)

    # ---- the headline measurement -------------------------------------------------------------
    run = make_run(full, np.asarray(my_ids, dtype=np.int64) + id_base)
    res = timed(run)
    plan, batch = run["plan"], run["batch"]
    # the LAST step of the timed region between the plan's own HIP events (first launch .. last launch done, on its launch stream):
    # when the plan is one launch this IS that kernel's duration, taken from the very passes ms_per_step is taken from
    last_ms, last_launches = (None, None) if dry else plan.last_kernel_ms()
    wg_pass = None if dry else ctx.wg_first_pass()      # (mode, pairs the first pass of the workgroup classes could not finish, pairs it scored)

    # per-launch device times (HIP events on the launch stream): extra, untimed passes so that reading
    # the events never sits inside the timed region
    kms, kms2 = [], []
    if not dry:
        plan.set_timing(1)                       # every launch as it is launched (one stream)
        for _ in range(min(3, max(1, args.steps))):
            plan.execute(run["out"].data_ptr(), stream)
            kms.append(plan.kernel_stats())
        plan.set_timing(2)                       # the multi-width one-wave launch class by class (detail table only)
        for _ in range(2):
            plan.execute(run["out"].data_ptr(), stream)
            kms2.append(plan.kernel_stats())
        plan.set_timing(0)

    # ---- N > 1: the other scaling curve, same run ----------------------------------------------
    other = None
    if world > 1 and not args.no_weak:
        if strong:       # weak: every rank a whole configuration of its own
            wl = synth.config_loci(args.workload, seed=synth.CONFIG_SEED + rank, n_loci=n_total, workers=gen_workers)[0]
            run2 = make_run(synth.pack_loci(wl)[0], np.arange(n_total, dtype=np.int64) + rank * n_total)
        else:            # strong: the rank's shard of configuration `seed`
            headers = synth.config_headers(args.workload, seed=synth.CONFIG_SEED, n_loci=n_total)
            ids2 = shard.shard_by_cost(shard.header_time_costs(headers, params.indel_flank_len, world=world), world)[rank]
            wl = synth.config_loci(args.workload, seed=synth.CONFIG_SEED, n_loci=n_total, ids=ids2, workers=gen_workers)[0]
            run2 = make_run(synth.pack_loci(wl)[0], np.asarray(ids2, dtype=np.int64))
        r2 = timed(run2)
        other = {"scaling": "weak" if strong else "strong", "value": r2["cells"] * args.steps / r2["elapsed"], "unit": "cells/s",
                 "ms_per_step": r2["elapsed"] / args.steps * 1e3, "loci_per_s": r2["loci"] * args.steps / r2["elapsed"],
                 "total_loci": int(r2["loci"]), "total_cells": r2["cells"]}
        run2["plan"].close()
        del wl, run2

    # ---- checks, outside the timed region ---------------------------------------------------------
    checks = {}
    exp_ids = exp_off = None
    if world > 1:
        exp_ids, exp_off = expected_global_offsets(batch, run["ids"], world, xdev, xgroup)

    def offsets_ok(og):
        return bool(np.array_equal(exp_ids, np.arange(len(exp_ids))) and np.array_equal(og.global_off, exp_off))

    if dry:
        # the gathered vector must hold every locus of the catalogue at its place (strong) / every rank's block (weak)
        step(run)
        if rank == 0:
            glob = (run["glob"] if world > 1 else run["out"][:plan.ll_size]).numpy()
            off = run["og"].global_off if world > 1 else batch.ll_off
            bad = sum(int(not np.array_equal(glob[off[g]:off[g + 1]], -(float(g) + 1.0) - np.arange(int(off[g + 1] - off[g])) / 1024.0))
                      for g in range(len(off) - 1))
            checks["dry_run_gather"] = {
                "total_loci": int(res["loci"]), "gathered_loci": len(off) - 1, "misplaced_loci": bad,
                "order_ok": bool(len(off) - 1 == (n_total if strong else n_total * world)) and (world == 1 or offsets_ok(run["og"]))}
        # a stand-in per-launch table of a realistic size (45 classes), so that the line below is assembled exactly as on a GPU
        kms = [[{"strip_width": 4 + k % 17, "lanes_per_pair": 2 << (k % 6), "family": ("packed", "one-per-wave", "workgroup", "exact")[k % 4],
                 "pairs": 1000 + k, "cells": 1.0e9 + k, "ms": 1.0 + 0.01 * k} for k in range(45)]]
    if not args.no_verify and not dry:
        # rank 0 holds the global, locus-ordered LL vector of the last step (N == 1: its own buffer)
        if world > 1:
            step(run)
            dev_sync()
        if rank == 0:
            if world > 1 and strong:
                glob = run["glob"].cpu().numpy()
                goff = run["og"].global_off
                order_ok = bool(len(goff) - 1 == n_total) and offsets_ok(run["og"])   # layout re-derived from every rank's own batch
                # bits against a single-GPU recomputation: every 8th locus of the catalogue, generated and scored by rank 0 alone
                ids = list(range(0, n_total, 8))
                sl, _ = synth.config_loci(args.workload, seed=synth.CONFIG_SEED, n_loci=n_total, ids=ids)
                sb, _ = synth.pack_loci(sl)
                ll1, _ = ctx.align_batch(sb)
                mism = pairs = 0
                sub = np.zeros(sb.ll_size)
                for k, l in enumerate(ids):
                    a = ll1[sb.ll_off[k]:sb.ll_off[k + 1]]
                    b = glob[goff[l]:goff[l + 1]]
                    pairs += a.size
                    mism += int((a.view(np.uint64) != b.view(np.uint64)).sum()) if a.size == b.size else a.size
                    if a.size == b.size:
                        sub[sb.ll_off[k]:sb.ll_off[k + 1]] = b
                rank_of = np.zeros(n_total, dtype=np.int64)
                for r in range(world):
                    rank_of[shards[r]] = r
                checks["single_gpu_check"] = {"order_ok": order_ok, "loci": len(ids), "checked_pairs": int(pairs), "mismatches": int(mism),
                                              "ranks_covered": int(len(set(rank_of[ids].tolist())))}
                checks["oracle_check"] = oracle_check(sb, sub, params)      # the gathered bits of those loci against the CPU oracle
            elif world == 1:
                ll_host = run["out"][:plan.ll_size].cpu().numpy()
                checks["oracle_check"] = oracle_check(batch, ll_host, params)

    if rank == 0:
        t7 = params.as_tuple()
        sym = (t7[1] == t7[3]) and (t7[5] == t7[6])
        symtxt = "true" if sym else "false"
        clock_hz = info["clock_mhz"] * 1e6
        peak = (info["n_cu"] * 64 * clock_hz / 1e12) or 1.0       # FP64 add/max lane-ops/s: 4 SIMD x 16 lanes/clk per CU
        all_ms = float(np.mean([sum(k["ms"] for k in s) for s in kms]))
        # rocprofv3 --pmc summaries of the latest round (profiles/): HBM traffic per launch, VALU issue share.  They
        # are counters of an EARLIER run of profiles/collect.sh: used only when that run loaded this very library.
        import glob as _glob
        import hashlib
        lib_id = {"version": _lib.lib().ltr_version().decode(), "source_id": _lib.source_id(),
                  "so_sha256_16": hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]}
        tj, tj_path = None, None
        for cand in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic*.json")))[::-1]:
            try:
                j = json.load(open(cand))
            except Exception:
                continue
            if (j.get("library") or {}).get("source_id") == lib_id["source_id"] and args.workload in j.get("source", ""):
                tj, tj_path = j, os.path.relpath(cand, ROOT)
                break

        def class_roofline(ci):
            """Roofline block of launch class `ci`: FP64 add/max the recurrence needs per cell x its nominal cells over the
            average duration of its launch (HIP events on the launch stream, the extra per-launch passes above)."""
            kk = kms[0][ci]
            ms = float(np.mean([s[ci]["ms"] for s in kms]))
            fp64_pc = 11.0 if sym else 13.0
            fam, w, lanes = kk.get("family"), kk["strip_width"], kk.get("lanes_per_pair", 64)
            if fam == "exact":
                fp64_pc += 3.0 if sym else 4.0             # + best-of-three (max(D,I) is shared with X when b == d), band penalty add, row maximum
                kname = f"ltr_dp_kernel<{w}, true, {symtxt}, {'true' if w != 8 else 'false'}>" if lanes == 64 else f"ltr_dp_wgx_kernel<{lanes // 64}, ...>"
            elif fam == "workgroup" and wg_pass and wg_pass[0] == 1 and lanes > 64:
                # the threshold kernels went first (the context has learnt that certificates fail on these reads): exact in one pass,
                # 13 operations a cell (11 + best-of-three + the compare), odd classes on the next even strip width
                fp64_pc += 2.0
                kname = f"ltr_dp_wg_kernel<{w + (w & 1)}, {lanes // 64}, true, true>"
            elif fam == "workgroup":
                kname = f"ltr_dp_wg_kernel<{w}, {lanes // 64}, true>"
            elif fam == "packed" and not kk.get("plan_kernel"):
                kname = f"ltr_dp_pack_kernel<{w}, {symtxt}>"
            elif kk.get("plan_kernel"):
                kname = f"ltr_dp_plan_kernel<{symtxt}>"         # every one-wave class and packed width of the plan in ONE persistent launch
            elif kk.get("ranges"):
                kname = f"ltr_dp_multi_kernel<{symtxt}>"        # the one-wave classes of strip widths 11 .. 20 in one persistent launch
            else:
                kname = f"ltr_dp_kernel<{w}, false, {symtxt}, true>"
            ach = kk["cells"] * fp64_pc / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            pk = (tj or {}).get("per_kernel", {}).get(kname)
            return {"bound": "valu-fp64", "achieved": ach, "peak": peak, "unit": "Tlane-op/s (FP64 add/max)",
                    "frac": ach / peak if peak else None,
                    "traffic": (pk["fetch_bytes"] + pk["write_bytes"]) if pk and "fetch_bytes" in pk and "write_bytes" in pk else None,
                    "kernel": kname, "lanes_per_pair": lanes, "kernel_ms": ms, "kernel_cells": kk["cells"], "kernel_pairs": kk["pairs"],
                    "ops_per_cell": fp64_pc,
                    # share of VALU issue slots busy during the launch: SQ_INSTS_VALU x 4 cycles / (SIMDs x busy cycles), rocprofv3 --pmc
                    "valu_issue_frac": (pk or {}).get("valu_issue_frac"),
                    # the committed counter summary these two fields come from (null: none collected with the library that ran here)
                    "counters_from": tj_path}

        dom = max(range(len(kms[0])), key=lambda k: kms[0][k]["cells"])
        roof = class_roofline(dom)
        if last_launches == 1 and last_ms:
            # one launch = the whole plan: its duration from the timed region itself (not from the extra per-launch passes)
            roof["kernel_ms_extra_passes"] = roof["kernel_ms"]
            roof["kernel_ms"] = float(last_ms)
            roof["kernel_ms_from"] = "the timed region's last step (the plan's own HIP events; ms_per_step is the wall-clock mean of all its steps)"
            roof["achieved"] = roof["kernel_cells"] * roof["ops_per_cell"] / (roof["kernel_ms"] * 1e-3) / 1e12
            roof["frac"] = roof["achieved"] / peak if peak else None
        dom_ms, dom_cells, fp64_pc = roof["kernel_ms"], roof["kernel_cells"], roof["ops_per_cell"]
        fast = [k for k in range(len(kms[0])) if kms[0][k].get("family") != "exact"]
        most = max(fast, key=lambda k: kms[0][k]["pairs"]) if fast else dom
        value = res["cells"] * args.steps / res["elapsed"]
        hbm = {"algorithmic_bytes_per_step": plan.input_bytes,
               "achieved_GBps": plan.input_bytes / (all_ms * 1e-3) / 1e9 if all_ms > 0 else None, "peak_GBps": 8000.0}
        # ---- the ONE stdout line (compact: the driver keeps a bounded tail) ----
        line = {
            "metric": "read x haplotype DP cells/s",
            "value": value,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (numpy PCG64, longtr_amd/synth.py, seed in config)",
            "config": {"workload": desc + (f"; config 4: cost-sharded over {world} GPUs, ordered gather to rank 0" if (world > 1 and strong) else ""),
                       "total_loci": int(res["loci"]), "total_pairs": int(res["pairs"]), "total_cells": res["cells"],
                       "seed": synth.CONFIG_SEED, "parallelism": f"loci-shard x{world}",
                       "shard_loci": [len(x) for x in shards] if shards is not None else None,
                       "alignment_params": args.params if args.params else ("ont f=g=-4.6" if ont else "default"), "pair_packing_mode": args.pair_packing},
            "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "kernel_ms_from", "kernel_cells",
                                              "ops_per_cell", "valu_issue_frac", "counters_from") if k in roof},
            "loci_per_s": res["loci"] * args.steps / res["elapsed"],            # resident plan: inputs in HBM, plan built
            "library": lib_id,
        }
        if wg_pass is not None and wg_pass[2] > 0:
            line["wg_first_pass"] = {"kernels": "threshold (exact in one pass)" if wg_pass[0] == 1 else "certificate (+ exact lists)",
                                     "learnt_from": f"{wg_pass[1]} of {wg_pass[2]} long pairs could not be finished by the first pass of the previous execute"}
        if args.workload == "config5":
            # every pair of the literal config 5 aborts on the row-maximum rule and leaves early; n*m cells are counted all the same
            line["roofline"]["note"] = "nominal cells: pairs abort early (row-maximum rule), frac is not a roofline fraction; see config5hifi"
        # the whole timed pass against the same peak (its launches overlap on several streams: ms_per_step, not the per-launch sum)
        line["roofline"]["whole_pass_frac"] = plan.cells * fp64_pc / (res["elapsed"] / args.steps) / 1e12 / peak if world == 1 else None
        line["roofline"]["hbm"] = hbm
        # ---- everything else: bench_detail.json next to this script ----
        detail = {
            "pairs_per_s": res["pairs"] * args.steps / res["elapsed"],
            "generator": "every locus from its own PCG64 stream keyed by (seed, configuration, locus index); every rank generates its shard only; not std::mt19937",
            # SURVEY 8d prices the recurrence as the reference writes it (22 lane-ops per cell); the kernel needs 11 (certificate
            # instead of the per-cell row maximum, LUT emission), so this ratio can exceed 1 -- it is not a roofline fraction
            "algorithmic_vs_reference_formulation": dom_cells * 22.0 / (dom_ms * 1e-3) / 1e12 / peak if dom_ms > 0 else None,
            "all_dp_kernels_ms": all_ms,      # sum of the per-launch times of an extra pass with every launch on one stream
            "roofline_dominant": roof,
            # the certificate class that holds most PAIRS (catalogue-shaped workloads: short repeats), priced the same way
            "roofline_most_pairs": class_roofline(most),
            # per launch as launched; then with the multi-width launch class by class (set_timing level 2)
            "kernels_by_class": [{"W": k["strip_width"], "lanes_per_pair": k.get("lanes_per_pair", 64), "family": k.get("family"), "pairs": k["pairs"], "cells": k["cells"],
                                  "ms": float(np.mean([s[i]["ms"] for s in kms2]))} for i, k in enumerate(kms2[0]) if k["pairs"] or k["family"] == "exact"] if kms2 else None,
            "kernels": [{"W": k["strip_width"], "lanes_per_pair": k.get("lanes_per_pair", 64), "family": k.get("family"), "pairs": k["pairs"], "cells": k["cells"],
                         "ms": float(np.mean([s[i]["ms"] for s in kms])), **({"ranges": k["ranges"]} if "ranges" in k else {})}
                        for i, k in enumerate(kms[0]) if k["pairs"] or k["family"] == "exact"],
            "device": info,
            "gen_s": t_gen,
            "plan_create_s": run["t_plan"],      # host packing + binning + H2D upload of rank 0's plan (outside the timed region)
        }
        if world > 1:
            line["backend"] = backend_txt                 # of the data exchange (the control plane is gloo whatever this says)
            line["world_size"] = dist.get_world_size()
            line["devices"] = devices                     # one per rank: PCI address + UUID -- N distinct entries = N distinct GPUs
            line["distinct_devices"] = len({d for d in devices if d})
        if other is not None:
            line["weak_scaling" if strong else "strong_scaling"] = {k: other[k] for k in ("value", "ms_per_step", "loci_per_s", "total_loci")}
            detail["weak_scaling" if strong else "strong_scaling"] = other
        if args.one_gpu:
            line["debug_one_gpu"] = True         # every rank on cuda:0, exchange over gloo: a check of the N > 1 code path, not a rate
        for k, v in checks.items():
            detail[k] = v
            line[k] = {kk: v[kk] for kk in ("order_ok", "offsets_ok", "loci", "checked_pairs", "pairs", "mismatches", "classes_covered", "ranks_covered") if kk in v}
        if world == 1 and not args.no_end_to_end and not dry:
            try:
                e2e = end_to_end(ctx, args, params)
                # (the raw loci are their own draw of the generator: compare by the cells the call's plans scored, not by loci)
                e2e["frac_of_resident_rate"] = e2e["cells_per_s"] / value
                detail["end_to_end"] = e2e
                line["loci_per_s_end_to_end"] = e2e["loci_per_s"]
                line["end_to_end_frac_of_resident"] = e2e["frac_of_resident_rate"]
            except Exception as e:
                detail["end_to_end"] = {"error": repr(e)}
        if world == 1 and not args.no_neighbours and not dry:
            try:
                detail["neighbours"] = neighbours(ctx, peak)
            except Exception as e:
                detail["neighbours"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                one, many = cpu_baselines(batch, params, min(args.cpu_budget_s, 0.3) if dry else args.cpu_budget_s)
                line["cpu_baseline"] = one
                detail["cpu_baseline_ncores"] = many
                line["speedup_vs_cpu_1thread"] = value / one["value"]
                if many and "value" in many:
                    line["cpu_baseline_ncores"] = {k: many[k] for k in ("value", "unit", "cores", "kind", "cpu")}
                    line["speedup_vs_cpu_ncores"] = value / many["value"]
            except Exception as e:  # the baseline is reporting, never the product path
                line["cpu_baseline"] = {"error": repr(e)[:200]}
        if dry:
            line["dry_run"], line["value"], line["roofline"]["kernel"] = True, None, "stand-in table (no GPU)"
            line.update(checks["dry_run_gather"])
            line["shard_sizes"] = line["config"]["shard_loci"]
            if other is not None:
                line["other"] = {"scaling": other["scaling"], "total_loci": other["total_loci"]}
        line["detail"] = write_detail(dict(line, **detail), args)
        print(fit_line(line), flush=True)
    plan.close()
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def end_to_end_ranks(args, ctx, dry, rank, world, strong, my_ids, id_base, n_total, gen_workers, params, desc, dev, xdev, xgroup,
                     backend_txt, devices, info, dev_sync):
    """--end-to-end: one step = the drop-in call on this rank's RAW loci (ltr_calc_hap_aln_probs: seq_stutter_genotyper.cpp:514-563,
    host buffers either side) + the ordered gather of the per-read rows to rank 0 (vcf_writer.cpp:7-36).  The host side of that call
    runs on the rank's host-thread budget (ltr_ctx_set_host_threads; rule: affinity mask, cgroup quota, LOCAL_WORLD_SIZE) -- eight
    ranks on one host are eight such budgets.  The same contract line; `value` = the cells the calls' plans scored per second."""
    import torch
    import torch.distributed as dist
    from longtr_amd import _lib, shard, synth
    loci, _ = synth.config_loci(args.workload, seed=synth.CONFIG_SEED + (0 if strong else rank), n_loci=n_total,
                                ids=(my_ids if (world > 1 and strong) else None), raw=True, workers=gen_workers)
    items = [(L.blocks(), L.raw_alns) for L in loci]
    packed = _lib.Context.pack_loci(items, contiguous=True)
    flat, off = packed["flat"], packed["flat_off"]
    gids = np.asarray(my_ids, dtype=np.int64) + id_base
    flat_t = torch.from_numpy(flat)
    og = shard.OrderedGather(np.diff(off), gids, xdev, group=xgroup) if world > 1 else None
    threads = _lib.lib().ltr_host_threads_rule(0) if ctx is None else ctx.host_threads()
    state = {"glob": None}

    def step():
        if dry:
            for k, g in enumerate(gids):
                flat[off[k]:off[k + 1]] = -(float(g) + 1.0) - np.arange(int(off[k + 1] - off[k])) / 1024.0
        else:
            ctx.calc_hap_aln_probs_packed(packed)
        if og is not None:
            state["glob"] = og(flat_t[:int(off[-1])].to(xdev) if xdev.type != "cpu" else flat_t[:int(off[-1])])

    for _ in range(args.warmup):
        step()
    dev_sync()
    if ctx is not None:
        ctx.timers(reset=True)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dev_sync()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    mark_stage("step")
    tm = ctx.timers() if ctx is not None else {"dp_cells": 0.0, "dp_pairs": 0, "dp_kernel_ms": 0.0}
    c = torch.tensor([tm["dp_cells"] / max(args.steps, 1), float(len(items)), float(tm["dp_pairs"]) / max(args.steps, 1), el, float(threads)], dtype=torch.float64)
    per_rank = [torch.zeros_like(c) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, c)
    else:
        per_rank = [c]
    el = max(float(x[3]) for x in per_rank)
    cells, nloci, npairs = (sum(float(x[i]) for x in per_rank) for i in (0, 1, 2))
    # the gathered rows in global locus order: every locus against a one-rank recomputation of a sample on rank 0
    check = None
    if rank == 0 and not args.no_verify:
        glob = state["glob"].cpu().numpy() if og is not None else flat[:int(off[-1])]
        goff = og.global_off if og is not None else off
        n_glob = len(goff) - 1
        if dry:
            bad = sum(int(not np.array_equal(glob[goff[g]:goff[g + 1]], -(float(g) + 1.0) - np.arange(int(goff[g + 1] - goff[g])) / 1024.0)) for g in range(n_glob))
            check = {"gathered_loci": n_glob, "misplaced_loci": bad, "order_ok": bool(n_glob == (n_total if strong else n_total * world))}
        elif strong:
            ids = list(range(0, n_total, max(1, n_total // 200)))
            sl, _ = synth.config_loci(args.workload, seed=synth.CONFIG_SEED, n_loci=n_total, ids=ids, raw=True)
            sp = _lib.Context.pack_loci([(L.blocks(), L.raw_alns) for L in sl], contiguous=True)
            ctx.calc_hap_aln_probs_packed(sp)
            mism = pairs = 0
            for k, l in enumerate(ids):
                a = sp["flat"][sp["flat_off"][k]:sp["flat_off"][k + 1]]
                b2 = glob[goff[l]:goff[l + 1]]
                pairs += a.size
                mism += int((a.view(np.uint64) != b2.view(np.uint64)).sum()) if a.size == b2.size else a.size
            check = {"order_ok": bool(n_glob == n_total), "loci": len(ids), "checked_pairs": int(pairs), "mismatches": int(mism)}
    if rank == 0:
        peak = (info["n_cu"] * 64 * info["clock_mhz"] * 1e6 / 1e12) or 1.0
        value = cells * args.steps / el if el > 0 else 0.0
        line = {"metric": "read x haplotype DP cells/s", "value": None if dry else value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic (numpy PCG64, longtr_amd/synth.py, seed in config)",
                "config": {"workload": desc + "; END TO END: ltr_calc_hap_aln_probs on raw alignments per rank (host in, host out) + ordered gather of the per-read rows",
                           "total_loci": int(nloci), "total_pairs": int(npairs), "total_cells": cells, "seed": synth.CONFIG_SEED, "parallelism": f"loci-shard x{world}",
                           "alignment_params": args.params if args.params else "workload default"},
                "roofline": {"bound": "valu-fp64", "achieved": value * 11.0 / 1e12, "peak": peak, "unit": "Tlane-op/s (FP64 add/max)", "frac": value * 11.0 / 1e12 / peak,
                             "traffic": None, "kernel": "the whole call (host preparation + ltr_dp_plan_kernel per chunk + fan-out)",
                             "kernel_ms": float(np.mean([float(tm["dp_kernel_ms"]) / max(args.steps, 1)]))},
                "cpu_baseline": None,
                "loci_per_s": nloci * args.steps / el if el > 0 else None,
                "host_threads_per_rank": [int(float(x[4])) for x in per_rank],
                "rank_ms_per_step": [float(x[3]) / args.steps * 1e3 for x in per_rank],
                "end_to_end": True}
        if world > 1:
            line["backend"], line["world_size"], line["devices"] = backend_txt, world, devices
            line["distinct_devices"] = len({d for d in (devices or []) if d})
        if args.one_gpu:
            line["debug_one_gpu"] = True
        if check is not None:
            line["gather_check"] = check
        if dry:
            line["dry_run"] = True
        print(fit_line(line), flush=True)
    if ctx is not None:
        ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def neighbours(ctx, peak):
    """The two kernels either side of the DP, each on a bounded batch, each with its own
    roofline block: device time from HIP events around the launches (ltr_timers), instruction-issue bound.

    * NW = ltr_haplotype_align_to_ref (NeedlemanWunsch::Align, NeedlemanWunsch.cpp:82-420) over the haplotypes of 3000
      config-3 loci (1000 loci are 1900 wavefronts per strip-width class: fewer than two a SIMD).  FP32; the step loop of
      ltr_nw_wave_kernel<8> issues 255 vector instructions per 8 cells (ISA count: 48 compares + 56 selects for the three
      trace-back pointers of a cell, 24 max3, 41 adds, 10 or3, moves), so the kernel's ceiling is peak / 32 cells/s.
    * short = the seeded stutter path (HapAligner.cpp:13-233) over 300 period-1 loci: flank cells at 13 FP64 add/max each
      (HapAligner.cpp:143-151: 3 + 2 + 1, 2 + 1 + 1, 2 + 1) -- the stutter-block row (13 artifact sizes per read position,
      chains of dependent look-ups) has no per-cell count and is in the time, not in the numerator."""
    from longtr_amd import _abi, synth
    out = {}
    loci, desc = synth.config_loci("config3", n_loci=3000)
    packed = ctx.pack_haplotypes([L.blocks() for L in loci])
    cells = float(sum(len(L.haplotypes[0]) * sum(len(h) for h in L.haplotypes) for L in loci))
    ctx.haplotype_align_to_ref_packed(packed, decode=False)
    ctx.timers(reset=True)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.haplotype_align_to_ref_packed(packed, decode=False)
    dt = (time.perf_counter() - t0) / 3
    kms = ctx.timers(reset=True)["nw_kernel_ms"] / 3
    ach = cells * 32.0 / (kms * 1e-3) / 1e12
    out["nw"] = {"workload": "haplotypes of 3000 config-3 loci against their reference allele", "cells": cells, "call_ms": dt * 1e3,
                 "kernel_ms": kms, "cells_per_s_call": cells / dt, "cells_per_s_kernel": cells / (kms * 1e-3),
                 "roofline": {"bound": "valu-issue", "achieved": ach, "peak": peak, "unit": "Tlane-op/s (vector instructions x 64 lanes)",
                              "frac": ach / peak, "ops_per_cell": 32.0, "traffic": None, "kernel": "ltr_nw_wave_kernel<W>"}}
    rng = np.random.default_rng(5)
    prm = _abi.default_params()
    prm.use_short_path = 1
    held = ctx.params
    sloci, flank_cells, all_cells = [], 0.0, 0.0
    for _ in range(300):
        tr, H, R = int(rng.integers(10, 60)), int(rng.integers(2, 5)), 30
        blocks, alns = synth.homopolymer_locus(rng, tr, H, R)
        sloci.append((blocks, alns))
        rows = sum(len(a["seq"]) for a in alns)
        for al in blocks[1]["alleles"][:H]:
            flank_cells += rows * (len(blocks[0]["alleles"][0]) + len(blocks[2]["alleles"][0]))
            all_cells += rows * (len(blocks[0]["alleles"][0]) + len(al) + len(blocks[2]["alleles"][0]))
    ctx.set_params(prm)
    split = None
    try:
        spacked = ctx.pack_loci(sloci)
        ctx.calc_hap_aln_probs_packed(spacked)
        ctx.timers(reset=True)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.calc_hap_aln_probs_packed(spacked)
        dt = (time.perf_counter() - t0) / 3
        kms = ctx.timers(reset=True)["short_kernel_ms"] / 3
        # ... and kernel by kernel (events between the launches: an extra, untimed pass)
        ctx.set_debug("short_split", 1)
        ctx.short_kernel_split(reset=True)
        for _ in range(3):
            ctx.calc_hap_aln_probs_packed(spacked)
        split = [x / 3 for x in ctx.short_kernel_split(reset=True)]
        ctx.set_debug("short_split", 0)
    finally:
        ctx.set_params(held)
    ach = flank_cells * 13.0 / (kms * 1e-3) / 1e12
    # The flank rows are the part with a per-cell operation count: their OWN launches' time prices them; the stutter-block row has
    # none (a walk per (read position, artifact size), chains of dependent look-ups) and is reported as time and share
    flank_ms = (split[0] + split[2]) if split else None
    per_kernel = None if not split else {
        "flank_rows": {"kernel": "ltr_short_prep_kernel + ltr_short_flank_kernel<false|true>", "ms": flank_ms, "cells": flank_cells, "ops_per_cell": 13.0,
                       "achieved": flank_cells * 13.0 / (flank_ms * 1e-3) / 1e12 if flank_ms else None,
                       "frac": flank_cells * 13.0 / (flank_ms * 1e-3) / 1e12 / peak if flank_ms else None, "bound": "valu-fp64"},
        "block_row": {"kernel": "ltr_short_block_kernel", "ms": split[1], "share_of_kernel_time": split[1] / max(sum(split), 1e-9),
                      "bound": "valu-issue (walks of StutterAlignerClass.cpp:59-154, each run twice; VALU issue 0.91-0.97 by rocprofv3 --pmc, profiles/r05/pmc_dispatch_neighbours.txt)"},
        "seed_log_sum": {"kernel": "ltr_short_final_kernel", "ms": split[3]}}
    out["short_path"] = {"per_kernel": per_kernel,"workload": "300 period-1 loci, 30 raw reads x 2-4 alleles, use_short_path", "read_x_haplotype_cells": all_cells,
                         "flank_cells": flank_cells, "call_ms": dt * 1e3, "kernel_ms": kms, "loci_per_s_call": 300 / dt,
                         "cells_per_s_call": all_cells / dt, "cells_per_s_kernel": all_cells / (kms * 1e-3),
                         "roofline": {"bound": "valu-fp64", "achieved": ach, "peak": peak, "unit": "Tlane-op/s (FP64 add/max)",
                                      "frac": ach / peak, "ops_per_cell": 13.0, "traffic": None,
                                      "kernel": "ltr_short_flank_kernel<W> x 2 + ltr_short_block_kernel + prep + final (one event pair)"}}
    return out


def end_to_end(ctx, args, params):
    """The drop-in as a caller sees it: ltr_calc_hap_aln_probs on RAW alignments (CIGARs in, per-read
    LL matrices out): pooling, trimming, haplotype strings, plan building, upload, DP, download and
    scatter all inside the timed call; host buffers either side (PCIe included)."""
    from longtr_amd import synth
    n = args.e2e_loci if args.e2e_loci is not None else (30000 if args.workload == "catalogue" else 6000)
    # (never more loci than the resident workload holds: the fraction of the resident rate compares like with like)
    n = min(n, args.loci if args.loci is not None else synth._DEFAULT_N[args.workload])
    loci, desc = synth.config_loci(args.workload, seed=synth.CONFIG_SEED, n_loci=n, raw=True)
    items = [(L.blocks(), L.raw_alns) for L in loci]
    packed = ctx.pack_loci(items)
    ctx.calc_hap_aln_probs_packed(packed)               # warm-up (device pool, tables)
    reps = 3
    ctx.timers(reset=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.calc_hap_aln_probs_packed(packed)
    dt = (time.perf_counter() - t0) / reps
    tm = ctx.timers()
    # Two cell counts.  `cells_scored` = what the call's own plans scored (ltr_timers.dp_cells): pairs of DISTINCT trimmed reads x
    # haplotypes -- pools that trim to the same bytes are scored once -- counted exactly like a resident plan counts its pairs, so
    # cells_scored / time over the resident rate is a fraction <= 1 by construction: the share of the resident rate the call
    # reaches on the DP it really runs.  `cells_nominal` = every read x every haplotype window (what the reference would run).
    cells_nominal = float(sum(sum(len(r) for r in L.trimmed_reads) * sum(max(len(h) - 60, 0) for h in L.haplotypes) for L in loci))
    cells = tm["dp_cells"] / reps
    out = {"loci_per_s": len(loci) / dt, "ms_per_call": dt * 1e3, "loci": len(loci), "cells": cells, "cells_per_s": cells / dt,
           "cells_nominal": cells_nominal, "pairs_scored": tm["dp_pairs"] // reps, "dp_kernel_ms_per_call": tm["dp_kernel_ms"] / reps,
           "what": "ltr_calc_hap_aln_probs(raw alignments) -> per-read LL matrices, host to host, " + desc}
    if tm:
        out["timers"] = tm
    return out


if __name__ == "__main__":
    sys.exit(main())
