#!/usr/bin/env python3
"""bench.py -- read x haplotype DP cells/s on BASELINE config 3 (10k synthetic loci, 30x, TR 20-1000 bp).

  python bench.py --gpus N --steps K --warmup W            (N == 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (every pooled read x candidate haplotype DP of the rank's
loci, inputs already resident in HBM) + the gather of per-locus results to rank 0 (the one
exchange step of the path; N > 1 only).  One process per GPU; loci shard with no data-path
collective, so scaling is weak: every rank owns its own 10k-locus batch (seed = base + rank).

Rank 0 prints ONE JSON line.  `roofline` prices the dominant DP launch (widest strip class)
against the FP64 vector-ALU peak -- this path is a scalar max-plus recurrence, VALU-bound by
~3 orders of magnitude over its HBM traffic; the HBM figure is reported next
to it.  `cpu_baseline` times the reference's own align_seq_to_hap (oracle/_ref, built from
the reference sources in the dev container) -- or the C port when that build is absent -- on
a bounded sample of the same workload, single thread.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--loci", type=int, default=10000, help="loci per GPU (BASELINE config 3 = 10000)")
    ap.add_argument("--workload", default="config3", choices=["config2", "config3", "config5"])
    ap.add_argument("--cpu-budget-s", type=float, default=20.0, help="CPU baseline sample budget (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(batch, params, budget_s):
    """Reference (or port) on host cores, single thread, on a bounded prefix of the same loci."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from longtr_amd import _abi, synth
    kind = "reference" if ol.have_ref() else "port"
    # pick loci until the estimated CPU time reaches the budget (reference: ~5e8 cells/s on this workload)
    est_rate = 5.0e8
    rl, hl = np.diff(batch.read_off), np.diff(batch.hap_off)
    chosen, cells = [], 0.0
    for l in range(batch.n_loci):
        m = rl[batch.locus_read_off[l]:batch.locus_read_off[l + 1]].astype(np.float64)
        n = hl[batch.locus_hap_off[l]:batch.locus_hap_off[l + 1]].astype(np.float64) - 60
        c = float(m.sum() * n.sum())
        if chosen and cells + c > est_rate * budget_s:
            break
        chosen.append(l)
        cells += c
    sub = []
    for l in chosen:
        reads = [batch.read_bytes[batch.read_off[r]:batch.read_off[r + 1]].tobytes()
                 for r in range(batch.locus_read_off[l], batch.locus_read_off[l + 1])]
        haps = [batch.hap_bytes[batch.hap_off[h]:batch.hap_off[h + 1]].tobytes()
                for h in range(batch.locus_hap_off[l], batch.locus_hap_off[l + 1])]
        sub.append((reads, haps))
    sb = _abi.PackedBatch(sub)
    nominal = synth.nominal_cells(sb, params.indel_flank_len)
    if kind == "reference":
        _, secs = ol.ref_align_batch(sb, params)        # std::chrono around align_seq_to_hap only
    else:
        t0 = time.perf_counter()
        ol.oracle_align_batch(sb, params)
        secs = time.perf_counter() - t0
    return {"value": nominal / secs, "unit": "cells/s", "cores": 1, "kind": kind,
            "sample": f"first {len(chosen)} loci of the workload ({nominal:.3e} nominal cells, {secs:.1f} s, "
                      f"align_seq_to_hap only, 1 thread)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from longtr_amd import _abi, _lib, synth
    params = _abi.make_params(synth.ONT_PARAMS) if args.workload == "config5" else _abi.default_params()
    t_gen = time.perf_counter()
    n_loci = args.loci if args.workload == "config3" else None
    loci, desc = synth.config_loci(args.workload, seed=synth.CONFIG_SEED + rank, n_loci=n_loci)
    batch, _ = synth.pack_loci(loci)
    t_gen = time.perf_counter() - t_gen

    ctx = _lib.Context(local_rank, params)
    t_plan = time.perf_counter()
    plan = ctx.plan(batch)                      # pack + H2D: inputs resident in HBM before timing
    t_plan = time.perf_counter() - t_plan
    info = ctx.device_info()
    out = torch.empty(max(plan.ll_size, 1), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    # the one exchange step: per-locus LL blocks of every rank -> rank 0, in locus order
    # (longtr_amd/shard.py::gather_ll, covered on CPU by tests/test_distributed_gloo.py)
    if world > 1:
        from longtr_amd import shard
        sizes_local = torch.from_numpy(np.diff(batch.ll_off)).to(dev)
        ids_local = torch.arange(batch.n_loci, dtype=torch.int64, device=dev) * world + rank   # interleaved global ids
        metas = shard.exchange_meta(plan.ll_size, batch.n_loci, dev)

    def step():
        plan.execute(out.data_ptr(), stream)
        if world > 1:
            shard.gather_ll_raw(out[:plan.ll_size], sizes_local, ids_local, metas)   # rank 0 now holds every locus

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # per-launch device times (HIP events on the launch stream): taken from extra, untimed passes so
    # that reading the events never sits inside the timed region
    kms = []
    plan.set_timing(True)
    for _ in range(min(3, max(1, args.steps))):
        plan.execute(out.data_ptr(), stream)
        kms.append(plan.kernel_stats())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([plan.cells, float(batch.n_loci), float(plan.num_pairs)], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot_cells, tot_loci, tot_pairs = (float(x) for x in c.tolist())
    else:
        tot_cells, tot_loci, tot_pairs = plan.cells, float(batch.n_loci), float(plan.num_pairs)

    if rank == 0:
        # dominant launch = the strip class with the most cells
        dom = max(range(len(kms[0])), key=lambda k: kms[0][k]["cells"])
        dom_ms = float(np.mean([s[dom]["ms"] for s in kms]))
        dom_cells = kms[0][dom]["cells"]
        all_ms = float(np.mean([sum(k["ms"] for k in s) for s in kms]))
        t7 = params.as_tuple()
        sym = (t7[1] == t7[3]) and (t7[5] == t7[6])
        fp64_pc = 11 if sym else 13
        dom_w = kms[0][dom]["strip_width"]
        dom_lanes = kms[0][dom].get("lanes_per_pair", 64)
        step_ovh = 12.0 if dom_lanes == 64 else 16.0
        symtxt = "true" if sym else "false"
        kname = f"ltr_dp_kernel<{dom_w}, false, {symtxt}, true>" if dom_lanes == 64 else f"ltr_dp_dual_kernel<{dom_w}, {symtxt}>"
        OPS_PER_CELL = 22.0                       # 19 FP64 add/max + 3 int/cvt lane-ops per cell
        clock_hz = info["clock_mhz"] * 1e6
        peak = info["n_cu"] * 64 * clock_hz / 1e12       # FP64 add/max lane-ops/s: 4 SIMD x 16 lanes/clk per CU
        achieved = dom_cells * OPS_PER_CELL / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic = None      # HBM bytes per launch from rocprofv3 --pmc passes (profiles/); not measurable in-process
        import glob
        tfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))      # latest round's PMC summary
        tf = tfs[-1] if tfs else ""
        if tf and os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                pk = tj.get("per_kernel", {}).get(kname)
                traffic = (pk["fetch_bytes"] + pk["write_bytes"]) if pk else tj.get("dominant_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "read x haplotype DP cells/s",
            "value": tot_cells * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": desc, "loci_per_gpu": batch.n_loci, "pairs_per_gpu": int(plan.num_pairs),
                       "cells_per_gpu": plan.cells, "seed": synth.CONFIG_SEED, "parallelism": f"loci-shard x{world}",
                       "alignment_params": "ont f=g=-4.6" if args.workload == "config5" else "default"},
            "loci_per_s": tot_loci * args.steps / elapsed,
            "pairs_per_s": tot_pairs * args.steps / elapsed,
            "roofline": {"bound": "valu-fp64", "achieved": achieved, "peak": peak, "unit": "Tlane-op/s (FP64 add/max)",
                         "frac": achieved / peak if peak else None, "traffic": traffic,
                         "kernel": kname,
                         "kernel_ms": dom_ms, "kernel_cells": dom_cells, "ops_per_cell": OPS_PER_CELL,
                         # what the kernel actually issues: per cell 11 FP64 add/max
                         # (13 for an asymmetric model) + 1/4 integer add (emission-table address of four
                         # cells), per wavefront step ~12 more VALU instructions (hand-off, certificate,
                         # bookkeeping; ~16 in the two-pairs-per-wave kernels) shared by the strip's W cells
                         # -- the share of the VALU issue slots in use
                         "executed": {"fp64_ops_per_cell": fp64_pc, "int_ops_per_cell": 0.25, "valu_ops_per_step": step_ovh,
                                      "lanes_per_pair": dom_lanes,
                                      "valu_issue_frac": dom_cells * (fp64_pc + 0.25 + step_ovh / dom_w)
                                      / (dom_ms * 1e-3) / 1e12 / peak if dom_ms > 0 else None},
                         "all_dp_kernels_ms": all_ms,
                         "hbm": {"algorithmic_bytes_per_step": plan.input_bytes,
                                 "achieved_GBps": plan.input_bytes / (all_ms * 1e-3) / 1e9 if all_ms > 0 else None,
                                 "peak_GBps": 8000.0}},
            "kernels": [{"W": k["strip_width"], "lanes_per_pair": k.get("lanes_per_pair", 64), "pairs": k["pairs"], "cells": k["cells"],
                         "ms": float(np.mean([s[i]["ms"] for s in kms]))} for i, k in enumerate(kms[0])],
            "device": info,
            "gen_s": t_gen,
            "plan_create_s": t_plan,          # host packing + binning + H2D upload (outside the timed region)
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(batch, params, args.cpu_budget_s)
                line["speedup_vs_cpu_1thread"] = line["value"] / line["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is reporting, never the product path
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line))
    plan.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
