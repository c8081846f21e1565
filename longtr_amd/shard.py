"""Locus sharding across the GPUs of one node and the single exchange step of the path.

Loci are independent units (reference: BamProcessor::process_regions carries no state between
loci, src/bam_processor.cpp:563-627; its documented scale-out is one process per BED shard,
README.md:78-82).  Here: one process per GPU, loci -> ranks by greedy cost balance, no data-path
collective, and ONE gather of the per-locus log-likelihood matrices to rank 0 in locus order
(the analogue of the position-ordered VCF heap, src/vcf_writer.cpp:7-36).  Works on RCCL
("nccl" backend, device tensors) and on gloo (CPU tensors; used by the CPU tests).
"""
import heapq
import os

import numpy as np
import torch
import torch.distributed as dist


def locus_costs(batch, indel_flank_len=5):
    """DP cells per locus (sum over pairs of n*m), the balance criterion."""
    rl = np.diff(batch.read_off).astype(np.float64)
    hl = np.diff(batch.hap_off).astype(np.float64) - 2 * (35 - indel_flank_len)
    out = np.zeros(batch.n_loci)
    for l in range(batch.n_loci):
        m = rl[batch.locus_read_off[l]:batch.locus_read_off[l + 1]].sum()
        n = np.maximum(hl[batch.locus_hap_off[l]:batch.locus_hap_off[l + 1]], 0).sum()
        out[l] = m * n
    return out


def shard_by_cost(costs, world):
    """Greedy longest-processing-time assignment.  Returns a list (per rank) of ascending locus ids."""
    heap = [(0.0, r) for r in range(world)]
    heapq.heapify(heap)
    shards = [[] for _ in range(world)]
    for l in np.argsort(-np.asarray(costs), kind="stable"):
        load, r = heapq.heappop(heap)
        shards[r].append(int(l))
        heapq.heappush(heap, (load + float(costs[l]), r))
    return [sorted(s) for s in shards]


def exchange_meta(ll_numel, n_loci, device, group=None):
    """(elements, loci) of every rank -- constant for a resident plan, so exchanged once."""
    world = dist.get_world_size(group)
    meta = torch.tensor([ll_numel, n_loci], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    return [(int(m[0]), int(m[1])) for m in metas]


def gather_ll_raw(ll_local, sizes_local, locus_ids_local, metas=None, group=None, dst=0):
    """The collective itself: two padded gathers (LL payload, per-locus sizes + ids) to `dst`.
    Returns (metas, recv_ll, recv_ix) on dst, None elsewhere.  Variable lengths are handled by
    padding to the longest rank (payload ~P*H*8 B per locus, ~12 MB per 10 k loci: far below one
    xGMI link, so the padding is free)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = ll_local.device
    if metas is None:
        metas = exchange_meta(ll_local.numel(), sizes_local.numel(), dev, group)
    max_ll = max(m[0] for m in metas)
    max_n = max(m[1] for m in metas)
    send_ll = torch.zeros(max(max_ll, 1), dtype=torch.float64, device=dev)
    send_ll[:ll_local.numel()] = ll_local
    send_ix = torch.full((2, max(max_n, 1)), -1, dtype=torch.int64, device=dev)
    send_ix[0, :sizes_local.numel()] = sizes_local.to(dev)
    send_ix[1, :sizes_local.numel()] = locus_ids_local.to(dev)
    recv_ll = [torch.empty_like(send_ll) for _ in range(world)] if rank == dst else None
    recv_ix = [torch.empty_like(send_ix) for _ in range(world)] if rank == dst else None
    dist.gather(send_ll, recv_ll, dst=dst, group=group)
    dist.gather(send_ix, recv_ix, dst=dst, group=group)
    if rank != dst:
        return None
    return metas, recv_ll, recv_ix


def gather_ll(ll_local, sizes_local, locus_ids_local, group=None, dst=0):
    """Gather per-locus LL blocks to `dst` and index them by global locus id.

    ll_local: 1-D float64 tensor, this rank's locus blocks back to back;
    sizes_local: 1-D int64 (elements per local locus); locus_ids_local: 1-D int64 global ids.
    Returns on dst: dict {global locus id: 1-D float64 tensor}; on other ranks None."""
    raw = gather_ll_raw(ll_local, sizes_local, locus_ids_local, None, group, dst)
    if raw is None:
        return None
    metas, recv_ll, recv_ix = raw
    out = {}
    for r in range(len(metas)):
        n = metas[r][1]
        sizes = recv_ix[r][0, :n].cpu().numpy()
        ids = recv_ix[r][1, :n].cpu().numpy()
        off = 0
        for s, l in zip(sizes, ids):
            out[int(l)] = recv_ll[r][off:off + int(s)]
            off += int(s)
    return dict(sorted(out.items()))


def pair_time_cost(n, C, pairs_in_batch=1 << 21, n_cu=256, params=None):
    """The plan's launch-time model of one pair, asked from the library itself (ltr_debug_pair_costs = ltrp::classify_pair's
    cost, csrc/ltr_plan.cpp): wavefront steps x (strip width + per-step overhead) x the share of the wavefront the pair holds --
    the cheapest of one pair per wave, the packed geometries of 2 .. 32 lanes per pair and, for reads beyond 1280 columns, the
    four- / eight-wave workgroup kernels -- for a batch of `pairs_in_batch` pairs on `n_cu` CUs in automatic mode.
    n: haplotype window, C: read columns (read length - 1); numpy arrays broadcast."""
    import ctypes as C_
    from . import _abi, _lib
    n_b, C_b = np.broadcast_arrays(np.asarray(n), np.asarray(C))
    if not os.path.exists(_lib.LIB_PATH):
        # a sharding host without the compiled library (it only needs a BALANCE criterion, not the product): the model's leading
        # term by hand -- wavefront steps x (strip width + per-step overhead) of one pair per wavefront
        steps = np.maximum(n_b, 1).astype(np.float64) + 63.0
        blocks = np.ceil(np.maximum(C_b, 1) / 1280.0)
        width = np.ceil(np.maximum(C_b, 1) / (64.0 * blocks))
        return blocks * steps * (width + 1.5)
    shape = n_b.shape
    win = np.ascontiguousarray(np.maximum(n_b, 1).ravel(), dtype=np.int32)
    rl = np.ascontiguousarray(np.maximum(C_b, 1).ravel() + 1, dtype=np.int32)
    hl = np.ascontiguousarray(win + 60, dtype=np.int32)
    out = np.zeros(win.size, dtype=np.float64)
    prm = params if params is not None else _abi.default_params()
    L = _lib.lib()
    L.ltr_debug_pair_costs.argtypes = [C_.c_void_p, C_.c_int, C_.c_int, C_.c_int64, C_.c_int64, C_.c_int64, C_.c_void_p, C_.c_void_p, C_.c_void_p, C_.c_void_p]
    n_long = int((rl > 1281).sum() * max(pairs_in_batch // max(win.size, 1), 1)) if win.size else 0
    rc = L.ltr_debug_pair_costs(C_.byref(prm), -1, int(n_cu), int(pairs_in_batch), n_long, win.size, win.ctypes.data, rl.ctypes.data, hl.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"ltr_debug_pair_costs: {rc}")
    return out.reshape(shape)


def locus_time_costs(batch, indel_flank_len=5):
    """Launch-time model per locus, a better balance criterion than raw cells: short reads pay relatively more
    fill/drain and set-up per cell.  One call into the library for all pairs of the batch."""
    rl = np.diff(batch.read_off).astype(np.int64)
    hl = np.maximum(np.diff(batch.hap_off).astype(np.int64) - 2 * (35 - indel_flank_len), 1)
    P = np.diff(batch.locus_read_off).astype(np.int64)
    H = np.diff(batch.locus_hap_off).astype(np.int64)
    npairs = P * H
    if int(npairs.sum()) == 0:
        return np.zeros(batch.n_loci)
    # pair (r, h) of locus l, reads outer: read index = locus_read_off[l] + k // H[l], haplotype = locus_hap_off[l] + k % H[l]
    loc = np.repeat(np.arange(batch.n_loci), npairs)
    k = np.arange(int(npairs.sum())) - np.repeat(np.cumsum(npairs) - npairs, npairs)
    m = rl[np.asarray(batch.locus_read_off)[loc] + k // H[loc]]
    n = hl[np.asarray(batch.locus_hap_off)[loc] + k % H[loc]]
    c = pair_time_cost(n, m - 1, pairs_in_batch=max(int(npairs.sum()), 1)) + 3.0
    return np.bincount(loc, weights=c, minlength=batch.n_loci)


def header_time_costs(headers, indel_flank_len=5, sub_rate=0.0015, indel_rate=0.0005, world=1, n_cu=256):
    """The same model from the generator's locus headers alone (synth.config_headers: repeat length, candidate
    alleles, reads) -- what a rank needs to shard a catalogue it has not generated: reads x alleles pairs of
    (TR + pads + flanks) bases a side, reads pooled by exact sequence (two true alleles + the reads with an error).
    world / n_cu: the model is asked for a batch of (all pairs / world) pairs on n_cu CUs -- what one rank's plan will hold (the
    packing rules depend on the batch's size per CU)."""
    tr, h, r = (headers[:, k].astype(np.float64) for k in range(3))
    side = tr + 10.0 + 2.0 * indel_flank_len
    pools = np.minimum(r, 2.0 + r * (1.0 - (1.0 - sub_rate - indel_rate) ** side) + 0.1 * r)
    per_rank = max(int(float((pools * h).sum()) / max(int(world), 1)), 1)
    return pools * h * (pair_time_cost(side, side - 1.0, pairs_in_batch=per_rank, n_cu=n_cu) + 3.0)


class OrderedGather:
    """The exchange step of a sharded, resident batch: every rank's per-locus LL blocks -> `dst`,
    laid out in GLOBAL locus order (the analogue of the position-ordered VCF heap,
    src/vcf_writer.cpp:7-36).  Sizes and ids never change for a resident plan, so they are
    exchanged once here; a call moves only the payload: one padded gather + one index_select on dst.

    sizes_local[k] = elements of the k-th local locus, ids_local[k] = its GLOBAL id; the union of
    all ranks' ids must be exactly 0..n_global-1."""

    def __init__(self, sizes_local, ids_local, device, group=None, dst=0):
        self.group, self.dst, self.device = group, dst, device
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        sizes_local = torch.as_tensor(sizes_local, dtype=torch.int64)
        ids_local = torch.as_tensor(ids_local, dtype=torch.int64)
        self.n_local = int(sizes_local.sum())
        meta = exchange_meta(self.n_local, sizes_local.numel(), device, group)
        self.metas = meta
        self.max_ll = max(max(m[0] for m in meta), 1)
        max_n = max(max(m[1] for m in meta), 1)
        send_ix = torch.full((2, max_n), -1, dtype=torch.int64, device=device)
        send_ix[0, :sizes_local.numel()] = sizes_local.to(device)
        send_ix[1, :ids_local.numel()] = ids_local.to(device)
        recv_ix = [torch.empty_like(send_ix) for _ in range(self.world)] if self.rank == dst else None
        dist.gather(send_ix, recv_ix, dst=dst, group=group)
        self.send = torch.zeros(self.max_ll, dtype=torch.float64, device=device)
        self.recv = None
        self.perm = None
        self.global_off = None
        ok = torch.ones(1, dtype=torch.int64, device=device)
        if self.rank == dst:
            sizes = [recv_ix[r][0, :meta[r][1]].cpu().numpy() for r in range(self.world)]
            ids = [recv_ix[r][1, :meta[r][1]].cpu().numpy() for r in range(self.world)]
            all_ids = np.concatenate(ids) if ids else np.zeros(0, dtype=np.int64)
            n_global = len(all_ids)
            if n_global and not np.array_equal(np.sort(all_ids), np.arange(n_global)):
                ok[0] = 0
        # every rank learns whether the ids partition 0..n-1: a ValueError on dst alone would leave the others in the next gather forever
        if self.world > 1:
            dist.broadcast(ok, src=dst, group=group)
        if int(ok[0]) == 0:
            raise ValueError("OrderedGather: the ranks' locus ids do not partition 0..n-1")
        if self.rank == dst:
            gsize = np.zeros(n_global, dtype=np.int64)
            for r in range(self.world):
                gsize[ids[r]] = sizes[r]
            goff = np.zeros(n_global + 1, dtype=np.int64)
            goff[1:] = np.cumsum(gsize)
            perm = np.zeros(int(goff[-1]), dtype=np.int64)       # global element -> position in the [world, max_ll] receive buffer
            for r in range(self.world):
                # rank r's elements sit back to back in row r of the receive buffer; element e of its k-th locus goes to goff[ids[k]] + e
                loff = np.zeros(len(sizes[r]) + 1, dtype=np.int64)
                loff[1:] = np.cumsum(sizes[r])
                total = int(loff[-1])
                if total:
                    dst_pos = np.repeat(goff[ids[r]] - loff[:-1], sizes[r]) + np.arange(total)
                    perm[dst_pos] = r * self.max_ll + np.arange(total)
            self.global_off = goff
            self.perm = torch.from_numpy(perm).to(device)
            self.recv = torch.empty((self.world, self.max_ll), dtype=torch.float64, device=device)

    def __call__(self, ll_local):
        """ll_local: this rank's LL blocks back to back (1-D float64).  Returns on dst the global,
        locus-ordered LL vector (offsets in .global_off); None elsewhere."""
        self.send[:self.n_local] = ll_local[:self.n_local]
        recv_list = list(self.recv.unbind(0)) if self.rank == self.dst else None
        dist.gather(self.send, recv_list, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        return self.recv.view(-1).index_select(0, self.perm)
