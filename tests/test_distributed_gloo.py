"""CPU, world_size 2, gloo: locus sharding + the gather of per-locus results (the N>1 path of
bench.py), with a stand-in compute so no GPU is needed."""
import os
import socket

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
