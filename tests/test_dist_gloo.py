"""world_size-2 / -3 gloo tests of the row-sharded path's EXCHANGE (hipporag_amd/dist.py: ShardedRetriever + TorchComm) on
CPU, for both forms of the per-sweep collective: the in-place all-gather of the owners' row blocks (default) and the
north star's literal all-reduce (foreign blocks zeroed, SUM over the bytes of the group region).

The HIP kernels cannot run here; every rank's compute is the numpy stand-in of tests/test_shard_orchestration.py
(FakeShardStages: the hrag_shard_* interface, same state-buffer layout contract).  Under test: that the two collectives
assemble the SAME replicated iterate -- results identical between them, on every rank, and equal to the single-process
oracle -- with uneven fill (70 queries = two full slabs + a partial one), one and two exchange groups, a DPR-fallback
query and a partial filter.  The reference has no counterpart (HippoRAG.py:459 is a serial loop)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hipporag_amd import dist as hd
from hipporag_amd import synth
from tests.test_shard_orchestration import FakeShardStages, _check, _problem, _run_rank


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, groups, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kg, index, sidx, qf, qp, b = _problem(world)
        outs = {}
        for collective in ("allgather", "allreduce"):
            rs = hd.ShardedRetriever(FakeShardStages(sidx, index, rank), hd.TorchComm(rank, world, collective=collective),
                                     groups=groups)
            outs[collective] = _run_rank(rs, qf, qp, b)
        for a, w in zip(outs["allreduce"], outs["allgather"]):
            assert torch.equal(a, w)                                  # the literal all-reduce IS the gather, bit for bit
        if rank == 0:
            _check(index, qf, qp, b, *outs["allreduce"])
        gathered = [torch.empty_like(outs["allreduce"][2]) for _ in range(world)]
        dist.all_gather(gathered, outs["allreduce"][2])
        for g in gathered:
            assert torch.equal(g, outs["allreduce"][2])               # every rank holds the same answer
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,groups", [(2, 2), (3, 1)])
def test_both_exchange_collectives_assemble_the_same_iterate(world, groups):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), groups, ret), nprocs=world, join=True)
    assert dict(ret) == {r: 1 for r in range(world)}


def test_torchcomm_rejects_an_unknown_collective_and_uneven_shards():
    with pytest.raises(ValueError):
        hd.TorchComm(0, 2, collective="broadcast")
    from types import SimpleNamespace
    lay = SimpleNamespace(own_offset=128, own_bytes=256, group_bytes=4096, slabs_per_group=1)
    with pytest.raises(ValueError, match="equal-sized row shards"):
        hd.TorchComm(1, 2).exchange(torch.zeros(4096, dtype=torch.uint8), lay, 0)


def test_even_shards_cover_everything():
    for n in (0, 1, 7, 125_000):
        for world in (1, 2, 3, 8):
            ev = hd.even_shards(n, world)
            assert ev[0][0] == 0 and ev[-1][1] == n and all(a[1] == b[0] for a, b in zip(ev, ev[1:]))
            assert max(h - l for l, h in ev) - min(h - l for l, h in ev) <= 1
