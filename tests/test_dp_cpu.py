"""The multi-GPU path on CPU ranks (gloo, world_size 2): question sharding and the ONE collective of a training step
(mean all-reduce of the flat [alpha.grad | icv.grad | kl] buffer), as licv.trainer issues them over RCCL on GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from licv.trainer import allreduce_mean_, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_alpha, n_icv = 4, 4 * 32
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.cat([torch.randn(n_alpha, generator=g), torch.randn(n_icv, generator=g), torch.tensor([float(rank + 1)])])
        mine = flat.clone()
        allreduce_mean_(flat)
        # every rank must hold the same mean; rank 0 checks it against a gathered reference
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        ref = torch.stack(gathered).mean(0)
        ok = torch.allclose(flat, ref, atol=1e-7)
        shards = shard_indices(11, rank, world)
        out.put((rank, ok, shards, float(flat[-1])))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_allreduce_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert res[0][3] == res[1][3] == 1.5                           # the logged kl scalar rides in the same buffer
    s0, s1 = res[0][2], res[1][2]
    assert not set(s0) & set(s1) and len(s0) == len(s1) == 5 and set(s0) | set(s1) == set(range(10))


def test_single_process_allreduce_is_a_noop_and_shards_cover_everything():
    x = torch.arange(5.0)
    assert torch.equal(allreduce_mean_(x.clone()), x)
    assert shard_indices(8, 0, 1) == list(range(8))
    assert shard_indices(10, 3, 4, drop_last=False) == [3, 7]
