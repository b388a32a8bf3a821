"""N > 1 path on CPU: world_size-2 gloo processes run ShardedRollout with a stand-in backbone
(CPU arithmetic, same rollout_into contract as the HIP backbones) and must reproduce the
single-process trajectory of the concatenated batch, rank-major."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dlwp_benchmark_amd.sharding import ShardedRollout, chunk_bounds, shard_bounds


class ToyBackbone:
    """x_{t+1} = x_t + 0.1 * tanh(roll(x_t)) + 0.01 * prescribed_t ; same interface as HipBackbone models."""
    context_size = 1

    def _check_inputs(self, c, p, g):
        return c, p, g

    def rollout_into(self, out, constants, prescribed, prognostic, step_begin=0, step_end=-1):
        k = out.shape[1]
        if step_end < 0:
            step_end = k
        for s in range(step_begin, step_end):
            x = prognostic[:, 0] if s == 0 else out[:, s - 1]
            inc = 0.1 * torch.tanh(torch.roll(x, 1, dims=-1))
            if prescribed is not None:
                inc = inc + 0.01 * prescribed[:, s]
            out[:, s] = x + inc
        return out


def _inputs(b, t):
    g = torch.Generator().manual_seed(7)
    return torch.randn(b, t, 1, 8, 16, generator=g), torch.randn(b, t, 2, 8, 16, generator=g)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        presc, prog = _inputs(6, 8)
        lo, hi = shard_bounds(6, world, rank)
        run = ShardedRollout(ToyBackbone(), world_size=world, rank=rank, chunks=3)
        out = run(prescribed=presc[lo:hi].contiguous(), prognostic=prog[lo:hi].contiguous())
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (1, 5, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    assert chunk_bounds(20, 4) == [(0, 5), (5, 10), (10, 15), (15, 20)]
    assert chunk_bounds(3, 8) == [(0, 1), (1, 2), (2, 3)]


@pytest.mark.timeout(120)
def test_two_rank_gather_matches_single_process():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    presc, prog = _inputs(6, 8)
    want = ShardedRollout(ToyBackbone(), world_size=1)(prescribed=presc, prognostic=prog)
    for r in range(world):
        assert torch.equal(ret[r], want), f"rank {r} gathered trajectory differs"
