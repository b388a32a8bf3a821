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


class ExclusiveBackbone(ToyBackbone):
    """like FNO2DModule: the rollout is a launch that needs the whole chip -- no collective may run beside it"""
    exclusive_launch = True


class ForwardOnlyBackbone:
    """like ConvLSTM: recurrent state, no ranged rollout_into"""
    context_size = 1
    _check_inputs = ToyBackbone._check_inputs

    def __call__(self, constants=None, prescribed=None, prognostic=None):
        out = torch.empty(prognostic.shape[0], prognostic.shape[1] - 1, *prognostic.shape[2:])
        return ToyBackbone().rollout_into(out, constants, prescribed, prognostic)


BACKBONES = {"toy": ToyBackbone, "exclusive": ExclusiveBackbone, "forward_only": ForwardOnlyBackbone}


def _worker(rank, world, port, ret, kind="toy"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        presc, prog = _inputs(6, 8)
        lo, hi = shard_bounds(6, world, rank)
        run = ShardedRollout(BACKBONES[kind](), world_size=world, rank=rank, chunks=3)
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
@pytest.mark.parametrize("kind", list(BACKBONES))
def test_two_rank_gather_matches_single_process(kind):
    """chunked overlapped gather (toy), rollout-then-gather for a backbone that needs the chip to itself (exclusive),
    and for one without a ranged rollout (forward_only)"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + 3 * list(BACKBONES).index(kind)
    mp.spawn(_worker, args=(world, port, ret, kind), nprocs=world, join=True)
    presc, prog = _inputs(6, 8)
    want = ShardedRollout(ToyBackbone(), world_size=1)(prescribed=presc, prognostic=prog)
    for r in range(world):
        assert torch.equal(ret[r], want), f"rank {r} gathered trajectory differs"


# ---------------------------------------------------------------------------------------------------------------
# bench.py's default collective (`--collect metrics`, SURVEY section 8e "when only RMSE is needed"): every rank reduces
# its shard to [4, K, C] double sums (on the GPU: dlwp_weighted_error_sums_f32) and ONE all-reduce moves sums + sample
# count.  Here the per-rank sums come from a CPU stand-in of that kernel; what is under test is the cross-rank part
# of RolloutMetrics (gloo, world_size 2, UNEQUAL shards) against the oracle's metric restatement on the whole batch.
# ---------------------------------------------------------------------------------------------------------------
class _CpuSumsMetrics:
    """RolloutMetrics with the HIP sums kernel replaced by its definition (include/dlwp_hip.h) in float64 torch."""

    def __new__(cls, lats, std, clim):
        from dlwp_benchmark_amd.metrics import RolloutMetrics

        class M(RolloutMetrics):
            def sums(self, out, target, into=None):
                w = self.latw.double()[None, None, None, :, None]
                s = (self.std.double() if self.std is not None else torch.ones(out.shape[2], dtype=torch.float64))
                s = s[None, None, :, None, None]
                o, t = out.double() * s, target.double() * s
                res = torch.zeros(4, out.shape[1], out.shape[2], dtype=torch.float64)
                res[0] = (w * (o - t) ** 2).sum(dim=(0, 3, 4))
                if self.clim is not None:
                    c = self.clim.double()[None] * s
                    res[1] = (w * (o - c) * (t - c)).sum(dim=(0, 3, 4))
                    res[2] = (w * (o - c) ** 2).sum(dim=(0, 3, 4))
                    res[3] = (w * (t - c) ** 2).sum(dim=(0, 3, 4))
                return res if into is None else into.add_(res)

        return M(lats, std=std, climatology=clim)


def _metric_inputs():
    g = torch.Generator().manual_seed(11)
    out = torch.randn(5, 3, 2, 8, 16, generator=g)
    tar = out + 0.3 * torch.randn(5, 3, 2, 8, 16, generator=g)
    clim = 0.1 * torch.randn(3, 2, 8, 16, generator=g)
    lats = torch.linspace(-78.75, 78.75, 8)
    std = torch.tensor([2.0, 0.5])
    return out, tar, clim, lats, std


def _metrics_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out, tar, clim, lats, std = _metric_inputs()
        lo, hi = shard_bounds(out.shape[0], world, rank)   # 3 + 2 samples
        scorer = _CpuSumsMetrics(lats, std, clim)
        res = scorer(out[lo:hi].contiguous(), tar[lo:hi].contiguous(), world_size=world)
        ret[rank] = (res["rmse"], res["acc"])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_metric_allreduce_matches_whole_batch_oracle():
    from oracle.restate.metrics import lat_weighted_metrics

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_metrics_worker, args=(world, port, ret), nprocs=world, join=True)
    out, tar, clim, lats, std = _metric_inputs()
    rmse, acc = lat_weighted_metrics(out.numpy(), tar.numpy(), lats.numpy(), std=std.numpy(), climatology=clim.numpy())
    single = _CpuSumsMetrics(lats, std, clim)(out, tar, world_size=1)   # same class, whole batch, no collective
    for r in range(world):
        got_rmse, got_acc = ret[r]
        # the cross-rank reduction itself: exact up to the order of two float64 additions
        assert torch.allclose(got_rmse, single["rmse"], rtol=1e-13, atol=0), f"rank {r} RMSE vs single process"
        assert torch.allclose(got_acc, single["acc"], rtol=1e-13, atol=1e-15), f"rank {r} ACC vs single process"
        # and against the oracle's restatement of evaluate.py:786-821 (RolloutMetrics keeps its latitude weights and
        # scales in float32, hence 1e-7)
        assert torch.allclose(got_rmse, torch.from_numpy(rmse), rtol=1e-7, atol=0), f"rank {r} RMSE"
        assert torch.allclose(got_acc, torch.from_numpy(acc), rtol=1e-7, atol=1e-9), f"rank {r} ACC"


# ---------------------------------------------------------------------------------------------------------------
# bench.py --config C3 / C4 / C5 under world > 1 (reference call site scripts/evaluate.py:205-244): the SAME
# make_runner / timed_region the GPU run uses, with the stand-in backbone and the CPU stand-in of the sums kernel --
# warm-up, reset, K timed rollouts with per-rank accumulation, ONE all-reduce, max-over-ranks clock.
# ---------------------------------------------------------------------------------------------------------------
def _bench_flow_worker(rank, world, port, ret, collect):
    import argparse
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        presc, prog = _inputs(3, 6)
        g = torch.Generator().manual_seed(100 + rank)            # every rank its own shard of initial conditions
        prog = prog + 0.1 * torch.randn(prog.shape, generator=g)
        args = argparse.Namespace(gather_chunks=2, collect=collect)
        scorer = _CpuSumsMetrics(torch.zeros(8), None, None)
        step0, finish, acc = bench.make_runner(ToyBackbone(), world, rank, args, prog, 8, 16, scorer=scorer)
        dt = bench.timed_region(lambda: step0(prescribed=presc), finish, steps=3, warmup=2, dist=dist, device="cpu", backend="gloo")
        ret[rank] = (dt, acc["samples"], acc["scores"]["rmse"] if acc["scores"] else None, acc["out"], prog)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("collect", ["metrics", "gather"])
def test_two_rank_bench_control_flow(collect):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000) + (7 if collect == "gather" else 0)
    mp.spawn(_bench_flow_worker, args=(world, port, ret, collect), nprocs=world, join=True)
    presc, _ = _inputs(3, 6)
    assert ret[0][0] == ret[1][0] > 0                          # ONE clock: the max over ranks, identical on both
    progs = torch.cat([ret[r][4] for r in range(world)])
    want = ShardedRollout(ToyBackbone(), world_size=1)(prescribed=torch.cat([presc, presc]), prognostic=progs)
    if collect == "metrics":
        assert ret[0][1] == 3 * 3                              # warm-up sums were reset: 3 timed rollouts x 3 samples
        # scores over ALL ranks' samples (3 identical rollouts accumulate to the same mean)
        rmse = torch.sqrt(((want.double() - progs[:, 1:].double()) ** 2).mean(dim=(0, 3, 4)))
        for r in range(world):
            assert torch.allclose(ret[r][2], rmse, rtol=1e-12)
    else:
        for r in range(world):
            assert torch.equal(ret[r][3], want)                # every rank holds the rank-major global trajectory
