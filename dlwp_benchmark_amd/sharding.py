"""Multi-GPU rollout: shard the batch of initial conditions, one process per GPU.

The reference is single-process / single-device (no DDP, no collectives
on the live path).  Rollouts of different initial conditions are independent (no cross-sample op
in any backbone in eval mode), so the path shards with NO collective inside the step; the only
exchange is collecting the trajectories ([B/N, K, Cg, H, W] per rank) for the evaluation metrics
(reference scripts/evaluate.py:243 `outputs.append(output.cpu())`): ONE all-gather per rollout
over RCCL/xGMI.  It is issued per time chunk with async_op=True -- ProcessGroupNCCL runs it on its
own HIP stream after the chunk's kernels -- so the transfer of chunk k overlaps the compute of
chunk k+1 and only the last chunk's gather is exposed.  Exception: a backbone whose rollout is a
persistent launch that needs EVERY compute unit resident at once (FNO2DModule, `exclusive_launch`)
must not share the chip with an RCCL kernel -- its workgroups spin on peers that the collective's
workgroups would keep off the chip -- so there the whole rollout runs first and the gather follows it.

Works with any process group backend (`gloo` in the CPU tests); with world_size == 1 it is the
plain device-resident rollout.
"""
from typing import Optional

import torch


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous split of `n_items` initial conditions: rank r gets [lo, hi)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_bounds(n_steps: int, chunks: int, min_len: int = 1):
    chunks = max(1, min(chunks, n_steps // max(min_len, 1) or 1))
    base, rem = divmod(n_steps, chunks)
    out, a = [], 0
    for c in range(chunks):
        b = a + base + (1 if c < rem else 0)
        out.append((a, b))
        a = b
    return out


class ShardedRollout:
    """Callable: runs `model`'s rollout on this rank's shard and returns the gathered global
    trajectory [world*B_local, K, Cg, H, W] (rank-major).  `model` must provide `_check_inputs`, `context_size`
    and either `rollout_into(out, constants, prescribed, prognostic, step_begin, step_end)` (FNO, Swin, Pangu,
    FourCastNet, U-Net: chunked, overlapped gather) or only `forward` (ConvLSTM, whose recurrent state rules out a
    ranged rollout: one gather after the rollout)."""

    def __init__(self, model, world_size: int = 1, rank: int = 0, chunks: int = 4, group=None, gather: bool = True):
        self.model = model
        self.world = world_size
        self.rank = rank
        self.chunks = chunks
        self.group = group
        # gather=False: every rank keeps its own shard [B_local, K, ...] (evaluation that only needs scores reduces
        # them on the device and all-reduces a few hundred bytes instead: dlwp_benchmark_amd.metrics.RolloutMetrics)
        self.gather = gather

    def __call__(self, constants: Optional[torch.Tensor] = None, prescribed: Optional[torch.Tensor] = None,
                 prognostic: torch.Tensor = None) -> torch.Tensor:
        m = self.model
        constants, prescribed, prognostic = m._check_inputs(constants, prescribed, prognostic)
        b, t, cg, h, w = prognostic.shape
        ctx = m.context_size
        k = t - ctx
        with torch.no_grad():
            local = torch.empty(b, k, cg, h, w, device=prognostic.device, dtype=prognostic.dtype)
            ranged = hasattr(m, "rollout_into")
            if self.world == 1 or not self.gather:
                if not ranged:
                    return m(constants=constants, prescribed=prescribed, prognostic=prognostic)
                m.rollout_into(local, constants, prescribed, prognostic, 0, k)
                return local
            import torch.distributed as dist

            if not ranged or getattr(m, "exclusive_launch", False):
                # no collective beside the rollout: run it whole, then ONE all-gather of the trajectory
                if ranged:
                    m.rollout_into(local, constants, prescribed, prognostic, 0, k)
                else:
                    local = m(constants=constants, prescribed=prescribed, prognostic=prognostic)
                host = local.is_cuda and dist.get_backend(self.group) != "nccl"
                send = local.cpu() if host else local.contiguous()
                recv = torch.empty((self.world * b,) + tuple(send.shape[1:]), device=send.device, dtype=send.dtype)
                dist.all_gather_into_tensor(recv, send, group=self.group)
                return recv.to(prognostic.device) if host else recv
            bounds = chunk_bounds(k, self.chunks)
            works, parts = [], []
            # a CPU-only backend (gloo rehearsal on a one-GPU box) cannot move device tensors: stage on the host
            host = local.is_cuda and dist.get_backend(self.group) != "nccl"
            for (a, e) in bounds:
                m.rollout_into(local, constants, prescribed, prognostic, a, e)
                send = local[:, a:e].contiguous()
                if host:
                    send = send.cpu()
                recv = torch.empty((self.world * b,) + tuple(send.shape[1:]), device=send.device, dtype=send.dtype)
                works.append(dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True))
                parts.append((send, recv))
            out = torch.empty(self.world * b, k, cg, h, w, device=prognostic.device, dtype=prognostic.dtype)
            ov = out.view(self.world, b, k, cg, h, w)
            for (a, e), wk, (_, recv) in zip(bounds, works, parts):
                wk.wait()
                ov[:, :, a:e].copy_(recv.view(self.world, b, e - a, cg, h, w), non_blocking=True)
            return out


class CapturedStep:
    """One evaluation step -- `fn(*tensors)`: a rank's rollout plus whatever device-only work follows it (metric sums) -- recorded
    into a HIP graph at the first call with a given set of input tensors and replayed afterwards.  What it removes is the host
    side of the step: the launches of a 20-step FNO rollout + its metric kernels are ~8 graph nodes, 90-130 us of gaps per 1.5 ms
    rollout when enqueued one by one.

    Contract (that of any captured graph): `fn` must not synchronise, must not move data to the host and must not run a collective;
    its inputs are read from the SAME device addresses at every call (an evaluation loop stages each batch into fixed buffers:
    dlwp_benchmark_amd.staging.DeviceStager), and from the second call on the returned tensors are the SAME objects, overwritten
    by every replay (the first call with a set of tensors runs eagerly, the second one records).  Fused-kernel checks cannot run inside a graph: the model is verified by the caller (`model.verify()`, once per
    evaluation) -- FNO modules are switched to `check="deferred"` here for that reason.  Every distinct set of input tensors (address,
    shape, dtype) gets its own recording, up to `max_graphs` (a double-buffered `DeviceStager(fixed_buffers=True)` alternates between
    two sets; with `split_size`, two per sub-batch)."""

    def __init__(self, fn, model=None, max_graphs: int = 8):
        self.fn = fn
        self.max_graphs = int(max_graphs)
        self._seen = {}          # key -> None (ran eagerly once) | (graph, outputs); insertion order = age
        self.replays = 0
        if model is not None and hasattr(model, "set_execution_form") and getattr(model, "check", None) == "per_call":
            model.set_execution_form(check="deferred")
        if model is not None and hasattr(model, "set_step_graphs"):
            model.set_step_graphs(False)          # a graph replay cannot be recorded into another graph: the step runs eagerly ONCE

    @staticmethod
    def _key_of(tensors):
        return tuple(None if t is None else (t.data_ptr(), tuple(t.shape), t.dtype, str(t.device)) for t in tensors)

    @property
    def _graph(self):            # the most recently used recording (tests)
        live = [v for v in self._seen.values() if v is not None]
        return live[-1][0] if live else None

    def __call__(self, *tensors):
        key = self._key_of(tensors)
        if key not in self._seen:
            # first call with these tensors: the step runs eagerly, ONCE (its side effects -- running sums -- count once, and plans,
            # packed weights and workspaces come into being outside any capture); the recording happens at the next call.
            # One recording per set of input addresses (a double-buffered stager alternates between two), oldest dropped first.
            dev = next(t.device for t in tensors if t is not None)
            if dev.type != "cuda":
                raise RuntimeError("CapturedStep records HIP graphs: the inputs must live on an MI355X device")
            while len(self._seen) >= self.max_graphs:
                self._seen.pop(next(iter(self._seen)))
            self._seen[key] = None
            return self.fn(*tensors)
        entry = self._seen.pop(key)               # re-inserted below: most recently used last
        if entry is None:
            dev = next(t.device for t in tensors if t is not None)
            with torch.cuda.device(dev):
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                # (thread-local error mode: other threads of the process -- a communicator's watchdog -- may touch the device meanwhile)
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):      # nothing executes here
                    out = self.fn(*tensors)
            entry = (graph, out)
        self._seen[key] = entry
        entry[0].replay()
        self.replays += 1
        return entry[1]
