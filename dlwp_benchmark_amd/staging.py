"""Input staging: overlap the host->device copy of the next batch of
initial conditions with the rollout of the current one.

The reference evaluation loop moves each DataLoader batch synchronously from pageable memory right
before the forward (`.to(device).split(split_size)`, scripts/evaluate.py:213-217) and maps the dataset's
NaN sentinel to None (:213-217, train.py:180-181).  `DeviceStager` wraps any iterable of
(constants, prescribed, prognostic, target) CPU tuples (the layout of
WeatherBenchDataset.__getitem__, data/datasets/datasets.py:330-416): batch i+1 is copied through pinned
buffers on a side HIP stream while batch i is being rolled out; `prescribed` for all K steps travels once.
"""
from typing import Iterable, Iterator, Optional, Tuple

import torch


def _none_if_sentinel(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The dataset marks an absent input with a NaN tensor (datasets.py:317, :378); the evaluation loop decides with
    `.isnan().any()` (scripts/evaluate.py:213-215, train.py:180-181): ANY NaN makes the whole input absent."""
    if t is None or (t.numel() > 0 and bool(torch.isnan(t).any())):
        return None
    return t


class DeviceStager:
    """`split_size` restates `.split(split_size)` of the evaluation loop (evaluate.py:212-217, with
    split_size = batch_size // gradient_accumulation_steps): every uploaded batch is handed on as consecutive
    sub-batches of at most `split_size` samples along dim 0 (views of ONE upload; None inputs stay None)."""

    def __init__(self, batches: Iterable[Tuple], device, split_size: Optional[int] = None, fixed_buffers: bool = False):
        """`fixed_buffers`: upload into TWO preallocated sets of device tensors used alternately (batch i into set i % 2), so that
        the tensors a consumer sees have one of two address sets for the whole evaluation -- what a recorded step
        (dlwp_benchmark_amd.sharding.CapturedStep, one graph per address set) needs.  A set is overwritten only after the work the
        consumer enqueued on it has finished (an event recorded when the consumer asks for the next batch)."""
        self.batches = batches
        self.fixed_buffers = bool(fixed_buffers)
        self._bufs = [None, None]
        self._free = [None, None]
        self.split_size = int(split_size) if split_size else None
        if self.split_size is not None and self.split_size < 1:
            raise ValueError("split_size must be >= 1")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DeviceStager stages onto a GPU")
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def _upload(self, batch, slot: int = 0):
        out = []
        bufs = self._bufs[slot] if self.fixed_buffers else None
        if self.fixed_buffers and (bufs is None or len(bufs) != len(batch)):
            bufs = self._bufs[slot] = [None] * len(batch)
        with torch.cuda.stream(self.copy_stream):
            if self.fixed_buffers and self._free[slot] is not None:
                self.copy_stream.wait_event(self._free[slot])      # the consumer's work on this set has finished
            for k, t in enumerate(batch):
                t = _none_if_sentinel(t)
                if t is None:
                    out.append(None)
                    continue
                pinned = t.contiguous().pin_memory() if not t.is_pinned() else t
                if not self.fixed_buffers:
                    out.append(pinned.to(self.device, non_blocking=True))
                    continue
                if bufs[k] is None or bufs[k].shape != pinned.shape or bufs[k].dtype != pinned.dtype:
                    bufs[k] = torch.empty(pinned.shape, dtype=pinned.dtype, device=self.device)   # (a new shape: a new address)
                bufs[k].copy_(pinned, non_blocking=True)
                out.append(bufs[k])
        ev = torch.cuda.Event()
        ev.record(self.copy_stream)
        return out, ev

    def __iter__(self) -> Iterator[Tuple]:
        it = iter(self.batches)
        slot = 0
        try:
            nxt = self._upload(next(it), slot)
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._upload(next(it), 1 - slot)       # in flight while the caller computes on `cur`
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            if not self.fixed_buffers:
                for t in cur:
                    if t is not None:
                        t.record_stream(torch.cuda.current_stream(self.device))
            if self.split_size is None:
                yield tuple(cur)
            else:
                n = max(t.shape[0] for t in cur if t is not None)
                for a in range(0, n, self.split_size):
                    yield tuple(t[a:a + self.split_size] if t is not None else None for t in cur)
            if self.fixed_buffers:
                # the consumer has enqueued everything it does with this set (it came back for the next batch): the upload that
                # reuses the set waits for this point of the consumer's stream
                self._free[slot] = torch.cuda.Event()
                self._free[slot].record(torch.cuda.current_stream(self.device))
            slot = 1 - slot
