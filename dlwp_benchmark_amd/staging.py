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

    def __init__(self, batches: Iterable[Tuple], device, split_size: Optional[int] = None):
        self.batches = batches
        self.split_size = int(split_size) if split_size else None
        if self.split_size is not None and self.split_size < 1:
            raise ValueError("split_size must be >= 1")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DeviceStager stages onto a GPU")
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def _upload(self, batch):
        out = []
        with torch.cuda.stream(self.copy_stream):
            for t in batch:
                t = _none_if_sentinel(t)
                if t is None:
                    out.append(None)
                    continue
                pinned = t.contiguous().pin_memory() if not t.is_pinned() else t
                out.append(pinned.to(self.device, non_blocking=True))
        ev = torch.cuda.Event()
        ev.record(self.copy_stream)
        return out, ev

    def __iter__(self) -> Iterator[Tuple]:
        it = iter(self.batches)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._upload(next(it))       # in flight while the caller computes on `cur`
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in cur:
                if t is not None:
                    t.record_stream(torch.cuda.current_stream(self.device))
            if self.split_size is None:
                yield tuple(cur)
                continue
            n = max(t.shape[0] for t in cur if t is not None)
            for a in range(0, n, self.split_size):
                yield tuple(t[a:a + self.split_size] if t is not None else None for t in cur)
