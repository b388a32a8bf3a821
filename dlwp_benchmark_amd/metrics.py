"""On-device evaluation metrics.

`RolloutMetrics` reduces a rollout [B, K, C, H, W] and its targets to latitude-weighted RMSE (and ACC
when a climatology is given) per lead time and variable -- the quantities reference
scripts/evaluate.py:786-821 computes with xarray after copying every trajectory to the host, with the
per-variable de-normalisation of evaluate.py:281-296 folded in as a scale (the means cancel).
Multi-GPU: the [4, K, C] double sums are all-reduced (a few hundred bytes) instead of gathering
trajectories.
"""
import math
from typing import Optional

import torch

from . import lib as _lib


def latitude_weights(lats_deg: torch.Tensor) -> torch.Tensor:
    """cos(lat_j) / mean_j cos(lat_j)  (evaluate.py:788-790, Eq. (2) of arXiv:2002.00469)."""
    w = torch.cos(torch.deg2rad(lats_deg.double()))
    return (w / w.mean()).float()


class RolloutMetrics:
    def __init__(self, lats_deg: torch.Tensor, std: Optional[torch.Tensor] = None,
                 climatology: Optional[torch.Tensor] = None, group=None):
        self.latw = latitude_weights(lats_deg)
        self.std = std.float() if std is not None else None
        self.clim = climatology.float() if climatology is not None else None
        self.group = group
        self._dev = {}   # device -> (latw, std, clim) copies made once

    def _on(self, dev):
        key = str(dev)
        if key not in self._dev:
            self._dev[key] = (self.latw.to(dev).contiguous(),
                              self.std.to(dev).contiguous() if self.std is not None else None,
                              self.clim.to(dev).contiguous() if self.clim is not None else None)
        return self._dev[key]

    def sums(self, out: torch.Tensor, target: torch.Tensor, into: Optional[torch.Tensor] = None) -> torch.Tensor:
        """double [4, K, C] sums of this rank's samples (see dlwp_weighted_error_sums_f32).  `into`: a [4, K, C] double tensor
        of running sums the new ones are ADDED to (dlwp_weighted_error_sums_acc_f32: one launch per batch, no zero-fill, no add
        kernel); it is returned."""
        _lib.require_cuda_tensor(out, "out")
        _lib.require_cuda_tensor(target, "target")
        out, target = out.contiguous(), target.contiguous()
        b, k, c, h, w = out.shape
        if target.shape != out.shape:
            raise _lib.DlwpError(f"target shape {tuple(target.shape)} != output shape {tuple(out.shape)}")
        dev = out.device
        latw, std, clim = self._on(dev)
        if into is not None:
            if not isinstance(into, torch.Tensor) or not into.is_cuda:
                raise _lib.DlwpError("running sums must be a tensor on an MI355X device")
            if tuple(into.shape) != (4, k, c) or into.dtype != torch.float64 or not into.is_contiguous() or into.device != dev:
                raise _lib.DlwpError(f"running sums must be a contiguous double [4, {k}, {c}] tensor on {dev}")
        sums = into if into is not None else torch.empty(4, k, c, dtype=torch.float64, device=dev)
        lib = _lib.load()
        fn = lib.dlwp_weighted_error_sums_acc_f32 if into is not None else lib.dlwp_weighted_error_sums_f32
        with torch.cuda.device(dev):
            _lib.check(fn(out.data_ptr(), target.data_ptr(), clim.data_ptr() if clim is not None else None, latw.data_ptr(),
                          std.data_ptr() if std is not None else None, sums.data_ptr(), b, k, c, h, w, _lib.stream_ptr()),
                       "dlwp_weighted_error_sums_f32")
        return sums

    def __call__(self, out: torch.Tensor, target: torch.Tensor, world_size: int = 1):
        """Returns {"rmse": [K, C], "acc": [K, C] or None} over ALL ranks' samples."""
        return self.finalize(self.sums(out, target), float(out.shape[0]), out.shape[-2] * out.shape[-1], world_size)

    def finalize(self, s: torch.Tensor, n_samples: float, cells: int, world_size: int = 1):
        """Scores from accumulated sums: `s` = this rank's [4, K, C] sums (of one batch, or ADDED UP over many -- the
        reference accumulates squared errors over the whole evaluation before taking the root, evaluate.py:786-821),
        n_samples = how many samples went into them, cells = H * W.  With world_size > 1 ONE all-reduce moves the sums
        and the sample count (shards may differ in size): the only collective of a sharded evaluation."""
        if world_size > 1:
            import torch.distributed as dist

            host = dist.get_backend(self.group) != "nccl"
            dev = s.device
            buf = torch.empty(s.numel() + 1, dtype=torch.float64, device=dev)
            buf[:-1].copy_(s.flatten())
            buf[-1:].fill_(n_samples)   # the count travels with the sums
            if host:
                buf = buf.cpu()
            dist.all_reduce(buf, group=self.group)
            buf = buf.to(dev)
            s, n_samples = buf[:-1].view_as(s), buf[-1:]
        count = n_samples * cells
        rmse = torch.sqrt(s[0] / count)
        acc = s[1] / torch.sqrt(s[2] * s[3]) if self.clim is not None else None
        return {"rmse": rmse, "acc": acc}
