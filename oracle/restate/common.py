"""Shared pieces of the CPU restatement (TEST INFRASTRUCTURE, see oracle/__init__.py).

`prepare_inputs` and `rollout` restate the autoregressive driver that every reference backbone
carries a copy of; the canonical (un-broken) copy is
/root/reference/src/dlwpbench/models/swintransformer/swin_transformer.py:679-737
(identical loops: fno.py:49-106, fourcastnet.py:294-361, panguweather.py:442-510,
unet.py:316-383).
"""
from typing import Callable, Optional

import torch


def prepare_inputs(constants: Optional[torch.Tensor], prescribed: Optional[torch.Tensor],
                   prognostic: Optional[torch.Tensor]) -> torch.Tensor:
    """swin_transformer.py:679-692: cat([constants[:,0], 'b t c h w -> b (t c) h w' of the
    prescribed and prognostic windows], dim=1)."""
    tensors = []
    if constants is not None:
        tensors.append(constants[:, 0])
    if prescribed is not None:
        b, t, c, h, w = prescribed.shape
        tensors.append(prescribed.reshape(b, t * c, h, w))
    if prognostic is not None:
        b, t, c, h, w = prognostic.shape
        tensors.append(prognostic.reshape(b, t * c, h, w))
    return torch.cat(tensors, dim=1)


def rollout(one_step: Callable[[torch.Tensor], torch.Tensor], context_size: int,
            constants: Optional[torch.Tensor], prescribed: Optional[torch.Tensor],
            prognostic: torch.Tensor) -> torch.Tensor:
    """swin_transformer.py:694-737.  Returns [B, T-context_size, Cg, H, W]."""
    outs = []
    ctx = context_size
    for t in range(ctx, prognostic.shape[1]):
        t_start = max(0, t - ctx)
        if t == ctx:
            prognostic_t = prognostic[:, t_start:t]
            x_t = prepare_inputs(constants,
                                 prescribed[:, t_start:t] if prescribed is not None else None,
                                 prognostic_t)
        else:
            prognostic_t = torch.cat(
                [prognostic[:, t_start:ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
            x_t = prepare_inputs(constants,
                                 prescribed[:, t - ctx:t] if prescribed is not None else None,
                                 prognostic_t)
        out = prognostic_t[:, -1] + one_step(x_t)
        outs.append(out)
    return torch.stack(outs, dim=1)
