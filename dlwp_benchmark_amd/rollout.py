"""Device-resident autoregressive driver shared by the backbones whose step is a Python-level
composition of kernels (Swin, Pangu, FourCastNet, U-Net).

Restates the loop every reference backbone carries (canonical copy
models/swintransformer/swin_transformer.py:694-737; `_prepare_inputs` :679-692) with the two host-side
costs removed: the trajectory is written in place into a preallocated [B, T-ctx, Cg, H, W] buffer
(the reference rebuilds `torch.stack(outs)` every step, O(T^2) copies, and the fno/afno/convlstm
variants move every step to the CPU: fno.py:104, fourcastnet.py:359, convlstm.py:249), and nothing in the
loop synchronises with the host.
"""
from typing import Callable, Optional

import torch


def assemble_input(constants: Optional[torch.Tensor], prescribed: Optional[torch.Tensor],
                   frames) -> torch.Tensor:
    """`_prepare_inputs`: cat([constants[:,0], prescribed window (t c), prognostic window (t c)], 1).
    `frames` is the list of the context_size prognostic frames [B, Cg, H, W] (views, no copies)."""
    parts = []
    if constants is not None:
        parts.append(constants[:, 0])
    if prescribed is not None:
        b, t, c, h, w = prescribed.shape
        parts.append(prescribed.reshape(b, t * c, h, w))
    parts.extend(frames)
    if not torch.is_grad_enabled() and parts[0].is_cuda:
        from . import ops

        return ops.concat_channels(parts)          # one 16-byte copy kernel (dlwp_concat_channels_f32)
    return torch.cat(parts, dim=1)


def rollout_into(one_step: Callable[[torch.Tensor], torch.Tensor], context_size: int, out: torch.Tensor,
                 constants: Optional[torch.Tensor], prescribed: Optional[torch.Tensor], prognostic: torch.Tensor,
                 step_begin: int = 0, step_end: int = -1) -> torch.Tensor:
    """Runs rollout steps [step_begin, step_end) writing out[:, s] in place; earlier steps must be
    present in `out`.  Frame f of the window of step s (f in [s, s+ctx)) is the input frame f when
    f < ctx, else out[:, f - ctx]."""
    ctx = context_size
    n_steps = prognostic.shape[1] - ctx
    if step_end < 0:
        step_end = n_steps
    for s in range(step_begin, step_end):
        t = s + ctx
        frames = [prognostic[:, f] if f < ctx else out[:, f - ctx] for f in range(s, t)]
        x_t = assemble_input(constants, prescribed[:, t - ctx:t] if prescribed is not None else None, frames)
        inc = one_step(x_t)
        torch.add(frames[-1], inc, out=out[:, s])
    return out


def rollout_train(one_step: Callable[[torch.Tensor], torch.Tensor], context_size: int, constants: Optional[torch.Tensor],
                  prescribed: Optional[torch.Tensor], prognostic: torch.Tensor) -> torch.Tensor:
    """The same loop with autograd alive (reference scripts/train.py:263-271 trains through it): nothing in place, the
    trajectory is stacked once at the end."""
    ctx = context_size
    outs = []
    for s in range(prognostic.shape[1] - ctx):
        t = s + ctx
        frames = [prognostic[:, f] if f < ctx else outs[f - ctx] for f in range(s, t)]
        x_t = assemble_input(constants, prescribed[:, t - ctx:t] if prescribed is not None else None, frames)
        outs.append(frames[-1] + one_step(x_t))
    return torch.stack(outs, dim=1)
