"""Shared host-side plumbing for the backbone mirrors.

Every reference backbone has the same boundary:
  * constructed as `eval(cfg.model.type)(**cfg.model)` (reference scripts/train.py:54,
    scripts/evaluate.py:140) -> constructors take **kwargs and ignore extras (`type`, `name`);
  * called by keyword, `model(constants=..., prescribed=..., prognostic=...)`
    (train.py:263-267, evaluate.py:235-239), returns [B, T-context, Cg, H, W] on the device of
    `prognostic`;
  * `.eval()` / `.train()` return the module (the reference Swin returns None,
    swin_transformer.py:739-742 -- fixed here).
"""
import torch

from .. import lib as _lib


class HipBackbone(torch.nn.Module):
    """Base class: input validation + workspace cache.  No CPU path exists by design."""

    def __init__(self):
        super().__init__()
        self._ws = None
        self.step_graphs = False     # replay one_step as a HIP graph (launch-bound backbones), see graphs.py
        self._graphed = None

    def set_step_graphs(self, on: bool = True):
        """Capture `one_step` into a HIP graph and replay it per rollout step (eval mode only)."""
        self.step_graphs = bool(on)
        self._graphed = None
        return self

    def _step_fn(self):
        if not self.step_graphs or self.training:
            return self.one_step
        # keyed on (version, pointer) of every parameter and buffer: load_state_dict / optimizer steps write IN PLACE
        # (same pointers), and a captured graph bakes in derived buffers (packed MLP operands, plans) made from the
        # old values -- a version bump re-captures, whose warm-up re-derives them
        key = self._param_key()
        if self._graphed is None or self._graphed[0] != key:
            from ..graphs import GraphedStep

            self._graphed = (key, GraphedStep(self.one_step))
        return self._graphed[1]

    def _check_inputs(self, constants, prescribed, prognostic):
        if prognostic is None:
            raise _lib.DlwpError("prognostic must be given (reference forward signature)")
        _lib.require_cuda_tensor(prognostic, "prognostic")
        _lib.require_cuda_tensor(constants, "constants")
        _lib.require_cuda_tensor(prescribed, "prescribed")
        c = constants.contiguous() if constants is not None else None
        p = prescribed.contiguous() if prescribed is not None else None
        return c, p, prognostic.contiguous()

    def _grad_mode(self) -> bool:
        """train.py:263-271: `.train()` + autograd recording -> the differentiable rollout (dlwp_benchmark_amd/training.py:
        the hot kernels run their HIP forward inside autograd Functions, the pointwise layers are torch operators)."""
        return self.training and torch.is_grad_enabled()

    def _forward_train(self, constants, prescribed, prognostic):
        from ..rollout import rollout_train

        return rollout_train(self.one_step, self.context_size, constants, prescribed, prognostic)

    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        return self._ws

    def _param_key(self):
        return tuple((p._version, p.data_ptr()) for p in list(self.parameters()) + list(self.buffers()))
