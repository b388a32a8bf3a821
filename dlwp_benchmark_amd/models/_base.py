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

    # ---- precision forms (per-module state, no process-wide switch) ------------------------------------------------
    # name -> (attention precision, Linear form, block-tail MLP form).  "fp32" is the parity path of every mirror;
    # "f16x3" is fp32-grade too (two-part f16 splits); "bf16" is what the reference gets from
    # autocast(bfloat16) and what BASELINE.json names for the Swin and Pangu configs.
    COMPUTE_PRECISIONS = {"fp32": ("fp32", "bf16x6", "bf16x6"), "f16x3": ("fp32", "f16x3", "f16x3"),
                          "bf16attn": ("bf16", "bf16x6", "bf16x6"), "bf16": ("bf16", "bf16", "bf16x6")}

    def _set_on_submodules(self, attr: str, value) -> int:
        n = 0
        for m in self.modules():
            if attr in m.__dict__:
                setattr(m, attr, value)
                n += 1
        self._graphed = None        # a captured step graph baked the OLD kernels in: re-capture
        return n

    def set_attention_precision(self, precision: str):
        """"fp32" (default, parity path; "fp32_mfma" / "bf16x6" force one of its two fp32-accurate forms) or "bf16"
        (bf16 MFMA operands, fp32 accumulate / softmax)."""
        if precision not in ("fp32", "fp32_mfma", "bf16x6", "bf16"):
            raise _lib.DlwpError(f"unknown attention precision {precision!r}")
        self._set_on_submodules("attention_precision", precision)
        return self

    def set_linear_form(self, form: str):
        """"bf16x6" (default): the blocks' Linears run dlwp_linear_f32 (fp32-accurate on the bf16 matrix pipe, fused epilogues);
        "f16x3": dlwp_linear_f16x3 (fp32-grade, two-part f16 splits); "bf16": dlwp_linear_bf16 (bf16 operands, fp32
        accumulation -- nn.Linear under autocast(bfloat16)); "rocblas": torch's fp32 GEMMs (the cross-check)."""
        from .. import ops

        if form not in ops.LINEAR_FORMS:
            raise _lib.DlwpError(f"unknown linear form {form!r}")
        self._set_on_submodules("linear_form", form)
        return self

    def set_mlp_form(self, form: str):
        """FourCastNet block tails: "bf16x6" (default; fp32 products from three-part bf16 splits) or "f16x3" (two-part f16
        splits, dlwp_afno_block_tail_f16x3).  Both fp32-GEMM accurate."""
        if form not in ("bf16x6", "f16x3"):
            raise _lib.DlwpError(f"unknown MLP form {form!r}")
        self._set_on_submodules("mlp_form", form)
        return self

    def set_compute_precision(self, name: str):
        """ONE knob over the three above -- also the constructor kwarg `compute_precision` (a key the reference ignores:
        every reference constructor swallows unknown keys through **kwargs, so a config that carries it still builds
        there): "fp32" | "f16x3" | "bf16attn" | "bf16".  A drop-in user selects the config's named precision in
        `configs/model/*.yaml` and never calls a setter."""
        if name not in self.COMPUTE_PRECISIONS:
            raise _lib.DlwpError(f"unknown compute_precision {name!r} (one of {sorted(self.COMPUTE_PRECISIONS)})")
        attn, lin, mlp = self.COMPUTE_PRECISIONS[name]
        self._set_on_submodules("attention_precision", attn)
        self._set_on_submodules("linear_form", lin)
        self._set_on_submodules("mlp_form", mlp)
        self.compute_precision = name
        return self

    def _init_compute_precision(self, kwargs: dict):
        """called at the end of a mirror's constructor with its **kwargs"""
        cp = kwargs.get("compute_precision")
        self.compute_precision = "fp32"
        if cp is not None:
            self.set_compute_precision(str(cp))

    def invalidate_packed(self):
        """Derived operands (packed bf16 / f16 weight images, FNO plans, step graphs) are keyed on (data_ptr, _version) of
        their source parameters.  Writes through `.data` (EMA / init code: `p.data.copy_()`, `p.data.mul_()`) keep both --
        call this after such writes.  `load_state_dict` and `_apply` (.to / .half / .float) call it themselves."""
        from .. import ops

        ops.bump_pack_epoch()       # part of every derived-operand key (ops.pack_epoch) and of _param_key below
        self._graphed = None
        return self

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_packed()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate_packed()
        return r

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
        from .. import ops

        return (ops.pack_epoch(),) + tuple((p._version, p.data_ptr()) for p in list(self.parameters()) + list(self.buffers()))
