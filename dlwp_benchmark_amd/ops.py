"""Thin Python wrappers over the C ABI (include/dlwp_hip.h) for ops used by several backbones.
Every wrapper takes CUDA tensors, passes raw pointers + the current HIP stream, and raises
`DlwpError` on a non-zero status.  No wrapper has a CPU path."""
import ctypes
import weakref
import functools
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import lib as _lib

BIG = 1 << 30


@dataclass
class WindowSpec:
    """Geometry of one (shifted-)window attention call, see struct dlwp_wattn_desc."""
    grid: Sequence[int]
    padded: Sequence[int]
    pad_lead: Sequence[int]
    window: Sequence[int]
    shift_fwd: Sequence[int]
    shift_back: Sequence[int]
    use_mask: bool
    mask_b1: Sequence[int]
    mask_b2: Sequence[int]
    bias_mode: int
    heads: int
    head_dim: int
    scale: float
    form: int = -1          # dlwp_window_attn_f32: -1 by window size, 0 fp32 MFMA, 1 bf16x6 (struct dlwp_wattn_desc.form)

    def to_c(self) -> "_lib.WAttnDesc":
        d = _lib.WAttnDesc()
        for name in ("grid", "padded", "pad_lead", "window", "shift_fwd", "shift_back", "mask_b1", "mask_b2"):
            arr = getattr(d, name)
            for i, v in enumerate(getattr(self, name)):
                arr[i] = int(v)
        d.use_mask = int(self.use_mask)
        d.bias_mode = int(self.bias_mode)
        d.heads, d.head_dim, d.scale = int(self.heads), int(self.head_dim), float(self.scale)
        d.form = int(self.form)
        return d


def window_attention(qkv: torch.Tensor, qkv_bias: Optional[torch.Tensor], table: torch.Tensor,
                     spec: WindowSpec, precision: str = "fp32", count_fallbacks: bool = False):
    """qkv [B, L, 3*C] (qkv Linear output, un-padded token order) -> [B, L, C].
    precision "fp32": fp32-accurate products (parity path; the form of the contractions by window size),
    "fp32_mfma" / "bf16x6": the same with the form forced (each is the other's cross-check);
    "bf16": bf16 MFMA operands, fp32 accumulate.
    count_fallbacks=True (diagnostics, synchronises): returns (out, workgroups of the fast path that left the exponent
    slack of their softmax offset and were recomputed with the exact row maximum)."""
    forms = {"fp32": spec.form, "fp32_mfma": 0, "bf16x6": 1, "bf16": -1}
    if precision not in forms:
        raise _lib.DlwpError(f"unknown attention precision {precision!r}")
    from . import training as _T
    if _T.wants_grad(qkv, qkv_bias, table) and not count_fallbacks:
        return _T.window_attention(qkv, qkv_bias, table, spec, precision)     # HIP forward, differentiable (training.py)
    if qkv.dtype == torch.bfloat16:
        # the hand-over of a block in the all-bf16 form: bf16 qkv in, bf16 attention output (dlwp_window_attn_bf16_io)
        if precision != "bf16" or count_fallbacks:
            raise _lib.DlwpError("window_attention: a bfloat16 qkv tensor goes with precision='bf16'")
        return _window_attention_bf16_io(qkv, qkv_bias, table, spec)
    _lib.require_cuda_tensor(qkv, "qkv")
    _lib.require_cuda_tensor(table, "bias table")
    _lib.require_cuda_tensor(qkv_bias, "qkv bias")
    qkv = qkv.contiguous()
    table = table.contiguous()
    b, l, c3 = qkv.shape
    c = spec.heads * spec.head_dim
    if c3 != 3 * c or l != spec.grid[0] * spec.grid[1] * spec.grid[2]:
        raise _lib.DlwpError(f"qkv shape {tuple(qkv.shape)} does not match grid {tuple(spec.grid)} x 3*{c}")
    out = torch.empty(b, l, c, device=qkv.device, dtype=torch.float32)
    lib = _lib.load()
    d = spec.to_c()
    d.form = forms[precision]
    with torch.cuda.device(qkv.device):
        bf16 = precision == "bf16"
        fn = lib.dlwp_window_attn_bf16 if bf16 else lib.dlwp_window_attn_f32
        # the caller (torch's caching allocator) owns the workspace of the fast path; 0 bytes = generic kernel
        nbytes = int(lib.dlwp_window_attn_workspace_bytes(ctypes.byref(d), b, 1 if bf16 else 0))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device) if nbytes else None
        _lib.check(fn(ctypes.byref(d), qkv.data_ptr(),
                      qkv_bias.contiguous().data_ptr() if qkv_bias is not None else None,
                      table.data_ptr(), out.data_ptr(), b, ws.data_ptr() if ws is not None else None, nbytes,
                      _lib.stream_ptr()),
                   "dlwp_window_attn_" + ("bf16" if precision == "bf16" else "f32"))
        if count_fallbacks:
            n = ctypes.c_int32(0)
            if ws is not None:
                _lib.check(lib.dlwp_window_attn_fallbacks(ctypes.byref(d), b, 1 if bf16 else 0, ws.data_ptr(),
                                                          _lib.stream_ptr(), ctypes.byref(n)), "dlwp_window_attn_fallbacks")
            return out, int(n.value)
    return out


def window_attention_io_supported(spec: WindowSpec, batch: int) -> bool:
    """True when dlwp_window_attn_bf16_io covers the descriptor (one of the two fast kernels takes it)."""
    d = spec.to_c()
    d.form = -1
    return int(_lib.load().dlwp_window_attn_workspace_bytes(ctypes.byref(d), int(batch), 1)) > 0


_BIAS16 = {}


def _bias_bf16(bias: torch.Tensor) -> torch.Tensor:
    """bfloat16 image of a qkv bias (read by the earth-window kernel for zero-padded tokens), converted once per parameter
    state instead of once per call: keyed like the packed weights (pointer, version, pack epoch of invalidate_packed())."""
    key = (bias.data_ptr(), bias._version, str(bias.device), pack_epoch())
    hit = _BIAS16.get(id(bias))
    if hit is None or hit[0] != key or hit[2]() is not bias:      # (the weak reference: ids are reused after a free)
        if len(_BIAS16) > 256:
            _BIAS16.clear()
        hit = (key, bias.detach().to(torch.bfloat16).contiguous(), weakref.ref(bias))
        _BIAS16[id(bias)] = hit
    return hit[1]


def _window_attention_bf16_io(qkv: torch.Tensor, qkv_bias: Optional[torch.Tensor], table: torch.Tensor, spec: WindowSpec):
    if not qkv.is_cuda:
        raise _lib.DlwpError("qkv must be a tensor on an MI355X device")
    _lib.require_cuda_tensor(table, "bias table")
    qkv, table = qkv.contiguous(), table.contiguous()
    b, l, c3 = qkv.shape
    c = spec.heads * spec.head_dim
    if c3 != 3 * c or l != spec.grid[0] * spec.grid[1] * spec.grid[2]:
        raise _lib.DlwpError(f"qkv shape {tuple(qkv.shape)} does not match grid {tuple(spec.grid)} x 3*{c}")
    bias16 = _bias_bf16(qkv_bias) if qkv_bias is not None else None
    out = torch.empty(b, l, c, device=qkv.device, dtype=torch.bfloat16)
    lib = _lib.load()
    d = spec.to_c()
    d.form = -1
    with torch.cuda.device(qkv.device):
        nbytes = int(lib.dlwp_window_attn_workspace_bytes(ctypes.byref(d), b, 1))
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=qkv.device)
        rc = lib.dlwp_window_attn_bf16_io(ctypes.byref(d), qkv.data_ptr(), bias16.data_ptr() if bias16 is not None else None,
                                          table.data_ptr(), out.data_ptr(), b, ws.data_ptr(), nbytes, _lib.stream_ptr())
    if rc == -2:     # DLWP_ERR_UNSUPPORTED: a descriptor only the generic kernel takes -- the same arithmetic on fp32 tensors
        return window_attention(qkv.float(), qkv_bias, table, spec, precision="bf16").to(torch.bfloat16)
    _lib.check(rc, "dlwp_window_attn_bf16_io")
    return out


def window_attention_backward(qkv: torch.Tensor, qkv_bias: Optional[torch.Tensor], table: torch.Tensor, spec: WindowSpec,
                              grad_out: torch.Tensor):
    """Gradients of window_attention (fp32) with respect to qkv, the qkv bias (through zero-padded tokens; None when the
    descriptor does not pad or no bias was given) and the bias table: dlwp_window_attn_bwd_f32, flash-style -- the scores are
    recomputed per tile, no [B, heads, N, N] tensor exists (reference backward: scripts/train.py:271 through
    swin_transformer.py:122-154 / panguweather.py:176-211)."""
    _lib.require_cuda_tensor(qkv, "qkv")
    _lib.require_cuda_tensor(table, "bias table")
    _lib.require_cuda_tensor(qkv_bias, "qkv bias")
    _lib.require_cuda_tensor(grad_out, "grad_out")
    qkv, table, grad_out = qkv.contiguous(), table.contiguous(), grad_out.contiguous()
    b, l, c3 = qkv.shape
    c = spec.heads * spec.head_dim
    if c3 != 3 * c or l != spec.grid[0] * spec.grid[1] * spec.grid[2] or tuple(grad_out.shape) != (b, l, c):
        raise _lib.DlwpError(f"qkv {tuple(qkv.shape)} / grad_out {tuple(grad_out.shape)} do not match grid {tuple(spec.grid)} x 3*{c}")
    padded = tuple(spec.padded) != tuple(spec.grid)
    gqkv = torch.empty_like(qkv)
    gtab = torch.empty_like(table)
    gbias = torch.empty(3 * c, device=qkv.device, dtype=torch.float32) if (padded and qkv_bias is not None) else None
    lib = _lib.load()
    d = spec.to_c()
    with torch.cuda.device(qkv.device):
        nbytes = int(lib.dlwp_window_attn_bwd_workspace_bytes(ctypes.byref(d), b))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device)
        _lib.check(lib.dlwp_window_attn_bwd_f32(ctypes.byref(d), qkv.data_ptr(),
                                                qkv_bias.contiguous().data_ptr() if qkv_bias is not None else None,
                                                table.data_ptr(), grad_out.data_ptr(), gqkv.data_ptr(),
                                                gbias.data_ptr() if gbias is not None else None, gtab.data_ptr(), b,
                                                ws.data_ptr(), nbytes, _lib.stream_ptr()), "dlwp_window_attn_bwd_f32")
    return gqkv, gbias, gtab


ACTS = {"none": 0, "gelu": 1, "tanh": 2, "relu": 3, "silu": 4}


def act_code(activation) -> int:
    """Maps the reference's activation spec (module instance or the config string that the
    reference `eval`s, e.g. "th.nn.GELU()", unet.py:292) to the kernel's activation id."""
    n = activation if isinstance(activation, str) else type(activation).__name__
    for key, tag in (("GELU", "gelu"), ("Tanh", "tanh"), ("LeakyReLU", None), ("ReLU", "relu"), ("SiLU", "silu"),
                     ("Identity", "none")):
        if key in n:
            if tag is None:
                break
            return ACTS[tag]
    raise _lib.DlwpError(f"activation {activation!r} has no fused kernel (supported: GELU, Tanh, ReLU, SiLU)")


def conv3x3_cyl(x0: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0,
                x1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CylinderPad(1) + Conv2d(3x3) + bias + activation on cat([x0, x1], 1) without the cat."""
    from . import training as _T
    if _T.wants_grad(x0, x1, weight, bias):
        return conv3x3(x0, weight, bias, act=act, x1=x1)
    _lib.require_cuda_tensor(x0, "x0")
    _lib.require_cuda_tensor(x1, "x1")
    _lib.require_cuda_tensor(weight, "weight")
    x0 = x0.contiguous()
    x1 = x1.contiguous() if x1 is not None else None
    weight = weight.contiguous()
    b, c0, h, w = x0.shape
    c1 = x1.shape[1] if x1 is not None else 0
    cout = weight.shape[0]
    if tuple(weight.shape[1:]) != (c0 + c1, 3, 3):
        raise _lib.DlwpError(f"weight {tuple(weight.shape)} does not match {c0}+{c1} input channels, 3x3")
    y = torch.empty(b, cout, h, w, device=x0.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x0.device):
        _lib.check(lib.dlwp_conv3x3_cyl_f32(x0.data_ptr(), c0, x1.data_ptr() if x1 is not None else None, c1,
                                            weight.data_ptr(), bias.contiguous().data_ptr() if bias is not None else None,
                                            y.data_ptr(), b, h, w, cout, act, _lib.stream_ptr()), "dlwp_conv3x3_cyl_f32")
    return y


def conv3x3_hpx(x0: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0,
                x1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HEALPixPadding(1) + Conv2d(3x3) + bias + activation on cat([x0, x1], 1); x [(B*12), C, H, W]."""
    from . import healpix as _hpx
    from . import training as _T
    if _T.wants_grad(x0, x1, weight, bias):
        return conv3x3(x0, weight, bias, act=act, x1=x1, hpx=True)

    _lib.require_cuda_tensor(x0, "x0")
    _lib.require_cuda_tensor(x1, "x1")
    _lib.require_cuda_tensor(weight, "weight")
    x0 = x0.contiguous()
    x1 = x1.contiguous() if x1 is not None else None
    weight = weight.contiguous()
    n, c0, h, w = x0.shape
    c1 = x1.shape[1] if x1 is not None else 0
    cout = weight.shape[0]
    if n % 12:
        raise _lib.DlwpError(f"leading dimension {n} is not (batch * 12 faces)")
    if tuple(weight.shape[1:]) != (c0 + c1, 3, 3):
        raise _lib.DlwpError(f"weight {tuple(weight.shape)} does not match {c0}+{c1} input channels, 3x3")
    table = _hpx.device_table(h, w, 1, x0.device)
    y = torch.empty(n, cout, h, w, device=x0.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x0.device):
        _lib.check(lib.dlwp_conv3x3_hpx_f32(x0.data_ptr(), c0, x1.data_ptr() if x1 is not None else None, c1,
                                            weight.data_ptr(), bias.contiguous().data_ptr() if bias is not None else None,
                                            y.data_ptr(), n, h, w, cout, act, table.data_ptr(), _lib.stream_ptr()),
                   "dlwp_conv3x3_hpx_f32")
    return y


def conv3x3(x0: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0,
            x1: Optional[torch.Tensor] = None, pre_act: int = 0, resid: Optional[torch.Tensor] = None,
            hpx: bool = False) -> torch.Tensor:
    """pad(1) + Conv2d(3x3) on cat([x0, x1], 1) with the input activation `pre_act` applied while staging and
    `resid` added before `act`; padding rule: CylinderPad, or HEALPixPadding when hpx (x [(B*12), C, H, W])."""
    for t, n in ((x0, "x0"), (x1, "x1"), (weight, "weight"), (resid, "resid")):
        _lib.require_cuda_tensor(t, n)
    from . import training as _T
    if _T.wants_grad(x0, x1, weight, bias, resid):
        return _T.conv3x3(x0, weight, bias, act=act, x1=x1, pre_act=pre_act, resid=resid, hpx=hpx)   # HIP forward, differentiable
    x0 = x0.contiguous()
    x1 = x1.contiguous() if x1 is not None else None
    weight = weight.contiguous()
    n, c0, h, w = x0.shape
    c1 = x1.shape[1] if x1 is not None else 0
    cout = weight.shape[0]
    if tuple(weight.shape[1:]) != (c0 + c1, 3, 3):
        raise _lib.DlwpError(f"weight {tuple(weight.shape)} does not match {c0}+{c1} input channels, 3x3")
    if resid is not None and (tuple(resid.shape) != (n, cout, h, w) or not resid.is_contiguous()):
        raise _lib.DlwpError("conv3x3: resid must be contiguous and shaped like the output")
    table = None
    if hpx:
        from . import healpix as _hpx

        if n % 12:
            raise _lib.DlwpError(f"leading dimension {n} is not (batch * 12 faces)")
        table = _hpx.device_table(h, w, 1, x0.device)
    y = torch.empty(n, cout, h, w, device=x0.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x0.device):
        _lib.check(lib.dlwp_conv3x3_ex_f32(x0.data_ptr(), c0, x1.data_ptr() if x1 is not None else None, c1, weight.data_ptr(),
                                           bias.contiguous().data_ptr() if bias is not None else None,
                                           resid.data_ptr() if resid is not None else None, y.data_ptr(), n, h, w, cout,
                                           int(pre_act), int(act), table.data_ptr() if table is not None else None,
                                           _lib.stream_ptr()), "dlwp_conv3x3_ex_f32")
    return y


def groupnorm_act(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], groups: int,
                  eps: float = 1e-5, act: int = 0) -> torch.Tensor:
    """act(GroupNorm(groups)(x)) for x [N, C, ...] (reference unet.py:739 + :761, :887-888), one launch."""
    _lib.require_cuda_tensor(x, "x")
    from . import training as _T
    if _T.wants_grad(x, weight, bias):
        return _T._ACT_FNS[int(act)](torch.nn.functional.group_norm(x, int(groups), weight, bias, eps))
    x = x.contiguous()
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    y = torch.empty_like(x)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.dlwp_groupnorm_act_f32(x.data_ptr(), weight.contiguous().data_ptr() if weight is not None else None,
                                              bias.contiguous().data_ptr() if bias is not None else None, y.data_ptr(), n, c,
                                              hw, int(groups), float(eps), int(act), _lib.stream_ptr()),
                   "dlwp_groupnorm_act_f32")
    return y


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, padding: int = 0,
           pre_act: int = 0, act: int = 0, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """zero-padded Conv2d (square kernel / stride / padding): the strided and 1x1 convolutions of unet.py:583-584, :879, :450."""
    for t, n in ((x, "x"), (weight, "weight"), (resid, "resid")):
        _lib.require_cuda_tensor(t, n)
    from . import training as _T
    if _T.wants_grad(x, weight, bias, resid):
        y = torch.nn.functional.conv2d(_T._ACT_FNS[int(pre_act)](x), weight, bias, stride=stride, padding=padding)
        return _T._ACT_FNS[int(act)](y if resid is None else y + resid)
    x, weight = x.contiguous(), weight.contiguous()
    n, cin, h, w = x.shape
    cout, cin_w, k, k2 = weight.shape
    if cin_w != cin or k != k2:
        raise _lib.DlwpError(f"conv2d: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}")
    oh, ow = (h + 2 * padding - k) // stride + 1, (w + 2 * padding - k) // stride + 1
    if resid is not None and (tuple(resid.shape) != (n, cout, oh, ow) or not resid.is_contiguous()):
        raise _lib.DlwpError("conv2d: resid must be contiguous and shaped like the output")
    y = torch.empty(n, cout, oh, ow, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.dlwp_conv2d_f32(x.data_ptr(), weight.data_ptr(), bias.contiguous().data_ptr() if bias is not None else None,
                                       resid.data_ptr() if resid is not None else None, y.data_ptr(), n, cin, h, w, cout, k,
                                       int(stride), int(padding), int(pre_act), int(act), _lib.stream_ptr()), "dlwp_conv2d_f32")
    return y


def conv_transpose2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, padding: int = 0,
                     act: int = 0) -> torch.Tensor:
    """ConvTranspose2d (weight [cin, cout, k, k]; unet.py:523 2x2 s2, :719 4x4 s2 p1)."""
    _lib.require_cuda_tensor(x, "x")
    _lib.require_cuda_tensor(weight, "weight")
    from . import training as _T
    if _T.wants_grad(x, weight, bias):
        return _T._ACT_FNS[int(act)](torch.nn.functional.conv_transpose2d(x, weight, bias, stride=stride, padding=padding))
    x, weight = x.contiguous(), weight.contiguous()
    n, cin, h, w = x.shape
    cin_w, cout, k, k2 = weight.shape
    if cin_w != cin or k != k2:
        raise _lib.DlwpError(f"conv_transpose2d: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}")
    oh, ow = (h - 1) * stride - 2 * padding + k, (w - 1) * stride - 2 * padding + k
    y = torch.empty(n, cout, oh, ow, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.dlwp_conv_transpose2d_f32(x.data_ptr(), weight.data_ptr(),
                                                 bias.contiguous().data_ptr() if bias is not None else None, y.data_ptr(), n,
                                                 cin, h, w, cout, k, int(stride), int(padding), int(act), _lib.stream_ptr()),
                   "dlwp_conv_transpose2d_f32")
    return y


def avgpool2x2(x: torch.Tensor) -> torch.Tensor:
    """AvgPool2d(kernel_size=2, stride=2) (unet.py:450)."""
    _lib.require_cuda_tensor(x, "x")
    if torch.is_grad_enabled() and x.requires_grad:
        return torch.nn.functional.avg_pool2d(x, 2)
    x = x.contiguous()
    n, c, h, w = x.shape
    y = torch.empty(n, c, h // 2, w // 2, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.dlwp_avgpool2x2_f32(x.data_ptr(), y.data_ptr(), n * c, h, w, _lib.stream_ptr()), "dlwp_avgpool2x2_f32")
    return y


def small_module(m: torch.nn.Module, x: torch.Tensor, act: int = 0) -> torch.Tensor:
    """Runs one of the U-Net family's non-3x3 layers through its HIP kernel: AvgPool2d(2), ConvTranspose2d, Conv2d
    (zero padding, square); raises for anything else so that nothing silently falls back to a torch op."""
    nn = torch.nn
    if isinstance(m, nn.AvgPool2d):
        k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
        s_ = m.stride if isinstance(m.stride, int) else m.stride[0]
        if k != 2 or s_ != 2 or m.padding not in (0, (0, 0)):
            raise _lib.DlwpError("only AvgPool2d(2, 2) has a kernel")
        return avgpool2x2(x)
    if isinstance(m, nn.ConvTranspose2d):
        if m.kernel_size[0] != m.kernel_size[1] or m.stride[0] != m.stride[1] or m.padding[0] != m.padding[1] or \
                m.output_padding != (0, 0) or m.dilation != (1, 1) or m.groups != 1:
            raise _lib.DlwpError("ConvTranspose2d: only square kernel / stride / padding without output_padding, dilation, groups")
        return conv_transpose2d(x, m.weight, m.bias, m.stride[0], m.padding[0], act)
    if isinstance(m, nn.Conv2d):
        if m.kernel_size[0] != m.kernel_size[1] or m.stride[0] != m.stride[1] or m.padding[0] != m.padding[1] or \
                m.dilation != (1, 1) or m.groups != 1 or m.padding_mode != "zeros":
            raise _lib.DlwpError("Conv2d: only square kernel / stride / zero padding without dilation or groups")
        return conv2d(x, m.weight, m.bias, m.stride[0], m.padding[0], act=act)
    raise _lib.DlwpError(f"no HIP kernel for {type(m).__name__}")


def healpix_pad(x: torch.Tensor, padding: int) -> torch.Tensor:
    """HEALPixPadding(padding) (reference utils/healpix.py:165-368): [(B*12), C, H, W] -> [(B*12), C, H+2p, W+2p]."""
    from . import healpix as _hpx

    _lib.require_cuda_tensor(x, "x")
    x = x.contiguous()
    n, c, h, w = x.shape
    if n % 12:
        raise _lib.DlwpError(f"leading dimension {n} is not (batch * 12 faces)")
    table = _hpx.device_table(h, w, int(padding), x.device)
    if torch.is_grad_enabled() and x.requires_grad:
        if int(padding) != 1:
            raise _lib.DlwpError("differentiable HEALPix padding is built for padding 1")
        from . import training as _T
        return _T._hpx_pad_torch(x, table)
    y = torch.empty(n, c, h + 2 * padding, w + 2 * padding, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.dlwp_healpix_pad_f32(x.data_ptr(), y.data_ptr(), table.data_ptr(), n, c, h, w, int(padding),
                                            _lib.stream_ptr()), "dlwp_healpix_pad_f32")
    return y


def convlstm_gates(gates: torch.Tensor, c_prev: torch.Tensor):
    _lib.require_cuda_tensor(gates, "gates")
    _lib.require_cuda_tensor(c_prev, "c_prev")
    if torch.is_grad_enabled() and (gates.requires_grad or c_prev.requires_grad):     # convlstm.py:96-109 with torch operators
        netin, ig, fg, og = torch.split(gates, gates.shape[1] // 4, dim=1)
        c_new = torch.sigmoid(fg) * c_prev + torch.sigmoid(ig) * torch.tanh(netin)
        return torch.sigmoid(og) * torch.tanh(c_new), c_new
    gates, c_prev = gates.contiguous(), c_prev.contiguous()
    b, c4, h, w = gates.shape
    hid = c4 // 4
    h_out, c_out = torch.empty_like(c_prev), torch.empty_like(c_prev)
    lib = _lib.load()
    with torch.cuda.device(gates.device):
        _lib.check(lib.dlwp_convlstm_gates_f32(gates.data_ptr(), c_prev.data_ptr(), h_out.data_ptr(), c_out.data_ptr(),
                                               b, hid, h, w, _lib.stream_ptr()), "dlwp_convlstm_gates_f32")
    return h_out, c_out


def afno2d_mix(xf_cf: torch.Tensor, w1, b1, w2, b2, num_blocks: int, sparsity_threshold: float,
               hard_thresholding_fraction: float) -> torch.Tensor:
    """xf_cf complex64 CHANNELS-FIRST [B, C, H, Wf] (rfft2 of a [B, C, H, W] tensor) -> mixed spectrum,
    same shape and layout."""
    if not xf_cf.is_cuda or xf_cf.dtype != torch.complex64:
        raise _lib.DlwpError("afno2d_mix needs a complex64 CUDA tensor")
    xc = xf_cf.contiguous()
    b, c, h, wf = xc.shape
    xr = torch.view_as_real(xc)
    yr = torch.empty_like(xr)
    lib = _lib.load()
    with torch.cuda.device(xc.device):
        _lib.check(lib.dlwp_afno2d_mix_f32(xr.data_ptr(), yr.data_ptr(), w1.contiguous().data_ptr(),
                                           b1.contiguous().data_ptr(), w2.contiguous().data_ptr(),
                                           b2.contiguous().data_ptr(), b, h, wf, c, num_blocks,
                                           float(sparsity_threshold), float(hard_thresholding_fraction),
                                           _lib.stream_ptr()), "dlwp_afno2d_mix_f32")
    return torch.view_as_complex(yr)


class _Fft2Plans:
    """hipFFT plan pairs (R2C + C2R) per (device, batch, H, W), created on first use and kept for the process."""

    def __init__(self):
        self._plans = {}

    def get(self, device, batch: int, h: int, w: int):
        key = (str(device), batch, h, w)
        if key not in self._plans:
            lib = _lib.load()
            handle = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.dlwp_fft2_plan_create(ctypes.byref(handle), batch, h, w), "dlwp_fft2_plan_create")
            self._plans[key] = handle
        return self._plans[key]


_fft2_plans = _Fft2Plans()


class _AfnoFftPlans:
    """twiddle tables of the hand-written kept-column FFTs per (device, H, W, kept columns)"""

    def __init__(self):
        self._plans = {}

    def get(self, device, h: int, w: int, kc: int):
        key = (str(device), h, w, kc)
        if key not in self._plans:
            lib = _lib.load()
            handle = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.dlwp_afno_fft_plan_create(ctypes.byref(handle), h, w, kc, _lib.stream_ptr()),
                           "dlwp_afno_fft_plan_create")
            self._plans[key] = handle
        return self._plans[key]


_afno_fft_plans = _AfnoFftPlans()


def afno_kept_cols(h: int, w: int, hard_thresholding_fraction: float) -> int:
    """columns of the half spectrum the filter keeps: fourcastnet.py:93-94 (`total_modes = H // 2 + 1` -- the
    reference takes it from the FIRST spatial axis -- `kept_modes = int(total_modes * fraction)`, columns [:kept])"""
    return max(0, min(int((h // 2 + 1) * hard_thresholding_fraction), w // 2 + 1))


def afno2d_filter_backward(x_cf: torch.Tensor, grad_y: torch.Tensor, w1, b1, w2, b2, num_blocks: int,
                           sparsity_threshold: float, hard_thresholding_fraction: float):
    """Gradients of afno2d_filter_cf with respect to (x_cf, w1, b1, w2, b2), or None when the grid is not one of the
    hand-written kept-column transforms / the block size is not 4, 8 or 16 (the caller then differentiates the torch form).
    grad_x = C2R(mix_bwd(R2C(x), R2C(grad_y))): the same two hand-written transforms as the forward around
    dlwp_afno2d_mix_bwd_f32 (which recomputes the per-point MLP); the weight gradients are sums over the spectrum points of
    per-point factors the kernel writes -- four complex einsums (reference backward: train.py:271 through fourcastnet.py:85-124)."""
    _lib.require_cuda_tensor(x_cf, "x_cf")
    _lib.require_cuda_tensor(grad_y, "grad_y")
    x_cf, grad_y = x_cf.contiguous(), grad_y.contiguous()
    b, c, h, w = x_cf.shape
    lib = _lib.load()
    kc = afno_kept_cols(h, w, hard_thresholding_fraction)
    bs = c // num_blocks
    if kc < 1 or not lib.dlwp_afno_fft_supported(h, w, kc) or bs not in (4, 8, 16):
        return None
    scale = 1.0 / float(h * w) ** 0.5
    with torch.cuda.device(x_cf.device):
        st = _lib.stream_ptr()
        plan = _afno_fft_plans.get(x_cf.device, h, w, kc)
        mk = lambda: torch.empty(b, c, h, kc, 2, device=x_cf.device, dtype=torch.float32)
        xf, gf, gxf, xin, o1, d1, d2 = mk(), mk(), mk(), mk(), mk(), mk(), mk()
        _lib.check(lib.dlwp_afno_rfft2_kept_f32(plan, x_cf.data_ptr(), xf.data_ptr(), b * c, st), "dlwp_afno_rfft2_kept_f32")
        _lib.check(lib.dlwp_afno_rfft2_kept_f32(plan, grad_y.data_ptr(), gf.data_ptr(), b * c, st), "dlwp_afno_rfft2_kept_f32")
        ptr = lambda t: t.detach().contiguous().data_ptr()
        _lib.check(lib.dlwp_afno2d_mix_bwd_f32(xf.data_ptr(), gf.data_ptr(), gxf.data_ptr(), xin.data_ptr(), o1.data_ptr(),
                                               d1.data_ptr(), d2.data_ptr(), ptr(w1), ptr(b1), ptr(w2), ptr(b2), b, h, kc, c,
                                               num_blocks, w, float(sparsity_threshold), float(hard_thresholding_fraction),
                                               scale, scale, st), "dlwp_afno2d_mix_bwd_f32")
        gx = torch.empty_like(x_cf)
        _lib.check(lib.dlwp_afno_irfft2_kept_f32(plan, gxf.data_ptr(), gx.data_ptr(), b * c, st), "dlwp_afno_irfft2_kept_f32")
    cv = lambda t: torch.view_as_complex(t).view(b, num_blocks, bs, h, kc)
    gw1 = torch.einsum("bnihk,bnohk->nio", cv(xin).conj(), cv(d1))
    gw2 = torch.einsum("bnihk,bnohk->nio", cv(o1).conj(), cv(d2))
    gb1 = cv(d1).sum(dim=(0, 3, 4))
    gb2 = cv(d2).sum(dim=(0, 3, 4))
    ri = lambda t: torch.stack([t.real, t.imag], dim=0).contiguous()       # reference layout [2, nb, bs(, bs)]
    return gx, ri(gw1), ri(gb1), ri(gw2), ri(gb2)


def afno2d_filter_cf(x_cf: torch.Tensor, w1, b1, w2, b2, num_blocks: int, sparsity_threshold: float,
                     hard_thresholding_fraction: float, use_rocfft: bool = False) -> torch.Tensor:
    """irfft2(mix(rfft2(x_cf, norm="ortho")), norm="ortho") for CHANNELS-FIRST x_cf [B, C, H, W]
    (fourcastnet.py:87-123 without the `+ bias` of :127).  Three launches: the hand-written forward transform that
    produces only the kept columns, the mixing kernel in place on that [B, C, H, kept] spectrum (it carries the two
    1/sqrt(HW) factors of norm="ortho"), the hand-written inverse transform.  Grids that are not instantiated (and
    use_rocfft=True, the cross-check) take the hipFFT path on the full half spectrum."""
    _lib.require_cuda_tensor(x_cf, "x_cf")
    from . import training as _T
    if _T.wants_grad(x_cf, w1, b1, w2, b2):
        return _T.afno_filter(x_cf, w1, b1, w2, b2, num_blocks, sparsity_threshold, hard_thresholding_fraction)
    x_cf = x_cf.contiguous()
    b, c, h, w = x_cf.shape
    lib = _lib.load()
    scale = 1.0 / float(h * w) ** 0.5
    kc = afno_kept_cols(h, w, hard_thresholding_fraction)
    y = torch.empty_like(x_cf)
    with torch.cuda.device(x_cf.device):
        st = _lib.stream_ptr()
        if not use_rocfft and kc >= 1 and lib.dlwp_afno_fft_supported(h, w, kc):
            plan = _afno_fft_plans.get(x_cf.device, h, w, kc)
            spec = torch.empty(b, c, h, kc, 2, device=x_cf.device, dtype=torch.float32)
            _lib.check(lib.dlwp_afno_rfft2_kept_f32(plan, x_cf.data_ptr(), spec.data_ptr(), b * c, st), "dlwp_afno_rfft2_kept_f32")
            _lib.check(lib.dlwp_afno2d_mix_scaled_f32(spec.data_ptr(), spec.data_ptr(), w1.contiguous().data_ptr(),
                                                      b1.contiguous().data_ptr(), w2.contiguous().data_ptr(),
                                                      b2.contiguous().data_ptr(), b, h, kc, c, num_blocks,
                                                      float(sparsity_threshold), float(hard_thresholding_fraction),
                                                      scale, scale, st), "dlwp_afno2d_mix_scaled_f32")
            _lib.check(lib.dlwp_afno_irfft2_kept_f32(plan, spec.data_ptr(), y.data_ptr(), b * c, st), "dlwp_afno_irfft2_kept_f32")
            return y
        wf = w // 2 + 1
        spec = torch.empty(b, c, h, wf, 2, device=x_cf.device, dtype=torch.float32)
        plan = _fft2_plans.get(x_cf.device, b * c, h, w)
        _lib.check(lib.dlwp_rfft2_f32(plan, x_cf.data_ptr(), spec.data_ptr(), st), "dlwp_rfft2_f32")
        _lib.check(lib.dlwp_afno2d_mix_scaled_f32(spec.data_ptr(), spec.data_ptr(), w1.contiguous().data_ptr(),
                                                  b1.contiguous().data_ptr(), w2.contiguous().data_ptr(),
                                                  b2.contiguous().data_ptr(), b, h, wf, c, num_blocks,
                                                  float(sparsity_threshold), float(hard_thresholding_fraction),
                                                  scale, scale, st), "dlwp_afno2d_mix_scaled_f32")
        _lib.check(lib.dlwp_irfft2_f32(plan, spec.data_ptr(), y.data_ptr(), st), "dlwp_irfft2_f32")
    return y


def layernorm_nhwc_to_nchw(x: torch.Tensor, weight, bias, eps: float) -> torch.Tensor:
    """x [B, H, W, C] -> LayerNorm over C, returned channels-first [B, C, H, W]."""
    _lib.require_cuda_tensor(x, "x")
    x = x.contiguous()
    b, h, w, c = x.shape
    y = torch.empty(b, c, h, w, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.dlwp_layernorm_nhwc_to_nchw_f32(x.data_ptr(), weight.contiguous().data_ptr(),
                                                       bias.contiguous().data_ptr(), y.data_ptr(), b, h * w, c,
                                                       float(eps), _lib.stream_ptr()), "dlwp_layernorm_nhwc_to_nchw_f32")
    return y


def afno_merge(f_nchw: torch.Tensor, l_nchw: torch.Tensor, x_nhwc: torch.Tensor, weight, bias, eps: float,
               sum_bias: Optional[torch.Tensor] = None, want_norm: bool = True):
    """(f + l) transposed to token-major + x -> (sum [+ sum_bias], LayerNorm(sum)), both [B, H, W, C].
    want_norm=False returns (sum, None): for a consumer that normalises on the fly (token_mlp(ln_eps=...))."""
    for t, n in ((f_nchw, "f"), (l_nchw, "l"), (x_nhwc, "x")):
        _lib.require_cuda_tensor(t, n)
    f_nchw, l_nchw, x_nhwc = f_nchw.contiguous(), l_nchw.contiguous(), x_nhwc.contiguous()
    b, h, w, c = x_nhwc.shape
    s = torch.empty_like(x_nhwc)
    n = torch.empty_like(x_nhwc) if want_norm else None
    lib = _lib.load()
    with torch.cuda.device(x_nhwc.device):
        _lib.check(lib.dlwp_afno_merge_f32(f_nchw.data_ptr(), l_nchw.data_ptr(), x_nhwc.data_ptr(),
                                           weight.contiguous().data_ptr() if want_norm else None,
                                           bias.contiguous().data_ptr() if want_norm else None,
                                           sum_bias.contiguous().data_ptr() if sum_bias is not None else None,
                                           s.data_ptr(), n.data_ptr() if want_norm else None, b, h * w, c, float(eps),
                                           _lib.stream_ptr()),
                   "dlwp_afno_merge_f32")
    return s, n


def patch_embed_1x1_supported(in_channels: int, channels: int) -> bool:
    return in_channels <= 32 and 4 <= channels <= 256 and channels % 4 == 0


def patch_embed_1x1(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                    pos: Optional[torch.Tensor]) -> torch.Tensor:
    """x [B, Cin, H, W], weight [C, Cin, 1, 1] (Conv2d with 1x1 patches), pos [H*W, C] or None ->
    tokens [B, H*W, C] = conv(x).flatten(2).transpose(1, 2) + pos   (fourcastnet.py:530-543, :286-288)."""
    _lib.require_cuda_tensor(x, "x")
    x = x.contiguous()
    b, cin, h, w = x.shape
    c = weight.shape[0]
    if pos is not None and (tuple(pos.shape) != (h * w, c) or not pos.is_contiguous()):
        raise _lib.DlwpError(f"patch embed: pos must be contiguous [{h * w}, {c}], got {tuple(pos.shape)}")
    out = torch.empty(b, h * w, c, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.dlwp_patch_embed_1x1_f32(x.data_ptr(), weight.contiguous().data_ptr(),
                                                bias.contiguous().data_ptr() if bias is not None else None,
                                                pos.data_ptr() if pos is not None else None, out.data_ptr(), b, cin,
                                                h * w, c, _lib.stream_ptr()), "dlwp_patch_embed_1x1_f32")
    return out


def concat_channels(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """torch.cat(parts, dim=1) for [B, Ci, H, W] float32 tensors whose (Ci, H, W) block is contiguous (any batch stride: views into the
    inputs and the trajectory buffer) -- `_prepare_inputs` of the rollout loop (swin_transformer.py:679-692) as one 16-byte copy kernel.
    Parts the kernel does not take (other dtypes, inner strides, > 8 parts, plane not a multiple of 4) go through torch.cat: the same
    copy, written by torch."""
    parts = list(parts)
    p0 = parts[0]
    b, h, w = p0.shape[0], p0.shape[2], p0.shape[3]
    plane = h * w
    ok = 2 <= len(parts) <= 8 and plane % 4 == 0 and b <= 65535        # (one part: torch's copy is one call with less host work)
    for t in parts:
        ok = ok and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.shape[0] == b and tuple(t.shape[2:]) == (h, w) \
            and t.stride(3) == 1 and t.stride(2) == w and t.stride(1) == plane and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
    if not ok:
        return torch.cat(parts, dim=1)
    n = len(parts)
    ctot = sum(int(t.shape[1]) for t in parts)
    out = torch.empty(b, ctot, h, w, device=p0.device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in parts])
    chans = (ctypes.c_int32 * n)(*[int(t.shape[1]) for t in parts])
    strides = (ctypes.c_int64 * n)(*[int(t.stride(0)) for t in parts])
    lib = _lib.load()
    with torch.cuda.device(p0.device):
        _lib.check(lib.dlwp_concat_channels_f32(ptrs, chans, strides, n, out.data_ptr(), b, plane, _lib.stream_ptr()),
                   "dlwp_concat_channels_f32")
    return out


def patch_recover_1x1_supported(channels: int, out_channels: int) -> bool:
    return channels % 4 == 0 and channels <= 256 and 0 < out_channels <= 16


def patch_recover_1x1(tokens: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], h: int, w: int) -> torch.Tensor:
    """tokens [B, H, W, C] (or [B, H*W, C]) token-major, weight [Cout, C] (the head Linear of a 1x1-patch backbone) ->
    [B, Cout, H, W] = head(tokens) rearranged "b h w c -> b c h w"   (fourcastnet.py:144, :296-303)."""
    _lib.require_cuda_tensor(tokens, "tokens")
    tokens = tokens.contiguous()
    b, c = tokens.shape[0], tokens.shape[-1]
    cout = weight.shape[0]
    if tokens.numel() != b * h * w * c:
        raise _lib.DlwpError(f"patch recover: tokens {tuple(tokens.shape)} do not hold {h} x {w} tokens per sample")
    out = torch.empty(b, cout, h, w, device=tokens.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(tokens.device):
        _lib.check(lib.dlwp_patch_recover_1x1_f32(tokens.data_ptr(), weight.detach().contiguous().data_ptr(),
                                                  bias.detach().contiguous().data_ptr() if bias is not None else None,
                                                  out.data_ptr(), b, h * w, c, cout, _lib.stream_ptr()),
                   "dlwp_patch_recover_1x1_f32")
    return out


def token_mlp_supported(channels: int, hidden: int) -> bool:
    """True when dlwp_token_mlp_f32 handles this (channels, hidden) pair."""
    return int(_lib.load().dlwp_token_mlp_packed_bytes(int(channels), int(hidden))) > 0


# Derived operands are keyed on (data_ptr, _version) of their source parameters; writes through `.data` change neither.
# HipBackbone.invalidate_packed() (called by load_state_dict / _apply, and by users after `.data` writes) bumps this epoch,
# which is part of every key below: all packed images, plans and step graphs of the process are re-derived on next use.
_PACK_EPOCH = [0]


def pack_epoch() -> int:
    return _PACK_EPOCH[0]


def bump_pack_epoch() -> int:
    _PACK_EPOCH[0] += 1
    return _PACK_EPOCH[0]


class TokenMlpWeights:
    """fc1 / fc2 weights of a token MLP in the operand layout of dlwp_token_mlp_f32, re-packed on the device
    whenever a parameter has been written to (optimizer step, load_state_dict, .to()).  With `ln_weight` / `ln_bias`
    (and `b1`) the affine part of the LayerNorm in front of fc1 is folded into the operands, for token_mlp(ln_eps=...)."""

    def __init__(self):
        self._key = None
        self._buf = None

    def get(self, w1: torch.Tensor, w2: torch.Tensor, ln_weight: Optional[torch.Tensor] = None,
            ln_bias: Optional[torch.Tensor] = None, b1: Optional[torch.Tensor] = None, merged: bool = False,
            f16x3: bool = False) -> torch.Tensor:
        """merged=True: the k-slot order afno_block_tail wants (a different permutation of W1's columns);
        f16x3=True: the two f16 images of the f16x3 product form (afno_block_tail(form="f16x3"))."""
        extra = [t for t in (ln_weight, ln_bias, b1) if t is not None]
        key = tuple((t.data_ptr(), t._version) for t in (w1, w2, *extra)) + (str(w1.device), ln_weight is not None, merged, f16x3, pack_epoch())
        if key != self._key:
            hid, c = w1.shape
            if tuple(w2.shape) != (c, hid):
                raise _lib.DlwpError(f"token MLP: fc2.weight {tuple(w2.shape)} does not match fc1.weight {tuple(w1.shape)}")
            if (ln_weight is None) != (ln_bias is None):
                raise _lib.DlwpError("token MLP: LayerNorm weight and bias must be given together")
            lib = _lib.load()
            nbytes = int(lib.dlwp_token_mlp_packed_bytes(c, hid))
            if nbytes == 0:
                raise _lib.DlwpError(f"token MLP: unsupported shape channels={c} hidden={hid}")
            buf = torch.empty(nbytes // 4, dtype=torch.int32, device=w1.device)

            def ptr(t):
                return t.detach().contiguous().data_ptr() if t is not None else None

            with torch.cuda.device(w1.device):
                packer = "dlwp_token_mlp_pack_f16x3" if f16x3 else "dlwp_token_mlp_pack_f32"
                _lib.check(getattr(lib, packer)(ptr(w1), ptr(w2), ptr(ln_weight), ptr(ln_bias), ptr(b1), c, hid,
                                                1 if merged else 0, buf.data_ptr(), _lib.stream_ptr()), packer)
            self._key, self._buf = key, buf
        return self._buf


def token_mlp(n: torch.Tensor, resid: Optional[torch.Tensor], packed: torch.Tensor, b1: Optional[torch.Tensor],
              b2: Optional[torch.Tensor], hidden: int, out: Optional[torch.Tensor] = None,
              ln_eps: Optional[float] = None, emit_norm=None):
    """out = resid + b2 + fc2(gelu(fc1(n)))  over the last dimension (fourcastnet.py:41-57, :191-192), one launch.
    With `ln_eps` the kernel first LayerNorms `n` (packed must carry the folded affine part and fc1 bias, see
    TokenMlpWeights.get; b1 is then unused).  `out` may be `resid` / `n` itself (in place).
    emit_norm = (weight, bias, eps) of the NEXT block's first LayerNorm: n must be [B, H, W, C]; returns
    (out, LayerNorm(out) channels-first [B, C, H, W]) -- the next block's `layernorm_nhwc_to_nchw` for free."""
    _lib.require_cuda_tensor(n, "n")
    n = n.contiguous()
    c = n.shape[-1]
    if resid is not None:
        _lib.require_cuda_tensor(resid, "resid")
        if resid.shape != n.shape or not resid.is_contiguous():
            raise _lib.DlwpError("token MLP: resid must be contiguous and shaped like n")
    if ln_eps is None and b1 is None:
        raise _lib.DlwpError("token MLP: fc1 bias missing")
    if out is None:
        out = torch.empty_like(n)
    elif out.shape != n.shape or not out.is_contiguous():
        raise _lib.DlwpError("token MLP: out must be contiguous and shaped like n")
    lib = _lib.load()
    common = (n.data_ptr(), resid.data_ptr() if resid is not None else None, packed.data_ptr(),
              b1.contiguous().data_ptr() if b1 is not None else None,
              b2.contiguous().data_ptr() if b2 is not None else None,
              out.data_ptr(), n.numel() // c, c, int(hidden), float(ln_eps) if ln_eps is not None else -1.0)
    with torch.cuda.device(n.device):
        if emit_norm is None:
            _lib.check(lib.dlwp_token_mlp_f32(*common, _lib.stream_ptr()), "dlwp_token_mlp_f32")
            return out
        if n.dim() != 4:
            raise _lib.DlwpError("token MLP emit_norm: n must be [B, H, W, C]")
        gamma, beta, eps = emit_norm
        b, h, w, _ = n.shape
        nxt = torch.empty(b, c, h, w, device=n.device, dtype=torch.float32)
        _lib.check(lib.dlwp_token_mlp_emit_norm_f32(*common, gamma.contiguous().data_ptr(), beta.contiguous().data_ptr(),
                                                    float(eps), nxt.data_ptr(), h * w, _lib.stream_ptr()),
                   "dlwp_token_mlp_emit_norm_f32")
    return out, nxt


def afno_block_tail(f_cf: torch.Tensor, l_cf: torch.Tensor, x_nhwc: torch.Tensor, packed: torch.Tensor,
                    b2: Optional[torch.Tensor], hidden: int, ln_eps: float, emit_norm=None, out: Optional[torch.Tensor] = None,
                    form: str = "bf16x6"):
    """Everything of an AFNO block after the inverse FFT, one launch (fourcastnet.py:127, :187, :191-192):
    sum = f_cf + l_cf + x;  out = sum + fc2(gelu(fc1(LayerNorm(sum)))).  f_cf / l_cf [B, C, H, W], x / out [B, H, W, C];
    packed = TokenMlpWeights.get(..., norm2.weight, norm2.bias, fc1.bias, merged=True).
    emit_norm = (weight, bias, eps) of the next block's norm1 -> returns (out, LayerNorm(out) [B, C, H, W]).
    form "bf16x6" (three-part bf16 splits, six products) or "f16x3" (two-part f16 splits, three products; `packed` must
    come from TokenMlpWeights.get(..., f16x3=True)) -- both fp32-GEMM accurate, the operands here are LayerNorm / GELU outputs."""
    if form not in ("bf16x6", "f16x3"):
        raise _lib.DlwpError(f"afno_block_tail: unknown form {form!r}")
    for t, nm in ((f_cf, "f_cf"), (l_cf, "l_cf"), (x_nhwc, "x")):
        _lib.require_cuda_tensor(t, nm)
    f_cf, l_cf, x_nhwc = f_cf.contiguous(), l_cf.contiguous(), x_nhwc.contiguous()
    b, h, w, c = x_nhwc.shape
    if tuple(f_cf.shape) != (b, c, h, w) or tuple(l_cf.shape) != (b, c, h, w):
        raise _lib.DlwpError("afno_block_tail: f_cf / l_cf must be [B, C, H, W] matching x [B, H, W, C]")
    if out is None:
        out = torch.empty_like(x_nhwc)
    nxt = torch.empty(b, c, h, w, device=x_nhwc.device, dtype=torch.float32) if emit_norm is not None else None
    gamma, beta, eps = emit_norm if emit_norm is not None else (None, None, 0.0)
    lib = _lib.load()
    with torch.cuda.device(x_nhwc.device):
        name = "dlwp_afno_block_tail_f16x3" if form == "f16x3" else "dlwp_afno_block_tail_f32"
        _lib.check(getattr(lib, name)(f_cf.data_ptr(), l_cf.data_ptr(), x_nhwc.data_ptr(), packed.data_ptr(),
                                                b2.contiguous().data_ptr() if b2 is not None else None, out.data_ptr(), b,
                                                h * w, c, int(hidden), float(ln_eps),
                                                gamma.contiguous().data_ptr() if gamma is not None else None,
                                                beta.contiguous().data_ptr() if beta is not None else None, float(eps),
                                                nxt.data_ptr() if nxt is not None else None, _lib.stream_ptr()),
                   name)
    return (out, nxt) if emit_norm is not None else out


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
               pre_bias: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """LayerNorm over the last dimension (any leading shape); pre_bias [C] is added to x before the statistics.
    out_dtype=torch.bfloat16: the result rounded to bfloat16 (dlwp_layernorm_prebias_bf16out) -- for a bf16-form Linear, which
    would round it the same way itself."""
    _lib.require_cuda_tensor(x, "x")
    from . import training as _T
    if _T.wants_grad(x, weight, bias, pre_bias):     # training: the pointwise layers are torch operators (training.py)
        return torch.nn.functional.layer_norm(x if pre_bias is None else x + pre_bias, (x.shape[-1],), weight, bias, eps)
    x = x.contiguous()
    c = x.shape[-1]
    rows = x.numel() // c
    ob16 = out_dtype == torch.bfloat16
    y = torch.empty_like(x, dtype=torch.bfloat16) if ob16 else torch.empty_like(x)
    lib = _lib.load()
    name = "dlwp_layernorm_prebias_bf16out" if ob16 else "dlwp_layernorm_prebias_f32"
    with torch.cuda.device(x.device):
        _lib.check(getattr(lib, name)(x.data_ptr(), pre_bias.contiguous().data_ptr() if pre_bias is not None else None,
                                      weight.contiguous().data_ptr(), bias.contiguous().data_ptr(),
                                      y.data_ptr(), rows, c, float(eps), _lib.stream_ptr()), name)
    return y


@functools.lru_cache(maxsize=None)
def linear_supported(in_features: int, out_features: int) -> bool:
    """True when dlwp_linear_f32 handles this Linear shape (in % 32 == 0, out % 4 == 0)."""
    return int(_lib.load().dlwp_linear_packed_bytes(int(out_features), int(in_features))) > 0


class LinearWeights:
    """A Linear weight [out, in] split into the three bf16 images dlwp_linear_f32 reads, re-split on the device whenever
    the parameter has been written to (optimizer step, load_state_dict, .to()).  Derived data: not in any state dict."""

    def __init__(self):
        self._key = None
        self._buf = None
        self._f16 = None         # the f16x3 images live in a LinearWeights of their own (same derivation rule)

    def get(self, weight: torch.Tensor, f16: bool = False) -> torch.Tensor:
        if f16:
            if self._f16 is None:
                self._f16 = LinearWeights()
            return self._f16._get(weight, "dlwp_linear_pack_f16x3")
        return self._get(weight, "dlwp_linear_pack_f32")

    def _get(self, weight: torch.Tensor, packer: str) -> torch.Tensor:
        key = (weight.data_ptr(), weight._version, str(weight.device), pack_epoch())
        if key != self._key:
            n, k = weight.shape
            lib = _lib.load()
            nbytes = int(lib.dlwp_linear_packed_bytes(n, k))
            if nbytes == 0:
                raise _lib.DlwpError(f"linear: unsupported shape out={n} in={k} (need in % 32 == 0 and out % 4 == 0)")
            buf = torch.empty(nbytes // 4, dtype=torch.int32, device=weight.device)
            with torch.cuda.device(weight.device):
                _lib.check(getattr(lib, packer)(weight.detach().contiguous().data_ptr(), n, k, buf.data_ptr(),
                                                _lib.stream_ptr()), packer)
            self._key, self._buf = key, buf
        return self._buf


def linear_raw(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x [..., K] @ weight[N, K]^T + bias through dlwp_linear_f32 (fp32-accurate) for a weight that is a plain tensor -- the
    three GEMMs of a Linear's training step (training._LinearFn); the weight is split on the device per call."""
    _lib.require_cuda_tensor(x, "x")
    _lib.require_cuda_tensor(weight, "weight")
    _lib.require_cuda_tensor(bias, "bias")
    x, weight = x.contiguous(), weight.contiguous()
    n, k = weight.shape
    if x.shape[-1] != k:
        raise _lib.DlwpError(f"linear: input width {x.shape[-1]} does not match in_features {k}")
    lib = _lib.load()
    nbytes = int(lib.dlwp_linear_packed_bytes(n, k))
    if nbytes == 0:
        raise _lib.DlwpError(f"linear: unsupported shape out={n} in={k} (need in % 32 == 0 and out % 4 == 0)")
    out = torch.empty((*x.shape[:-1], n), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        packed = torch.empty(nbytes // 4, dtype=torch.int32, device=x.device)
        _lib.check(lib.dlwp_linear_pack_f32(weight.data_ptr(), n, k, packed.data_ptr(), _lib.stream_ptr()), "dlwp_linear_pack_f32")
        _lib.check(lib.dlwp_linear_f32(x.data_ptr(), packed.data_ptr(), bias.contiguous().data_ptr() if bias is not None else None,
                                       None, out.data_ptr(), x.numel() // k, k, n, 0, _lib.stream_ptr()), "dlwp_linear_f32")
    return out


def linear(x: torch.Tensor, m: torch.nn.Linear, act: int = 0, resid: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, precision: str = "fp32", out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """act(x @ m.weight.T + m.bias) + resid over the last dimension in one launch; act 0 none / 1 exact GELU.
    precision "bf16" only: x may BE a bfloat16 tensor, or out_dtype=torch.bfloat16 asks for a bfloat16 result (no residual) --
    dlwp_linear_bf16_io, the hand-over of an MLP's hidden activation at half the bytes, bit-identical to the fp32 hand-over.
    precision "fp32": dlwp_linear_f32, fp32-accurate GEMM on the bf16 matrix pipe (six products of exact three-way splits);
    "bf16": dlwp_linear_bf16, bf16 operands and fp32 accumulation (what autocast(bfloat16) makes of nn.Linear);
    "f16x3": dlwp_linear_f16x3, fp32-GEMM accuracy from exact two-part f16 splits (three products; |x| < 65504).
    `out` may be `resid` (in-place residual add).  With gradients wanted the torch operators run instead (training.py's
    convention)."""
    if precision not in ("fp32", "bf16", "f16x3"):
        raise _lib.DlwpError(f"linear: unknown precision {precision!r}")
    from . import training as _T
    if _T.wants_grad(x, m.weight, m.bias, resid):
        rows = x.numel() // max(x.shape[-1], 1)
        if x.is_cuda and x.dtype == torch.float32 and _T._LinearFn.supported(rows, m.in_features, m.out_features):
            y = _T.linear_fn(x, m.weight, m.bias)         # HIP GEMMs forward and backward (training._LinearFn)
        else:
            y = torch.nn.functional.linear(x, m.weight, m.bias)
        if act == 1:
            y = torch.nn.functional.gelu(y)
        elif act != 0:
            raise _lib.DlwpError(f"linear: activation {act} not supported")
        return y if resid is None else y + resid
    x_bf16 = x.dtype == torch.bfloat16
    o_bf16 = out_dtype == torch.bfloat16 or (out is not None and out.dtype == torch.bfloat16)
    if x_bf16 or o_bf16:
        if precision != "bf16" or (o_bf16 and resid is not None):
            raise _lib.DlwpError("linear: bfloat16 tensors are taken by precision='bf16', without a residual on a bfloat16 output")
        if not x.is_cuda:
            raise _lib.DlwpError("linear: x must be a CUDA tensor")
    else:
        _lib.require_cuda_tensor(x, "x")
    if act not in (0, 1):
        raise _lib.DlwpError(f"linear: activation {act} not supported")
    x = x.contiguous()
    k, n = m.in_features, m.out_features
    if x.shape[-1] != k:
        raise _lib.DlwpError(f"linear: input width {x.shape[-1]} does not match in_features {k}")
    cache = m.__dict__.get("_dlwp_packed")
    if cache is None:
        cache = LinearWeights()
        m.__dict__["_dlwp_packed"] = cache       # plain attribute: neither parameter nor buffer, not in the state dict
    packed = cache.get(m.weight, f16=precision == "f16x3")
    shape = (*x.shape[:-1], n)
    if resid is not None:
        _lib.require_cuda_tensor(resid, "resid")
        if tuple(resid.shape) != shape or not resid.is_contiguous():
            raise _lib.DlwpError("linear: resid must be contiguous and shaped like the output")
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.bfloat16 if o_bf16 else torch.float32)
    elif tuple(out.shape) != shape or not out.is_contiguous():
        raise _lib.DlwpError("linear: out must be contiguous and shaped like the output")
    if out.data_ptr() == x.data_ptr():
        raise _lib.DlwpError("linear: out must not alias x")
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if x_bf16 or o_bf16:
            _lib.check(lib.dlwp_linear_bf16_io(x.data_ptr(), packed.data_ptr(),
                                               m.bias.contiguous().data_ptr() if m.bias is not None else None,
                                               resid.data_ptr() if resid is not None else None, out.data_ptr(), x.numel() // k,
                                               k, n, int(act), int(x_bf16), int(o_bf16), _lib.stream_ptr()), "dlwp_linear_bf16_io")
            return out
        name = {"fp32": "dlwp_linear_f32", "bf16": "dlwp_linear_bf16", "f16x3": "dlwp_linear_f16x3"}[precision]
        _lib.check(getattr(lib, name)(x.data_ptr(), packed.data_ptr(),
                                      m.bias.contiguous().data_ptr() if m.bias is not None else None,
                                      resid.data_ptr() if resid is not None else None, out.data_ptr(), x.numel() // k, k, n,
                                      int(act), _lib.stream_ptr()), name)
    return out


def linear_any(x: torch.Tensor, m: torch.nn.Linear, act: int = 0) -> torch.Tensor:
    """act(m(x)) for any Linear: through linear() (HIP kernel; with gradients wanted its differentiable form) on a GPU, the
    module itself elsewhere (CPU construction / registry tests) or when the kernel does not take the shape."""
    from . import training as _T
    if x.is_cuda and x.dtype == torch.float32 and (_T.wants_grad(x, m.weight, m.bias) or linear_supported(m.in_features, m.out_features)):
        return linear(x, m, act=act)
    y = m(x)
    return torch.nn.functional.gelu(y) if act == 1 else y


LINEAR_FORMS = ("bf16x6", "f16x3", "bf16", "rocblas")
_FORM_PRECISION = {"bf16x6": "fp32", "f16x3": "f16x3", "bf16": "bf16"}


def form_precision(form: str) -> str:
    """the `precision` argument of linear() a model's linear_form stands for"""
    return _FORM_PRECISION[form]


def linear_as(form: str, x: torch.Tensor, m: torch.nn.Linear) -> torch.Tensor:
    """m(x) through dlwp_linear_f32 (form "bf16x6") / dlwp_linear_bf16 ("bf16") when the shape is covered, else the module
    itself (rocBLAS fp32)."""
    if form in _FORM_PRECISION and x.is_cuda and linear_supported(m.in_features, m.out_features):
        return linear(x, m, precision=_FORM_PRECISION[form])
    return m(x)


class ConvAsLinear:
    """A Conv2d / ConvTranspose2d whose kernel equals its stride (no overlap, no padding) as a Linear over the channels of
    TOKEN-MAJOR data: ConvTranspose2d(Cin, Cout, k, k) = Linear(Cin -> k k Cout) + a pixel shuffle, Conv2d(Cin, Cout, 1) =
    Linear(Cin -> Cout).  The derived weight / bias follow the module's parameters (data_ptr, version); out_features is
    padded to a multiple of 4 with zero rows (dlwp_linear_f32's store width).  Used by the Swin decoder
    (swin_transformer.py:600-612, :672-677), which the reference runs as MIOpen convolutions on channels-first copies."""

    def __init__(self, conv: torch.nn.Module):
        self.conv = conv
        self.transposed = isinstance(conv, torch.nn.ConvTranspose2d)
        k = conv.kernel_size
        if k[0] != k[1] or tuple(conv.stride) != tuple(k) or conv.padding not in (0, (0, 0)) or conv.groups != 1 or \
                conv.dilation not in (1, (1, 1)) or (not self.transposed and k[0] != 1) or \
                (self.transposed and conv.output_padding not in (0, (0, 0))):
            raise _lib.DlwpError(f"ConvAsLinear: {conv} is not a kernel = stride convolution")
        self.k = k[0]
        self.in_features = conv.in_channels
        self.cout = conv.out_channels
        n = self.k * self.k * self.cout if self.transposed else self.cout
        self.out_features = (n + 3) // 4 * 4
        self._n = n
        self._key = None
        self.weight = None
        self.bias = None

    def refresh(self):
        w, b = self.conv.weight, self.conv.bias
        key = (w.data_ptr(), w._version, str(w.device), None if b is None else (b.data_ptr(), b._version), pack_epoch())
        if key == self._key:
            return
        with torch.no_grad():
            if self.transposed:      # [Cin, Cout, k, k] -> [(di k + dj) Cout + co][ci]
                wl = w.permute(2, 3, 1, 0).reshape(self._n, self.in_features)
                bl = None if b is None else b.repeat(self.k * self.k)
            else:                    # [Cout, Cin, 1, 1]
                wl = w.reshape(self._n, self.in_features)
                bl = b
            if self.out_features != self._n:
                wl = torch.cat([wl, wl.new_zeros(self.out_features - self._n, self.in_features)])
                bl = None if bl is None else torch.cat([bl, bl.new_zeros(self.out_features - self._n)])
            self.weight = wl.contiguous()
            self.bias = None if bl is None else bl.contiguous()
        self._key = key

    def __call__(self, x: torch.Tensor, h: int, w: int, act: int = 0, precision: str = "fp32") -> torch.Tensor:
        """x [B, h*w, Cin] token-major -> [B, (h k)*(w k), Cout] token-major."""
        self.refresh()
        b = x.shape[0]
        y = linear(x, self, act=act, precision=precision)
        if self.out_features != self._n:
            y = y[..., :self._n]
        if self.transposed and self.k > 1:
            k = self.k
            y = y.reshape(b, h, w, k, k, self.cout).permute(0, 1, 3, 2, 4, 5).reshape(b, h * k * w * k, self.cout)
        return y.contiguous()


def attention_block_linears_supported(dim: int, hidden: int) -> bool:
    return linear_supported(dim, 3 * dim) and linear_supported(dim, dim) and linear_supported(dim, hidden) \
        and linear_supported(hidden, dim)


def attention_block_tail(x: torch.Tensor, attn_out: torch.Tensor, proj: torch.nn.Linear, norm2: torch.nn.LayerNorm,
                         fc1: torch.nn.Linear, fc2: torch.nn.Linear, precision: str = "fp32") -> torch.Tensor:
    """`x = x + proj(attn_out); x = x + fc2(gelu(fc1(norm2(x))))` of a Swin / Pangu block (swin_transformer.py:254-262,
    panguweather.py:318-322), IN PLACE on x: three dlwp_linear_f32 launches (bias, GELU and both residual adds in the GEMM
    epilogues) and one LayerNorm."""
    linear(attn_out, proj, resid=x, out=x, precision=precision)
    # bf16 form: LayerNorm output and hidden activation cross HBM as bfloat16 (their consumers round to bf16 anyway: bit-identical)
    b16 = torch.bfloat16 if precision == "bf16" else None
    n2 = layer_norm(x, norm2.weight, norm2.bias, norm2.eps, out_dtype=b16)
    hid = linear(n2, fc1, act=1, precision=precision, out_dtype=b16)
    linear(hid, fc2, resid=x, out=x, precision=precision)
    return x


def residual_block_tail(x: torch.Tensor, pend: Optional[torch.Tensor], attn_out: torch.Tensor, proj: torch.nn.Linear,
                        norm2: torch.nn.LayerNorm, fc1: torch.nn.Linear, fc2: torch.nn.Linear):
    """`x = x + proj(attn_out); x = x + fc2(gelu(fc1(norm2(x))))` of a Swin / Pangu block (swin_transformer.py:254-262,
    panguweather.py:318-322) with both residual adds as the beta = 1 accumulation of the GEMMs, IN PLACE on x, and
    the Linear biases deferred: x holds (true x - pend); returns the new pend.  The caller adds pend once per layer."""
    c = x.shape[-1]
    x2 = x.view(-1, c)
    x2.addmm_(attn_out.reshape(-1, c), proj.weight.t())
    if proj.bias is not None:
        pend = proj.bias if pend is None else pend + proj.bias
    n2 = layer_norm(x, norm2.weight, norm2.bias, norm2.eps, pre_bias=pend)
    hid = torch.nn.functional.gelu(torch.nn.functional.linear(n2, fc1.weight, fc1.bias))
    x2.addmm_(hid.view(-1, hid.shape[-1]), fc2.weight.t())
    if fc2.bias is not None:
        pend = fc2.bias if pend is None else pend + fc2.bias
    return pend


class HipLayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm (same parameters / state-dict names) whose forward runs dlwp_layernorm_f32."""

    def forward(self, x):
        if len(self.normalized_shape) != 1 or self.weight is None or self.bias is None or x.shape[-1] % 4:
            raise _lib.DlwpError("HipLayerNorm needs a 1-D affine normalized_shape with channels % 4 == 0")
        return layer_norm(x, self.weight, self.bias, self.eps)
