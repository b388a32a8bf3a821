"""Thin Python wrappers over the C ABI (include/dlwp_hip.h) for ops used by several backbones.
Every wrapper takes CUDA tensors, passes raw pointers + the current HIP stream, and raises
`DlwpError` on a non-zero status.  No wrapper has a CPU path."""
import ctypes
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

    def to_c(self) -> "_lib.WAttnDesc":
        d = _lib.WAttnDesc()
        for name in ("grid", "padded", "pad_lead", "window", "shift_fwd", "shift_back", "mask_b1", "mask_b2"):
            arr = getattr(d, name)
            for i, v in enumerate(getattr(self, name)):
                arr[i] = int(v)
        d.use_mask = int(self.use_mask)
        d.bias_mode = int(self.bias_mode)
        d.heads, d.head_dim, d.scale = int(self.heads), int(self.head_dim), float(self.scale)
        return d


def window_attention(qkv: torch.Tensor, qkv_bias: Optional[torch.Tensor], table: torch.Tensor,
                     spec: WindowSpec) -> torch.Tensor:
    """qkv [B, L, 3*C] (qkv Linear output, un-padded token order) -> [B, L, C]."""
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
    with torch.cuda.device(qkv.device):
        _lib.check(lib.dlwp_window_attn_f32(ctypes.byref(d), qkv.data_ptr(),
                                            qkv_bias.contiguous().data_ptr() if qkv_bias is not None else None,
                                            table.data_ptr(), out.data_ptr(), b, _lib.stream_ptr()),
                   "dlwp_window_attn_f32")
    return out
