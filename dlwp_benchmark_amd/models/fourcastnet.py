"""FourCastNet (= AFNONet) -- drop-in for reference models/fourcastnet/fourcastnet.py:214-361,
registered as `FourCastNet` like the reference registry (models/__init__.py:6).  Same constructor
kwargs (:215-234), state-dict names (`patch_embed.proj`, `pos_embed`, `blocks.{i}.{norm1,filter.{w1,b1,w2,b2},
norm2,mlp.{fc1,fc2}}`, the unused `norm`, `head`) and forward signature.

Per block: LayerNorm 1 -> channels-first (HIP), hipFFT R2C, the AFNO2D frequency-domain work (:87-121: four
full-size zero buffers, slice-assigns, eight einsums, ReLU, softshrink) as ONE in-place HIP kernel that also carries
the "ortho" factors, hipFFT C2R, one merge kernel (+ bias path, first skip, back to tokens) and one token-MLP kernel
(LayerNorm 2 -> fc1 -> GELU -> fc2 -> second skip).  Patch + position embedding is one kernel for 1x1 patches, and so is
the head + rearrange at the other end (dlwp_patch_recover_1x1_f32; other patch sizes: torch GEMM + view).
The rollout is device resident and does NOT reproduce the reference's crash on the second step
(`.to()` on a list, :336-340) nor its per-step `.cpu()` (:359).
"""
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import lib as _lib
from .. import ops
from ..rollout import rollout_into
from ._base import HipBackbone
from .swin import _Mlp


class AFNO2D(nn.Module):
    def __init__(self, hidden_size, num_blocks=8, sparsity_threshold=0.01, hard_thresholding_fraction=1,
                 hidden_size_factor=1):
        super().__init__()
        if hidden_size % num_blocks:
            raise ValueError(f"hidden_size {hidden_size} should be divisble by num_blocks {num_blocks}")
        if hidden_size_factor != 1:
            raise NotImplementedError("hidden_size_factor != 1 is not used by the reference")
        self.hidden_size, self.num_blocks = hidden_size, num_blocks
        self.block_size = hidden_size // num_blocks
        self.sparsity_threshold = sparsity_threshold
        self.hard_thresholding_fraction = hard_thresholding_fraction
        bs = self.block_size
        self.w1 = nn.Parameter(0.02 * torch.randn(2, num_blocks, bs, bs))
        self.b1 = nn.Parameter(0.02 * torch.randn(2, num_blocks, bs))
        self.w2 = nn.Parameter(0.02 * torch.randn(2, num_blocks, bs, bs))
        self.b2 = nn.Parameter(0.02 * torch.randn(2, num_blocks, bs))

    def filter_cf(self, x_cf):
        """x_cf CHANNELS-FIRST [B, C, H, W] -> irfft2(mix(rfft2(x_cf))) (without the `+ bias` of :127)."""
        return ops.afno2d_filter_cf(x_cf, self.w1, self.b1, self.w2, self.b2, self.num_blocks, self.sparsity_threshold,
                                    self.hard_thresholding_fraction)

    def forward(self, x):
        """x [B, H, W, C] -> irfft2(mix(rfft2(x))) + x   (fourcastnet.py:78-127)"""
        x_cf = x.permute(0, 3, 1, 2).contiguous()
        return self.filter_cf(x_cf).permute(0, 2, 3, 1) + x


class _Block(nn.Module):
    def __init__(self, dim, mlp_ratio, num_blocks, sparsity_threshold, hard_thresholding_fraction):
        super().__init__()
        self.norm1 = ops.HipLayerNorm(dim, eps=1e-6)
        self.filter = AFNO2D(dim, num_blocks, sparsity_threshold, hard_thresholding_fraction)
        self.norm2 = ops.HipLayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self._mlp_packed = ops.TokenMlpWeights()   # derived operand layout, not part of the state dict
        self._mlp_fused = None
        self.mlp_form = "bf16x6"                   # or "f16x3": the block tail's products from two-part f16 splits (FourCastNet.set_mlp_form)

    def forward(self, x, l_cf=None, next_norm=None):
        """fourcastnet.py:180-193 (double skip).  LayerNorm1 writes channels-first for the FFT; the inverse
        layout change is fused with `+ bias`, the first skip and LayerNorm2 (dlwp_afno_merge_f32).
        l_cf: norm1(x) channels-first if the previous block's MLP kernel already produced it; next_norm: the next
        block's norm1, to be produced by this block's MLP kernel.  Returns (x_out, l_cf of the next block or None)."""
        if self.training and torch.is_grad_enabled():
            # training (train.py:263-271): the block as the reference composes it (fourcastnet.py:180-193); the filter runs
            # its HIP kernels forward AND backward inside an autograd Function (training._AfnoFilterFn), so do the MLP's two Linears
            # (ops.linear_any -> training._LinearFn); LayerNorm / GELU / adds are torch operators
            residual = x
            x = self.filter(self.norm1(x)) + residual
            return x + ops.linear_any(ops.linear_any(self.norm2(x), self.mlp.fc1, act=1), self.mlp.fc2), None
        if l_cf is None:
            l_cf = ops.layernorm_nhwc_to_nchw(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        f_cf = self.filter.filter_cf(l_cf)
        m = self.mlp
        if self._mlp_fused is None:
            self._mlp_fused = ops.token_mlp_supported(x.shape[-1], m.fc1.out_features)
        if self._mlp_fused:
            hw = x.shape[1] * x.shape[2]
            emit = None
            if next_norm is not None and hw % 32 == 0:
                emit = (next_norm.weight, next_norm.bias, next_norm.eps)
            if hw % 32 == 0:
                # `+ bias`, first skip, LayerNorm2, fc1 -> GELU -> fc2, second skip (:127, :187, :191-192) and the next
                # block's norm1 in ONE launch, in place on x (norm2's affine part lives in the packed fc1 operands)
                packed = self._mlp_packed.get(m.fc1.weight, m.fc2.weight, self.norm2.weight, self.norm2.bias, m.fc1.bias,
                                              merged=True, f16x3=self.mlp_form == "f16x3")
                res = ops.afno_block_tail(f_cf, l_cf, x, packed, m.fc2.bias, m.fc1.out_features, self.norm2.eps,
                                          emit_norm=emit, out=x if x.is_contiguous() else None, form=self.mlp_form)
                return res if emit is not None else (res, None)
            # odd token counts: merge kernel (sum only) + token MLP with LayerNorm2 fused
            s, _ = ops.afno_merge(f_cf, l_cf, x, None, None, self.norm2.eps, want_norm=False)
            packed = self._mlp_packed.get(m.fc1.weight, m.fc2.weight, self.norm2.weight, self.norm2.bias, m.fc1.bias)
            return ops.token_mlp(s, s, packed, None, m.fc2.bias, m.fc1.out_features, out=s, ln_eps=self.norm2.eps), None
        # other widths: second skip folded into the fc2 GEMM -- the merge kernel stores sum + fc2.bias, addmm adds
        # onto it (beta = 1)
        s, n = ops.afno_merge(f_cf, l_cf, x, self.norm2.weight, self.norm2.bias, self.norm2.eps, sum_bias=m.fc2.bias)
        hid = torch.nn.functional.gelu(torch.nn.functional.linear(n, m.fc1.weight, m.fc1.bias))
        c = x.shape[-1]
        s.view(-1, c).addmm_(hid.view(-1, hid.shape[-1]), m.fc2.weight.t())   # in place: no copy of the addend
        return s, None


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = tuple(img_size), tuple(patch_size)
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)

    def forward(self, x):
        if tuple(x.shape[2:]) != self.img_size:
            raise _lib.DlwpError(f"Input image size {tuple(x.shape[2:])} doesn't match model {self.img_size}")
        return self.proj(x).flatten(2).transpose(1, 2)


class FourCastNet(HipBackbone):
    def __init__(self, img_height=720, img_width=1440, patch_size=(16, 16), constant_channels: int = 4,
                 prescribed_channels: int = 0, prognostic_channels: int = 1, filter="AFNO2D", embed_dim=768, depth=12,
                 mlp_ratio=4., drop_rate=0., drop_path_rate=0., num_blocks=16, sparsity_threshold=0.01,
                 hard_thresholding_fraction=1.0, context_size: int = 1, use_pos_embed: bool = True, **kwargs):
        super().__init__()
        if filter != "AFNO2D":
            raise NotImplementedError(f"filter {filter!r}: only the in-tree AFNO2D filter is on the hot path")
        self.img_size = (int(img_height), int(img_width))
        self.patch_size = tuple(int(p) for p in patch_size)
        self.context_size = int(context_size)
        self.embed_dim, self.out_chans, self.use_pos_embed = embed_dim, prognostic_channels, use_pos_embed
        in_chans = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        self.patch_embed = _PatchEmbed(self.img_size, self.patch_size, in_chans, embed_dim)
        if use_pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
        self.h, self.w = self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]
        self.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio, num_blocks, sparsity_threshold,
                                            hard_thresholding_fraction) for _ in range(depth)])
        self.norm = ops.HipLayerNorm(embed_dim, eps=1e-6)   # in the reference state dict, never applied (:283-293)
        self.head = nn.Linear(embed_dim, self.out_chans * self.patch_size[0] * self.patch_size[1], bias=False)
        self._init_compute_precision(kwargs)     # `compute_precision: bf16` in configs/model/*.yaml (HipBackbone.set_compute_precision)

    def one_step(self, x: torch.Tensor) -> torch.Tensor:
        b = x.shape[0]
        proj = self.patch_embed.proj
        if self.training and torch.is_grad_enabled():
            x = self.patch_embed(x)
            x = x + self.pos_embed if self.use_pos_embed else x
        elif self.patch_size == (1, 1) and ops.patch_embed_1x1_supported(proj.in_channels, self.embed_dim):
            if tuple(x.shape[2:]) != self.img_size:
                raise _lib.DlwpError(f"Input image size {tuple(x.shape[2:])} doesn't match model {self.img_size}")
            x = ops.patch_embed_1x1(x, proj.weight, proj.bias, self.pos_embed[0] if self.use_pos_embed else None)
        else:
            x = self.patch_embed(x)   # a transposed view: materialise token-major ONCE, with the add
            tok = torch.empty(x.shape, device=x.device, dtype=x.dtype)
            x = torch.add(x, self.pos_embed, out=tok) if self.use_pos_embed else tok.copy_(x)
        x = x.reshape(b, self.h, self.w, self.embed_dim)
        l_cf = None   # norm1 of the next block, channels-first, when the previous block's MLP kernel produced it
        for i, blk in enumerate(self.blocks):
            nxt = self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None
            x, l_cf = blk(x, l_cf, nxt)
        p1, p2 = self.patch_size
        if (p1, p2) == (1, 1) and not (self.training and torch.is_grad_enabled()) and \
                ops.patch_recover_1x1_supported(self.embed_dim, self.out_chans):
            return ops.patch_recover_1x1(x, self.head.weight, self.head.bias, self.h, self.w)   # head + rearrange in one pass
        x = self.head(x)
        return x.view(b, self.h, self.w, p1, p2, self.out_chans).permute(0, 5, 1, 3, 2, 4).reshape(
            b, self.out_chans, self.h * p1, self.w * p2)

    def rollout_into(self, out, constants, prescribed, prognostic, step_begin=0, step_end=-1):
        return rollout_into(self._step_fn(), self.context_size, out, constants, prescribed, prognostic, step_begin, step_end)

    def forward(self, constants: Optional[torch.Tensor] = None, prescribed: Optional[torch.Tensor] = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        constants, prescribed, prognostic = self._check_inputs(constants, prescribed, prognostic)
        if self._grad_mode():
            return self._forward_train(constants, prescribed, prognostic)
        with torch.no_grad():
            b, t, cg, h, w = prognostic.shape
            if t <= self.context_size:
                raise _lib.DlwpError(f"need more than context_size={self.context_size} frames, got {t}")
            out = torch.empty(b, t - self.context_size, cg, h, w, device=prognostic.device, dtype=torch.float32)
            self.rollout_into(out, constants, prescribed, prognostic)
        return out


AFNONet = FourCastNet
