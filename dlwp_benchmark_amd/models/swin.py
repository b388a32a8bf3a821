"""SwinTransformer -- drop-in for reference models/swintransformer/swin_transformer.py:466-742 on
MI355X (inference rollout).  Same class name, constructor kwargs (:490-516), state-dict names and
shapes (incl. the int64 `relative_position_index` buffers) and forward signature.

What runs where:
  * everything between the qkv and proj Linears of every block -- pad / roll / window partition /
    bias gather / per-call shift-mask build / softmax(QK^T)V / reverse / roll / crop
    (:217-251, :122-154, :383-401) -- is ONE HIP kernel, `dlwp_window_attn_f32` (flash-style
    streaming: the reference "window" is the whole map, N = 2048, so the N x N scores never exist);
    the index buffer is kept only for checkpoint compatibility and is never read on the device;
  * Linear / LayerNorm / (transposed) convolutions go to rocBLAS / MIOpen through torch on the GPU;
  * the rollout loop is device resident (dlwp_benchmark_amd/rollout.py).
`.train()` returns self (the reference returns None, :739-742).
"""
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import lib as _lib
from .. import ops
from ..rollout import rollout_into
from ._base import HipBackbone


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None):
        super().__init__()
        wh, ww = int(window_size[0]), int(window_size[1])
        self.window_size = (wh, ww)
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += wh - 1
        rel[:, :, 1] += ww - 1
        rel[:, :, 0] *= 2 * ww - 1
        self.register_buffer("relative_position_index", rel.sum(-1))   # checkpoint compatibility only
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio, qkv_bias, qk_scale, norm_layer):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.window_size = (int(window_size[0]), int(window_size[1]))
        self.shift_size = (int(shift_size[0]), int(shift_size[1]))
        self.norm1 = norm_layer(dim)
        self.attn = _WindowAttention(dim, self.window_size, num_heads, qkv_bias, qk_scale)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.attention_precision = "fp32"
        self.linear_form = "bf16x6"      # "bf16x6": dlwp_linear_f32; "bf16": dlwp_linear_bf16; "rocblas": fp32 rocBLAS GEMMs

    def forward(self, x, h, w, pend=None):
        """x holds (true x - pend); returns (x, pend) in the same convention (pend None = nothing pending)."""
        wh, ww = self.window_size
        if h % wh or w % ww:
            raise _lib.DlwpError("feature map is not a multiple of the window: the reference pads with swapped axes "
                                 "here (swin_transformer.py:219-222); not reproduced on the device")
        sh, sw = self.shift_size
        shifted = sh > 0 or sw > 0
        spec = ops.WindowSpec(
            grid=(1, h, w), padded=(1, h, w), pad_lead=(0, 0, 0), window=(1, wh, ww),
            shift_fwd=(0, sh, sw), shift_back=(0, sh, sw), use_mask=shifted,
            # region ids from the slices of swin_transformer.py:385-390 (shift = window // 2 there)
            mask_b1=(ops.BIG, h - wh, w - ww), mask_b2=(ops.BIG, h - wh // 2, w - ww // 2),
            bias_mode=0, heads=self.num_heads, head_dim=self.dim // self.num_heads, scale=self.attn.scale)
        deferred = isinstance(self.norm1, ops.HipLayerNorm) and isinstance(self.norm2, ops.HipLayerNorm) and x.is_contiguous() \
            and not (self.training and torch.is_grad_enabled())     # the in-place residual form is inference only
        if not deferred:
            if pend is not None:
                x = x + pend
            # (training: the four Linears and the attention run their HIP kernels forward AND backward -- ops.linear_any ->
            # training._LinearFn, ops.window_attention -> training._WindowAttentionFn; GELU / LayerNorm / adds are torch operators)
            qkv = ops.linear_any(self.norm1(x), self.attn.qkv)
            a = ops.window_attention(qkv, self.attn.qkv.bias, self.attn.relative_position_bias_table, spec,
                                     precision=self.attention_precision)
            x = x + ops.linear_any(a, self.attn.proj)
            return x + ops.linear_any(ops.linear_any(self.norm2(x), self.mlp.fc1, act=1), self.mlp.fc2), None
        if self.linear_form != "rocblas" and ops.attention_block_linears_supported(self.dim, self.mlp.fc1.out_features):
            # the four Linears as dlwp_linear_f32 (fp32-accurate GEMM on the bf16 pipe) or dlwp_linear_bf16, bias / GELU /
            # residual adds in their epilogues, in place on x
            prec = ops.form_precision(self.linear_form)
            if pend is not None:
                x.add_(pend)
            # all-bf16 form (bf16 Linear operands AND bf16 attention): qkv and the attention output cross HBM as bfloat16 too
            # (dlwp_linear_bf16_io -> dlwp_window_attn_bf16_io -> dlwp_linear_bf16_io) where the fast attention kernels take the call
            io16 = prec == "bf16" and self.attention_precision == "bf16" and ops.window_attention_io_supported(spec, x.shape[0])
            qkv = ops.linear(ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps,
                                            out_dtype=torch.bfloat16 if prec == "bf16" else None), self.attn.qkv, precision=prec,
                             out_dtype=torch.bfloat16 if io16 else None)
            a = ops.window_attention(qkv, self.attn.qkv.bias, self.attn.relative_position_bias_table, spec,
                                     precision=self.attention_precision)
            return ops.attention_block_tail(x, a, self.attn.proj, self.norm2, self.mlp.fc1, self.mlp.fc2, precision=prec), None
        # rocBLAS form: residual adds as GEMM accumulation in place on x, Linear biases deferred into `pend`
        # (ops.residual_block_tail)
        qkv = self.attn.qkv(ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, pre_bias=pend))
        a = ops.window_attention(qkv, self.attn.qkv.bias, self.attn.relative_position_bias_table, spec,
                                 precision=self.attention_precision)
        pend = ops.residual_block_tail(x, pend, a, self.attn.proj, self.norm2, self.mlp.fc1, self.mlp.fc2)
        return x, pend


class _PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)
        self.linear_form = "bf16x6"

    def forward(self, x, h, w):
        b, l, c = x.shape
        if h % 2 or w % 2:
            raise _lib.DlwpError("odd feature map in PatchMerging (the reference pads with an undefined mode, "
                                 "swin_transformer.py:293-296)")
        x = x.view(b, h, w, c)
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(b, -1, 4 * c)
        return ops.linear_as(self.linear_form, self.norm(x), self.reduction)


class _BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, norm_layer, downsample):
        super().__init__()
        ws = (int(window_size[0]), int(window_size[1]))
        self.blocks = nn.ModuleList([
            _Block(dim, num_heads, ws, (0, 0) if i % 2 == 0 else (ws[0] // 2, ws[1] // 2), mlp_ratio, qkv_bias, qk_scale,
                   norm_layer) for i in range(depth)])
        self.downsample = _PatchMerging(dim, norm_layer) if downsample else None

    def forward(self, x, h, w):
        pend = None
        for blk in self.blocks:
            x, pend = blk(x, h, w, pend)
        if pend is not None:
            x.add_(pend)   # the layer's deferred Linear biases, once
        if self.downsample is not None:
            return x, self.downsample(x, h, w), (h + 1) // 2, (w + 1) // 2
        return x, x, h, w


class _PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim, norm_layer):
        super().__init__()
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        _, _, h, w = x.shape
        ph, pw = self.patch_size
        if w % pw:
            x = F.pad(x, (0, pw - w % pw), mode="circular")
        if h % ph:
            x = F.pad(x, (0, 0, 0, ph - h % ph))
        x = self.proj(x)
        wh, ww = x.shape[2], x.shape[3]
        x = x.flatten(2).transpose(1, 2)
        if self.norm is not None:
            x = self.norm(x)
        return x, wh, ww


class SwinTransformer(HipBackbone):
    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 context_size: int = 10, img_height=224, img_width=196, patch_size=4, embed_dim=96,
                 depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer="nn.LayerNorm", ape=False,
                 patch_norm=True, frozen_stages=-1, use_checkpoint=False, mesh="equirectangular", **kwargs):
        super().__init__()
        if mesh not in ("equirectangular", "healpix"):
            raise ValueError(f"unknown mesh {mesh!r}")
        # mesh == "healpix" (SwinTransformerHPX): the reference takes the same code path whenever the map divides by
        # patch and window (swin_transformer.py:220-251), and crashes otherwise (:451 `hpx_pad` does not exist)
        self.mesh = mesh
        if ape:
            raise NotImplementedError("absolute position embedding (ape=True) is not used by any reference config")
        self.context_size = int(context_size)
        depths, num_heads = list(depths), list(num_heads)
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.patch_size = patch_size if isinstance(patch_size, int) else int(patch_size)
        in_chans = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        # configs pass the string "nn.LayerNorm" (swintransformer.yaml:21); same parameters, HIP forward
        norm = ops.HipLayerNorm if (isinstance(norm_layer, str) or norm_layer is nn.LayerNorm) else norm_layer
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim, norm if patch_norm else None)
        res = np.array((img_height // self.patch_size, img_width // self.patch_size))
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(_BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], res, mlp_ratio, qkv_bias,
                                           qk_scale, norm, downsample=i < self.num_layers - 1))
            res = res // 2
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in range(self.num_layers):
            self.add_module(f"norm{i}", norm(self.num_features[i]))
        self.decoder = nn.ModuleList()
        for idx, i_layer in enumerate(range(self.num_layers)[::-1]):
            ch = int(embed_dim * 2 ** i_layer)
            k = self.patch_size if i_layer == 0 else 2
            self.decoder.append(nn.Sequential(
                nn.ConvTranspose2d(ch if idx == 0 else ch * 2, ch if i_layer == 0 else ch // 2, kernel_size=k, stride=k),
                nn.GELU()))
        self.final = nn.Conv2d(embed_dim, prognostic_channels, kernel_size=1)
        self._init_compute_precision(kwargs)     # `compute_precision: bf16` in configs/model/*.yaml (HipBackbone.set_compute_precision)

    def train(self, mode: bool = True):
        super().train(mode)
        return self

    def _token_decoder(self):
        """The decoder (kernel = stride transposed convolutions + GELU, :600-612) and the 1x1 head as Linears over token-major
        data (ops.ConvAsLinear), when every layer qualifies; None otherwise (the modules themselves run, channels-first)."""
        if self.__dict__.get("_tok_dec") is None:
            dec = None
            try:
                dec = ([ops.ConvAsLinear(seq[0]) for seq in self.decoder], ops.ConvAsLinear(self.final))
                if not all(isinstance(seq[1], nn.GELU) and seq[1].approximate == "none" for seq in self.decoder) or \
                        not all(ops.linear_supported(m.in_features, m.out_features) for m in dec[0] + [dec[1]]):
                    dec = None
            except _lib.DlwpError:
                dec = None
            self.__dict__["_tok_dec"] = dec if dec is not None else False
        return self.__dict__["_tok_dec"] or None

    def one_step(self, x: torch.Tensor) -> torch.Tensor:
        """swin_transformer.py:645-677."""
        dec = None if (self.training and torch.is_grad_enabled()) else self._token_decoder()
        form = next((m.linear_form for m in self.modules() if hasattr(m, "linear_form")), "bf16x6")
        if dec is not None and form != "rocblas" and x.is_cuda:
            return self._one_step_tokens(x, dec, ops.form_precision(form))
        x, h, w = self.patch_embed(x)
        outs = []
        for i, layer in enumerate(self.layers):
            x_out, x, hn, wn = layer(x, h, w)
            x_out = getattr(self, f"norm{i}")(x_out)
            outs.append(x_out.view(-1, h, w, self.num_features[i]).permute(0, 3, 1, 2).contiguous())
            h, w = hn, wn
        outs = outs[::-1]
        xo = None
        for idx, layer in enumerate(self.decoder):
            xo = layer(outs[idx] if idx == 0 else torch.cat([outs[idx], xo], dim=1))
        return self.final(xo)

    def _one_step_tokens(self, x, dec, precision):
        """one_step with everything token-major: no channels-first copies, the decoder's convolutions as dlwp_linear_* with
        their GELU in the epilogue, the 1x1-patch embedding as dlwp_patch_embed_1x1_f32."""
        pe = self.patch_embed
        b = x.shape[0]
        if pe.patch_size == (1, 1) and ops.patch_embed_1x1_supported(pe.proj.in_channels, self.embed_dim):
            h, w = x.shape[2], x.shape[3]
            t = ops.patch_embed_1x1(x, pe.proj.weight, pe.proj.bias, None)
            x = pe.norm(t) if pe.norm is not None else t
        else:
            x, h, w = pe(x)
        outs = []
        for i, layer in enumerate(self.layers):
            x_out, x, hn, wn = layer(x, h, w)
            outs.append((getattr(self, f"norm{i}")(x_out), h, w))
            h, w = hn, wn
        outs = outs[::-1]
        xo = None
        for idx, lin in enumerate(dec[0]):
            t, h, w = outs[idx]
            xo = lin(t if idx == 0 else torch.cat([t, xo], dim=-1), h, w, act=1, precision=precision)
            h, w = h * lin.k, w * lin.k
        y = dec[1](xo, h, w, precision="fp32")                         # [B, h*w, Cg]; the output head stays fp32-accurate
        return y.view(b, h, w, -1).permute(0, 3, 1, 2).contiguous()

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


class SwinTransformerHPX(SwinTransformer):
    """reference swin_transformer.py:745-878: Swin on the HEALPix mesh.  The 12 faces [.., 12, h, w] are laid out as
    a 3 x 4 rectangle (north / equatorial / south bands, `_faces2rect` :826-834), the equirectangular backbone runs
    on that [3h, 4w] map (`img_height`, `img_width` are the RECTANGLE's size) and the result is cut back into faces
    (`_reshape_output` :867-878).  The layout change commutes with the residual add of the rollout, so it is done once
    on the inputs and once on the trajectory instead of every step."""

    def __init__(self, *args, mesh="healpix", **kwargs):
        super().__init__(*args, mesh="healpix", **kwargs)

    @staticmethod
    def faces_to_rect(t: torch.Tensor) -> torch.Tensor:
        """[B, T, C, 12, h, w] -> [B, T, C, 3h, 4w]"""
        b, tt, c, f, h, w = t.shape
        return t.reshape(b, tt, c, 3, 4, h, w).permute(0, 1, 2, 3, 5, 4, 6).reshape(b, tt, c, 3 * h, 4 * w)

    @staticmethod
    def rect_to_faces(t: torch.Tensor) -> torch.Tensor:
        """[B, T, C, 3h, 4w] -> [B, T, C, 12, h, w]"""
        b, tt, c, hh, ww = t.shape
        h, w = hh // 3, ww // 4
        return t.reshape(b, tt, c, 3, h, 4, w).permute(0, 1, 2, 3, 5, 4, 6).reshape(b, tt, c, 12, h, w)

    def forward(self, constants: Optional[torch.Tensor] = None, prescribed: Optional[torch.Tensor] = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        for name, t in (("constants", constants), ("prescribed", prescribed), ("prognostic", prognostic)):
            if t is not None and (t.dim() != 6 or t.shape[3] != 12):
                raise _lib.DlwpError(f"{name}: expected [B, T, C, 12, H, W], got {tuple(t.shape)}")
        if prognostic is None:
            raise _lib.DlwpError("prognostic is required")
        rect = lambda t: self.faces_to_rect(t).contiguous() if t is not None else None
        out = super().forward(constants=rect(constants), prescribed=rect(prescribed), prognostic=rect(prognostic))
        return self.rect_to_faces(out).contiguous()
