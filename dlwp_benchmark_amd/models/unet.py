"""UNet (classic, equirectangular) and ConvLSTM -- drop-ins for reference models/unet/unet.py:274-383 +
:429-555 and models/convlstm/convlstm.py:114-251 on MI355X (inference rollout).  Same class names,
constructor kwargs, `nn.Sequential` index layout (so state-dict keys match: `encoder.layers.{l}.{k}`,
`decoder.layers.{l}.{k}`, `decoder.output_layer`; `encoder.{1,4,7}`, `clstm.{i}.conv.1`, `decoder.1`) and
forward signature.

Every `CylinderPad(1) -> Conv2d(3x3) -> activation` triple is ONE HIP kernel (`dlwp_conv3x3_cyl_f32`:
halo staged in LDS with the wrap/zero rule applied at load time, bias + activation in the epilogue); the
skip-connection `torch.cat` (unet.py:553) and the ConvLSTM `cat((x, h_prev))` (convlstm.py:94) are folded
into the kernel's two-segment input; the LSTM gate math (convlstm.py:96-109) is one fused kernel.
AvgPool / ConvTranspose2d(2x2, s2) / the 1x1 head go through torch on the GPU.

Reference defect NOT reproduced: the shipped UNet encoder pads twice (CylinderPad(1) AND padding=1,
unet.py:456-462) and crashes on the first skip concat; this mirror uses the consistent `padding=0`
semantics the decoder (:512-518) and ConvLSTM use -- the same workaround the oracle fixtures were
generated with.
"""
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import lib as _lib
from .. import ops
from ..rollout import rollout_into
from ._base import HipBackbone


class CylinderPad(nn.Module):
    """Placeholder keeping the reference's Sequential indices (utils/utils.py:11-26); the padding
    itself happens inside dlwp_conv3x3_cyl_f32."""

    def __init__(self, padding: int = 1):
        super().__init__()
        self.p = padding


class HEALPixPadding(nn.Module):
    """reference utils/healpix.py:165-368 as one table-driven gather kernel; x [(B*12), C, H, W]."""

    def __init__(self, padding: int = 1, **kwargs):
        super().__init__()
        self.p = int(padding)
        if self.p <= 0:
            raise ValueError("padding must be positive")

    def forward(self, x):
        return ops.healpix_pad(x, self.p)


class HEALPixLayer(nn.Module):
    """reference utils/healpix.py:69-114 for `layer=torch.nn.Conv2d`, kernel_size 3: keeps the reference's
    `layers = Sequential(HEALPixPadding, Conv2d(padding=0))` tree (state-dict key `layers.1.*`); executed as one
    fused kernel by `_run_stack`, or on its own through `forward`."""

    def __init__(self, layer=nn.Conv2d, **kwargs):
        super().__init__()
        kwargs = dict(kwargs)
        kwargs.pop("enable_nhwc", None), kwargs.pop("enable_healpixpad", None)
        if layer is ResidualBlock or layer == "ResidualBlock":
            # not a convolution class: the reference adds no padding layer of its own (healpix.py:86-97), the block
            # pads in front of its two convolutions itself
            self.layers = nn.Sequential(ResidualBlock(**kwargs))
            return
        if layer is not nn.Conv2d or kwargs.get("kernel_size", 3) != 3 or kwargs.get("dilation", 1) != 1:
            raise NotImplementedError("only HEALPixLayer(Conv2d, kernel_size=3, dilation=1) and "
                                      "HEALPixLayer(ResidualBlock) have fused kernels")
        kwargs["padding"] = 0
        self.layers = nn.Sequential(HEALPixPadding(1), nn.Conv2d(**kwargs))

    def forward(self, x, act: int = 0, x1=None):
        if isinstance(self.layers[0], ResidualBlock):
            return self.layers[0](x)
        conv = self.layers[1]
        return ops.conv3x3_hpx(x, conv.weight, conv.bias, act, x1=x1)


class ResidualBlock(nn.Module):
    """reference models/unet/unet.py:839-901 on the HEALPix mesh: pre-activation wide residual block.  Each
    `HEALPixPadding(1) -> Conv2d(3x3, padding 0)` is one dlwp_conv3x3_hpx_f32 launch, the activation in front of the
    second convolution rides in the first one's epilogue when there is no GroupNorm between them."""

    def __init__(self, in_channels: int, out_channels: int, activation=None, norm: bool = False, n_groups: int = 1,
                 kernel_size=3, padding=1, mesh=None):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("only 3x3 residual blocks have a fused kernel")
        activation = _resolve_activation(activation) if activation is not None else nn.GELU()
        if not isinstance(activation, nn.GELU):
            raise NotImplementedError("ResidualBlock: only GELU (the reference default) is wired to the fused kernels")
        self.activation = activation
        self.mesh = mesh
        # unet.py:867-872: HEALPixPadding on the HEALPix mesh, CylinderPad otherwise; either way the padding happens
        # inside the fused convolution kernel
        self.cylinder_pad = HEALPixPadding(padding=1) if mesh == "healpix" else CylinderPad(1)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=0)
        with torch.no_grad():   # zero_module (unet.py:762-766)
            self.conv2.weight.zero_(), self.conv2.bias.zero_()
        self.shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=(1, 1)) if in_channels != out_channels \
            else nn.Identity()
        self.norm1 = nn.GroupNorm(n_groups, in_channels) if norm else nn.Identity()
        self.norm2 = nn.GroupNorm(n_groups, out_channels) if norm else nn.Identity()

    def forward(self, x):
        """unet.py:884-901 in two or four launches: without norms `act(x)` rides in conv1's input staging and the
        second activation in its epilogue; with GroupNorm each `act(norm(.))` is one dlwp_groupnorm_act_f32 launch.
        The shortcut (identity or 1x1 convolution) is added in conv2's epilogue."""
        gelu = ops.act_code(self.activation)
        hpx = self.mesh == "healpix"
        x = x.contiguous()
        short = x if isinstance(self.shortcut, nn.Identity) else ops.conv2d(x, self.shortcut.weight, self.shortcut.bias)
        n1, n2 = self.norm1, self.norm2
        if isinstance(n1, nn.Identity):
            h, pre = x, gelu
        else:
            h, pre = ops.groupnorm_act(x, n1.weight, n1.bias, n1.num_groups, n1.eps, gelu), 0
        if isinstance(n2, nn.Identity):
            h = ops.conv3x3(h, self.conv1.weight, self.conv1.bias, act=gelu, pre_act=pre, hpx=hpx)
        else:
            h = ops.conv3x3(h, self.conv1.weight, self.conv1.bias, act=0, pre_act=pre, hpx=hpx)
            h = ops.groupnorm_act(h, n2.weight, n2.bias, n2.num_groups, n2.eps, gelu)
        return ops.conv3x3(h, self.conv2.weight, self.conv2.bias, act=0, resid=short, hpx=hpx)


class MiddleBlock(nn.Module):
    """unet.py:904-946 (attention is an Identity in the reference)."""

    def __init__(self, in_channels: int, attention: bool = False, activation=None, norm: bool = False, mesh=None):
        super().__init__()
        self.res1 = ResidualBlock(in_channels, in_channels, activation=activation, norm=norm, mesh=mesh)
        self.attn = nn.Identity()
        self.res2 = ResidualBlock(in_channels, in_channels, activation=activation, norm=norm, mesh=mesh)

    def forward(self, x):
        return self.res2(self.res1(x))


def _resolve_activation(activation):
    if isinstance(activation, str):
        name = activation
        for key, mod in (("GELU", nn.GELU), ("Tanh", nn.Tanh), ("LeakyReLU", None), ("ReLU", nn.ReLU), ("SiLU", nn.SiLU)):
            if key in name:
                if mod is None:
                    break
                return mod()
        raise _lib.DlwpError(f"activation {activation!r} has no fused kernel (supported: GELU, Tanh, ReLU, SiLU)")
    return activation


_STRUCTURAL = (CylinderPad, HEALPixLayer, nn.Conv2d, nn.ConvTranspose2d, nn.AvgPool2d)


def _run_stack(seq: nn.Sequential, x, skip=None):
    """Executes a reference-shaped Sequential of [AvgPool] (CylinderPad, Conv2d, act)* [ConvTranspose2d]
    with each (pad, conv, act) triple fused into one kernel call; `skip` is concatenated in front
    of x for the first conv (torch.cat([skip, x], 1), unet.py:553)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, HEALPixLayer):
            act, step = 0, 1
            if i + 1 < len(mods) and not isinstance(mods[i + 1], _STRUCTURAL):
                act, step = ops.act_code(mods[i + 1]), 2
            x = m(skip, act, x1=x) if skip is not None else m(x, act)
            skip = None
            i += step
        elif isinstance(m, CylinderPad):
            conv = mods[i + 1]
            act = 0
            step = 2
            if i + 2 < len(mods) and not isinstance(mods[i + 2], _STRUCTURAL):
                act = ops.act_code(mods[i + 2])
                step = 3
            if skip is not None:
                x = ops.conv3x3_cyl(skip, conv.weight, conv.bias, act, x1=x)
                skip = None
            else:
                x = ops.conv3x3_cyl(x, conv.weight, conv.bias, act)
            i += step
        else:
            if skip is not None:
                x = torch.cat([skip, x], dim=1)
                skip = None
            x = ops.small_module(m, x)      # AvgPool2d(2) / ConvTranspose2d / plain Conv2d: HIP kernels, no torch op
            i += 1
    return x


def _conv_block(c_in, c_out, activation, mesh):
    if mesh == "healpix":
        return [HEALPixLayer(layer=nn.Conv2d, in_channels=c_in, out_channels=c_out, kernel_size=3, padding=1), activation]
    return [CylinderPad(1), nn.Conv2d(c_in, c_out, kernel_size=3, padding=0), activation]


class _UNetEncoder(nn.Module):
    def __init__(self, in_channels, hidden_channels, n_convolutions, activation, mesh="equirectangular"):
        super().__init__()
        layers = []
        channels = [in_channels] + list(hidden_channels)
        for c_idx in range(len(channels) - 1):
            layer = []
            c_in, c_out = channels[c_idx], channels[c_idx + 1]
            if c_idx > 0:
                layer.append(nn.AvgPool2d(kernel_size=2, stride=2, padding=0))
            n_convs = n_convolutions // 2 if c_idx == len(hidden_channels) - 1 else n_convolutions
            for n_conv in range(n_convs):
                layer += _conv_block(c_in if n_conv == 0 else c_out, c_out, activation, mesh)
            layers.append(nn.Sequential(*layer))
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        outs = []
        for layer in self.layers:
            x = _run_stack(layer, x)
            outs.append(x)
        return outs


class _UNetDecoder(nn.Module):
    def __init__(self, hidden_channels, out_channels, n_convolutions, activation, mesh="equirectangular"):
        super().__init__()
        hidden = list(hidden_channels)[::-1]
        layers = []
        c_out = hidden[0]
        for c_idx in range(len(hidden)):
            layer = []
            c_in = c_out = hidden[c_idx]
            n_convs = n_convolutions // 2 if c_idx == 0 else n_convolutions
            for n_conv in range(n_convs):
                c_in_ = c_in if c_idx == 0 else 2 * hidden[c_idx]
                layer += _conv_block(c_in_ if n_conv == 0 else c_out, c_out, activation, mesh)
            if c_idx < len(hidden) - 1:
                layer.append(nn.ConvTranspose2d(c_out, hidden[c_idx + 1], kernel_size=2, stride=2))
            layers.append(nn.Sequential(*layer))
        self.layers = nn.ModuleList(layers)
        self.output_layer = nn.Conv2d(c_out, out_channels, kernel_size=1)

    def forward(self, x, skips):
        for l_idx, layer in enumerate(self.layers):
            x = _run_stack(layer, x, skip=skips[l_idx] if l_idx > 0 else None)
        return ops.small_module(self.output_layer, x)


class UNet(HipBackbone):
    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 hidden_channels: list = [8, 16, 32], n_convolutions: int = 2, activation=nn.GELU(),
                 context_size: int = 1, mesh: str = "equirectangular", **kwargs):
        super().__init__()
        if mesh not in ("equirectangular", "healpix"):
            raise ValueError(f"unknown mesh {mesh!r}")
        activation = _resolve_activation(activation)
        ops.act_code(activation)  # fail early if there is no fused kernel for it
        self.context_size = int(context_size)
        in_channels = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        self.encoder = _UNetEncoder(in_channels, list(hidden_channels), n_convolutions, activation, mesh)
        self.decoder = _UNetDecoder(list(hidden_channels), prognostic_channels, n_convolutions, activation, mesh)

    def one_step(self, x):
        enc = self.encoder(x)
        return self.decoder(x=enc[-1], skips=enc[::-1])

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


class UNetHPX(UNet):
    """reference models/unet/unet.py:386-426: the classic U-Net on the HEALPix mesh.  Tensors carry a face
    axis, constants [B, 1, C, 12, H, W], prescribed / prognostic [B, T, C, 12, H, W]; faces fold into the batch
    for the backbone (:413-426) and every HEALPixLayer(Conv2d) + activation is one dlwp_conv3x3_hpx_f32 launch."""

    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 hidden_channels: list = [8, 16, 32], n_convolutions: int = 2, activation=nn.GELU(),
                 context_size: int = 1, mesh: str = "healpix", **kwargs):
        super().__init__(constant_channels=constant_channels, prescribed_channels=prescribed_channels,
                         prognostic_channels=prognostic_channels, hidden_channels=hidden_channels,
                         n_convolutions=n_convolutions, activation=activation, context_size=context_size, mesh="healpix")

    @staticmethod
    def _fold(t):
        """[B, T, C, F, H, W] -> [(B F), (T C), H, W]"""
        b, tt, c, f, h, w = t.shape
        return t.permute(0, 3, 1, 2, 4, 5).reshape(b * f, tt * c, h, w)

    def forward(self, constants: Optional[torch.Tensor] = None, prescribed: Optional[torch.Tensor] = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        for name, t in (("constants", constants), ("prescribed", prescribed), ("prognostic", prognostic)):
            if t is not None:
                _lib.require_cuda_tensor(t, name)
                if t.dim() != 6 or t.shape[3] != 12:
                    raise _lib.DlwpError(f"{name}: expected [B, T, C, 12, H, W], got {tuple(t.shape)}")
        if prognostic is None:
            raise _lib.DlwpError("prognostic is required")
        ctx = self.context_size
        b, t_total, cg, f, h, w = prognostic.shape
        if t_total <= ctx:
            raise _lib.DlwpError(f"need more than context_size={ctx} frames, got {t_total}")
        if self._grad_mode():
            from ..rollout import rollout_train

            fold = lambda t: t.permute(0, 3, 1, 2, 4, 5).reshape(b * f, t.shape[1], t.shape[2], h, w).float()
            out = rollout_train(self.one_step, ctx, fold(constants) if constants is not None else None,
                                fold(prescribed) if prescribed is not None else None, fold(prognostic))
            return out.reshape(b, f, t_total - ctx, cg, h, w).permute(0, 2, 3, 1, 4, 5)
        with torch.no_grad():
            # face-folded working layout [(B F), T, C, H, W]: the generic rollout then runs unchanged
            fold5 = lambda t: t.permute(0, 3, 1, 2, 4, 5).reshape(b * f, t.shape[1], t.shape[2], h, w).float().contiguous()
            out = torch.empty(b * f, t_total - ctx, cg, h, w, device=prognostic.device, dtype=torch.float32)
            rollout_into(self._step_fn(), ctx, out, fold5(constants) if constants is not None else None,
                         fold5(prescribed) if prescribed is not None else None, fold5(prognostic))
            return out.reshape(b, f, t_total - ctx, cg, h, w).permute(0, 2, 3, 1, 4, 5).contiguous()


def _res_layer(c_in, c_out, mesh):
    """unet.py:588-610 / :663-683: the residual block of an encoder / decoder level, wrapped in a HEALPixLayer on the
    HEALPix mesh (state-dict key `...layers.0.conv1`), bare otherwise (`...conv1`)."""
    if mesh == "healpix":
        return HEALPixLayer(layer=ResidualBlock, in_channels=c_in, out_channels=c_out, kernel_size=3, padding=0, mesh="healpix")
    return ResidualBlock(in_channels=c_in, out_channels=c_out, kernel_size=3, padding=1)


class _ModernUNetEncoder(nn.Module):
    """unet.py:559-632."""

    def __init__(self, in_channels, hidden_channels, mesh="healpix"):
        super().__init__()
        self.attn = nn.Identity()
        channels = [in_channels] + list(hidden_channels)
        layers = []
        for c_idx in range(len(channels) - 1):
            c_in, c_out = channels[c_idx], channels[c_idx + 1]
            first = nn.Conv2d(c_in, c_in, (3, 3), (2, 2), (1, 1)) if c_idx > 0 else nn.Conv2d(c_in, c_in, (1, 1), (1, 1), (0, 0))
            layers.append(nn.Sequential(first, _res_layer(c_in, c_out, mesh), self.attn))
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        for layer in self.layers:
            x = layer[1](ops.small_module(layer[0], x))    # strided 3x3 / 1x1 convolution, then the residual block
        return x


class _ModernUNetDecoder(nn.Module):
    """unet.py:634-760, healpix branch, as it RUNS: the skip concatenation of :747-751 tests for ResidualBlock
    instances, the sub-modules are HEALPixLayer wrappers, so no skip is ever concatenated (the channel counts of
    the constructor are consistent with exactly that)."""

    def __init__(self, hidden_channels, out_channels, activation, mesh="healpix"):
        super().__init__()
        final_out = 2 * hidden_channels[0]
        hidden = list(hidden_channels)[::-1]
        self.attn = nn.Identity()
        self.activation = activation
        layers = []
        c_out2 = None
        for c_idx in range(len(hidden)):
            c_out = hidden[c_idx]
            c_in_ = c_out if c_idx == 0 else 2 * hidden[c_idx]
            c_out2 = 2 * hidden[c_idx + 1] if c_idx + 1 < len(hidden) else 2 * hidden[c_idx]
            layer = [_res_layer(c_in_, c_out, mesh), self.attn, _res_layer(c_out, c_out2, mesh)]
            if c_idx < len(hidden) - 1:
                layer.append(nn.ConvTranspose2d(c_out2, c_out2, (4, 4), (2, 2), (1, 1)))
            layers.append(nn.Sequential(*layer))
        self.layers = nn.ModuleList(layers)
        self.output_layer = nn.Conv2d(c_out2, out_channels, kernel_size=1)
        with torch.no_grad():   # zero_module
            self.output_layer.weight.zero_(), self.output_layer.bias.zero_()
        self.final_norm = nn.GroupNorm(8, final_out)

    def forward(self, x):
        for layer in self.layers:
            for sub in layer:
                x = ops.small_module(sub, x) if isinstance(sub, nn.ConvTranspose2d) else sub(x)
        fn = self.final_norm
        x = ops.groupnorm_act(x, fn.weight, fn.bias, fn.num_groups, fn.eps, ops.act_code(self.activation))
        return ops.small_module(self.output_layer, x)


class MUNetHPX(UNetHPX):
    """reference models/unet/unet.py:205-269 (`MUNetHPX`, the PDE-Refiner style ModernUNet :72-203 on the HEALPix
    mesh).  Same constructor kwargs, module tree / state-dict keys and
    forward(constants, prescribed, prognostic) with face-carrying tensors [B, T, C, 12, H, W].  `recurrent=True`
    (a hard-coded cuda:0 ConvLSTM cell, :689-703) and attention (an Identity in the reference) are not built."""

    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 hidden_channels: list = [64, 128, 256, 1024], activation=nn.GELU(), context_size: int = 1,
                 mesh: str = "healpix", attention: bool = False, norm: bool = False, recurrent: bool = False, **kwargs):
        HipBackbone.__init__(self)
        if recurrent:
            raise NotImplementedError("MUNetHPX(recurrent=True) is not built")
        activation = _resolve_activation(activation)
        if not isinstance(activation, nn.GELU):
            raise NotImplementedError("MUNetHPX: only GELU is wired to the fused kernels")
        self.context_size = int(context_size)
        self.mesh = "healpix"
        in_channels = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        self.encoder = _ModernUNetEncoder(in_channels, list(hidden_channels))
        self.middle = MiddleBlock(in_channels=hidden_channels[-1], norm=norm, activation=activation, mesh="healpix")
        self.decoder = _ModernUNetDecoder(list(hidden_channels), prognostic_channels, activation)

    def one_step(self, x):
        return self.decoder(self.middle(self.encoder(x)))


class ModernUNet(UNet):
    """Registry name `ModernUNet` (reference models/unet/unet.py:72-203; the `type` of seven configs/model/modernunet*.yaml).
    The reference class CANNOT be constructed with mesh="equirectangular" -- the only mesh those configs give it --
    because its decoder defines `c_out2` inside the healpix branch only (unet.py:705-719: NameError), and with the
    skip concatenation of :747-751 active its channel counts would not line up either.  There is therefore NO
    reference behaviour to be faithful to on the lat-lon grid (PARITY UNPINNED by necessity).
    What this class does: mesh="healpix" is MUNetHPX (pinned by fixtures of the real class); mesh="equirectangular"
    builds the same network as it RUNS on the HEALPix mesh (no skip concatenation, :747-751) with the lat-lon
    residual block of :867-872 (CylinderPad instead of HEALPixPadding; strided / transposed convolutions zero-padded
    as constructed at :583, :719), so the reference's model configs construct and roll out through the HIP kernels."""

    def __new__(cls, *args, **kwargs):
        if cls is ModernUNet and kwargs.get("mesh", "equirectangular") == "healpix":
            return MUNetHPX(*args, **kwargs)
        return super().__new__(cls)

    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 hidden_channels: list = [64, 128, 256, 1024], activation=nn.GELU(), context_size: int = 1,
                 mesh: str = "equirectangular", attention: bool = False, norm: bool = False, recurrent: bool = False,
                 **kwargs):
        HipBackbone.__init__(self)
        if mesh != "equirectangular":
            raise ValueError(f"unknown mesh {mesh!r}")
        if recurrent:
            raise NotImplementedError("ModernUNet(recurrent=True) is not built")
        activation = _resolve_activation(activation)
        if not isinstance(activation, nn.GELU):
            raise NotImplementedError("ModernUNet: only GELU is wired to the fused kernels")
        self.context_size = int(context_size)
        self.mesh = mesh
        hidden_channels = [int(c) for c in hidden_channels]
        if (2 * hidden_channels[0]) % 8:
            raise ValueError("final GroupNorm(8, 2 * hidden_channels[0]) needs hidden_channels[0] % 4 == 0 (unet.py:739)")
        in_channels = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        self.encoder = _ModernUNetEncoder(in_channels, hidden_channels, mesh=None)
        self.middle = MiddleBlock(in_channels=hidden_channels[-1], norm=norm, activation=activation, mesh=None)
        self.decoder = _ModernUNetDecoder(hidden_channels, prognostic_channels, activation, mesh=None)

    def one_step(self, x):
        return self.decoder(self.middle(self.encoder(x)))


class _ConvLSTMCell(nn.Module):
    def __init__(self, input_size, hidden_size, bias=True, mesh="equirectangular"):
        super().__init__()
        self.hidden_size = hidden_size
        if mesh == "healpix":   # convlstm.py:56-63: state-dict key conv.layers.1.*
            self.conv = HEALPixLayer(layer=nn.Conv2d, in_channels=input_size + hidden_size, out_channels=hidden_size * 4,
                                     kernel_size=3, padding=1, bias=bias)
        else:
            self.conv = nn.Sequential(CylinderPad(1), nn.Conv2d(input_size + hidden_size, hidden_size * 4, kernel_size=3,
                                                                stride=1, padding=0, bias=bias))

    def gates(self, x, h_prev):
        """conv3x3(cat(x, h_prev)) without the cat (convlstm.py:94)."""
        if isinstance(self.conv, HEALPixLayer):
            return self.conv(x, 0, x1=h_prev)
        conv = self.conv[1]
        return ops.conv3x3_cyl(x, conv.weight, conv.bias, 0, x1=h_prev)


class ConvLSTM(HipBackbone):
    """reference models/convlstm/convlstm.py:114-251.  The loop runs from t = 0 (teacher forcing while
    t < context_size) because the LSTM state is carried from the first frame on; there is therefore no ranged
    `rollout_into` -- a sharded evaluation collects its trajectory after the whole rollout (sharding.py)."""

    def __init__(self, batch_size: int = 16, constant_channels: int = 4, prescribed_channels: int = 0,
                 prognostic_channels: int = 1, hidden_sizes: list = [16, 16], height: int = 32, width: int = 64,
                 device=None, bias: bool = True, context_size: int = 1, mesh: str = "equirectangular", **kwargs):
        super().__init__()
        if mesh not in ("equirectangular", "healpix"):
            raise ValueError(f"unknown mesh {mesh!r}")
        self.mesh = mesh
        self.hidden_sizes = list(hidden_sizes)
        self.context_size = int(context_size)
        in_size = constant_channels + prescribed_channels + prognostic_channels
        h0 = self.hidden_sizes[0]
        # registration order encoder, clstm, decoder as in the reference (:148-193): state_dict() iterates in that order
        hpx = lambda ci, co: HEALPixLayer(layer=nn.Conv2d, in_channels=ci, out_channels=co, kernel_size=3, padding=1)
        if mesh == "healpix":   # convlstm.py:155-161
            self.encoder = nn.Sequential(hpx(in_size, h0), nn.Tanh(), hpx(h0, h0), nn.Tanh(), hpx(h0, h0))
        else:
            self.encoder = nn.Sequential(
                CylinderPad(1), nn.Conv2d(in_size, h0, kernel_size=3, padding=0), nn.Tanh(),
                CylinderPad(1), nn.Conv2d(h0, h0, kernel_size=3, padding=0), nn.Tanh(),
                CylinderPad(1), nn.Conv2d(h0, h0, kernel_size=3, padding=0))
        self.clstm = nn.Sequential(*[_ConvLSTMCell(hs, hs, bias, mesh) for hs in self.hidden_sizes])
        if mesh == "healpix":   # convlstm.py:186-193
            self.decoder = hpx(self.hidden_sizes[-1], prognostic_channels)
        else:
            self.decoder = nn.Sequential(CylinderPad(1), nn.Conv2d(self.hidden_sizes[-1], prognostic_channels,
                                                                   kernel_size=3, padding=0))

    def _decode(self, x):
        return self.decoder(x) if isinstance(self.decoder, HEALPixLayer) else _run_stack(self.decoder, x)

    def _rollout(self, constants, prescribed, prognostic):
        """convlstm.py:210-251 on [N, T, C, H, W] tensors (N = batch, or batch * 12 faces): loop from t = 0, teacher
        forcing while t < context_size, (h, c) carried across steps, outputs[context_size:] returned.  The states
        live in local tensors instead of module attributes (:108-109), so the module is re-entrant."""
        b, t_total, cg, hgt, wid = prognostic.shape
        ctx = self.context_size
        if t_total <= ctx:
            raise _lib.DlwpError(f"need more than context_size={ctx} frames, got {t_total}")
        dev = prognostic.device
        hs = [torch.zeros(b, n, hgt, wid, device=dev) for n in self.hidden_sizes]
        cs = [torch.zeros(b, n, hgt, wid, device=dev) for n in self.hidden_sizes]
        grad = self._grad_mode()       # training: nothing in place, the trajectory is stacked at the end
        outs = []
        out = None if grad else torch.empty(b, t_total - ctx, cg, hgt, wid, device=dev, dtype=torch.float32)
        prev = None
        for t in range(t_total):
            prog_t = prognostic[:, t] if t < ctx else prev
            parts = []
            if constants is not None:
                parts.append(constants[:, 0])
            if prescribed is not None:
                parts.append(prescribed[:, t])
            parts.append(prog_t)
            x = _run_stack(self.encoder, torch.cat(parts, dim=1))
            for i, cell in enumerate(self.clstm):
                hs[i], cs[i] = ops.convlstm_gates(cell.gates(x, hs[i]), cs[i])
                x = hs[i]
            inc = self._decode(x)
            if grad:
                prev = prog_t + inc
                if t >= ctx:
                    outs.append(prev)
            elif t >= ctx:
                torch.add(prog_t, inc, out=out[:, t - ctx])
                prev = out[:, t - ctx]
            else:
                prev = prog_t + inc
        return torch.stack(outs, dim=1) if grad else out

    def forward(self, constants: Optional[torch.Tensor] = None, prescribed: Optional[torch.Tensor] = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        constants, prescribed, prognostic = self._check_inputs(constants, prescribed, prognostic)
        if self._grad_mode():
            return self._rollout(constants, prescribed, prognostic)
        with torch.no_grad():
            return self._rollout(constants, prescribed, prognostic)


class ConvLSTMHPX(ConvLSTM):
    """reference models/convlstm/convlstm.py:258-305: ConvLSTM on the HEALPix mesh.  Tensors carry a face axis
    (constants [B, 1, C, 12, H, W], prescribed / prognostic [B, T, C, 12, H, W]); faces fold into the batch
    (`_prepare_inputs` :293-305), every HEALPixLayer(Conv2d) is one dlwp_conv3x3_hpx_f32 launch with the Tanh in
    its epilogue and `cat(x, h)` folded into its two-segment input."""

    def __init__(self, batch_size: int = 16, constant_channels: int = 4, prescribed_channels: int = 0,
                 prognostic_channels: int = 1, hidden_sizes: list = [16, 16], height: int = 32, width: int = 64,
                 device=None, bias: bool = True, context_size: int = 1, mesh: str = "healpix", **kwargs):
        super().__init__(batch_size=batch_size, constant_channels=constant_channels,
                         prescribed_channels=prescribed_channels, prognostic_channels=prognostic_channels,
                         hidden_sizes=hidden_sizes, height=height, width=width, device=device, bias=bias,
                         context_size=context_size, mesh="healpix")

    def forward(self, constants: Optional[torch.Tensor] = None, prescribed: Optional[torch.Tensor] = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        for name, t in (("constants", constants), ("prescribed", prescribed), ("prognostic", prognostic)):
            if t is not None:
                _lib.require_cuda_tensor(t, name)
                if t.dim() != 6 or t.shape[3] != 12:
                    raise _lib.DlwpError(f"{name}: expected [B, T, C, 12, H, W], got {tuple(t.shape)}")
        if prognostic is None:
            raise _lib.DlwpError("prognostic is required")
        b, t_total, cg, f, h, w = prognostic.shape
        with torch.set_grad_enabled(self._grad_mode()):
            fold5 = lambda t: t.permute(0, 3, 1, 2, 4, 5).reshape(b * f, t.shape[1], t.shape[2], h, w).float().contiguous()
            out = self._rollout(fold5(constants) if constants is not None else None,
                                fold5(prescribed) if prescribed is not None else None, fold5(prognostic))
            return out.reshape(b, f, out.shape[1], cg, h, w).permute(0, 2, 3, 1, 4, 5).contiguous()
