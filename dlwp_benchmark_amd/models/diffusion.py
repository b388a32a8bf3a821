"""PDE-Refiner style diffusion backbones -- drop-in for reference
models/diffusion_models/modern_unet/modern_unet.py (`DiffModernUNet` :46-293, `DiffMUNetHPX` :295-327, registered in the
reference registry, models/__init__.py:15; configs/model/diffusion_modernunet*.yaml): a time-conditioned ModernUNet evaluated
`num_refinement_step` times per forecast step inside the autoregressive rollout ("the same rollout
wrapped in denoising iterations").  Same constructor kwargs, module tree / state-dict names and
`forward(constants, prescribed, prognostic, noise_scheduler, target)` signature.

The network runs on the hand-written kernels of the U-Net family: every `pad -> Conv2d(3x3)` is one dlwp_conv3x3_ex_f32 launch
(CylinderPad or HEALPixPadding inside the kernel, the activation in front of it applied while its input is staged, the shortcut
added in the epilogue), GroupNorm(+GELU) is dlwp_groupnorm_act_f32, the stride-2 / transposed / 1x1 convolutions are
dlwp_conv2d_f32 / dlwp_conv_transpose2d_f32.  What stays in torch: the sinusoidal time embedding and its two tiny Linears
([B, 64] operands), the AdaGN affine `h * (1 + scale) + shift` (one elementwise pass) and the noise scheduler itself, which the
caller supplies (the reference scripts pass diffusers' DDPMScheduler, scripts/evaluate.py:186-202; any object with `.timesteps`
and `.step(pred, k, sample).prev_sample` works).  The reference draws the start noise with `th.randn(shape)` on the host and
moves it to the device (:186); so does this mirror, so a seeded run sees the same noise.

Not mirrored: the optional `AttentionBlock` (`attention=True`; both reference configs run without it and the encoder / decoder
never receive the flag's blocks in a usable form) -- constructing with attention=True raises."""
import math
from typing import Optional

import torch
from torch import nn

from .. import lib as _lib
from .. import ops
from .unet import CylinderPad, HEALPixPadding, _resolve_activation


def fourier_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """modern_unet.py:10-31 (sinusoidal timestep embedding, cos | sin halves)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _zero_module(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


class ResidualBlock(nn.Module):
    """modern_unet.py:560-650: conditioned wide residual block.  `use_scale_shift_norm` (AdaGN): the embedding gives a
    per-(sample, channel) scale and shift applied between the two convolutions; otherwise it is added as a bias."""

    def __init__(self, in_channels: int, out_channels: int, cond_channels: int, activation=None, norm: bool = False,
                 n_groups: int = 4, use_scale_shift_norm: bool = True, kernel_size=3, padding=1, mesh=None):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("only 3x3 residual blocks have a fused kernel")
        activation = _resolve_activation(activation) if activation is not None else nn.GELU()
        if not isinstance(activation, nn.GELU):
            raise NotImplementedError("ResidualBlock: only GELU (the reference default) is wired to the fused kernels")
        self.activation = activation
        self.use_scale_shift_norm = use_scale_shift_norm
        self.mesh = mesh
        self.cylinder_pad = HEALPixPadding(padding=1) if mesh == "healpix" else CylinderPad(padding)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=0)
        self.conv2 = _zero_module(nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=0))
        self.shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=(1, 1)) if in_channels != out_channels else nn.Identity()
        self.norm1 = nn.GroupNorm(n_groups, in_channels) if norm else nn.Identity()
        self.norm2 = nn.GroupNorm(n_groups, out_channels) if norm else nn.Identity()
        self.cond_emb = nn.Linear(cond_channels, 2 * out_channels if use_scale_shift_norm else out_channels)

    def forward(self, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
        """modern_unet.py:620-650.  `emb` [B', cond]: B' = the leading dimension of x (on the HEALPix mesh the caller has
        already repeated it per face -- the reference cannot broadcast [B, C] against [(B 12), C, h, w] either)."""
        gelu = ops.act_code(self.activation)
        hpx = self.mesh == "healpix"
        x = x.contiguous()
        short = x if isinstance(self.shortcut, nn.Identity) else ops.conv2d(x, self.shortcut.weight, self.shortcut.bias)
        n1, n2 = self.norm1, self.norm2
        if isinstance(n1, nn.Identity):
            h = ops.conv3x3(x, self.conv1.weight, self.conv1.bias, pre_act=gelu, hpx=hpx)
        else:
            h = ops.groupnorm_act(x, n1.weight, n1.bias, n1.num_groups, n1.eps, gelu)
            h = ops.conv3x3(h, self.conv1.weight, self.conv1.bias, hpx=hpx)
        emb_out = self.cond_emb(emb)[:, :, None, None]
        if not isinstance(n2, nn.Identity) and self.use_scale_shift_norm:
            h = ops.groupnorm_act(h, n2.weight, n2.bias, n2.num_groups, n2.eps, 0)
        if self.use_scale_shift_norm:
            scale, shift = torch.chunk(emb_out, 2, dim=1)
            h = torch.addcmul(shift, h, 1 + scale)                     # norm2(h) * (1 + scale) + shift
            return ops.conv3x3(h, self.conv2.weight, self.conv2.bias, pre_act=gelu, resid=short, hpx=hpx)
        h = h + emb_out
        if isinstance(n2, nn.Identity):
            return ops.conv3x3(h, self.conv2.weight, self.conv2.bias, pre_act=gelu, resid=short, hpx=hpx)
        h = ops.groupnorm_act(h, n2.weight, n2.bias, n2.num_groups, n2.eps, gelu)
        return ops.conv3x3(h, self.conv2.weight, self.conv2.bias, resid=short, hpx=hpx)


class ConditionalHEALPixLayer(nn.Module):
    """utils/healpix.py:117-162 around a ResidualBlock: `layers.0` is the block (it pads inside its kernels)."""

    def __init__(self, layer=ResidualBlock, **kwargs):
        super().__init__()
        if layer is not ResidualBlock:
            raise NotImplementedError("ConditionalHEALPixLayer: only ResidualBlock is used by the reference networks")
        self.layers = nn.Sequential(layer(**kwargs))

    def forward(self, x, emb=None):
        return self.layers[0](x, emb)


class MiddleBlock(nn.Module):
    """modern_unet.py:653-706 (the attention slot is an Identity)."""

    def __init__(self, in_channels: int, time_embed_dim: int, attention: bool = False, activation=None, norm: bool = False,
                 use_scale_shift_norm: bool = True, mesh=None):
        super().__init__()
        if attention:
            raise NotImplementedError("MiddleBlock: attention=True is not built (unused by the reference configs)")
        kw = dict(cond_channels=time_embed_dim, activation=activation, norm=norm, use_scale_shift_norm=use_scale_shift_norm, mesh=mesh)
        self.res1 = ResidualBlock(in_channels, in_channels, **kw)
        self.attn = nn.Identity()
        self.res2 = ResidualBlock(in_channels, in_channels, **kw)

    def forward(self, x, emb):
        return self.res2(self.res1(x, emb), emb)


def _cond_layer(c_in, c_out, time_embed_dim, mesh, use_scale_shift_norm):
    if mesh == "healpix":    # (the reference does not forward use_scale_shift_norm here: the block's default, True, applies)
        return ConditionalHEALPixLayer(layer=ResidualBlock, in_channels=c_in, out_channels=c_out, cond_channels=time_embed_dim,
                                       kernel_size=3, padding=1, mesh=mesh)
    return ResidualBlock(in_channels=c_in, out_channels=c_out, cond_channels=time_embed_dim, kernel_size=3, padding=1,
                         use_scale_shift_norm=use_scale_shift_norm)


class ModernUNetEncoder(nn.Module):
    """modern_unet.py:325-406: per level [Conv2d(3x3, stride 2) below the top], conditioned ResidualBlock, Identity."""

    def __init__(self, in_channels=2, hidden_channels=(64, 128, 256, 1024), time_embed_dim=1024, activation=None,
                 attention: bool = False, mesh: str = "equirectangular", use_scale_shift_norm=True):
        super().__init__()
        if attention:
            raise NotImplementedError("attention=True is not built (unused by the reference configs)")
        channels = [in_channels] + list(hidden_channels)
        layers = []
        for i in range(len(channels) - 1):
            layer = []
            if i > 0:
                layer.append(nn.Conv2d(channels[i], channels[i], (3, 3), (2, 2), (1, 1)))
            layer.append(_cond_layer(channels[i], channels[i + 1], time_embed_dim, mesh, use_scale_shift_norm))
            layer.append(nn.Identity())
            layers.append(nn.Sequential(*layer))
        self.attn = nn.Identity()
        self.layers = nn.ModuleList(layers)

    def forward(self, x, emb):
        outs = []
        for layer in self.layers:
            for m in layer:
                if isinstance(m, nn.Conv2d):
                    x = ops.small_module(m, x)
                elif not isinstance(m, nn.Identity):
                    x = m(x, emb)
            outs.append(x)
        return outs


class ModernUNetDecoder(nn.Module):
    """modern_unet.py:408-497: per level conditioned ResidualBlock (on cat([skip, x]) below the bottom), Identity,
    [ConvTranspose2d(4, 2, 1) above the top]; then GroupNorm(4) -> activation -> 1x1 output convolution."""

    def __init__(self, hidden_channels=(64, 128, 256, 1024), out_channels=2, time_embed_dim=1024, activation=None,
                 attention: bool = False, mesh: str = "equirectangular", use_scale_shift_norm=True):
        super().__init__()
        if attention:
            raise NotImplementedError("attention=True is not built (unused by the reference configs)")
        hidden = list(hidden_channels)[::-1]
        self.activation = _resolve_activation(activation) if activation is not None else nn.GELU()
        layers = []
        for i, c in enumerate(hidden):
            layer = [_cond_layer(c if i == 0 else 2 * c, c, time_embed_dim, mesh, use_scale_shift_norm), nn.Identity()]
            if i < len(hidden) - 1:
                layer.append(nn.ConvTranspose2d(c, hidden[i + 1], (4, 4), (2, 2), (1, 1)))
            layers.append(nn.Sequential(*layer))
        self.attn = nn.Identity()
        self.layers = nn.ModuleList(layers)
        self.output_layer = _zero_module(nn.Conv2d(hidden[-1], out_channels, kernel_size=1))
        self.final_norm = nn.GroupNorm(4, hidden[-1])

    def forward(self, x, skips, emb):
        for i, layer in enumerate(self.layers):
            if i > 0:
                x = torch.cat([skips[i], x], dim=1)
            for m in layer:
                if isinstance(m, nn.ConvTranspose2d):
                    x = ops.small_module(m, x)
                elif not isinstance(m, nn.Identity):
                    x = m(x, emb)
        fn = self.final_norm
        x = ops.groupnorm_act(x, fn.weight, fn.bias, fn.num_groups, fn.eps, ops.act_code(self.activation))
        return ops.small_module(self.output_layer, x)


class DiffModernUNet(nn.Module):
    """modern_unet.py:46-293."""

    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 hidden_channels=(64, 128, 256, 1024), activation=None, context_size: int = 1, mesh: str = "equirectangular",
                 attention: bool = False, norm: bool = False, use_scale_shift_norm=True, predict_diff=True,
                 num_refinement_step=5, **kwargs):
        super().__init__()
        if attention:
            raise NotImplementedError("attention=True is not built (unused by the reference configs)")
        activation = _resolve_activation(activation) if activation is not None else nn.GELU()
        self.context_size = context_size
        self.mesh = mesh
        self.hidden_channels = list(hidden_channels)
        time_embed_dim = self.hidden_channels[0] * 4
        self.activation = activation
        self.predict_diff = predict_diff
        self.num_refinement_step = num_refinement_step
        self.time_embed = nn.Sequential(nn.Linear(self.hidden_channels[0], time_embed_dim), self.activation,
                                        nn.Linear(time_embed_dim, time_embed_dim))
        in_channels = constant_channels + (prescribed_channels + prognostic_channels) * context_size + prognostic_channels * context_size
        self.encoder = ModernUNetEncoder(in_channels=in_channels, hidden_channels=self.hidden_channels,
                                         time_embed_dim=time_embed_dim, activation=activation, attention=attention, mesh=mesh,
                                         use_scale_shift_norm=use_scale_shift_norm)
        self.middle = MiddleBlock(in_channels=self.hidden_channels[-1], time_embed_dim=time_embed_dim, norm=norm,
                                  activation=activation, use_scale_shift_norm=use_scale_shift_norm, mesh=mesh)
        self.decoder = ModernUNetDecoder(hidden_channels=self.hidden_channels, out_channels=prognostic_channels,
                                         time_embed_dim=time_embed_dim, activation=activation, mesh=mesh,
                                         use_scale_shift_norm=use_scale_shift_norm)

    # ---- modern_unet.py:120-139
    def _prepare_inputs(self, constants=None, prescribed=None, prognostic=None) -> torch.Tensor:
        tensors = []
        if constants is not None:
            tensors.append(constants[:, 0])
        if prescribed is not None:
            tensors.append(prescribed.flatten(1, 2))
        if prognostic is not None:
            tensors.append(prognostic.flatten(1, 2))
        return torch.cat(tensors, dim=1)

    def _fold_faces(self, t):      # [B, T, C, F, H, W] -> [(B F), T, C, H, W]
        b, tt, c, f, h, w = t.shape
        return t.permute(0, 3, 1, 2, 4, 5).reshape(b * f, tt, c, h, w)

    # ---- modern_unet.py:141-176
    def single_forward(self, constants, prescribed, prognostic, y_noised, time):
        time = time * (1000 / self.num_refinement_step)
        if prognostic.ndim == 6:
            prognostic = self._fold_faces(prognostic)
        if y_noised.ndim == 5:
            y_noised = y_noised.expand(-1, prognostic.shape[1], -1, -1, -1)
        elif y_noised.ndim == 6:
            y_noised = self._fold_faces(y_noised.expand(-1, prognostic.shape[1], -1, -1, -1, -1))
        prognostic_t = torch.cat([prognostic, y_noised], dim=2)
        x_t = self._prepare_inputs(constants=constants, prescribed=prescribed, prognostic=prognostic_t).contiguous()
        emb = self.time_embed(fourier_embedding(time, self.hidden_channels[0]))
        enc = self.encoder(x_t, emb)
        mid = self.middle(enc[-1], emb)
        return self.decoder(mid, enc[::-1], emb)

    # ---- modern_unet.py:178-211
    def diffusion_forward(self, constants, prescribed, prognostic, noise_scheduler, target_shape):
        if prognostic.ndim == 6:
            prognostic = self._fold_faces(prognostic)
        if len(target_shape) == 6:
            b, t, c, f, h, w = target_shape
            target_shape = (b * f, t, c, h, w)
        y_noised = torch.randn(target_shape).to(device=prognostic.device)      # host RNG, as the reference (:186)
        for k in noise_scheduler.timesteps:
            k_tensor = torch.full((prognostic.shape[0],), int(k), dtype=torch.long, device=prognostic.device)
            pred = self.single_forward(constants, prescribed, prognostic, y_noised, time=k_tensor).unsqueeze(1)
            y_noised = noise_scheduler.step(pred, k, y_noised).prev_sample
        return y_noised.to(prognostic.device).flatten(1, 2)

    # ---- modern_unet.py:214-293
    def forward(self, constants: Optional[torch.Tensor] = None, prescribed: Optional[torch.Tensor] = None,
                prognostic: torch.Tensor = None, noise_scheduler=None, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        if prognostic is None or noise_scheduler is None:
            raise _lib.DlwpError("prognostic and noise_scheduler are required")
        _lib.require_cuda_tensor(prognostic, "prognostic")
        prescribed_input = prescribed
        if self.mesh == "healpix":
            bsz, nf = prognostic.shape[0], prognostic.shape[3]
        outs = []
        with torch.no_grad():
            for t in range(self.context_size, prognostic.shape[1]):
                t_start = max(0, t - self.context_size)
                if t == self.context_size:
                    prognostic_t = prognostic[:, t_start:t]
                    prescribed = prescribed[:, t_start:t] if prescribed is not None else None
                else:
                    prognostic_t = torch.cat([prognostic[:, t_start:self.context_size],
                                              torch.stack(outs, dim=1)[:, -self.context_size:]], dim=1)
                    prescribed = prescribed_input[:, t - self.context_size:t] if prescribed_input is not None else None
                tshape = prognostic[:, t].unsqueeze(1).shape
                out = self.diffusion_forward(constants, prescribed, prognostic_t, noise_scheduler, tshape)
                if self.mesh == "healpix":
                    out = out.reshape(bsz, nf, *out.shape[1:]).permute(0, 2, 1, 3, 4)
                outs.append(prognostic_t[:, -1] + out)      # the network predicts the residual (:289)
        return torch.stack(outs, dim=1)


class DiffMUNetHPX(DiffModernUNet):
    """modern_unet.py:295-327: the same network on [.., 12, h, w] HEALPix data, faces folded into the batch dimension."""

    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 hidden_channels=(64, 128, 256, 1024), activation=None, context_size: int = 1, mesh: str = "healpix",
                 attention: bool = False, norm: bool = False, use_scale_shift_norm=True, predict_diff=True,
                 num_refinement_step=5, **kwargs):
        super().__init__(constant_channels=constant_channels, prescribed_channels=prescribed_channels,
                         prognostic_channels=prognostic_channels, hidden_channels=hidden_channels, activation=activation,
                         context_size=context_size, mesh="healpix", attention=attention, norm=norm,
                         use_scale_shift_norm=use_scale_shift_norm, predict_diff=predict_diff,
                         num_refinement_step=num_refinement_step)

    def _prepare_inputs(self, constants=None, prescribed=None, prognostic=None) -> torch.Tensor:
        """:314-327: [(B F), (T C), H, W]."""
        def fold(t):      # [B, T, C, F, H, W] -> [(B F), (T C), H, W]
            b, tt, c, f, h, w = t.shape
            return t.permute(0, 3, 1, 2, 4, 5).reshape(b * f, tt * c, h, w)

        tensors = []
        if constants is not None:
            tensors.append(fold(constants[:, :1]))
        if prescribed is not None:
            tensors.append(fold(prescribed))
        if prognostic is not None:
            tensors.append(fold(prognostic) if prognostic.ndim == 6 else prognostic.flatten(1, 2))
        return torch.cat(tensors, dim=1)
