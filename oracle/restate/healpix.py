"""CPU restatement of the HEALPix face padding and the HEALPix U-Net (TEST INFRASTRUCTURE).

`healpix_pad` restates reference utils/healpix.py:165-368 (`HEALPixPadding`): every one of the 12 faces
[..., 12, H, W] is padded by p cells taken from its 8 neighbours; polar faces contribute rotated, the two
corners an equatorial face has no neighbour for are synthesised (`tl` :316-343, `br` :345-368: off-diagonal
cells copied from the two adjacent faces, diagonal cells their mean).  `unet_hpx_rollout` restates
UNetHPX (models/unet/unet.py:386-426 on top of :274-383, :429-555 with HEALPixLayer(Conv2d), healpix.py:69-114).
PINNED by tests/golden/healpix_pad_*.npz and model_unethpx_*.npz (real reference).
"""
import torch
import torch.nn.functional as F

from .unet import _act


def _rot(t, k):
    return torch.rot90(t, k, dims=(-2, -1))


def _synth_tl(t, l, p):
    ret = torch.zeros_like(t)[..., :p, :p].clone()
    ret[..., -1, -1] = 0.5 * t[..., -1, 0] + 0.5 * l[..., 0, -1]
    for i in range(1, p):
        ret[..., -i - 1, -i:] = t[..., -i - 1, :i]
        ret[..., -i:, -i - 1] = l[..., :i, -i - 1]
        ret[..., -i - 1, -i - 1] = 0.5 * t[..., -i - 1, 0] + 0.5 * l[..., 0, -i - 1]
    return ret


def _synth_br(b, r, p):
    ret = torch.zeros_like(b)[..., :p, :p].clone()
    ret[..., 0, 0] = 0.5 * b[..., 0, -1] + 0.5 * r[..., -1, 0]
    for i in range(1, p):
        ret[..., :i, i] = r[..., -i:, i]
        ret[..., i, :i] = b[..., i, -i:]
        ret[..., i, i] = 0.5 * b[..., i, -1] + 0.5 * r[..., -1, i]
    return ret


def healpix_pad(x, p):
    """x [(B*12), C, H, W] -> [(B*12), C, H+2p, W+2p]."""
    nf, c, h, w = x.shape
    d = x.reshape(-1, 12, c, h, w)
    f = [d[:, i] for i in range(12)]
    out = []
    for k in range(4):            # northern faces 0-3
        t, tl, l, b, br, r = f[(k + 1) % 4], f[(k + 2) % 4], f[(k + 3) % 4], f[4 + k], f[8 + k], f[4 + (k + 1) % 4]
        col = torch.cat((_rot(t, 1)[..., -p:, :], f[k], b[..., :p, :]), dim=-2)
        left = torch.cat((_rot(tl, 2)[..., -p:, -p:], _rot(l, -1)[..., -p:], l[..., :p, -p:]), dim=-2)
        right = torch.cat((t[..., -p:, :p], r[..., :p], br[..., :p, :p]), dim=-2)
        out.append(torch.cat((left, col, right), dim=-1))
    for k in range(4):            # equatorial faces 4-7
        t, l, bl, b, r, tr = f[k], f[(k + 3) % 4], f[4 + (k + 3) % 4], f[8 + (k + 3) % 4], f[8 + k], f[4 + (k + 1) % 4]
        col = torch.cat((t[..., -p:, :], f[4 + k], b[..., :p, :]), dim=-2)
        left = torch.cat((_synth_tl(t, l, p)[..., -p:, -p:], l[..., -p:], bl[..., :p, -p:]), dim=-2)
        right = torch.cat((tr[..., -p:, :p], r[..., :p], _synth_br(b, r, p)[..., :p, :p]), dim=-2)
        out.append(torch.cat((left, col, right), dim=-1))
    for k in range(4):            # southern faces 8-11
        t, tl, l, b, br, r = f[4 + (k + 1) % 4], f[k], f[4 + k], f[8 + (k + 3) % 4], f[8 + (k + 2) % 4], f[8 + (k + 1) % 4]
        col = torch.cat((t[..., -p:, :], f[8 + k], _rot(b, 1)[..., :p, :]), dim=-2)
        left = torch.cat((tl[..., -p:, -p:], l[..., -p:], b[..., :p, -p:]), dim=-2)
        right = torch.cat((r[..., -p:, :p], _rot(r, -1)[..., :p], _rot(br, 2)[..., :p, :p]), dim=-2)
        out.append(torch.cat((left, col, right), dim=-1))
    return torch.stack(out, dim=1).reshape(nf, c, h + 2 * p, w + 2 * p)


def unet_hpx_one_step(sd, cfg, x):
    """x [(B*12), Cin, H, W]; encoder/decoder of unet.py:429-555 with HEALPixLayer(Conv2d) blocks: the outer
    Sequential holds (HEALPixLayer, activation) pairs, the conv weight sits at `....layers.1`."""
    hidden = list(cfg["hidden_channels"])
    nconv = cfg.get("n_convolutions", 2)
    act = _act(cfg.get("activation", "th.nn.ReLU()"))
    nl = len(hidden)
    conv = lambda t, name: F.conv2d(healpix_pad(t, 1), sd[name + ".layers.1.weight"], sd[name + ".layers.1.bias"])
    skips = []
    for li in range(nl):
        idx = 0
        if li > 0:
            x = F.avg_pool2d(x, 2, 2)
            idx = 1
        for _ in range(nconv // 2 if li == nl - 1 else nconv):
            x = act(conv(x, f"encoder.layers.{li}.{idx}"))
            idx += 2
        skips.append(x)
    skips = skips[::-1]
    for li in range(nl):
        if li > 0:
            x = torch.cat([skips[li], x], dim=1)
        idx = 0
        for _ in range(nconv // 2 if li == 0 else nconv):
            x = act(conv(x, f"decoder.layers.{li}.{idx}"))
            idx += 2
        if li < nl - 1:
            x = F.conv_transpose2d(x, sd[f"decoder.layers.{li}.{idx}.weight"], sd[f"decoder.layers.{li}.{idx}.bias"], stride=2)
    return F.conv2d(x, sd["decoder.output_layer.weight"], sd["decoder.output_layer.bias"])


def unet_hpx_rollout(sd, cfg, constants, prescribed, prognostic):
    """unet.py:331-383 with the HPX `_prepare_inputs` (:413-426): tensors carry a face axis
    [B, T, C, 12, H, W]; faces are folded into the batch for the backbone."""
    return _hpx_rollout(lambda x: unet_hpx_one_step(sd, cfg, x), cfg, constants, prescribed, prognostic)


def _hpx_rollout(one_step, cfg, constants, prescribed, prognostic):
    ctx = cfg["context_size"]
    b = prognostic.shape[0]
    fold_c = lambda t: t.permute(0, 2, 1, 3, 4).reshape(-1, t.shape[1], t.shape[3], t.shape[4])          # b c f h w
    fold_t = lambda t: t.permute(0, 3, 1, 2, 4, 5).reshape(-1, t.shape[1] * t.shape[2], t.shape[4], t.shape[5])
    outs = []
    for t in range(ctx, prognostic.shape[1]):
        t0 = max(0, t - ctx)
        if t == ctx:
            prog_t = prognostic[:, t0:t]
            presc_t = prescribed[:, t0:t] if prescribed is not None else None
        else:
            prog_t = torch.cat([prognostic[:, t0:ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
            presc_t = prescribed[:, t - ctx:t] if prescribed is not None else None
        parts = []
        if constants is not None:
            parts.append(fold_c(constants[:, 0]))
        if presc_t is not None:
            parts.append(fold_t(presc_t))
        parts.append(fold_t(prog_t))
        y = one_step(torch.cat(parts, dim=1))
        y = y.reshape(b, 12, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)                   # b tc f h w
        outs.append(prog_t[:, -1] + y)
    return torch.stack(outs, dim=1)


# ------------------------------------------------------------------------------------------
# ModernUNet on the HEALPix mesh (MUNetHPX), SURVEY.md 8a row a16
# ------------------------------------------------------------------------------------------
def residual_block(sd, prefix, x, norm_groups=None):
    """unet.py:839-901 `ResidualBlock` with mesh == "healpix": pre-activation, HEALPixPadding(1) in front of both
    3x3 convolutions (padding 0), 1x1 shortcut when the channel count changes; GroupNorm only if the block was
    built with norm=True (the MiddleBlock of a norm=True model, n_groups = 1)."""
    def norm(t, name):
        if norm_groups is None:
            return t
        return F.group_norm(t, norm_groups, sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"])

    h = F.gelu(norm(x, "norm1"))
    h = F.conv2d(healpix_pad(h, 1), sd[prefix + ".conv1.weight"], sd[prefix + ".conv1.bias"])
    h = F.gelu(norm(h, "norm2"))
    h = F.conv2d(healpix_pad(h, 1), sd[prefix + ".conv2.weight"], sd[prefix + ".conv2.bias"])
    if prefix + ".shortcut.weight" in sd:
        x = F.conv2d(x, sd[prefix + ".shortcut.weight"], sd[prefix + ".shortcut.bias"])
    return h + x


def munet_hpx_one_step(sd, cfg, x):
    """ModernUNetEncoder (unet.py:559-632) -> MiddleBlock (:904-946) -> ModernUNetDecoder (:634-760) as they RUN
    on the HEALPix mesh.  Kept on purpose: the decoder never concatenates skips (its isinstance test looks for
    ResidualBlock but the sub-modules are HEALPixLayer wrappers, :747-751), the first decoder block of every level
    but the first therefore receives the 2*hidden channels of the level below; down-sampling is a plain zero-padded
    3x3 stride-2 convolution per face, up-sampling a ConvTranspose2d(4, 2, 1) per face."""
    hidden = list(cfg["hidden_channels"])
    norm = cfg.get("norm", False)
    nl = len(hidden)
    for li in range(nl):
        w, b = sd[f"encoder.layers.{li}.0.weight"], sd[f"encoder.layers.{li}.0.bias"]
        x = F.conv2d(x, w, b, stride=2, padding=1) if li > 0 else F.conv2d(x, w, b)
        x = residual_block(sd, f"encoder.layers.{li}.1.layers.0", x)
    x = residual_block(sd, "middle.res1", x, 1 if norm else None)
    x = residual_block(sd, "middle.res2", x, 1 if norm else None)
    for li in range(nl):
        x = residual_block(sd, f"decoder.layers.{li}.0.layers.0", x)
        x = residual_block(sd, f"decoder.layers.{li}.2.layers.0", x)
        if li < nl - 1:
            x = F.conv_transpose2d(x, sd[f"decoder.layers.{li}.3.weight"], sd[f"decoder.layers.{li}.3.bias"], stride=2,
                                   padding=1)
    x = F.gelu(F.group_norm(x, 8, sd["decoder.final_norm.weight"], sd["decoder.final_norm.bias"]))
    return F.conv2d(x, sd["decoder.output_layer.weight"], sd["decoder.output_layer.bias"])


def munet_hpx_rollout(sd, cfg, constants, prescribed, prognostic):
    """ModernUNet.forward (unet.py:139-203) with MUNetHPX._prepare_inputs (:234-269)."""
    return _hpx_rollout(lambda x: munet_hpx_one_step(sd, cfg, x), cfg, constants, prescribed, prognostic)


def convlstm_hpx_rollout(sd, cfg, constants, prescribed, prognostic):
    """reference models/convlstm/convlstm.py:210-251 with mesh="healpix" (ConvLSTMHPX :258-305): tensors
    [B, T, C, 12, H, W], faces folded into the batch per step (:293-305), every convolution behind HEALPixPadding(1)
    (utils/healpix.py:69-114), LSTM state carried from t = 0, teacher forcing while t < context_size."""
    hidden = list(cfg["hidden_sizes"])
    ctx = cfg["context_size"]
    b, t_total, cg, f, hgt, wid = prognostic.shape
    fold = lambda t: t.permute(0, 2, 1, 3, 4).reshape(b * f, t.shape[1], hgt, wid)    # "b c f h w -> (b f) c h w"
    hs = [torch.zeros(b * f, n, hgt, wid) for n in hidden]
    cs = [torch.zeros(b * f, n, hgt, wid) for n in hidden]
    conv = lambda x, name: F.conv2d(healpix_pad(x, 1), sd[name + ".layers.1.weight"], sd.get(name + ".layers.1.bias"))
    outs = []
    for t in range(t_total):
        prog_t = prognostic[:, t] if t < ctx else outs[-1]
        parts = []
        if constants is not None:
            parts.append(fold(constants[:, 0]))
        if prescribed is not None:
            parts.append(fold(prescribed[:, t]))
        parts.append(fold(prog_t))
        x = torch.cat(parts, dim=1)
        x = torch.tanh(conv(x, "encoder.0"))
        x = torch.tanh(conv(x, "encoder.2"))
        x = conv(x, "encoder.4")
        for i, n in enumerate(hidden):
            g = conv(torch.cat((x, hs[i]), dim=1), f"clstm.{i}.conv")
            netin, ig, fg, og = torch.split(g, n, dim=1)
            cs[i] = torch.sigmoid(fg) * cs[i] + torch.sigmoid(ig) * torch.tanh(netin)
            hs[i] = torch.sigmoid(og) * torch.tanh(cs[i])
            x = hs[i]
        out = conv(x, "decoder")
        out = out.reshape(b, f, cg, hgt, wid).permute(0, 2, 1, 3, 4)                  # "(b f) c h w -> b c f h w"
        outs.append(prog_t + out)
    return torch.stack(outs[ctx:], dim=1)
