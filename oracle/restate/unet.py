"""CPU restatement of the (classic) U-Net and the ConvLSTM (TEST INFRASTRUCTURE).

`unet_one_step`: reference models/unet/unet.py:429-555 (UNetEncoder / UNetDecoder) on the
equirectangular mesh.  The reference class as shipped crashes (encoder applies CylinderPad(1) AND
Conv2d(padding=1), :456-462, so every conv grows H and W by 2, while the decoder uses padding=0,
:512-518).  The oracle follows the documented runtime workaround of SURVEY.md section 8c: encoder
convs run with padding 0 (consistent with the decoder and with convlstm.py:47-55); the golden
fixtures are generated from the real reference class with `padding=(0,0)` set on the encoder convs
after construction.  PINNED by tests/golden/unet_*.npz.

`convlstm_rollout`: reference models/convlstm/convlstm.py:210-251 (loop starts at t = 0 with teacher
forcing for t < context_size, (h, c) carried between steps, outputs[context_size:] returned) and
ConvLSTMCell.forward (:82-111).  PINNED by tests/golden/convlstm_*.npz.
"""
import torch
import torch.nn.functional as F

from .common import rollout


def cylinder_pad(x, p=1):
    """utils/utils.py:11-26: circular along longitude (W), zeros along latitude (H)."""
    x = F.pad(x, (p, p, 0, 0), mode="circular")
    return F.pad(x, (0, 0, p, p))


def _act(name):
    n = str(name)
    if "GELU" in n:
        return F.gelu
    if "ReLU" in n and "Leaky" not in n:
        return F.relu
    if "Tanh" in n:
        return torch.tanh
    if "SiLU" in n:
        return F.silu
    raise ValueError(f"unsupported activation {name}")


def unet_one_step(sd, cfg, x):
    hidden = list(cfg["hidden_channels"])
    nconv = cfg.get("n_convolutions", 2)
    act = _act(cfg.get("activation", "th.nn.GELU()"))
    nl = len(hidden)
    skips = []
    for li in range(nl):
        idx = 0
        if li > 0:
            x = F.avg_pool2d(x, 2, 2)
            idx = 1
        n_here = nconv // 2 if li == nl - 1 else nconv
        for _ in range(n_here):
            # Sequential: [AvgPool] (CylinderPad, Conv2d, activation)*n -> conv is at idx+1
            x = act(F.conv2d(cylinder_pad(x), sd[f"encoder.layers.{li}.{idx + 1}.weight"],
                             sd[f"encoder.layers.{li}.{idx + 1}.bias"]))
            idx += 3
        skips.append(x)
    skips = skips[::-1]
    for li in range(nl):
        if li > 0:
            x = torch.cat([skips[li], x], dim=1)
        n_here = nconv // 2 if li == 0 else nconv
        idx = 0
        for _ in range(n_here):
            x = act(F.conv2d(cylinder_pad(x), sd[f"decoder.layers.{li}.{idx + 1}.weight"],
                             sd[f"decoder.layers.{li}.{idx + 1}.bias"]))
            idx += 3
        if li < nl - 1:
            x = F.conv_transpose2d(x, sd[f"decoder.layers.{li}.{idx}.weight"], sd[f"decoder.layers.{li}.{idx}.bias"],
                                   stride=2)
    return F.conv2d(x, sd["decoder.output_layer.weight"], sd["decoder.output_layer.bias"])


def unet_rollout(sd, cfg, constants, prescribed, prognostic):
    return rollout(lambda xt: unet_one_step(sd, cfg, xt), cfg["context_size"], constants, prescribed, prognostic)


def convlstm_rollout(sd, cfg, constants, prescribed, prognostic):
    hidden = list(cfg["hidden_sizes"])
    ctx = cfg["context_size"]
    b, t_total = prognostic.shape[0], prognostic.shape[1]
    hgt, wid = prognostic.shape[-2], prognostic.shape[-1]
    hs = [torch.zeros(b, n, hgt, wid) for n in hidden]
    cs = [torch.zeros(b, n, hgt, wid) for n in hidden]
    outs = []
    conv = lambda x, name: F.conv2d(cylinder_pad(x), sd[name + ".weight"], sd.get(name + ".bias"))
    for t in range(t_total):
        prog_t = prognostic[:, t] if t < ctx else outs[-1]
        parts = []
        if constants is not None:
            parts.append(constants[:, 0])
        if prescribed is not None:
            parts.append(prescribed[:, t])
        parts.append(prog_t)
        x = torch.cat(parts, dim=1)
        x = torch.tanh(conv(x, "encoder.1"))
        x = torch.tanh(conv(x, "encoder.4"))
        x = conv(x, "encoder.7")
        for i, n in enumerate(hidden):
            g = conv(torch.cat((x, hs[i]), dim=1), f"clstm.{i}.conv.1")
            netin, ig, fg, og = torch.split(g, n, dim=1)
            c_new = torch.sigmoid(fg) * cs[i] + torch.sigmoid(ig) * torch.tanh(netin)
            h_new = torch.sigmoid(og) * torch.tanh(c_new)
            hs[i], cs[i] = h_new, c_new
            x = h_new
        out = conv(x, "decoder.1")
        outs.append(prog_t + out)
    return torch.stack(outs[ctx:], dim=1)
