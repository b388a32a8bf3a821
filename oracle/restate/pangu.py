"""CPU restatement of the (2-D variant of) Pangu-Weather backbone (TEST INFRASTRUCTURE).

State-dict driven restatement of reference models/panguweather/panguweather.py + utils/ in eval
mode.  PINNED by tests/golden/pangu_*.npz (real reference).

Quirks kept bit-for-bit (SURVEY.md header item 3):
  * one pressure "level" only: img_size = (1, n_lat/p, n_lon/p) (:403), zero-padded to 2 for the
    (2,6,12) window (utils/pad.py:21-24) -- the padded level takes part in the softmax;
  * forward cyclic shift rolls longitude by shift_LAT (`shifts=(-pl,-lat,-lat)`, :291) while the
    reverse roll uses shift_lon (:310);
  * earth-specific bias table indexed by window type = (pl_window, lat_window) and shared over
    longitude (:158, :189-196); shift mask from utils/shift_window_mask.py:37-73 (0 / -100).
"""
import torch
import torch.nn.functional as F

from .common import rollout


def get_pad3d(res, win):
    pl, lat, lon = res
    wpl, wlat, wlon = win
    out = [0, 0, 0, 0, 0, 0]  # left right top bottom front back
    if pl % wpl:
        p = wpl - pl % wpl
        out[4], out[5] = p // 2, p - p // 2
    if lat % wlat:
        p = wlat - lat % wlat
        out[2], out[3] = p // 2, p - p // 2
    if lon % wlon:
        p = wlon - lon % wlon
        out[0], out[1] = p // 2, p - p // 2
    return tuple(out)


def earth_position_index(win):
    """utils/earth_position_index.py:4-45."""
    wpl, wlat, wlon = win
    zi, zj = torch.arange(wpl), -torch.arange(wpl) * wpl
    hi, hj = torch.arange(wlat), -torch.arange(wlat) * wlat
    w = torch.arange(wlon)
    c1 = torch.stack(torch.meshgrid(zi, hi, w, indexing="ij")).flatten(1)
    c2 = torch.stack(torch.meshgrid(zj, hj, w, indexing="ij")).flatten(1)
    co = (c1[:, :, None] - c2[:, None, :]).permute(1, 2, 0).contiguous()
    co[:, :, 2] += wlon - 1
    co[:, :, 1] *= 2 * wlon - 1
    co[:, :, 0] *= (2 * wlon - 1) * wlat * wlat
    return co.sum(-1)


def window_partition(x, win):
    b, pl, lat, lon, c = x.shape
    wpl, wlat, wlon = win
    x = x.view(b, pl // wpl, wpl, lat // wlat, wlat, lon // wlon, wlon, c)
    return x.permute(0, 5, 1, 3, 2, 4, 6, 7).contiguous().view(-1, (pl // wpl) * (lat // wlat), wpl, wlat, wlon, c)


def window_reverse(wins, win, pl, lat, lon):
    wpl, wlat, wlon = win
    b = int(wins.shape[0] / (lon / wlon))
    x = wins.view(b, lon // wlon, pl // wpl, lat // wlat, wpl, wlat, wlon, -1)
    return x.permute(0, 2, 4, 3, 5, 1, 6, 7).contiguous().view(b, pl, lat, lon, -1)


def shift_window_mask(res, win, shift):
    pl, lat, lon = res
    wpl, wlat, wlon = win
    spl, slat, slon = shift
    img = torch.zeros(1, pl, lat, lon + slon, 1)
    cnt = 0
    for a in (slice(0, -wpl), slice(-wpl, -spl), slice(-spl, None)):
        for b_ in (slice(0, -wlat), slice(-wlat, -slat), slice(-slat, None)):
            for c in (slice(0, -wlon), slice(-wlon, -slon), slice(-slon, None)):
                img[:, a, b_, c, :] = cnt
                cnt += 1
    img = img[:, :, :, :lon, :]
    mw = window_partition(img, win)
    mw = mw.view(mw.shape[0], mw.shape[1], wpl * wlat * wlon)
    m = mw.unsqueeze(2) - mw.unsqueeze(3)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def earth_attention(x, sd, prefix, win, num_heads, types, mask):
    """panguweather.py:176-211.  x [B*nLon, nW, N, C]."""
    b_, nw_, n, c = x.shape
    hd = c // num_heads
    qkv = F.linear(x, sd[prefix + ".qkv.weight"], sd[prefix + ".qkv.bias"])
    qkv = qkv.reshape(b_, nw_, n, 3, num_heads, hd).permute(3, 0, 4, 1, 2, 5)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = sd.get(prefix + ".earth_position_index")
    if idx is None:
        idx = earth_position_index(win)
    bias = sd[prefix + ".earth_position_bias_table"][idx.view(-1)].view(n, n, types, -1).permute(3, 2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nlon = mask.shape[0]
        attn = attn.view(b_ // nlon, nlon, num_heads, nw_, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, num_heads, nw_, n, n)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).permute(0, 2, 3, 1, 4).reshape(b_, nw_, n, c)
    return F.linear(out, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"])


def earth_block(x, sd, prefix, res, win, shift, num_heads):
    """panguweather.py:275-323."""
    pl, lat, lon = res
    b, l, c = x.shape
    ln = lambda t, n: F.layer_norm(t, (c,), sd[f"{prefix}.{n}.weight"], sd[f"{prefix}.{n}.bias"])
    shortcut = x
    x = ln(x, "norm1").view(b, pl, lat, lon, c)
    pad = get_pad3d(res, win)
    x = F.pad(x.permute(0, 4, 1, 2, 3), pad).permute(0, 2, 3, 4, 1)
    _, plp, latp, lonp, _ = x.shape
    roll = bool(shift[0] and shift[1] and shift[2])
    types = (plp // win[0]) * (latp // win[1])
    if roll:
        x = torch.roll(x, shifts=(-shift[0], -shift[1], -shift[1]), dims=(1, 2, 3))   # [sic] :291
        mask = sd.get(prefix + ".attn_mask")
        if mask is None:
            mask = shift_window_mask((plp, latp, lonp), win, shift)
    else:
        mask = None
    xw = window_partition(x, win)
    xw = xw.view(xw.shape[0], xw.shape[1], win[0] * win[1] * win[2], c)
    aw = earth_attention(xw, sd, prefix + ".attn", win, num_heads, types, mask)
    aw = aw.view(aw.shape[0], aw.shape[1], win[0], win[1], win[2], c)
    x = window_reverse(aw, win, plp, latp, lonp)
    if roll:
        x = torch.roll(x, shifts=shift, dims=(1, 2, 3))                               # :310
    # crop3d
    fr, to_, le = pad[4], pad[2], pad[0]
    x = x[:, fr:fr + pl, to_:to_ + lat, le:le + lon]
    x = shortcut + x.reshape(b, pl * lat * lon, c)
    h = ln(x, "norm2")
    h = F.linear(h, sd[prefix + ".mlp.fc1.weight"], sd[prefix + ".mlp.fc1.bias"])
    h = F.linear(F.gelu(h), sd[prefix + ".mlp.fc2.weight"], sd[prefix + ".mlp.fc2.bias"])
    return x + h


def basic_layer(x, sd, prefix, res, depth, num_heads, win):
    for i in range(depth):
        shift = (0, 0, 0) if i % 2 == 0 else (1, 3, 6)     # :227: default shift when None is passed
        x = earth_block(x, sd, f"{prefix}.blocks.{i}", res, win, shift, num_heads)
    return x


def down_sample(x, sd, res_in, res_out):
    """panguweather.py:80-130."""
    b, n, c = x.shape
    pl, lat, lon = res_in
    _, olat, olon = res_out
    hp, wp = olat * 2 - lat, olon * 2 - lon
    x = x.reshape(b, pl, lat, lon, c)
    x = F.pad(x.permute(0, 4, 1, 2, 3), (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2, 0, 0)).permute(0, 2, 3, 4, 1)
    x = x.reshape(b, pl, olat, 2, olon, 2, c).permute(0, 1, 2, 4, 3, 5, 6).reshape(b, pl * olat * olon, 4 * c)
    x = F.layer_norm(x, (4 * c,), sd["downsample.norm.weight"], sd["downsample.norm.bias"])
    return F.linear(x, sd["downsample.linear.weight"])


def up_sample(x, sd, res_in, res_out):
    """panguweather.py:30-77."""
    b, n, c = x.shape
    pl, lat, lon = res_in
    _, olat, olon = res_out
    x = F.linear(x, sd["upsample.linear1.weight"])
    x = x.reshape(b, pl, lat, lon, 2, 2, c // 2).permute(0, 1, 2, 4, 3, 5, 6).reshape(b, pl, lat * 2, lon * 2, -1)
    ph, pw = lat * 2 - olat, lon * 2 - olon
    x = x[:, :pl, ph // 2: 2 * lat - (ph - ph // 2), pw // 2: 2 * lon - (pw - pw // 2), :]
    x = x.reshape(b, -1, x.shape[-1])
    x = F.layer_norm(x, (x.shape[-1],), sd["upsample.norm.weight"], sd["upsample.norm.bias"])
    return F.linear(x, sd["upsample.linear2.weight"])


def pangu_one_step(sd, cfg, x):
    """panguweather.py:512-535."""
    p = tuple(cfg["patch_size"])
    win = tuple(cfg["window_size"])
    heads = list(cfg["num_heads"])
    nlat, nlon = cfg["n_lat"], cfg["n_lon"]
    hr, wr = nlat % p[0], nlon % p[1]
    pt = pb = pl_ = pr = 0
    if hr:
        hp = p[0] - hr
        pt, pb = hp // 2, hp - hp // 2
    if wr:
        wp = p[1] - wr
        pl_, pr = wp // 2, wp - wp // 2
    x = F.pad(x, (pl_, pr, pt, pb))
    x = F.conv2d(x, sd["patchembed2d.proj.weight"], sd["patchembed2d.proj.bias"], stride=p).unsqueeze(2)
    b, c, pl, lat, lon = x.shape
    x = x.reshape(b, c, -1).transpose(1, 2)
    res = (1, nlat // p[0], nlon // p[1])
    res2 = (1, res[1] // 2, res[2] // 2)
    x = basic_layer(x, sd, "layer1", res, 2, heads[0], win)
    skip = x
    x = down_sample(x, sd, res, res2)
    x = basic_layer(x, sd, "layer2", res2, 6, heads[1], win)
    x = basic_layer(x, sd, "layer3", res2, 6, heads[2], win)
    x = up_sample(x, sd, res2, res)
    x = basic_layer(x, sd, "layer4", res, 2, heads[3], win)
    out = torch.cat([x, skip], dim=-1).transpose(1, 2).reshape(b, -1, pl, lat, lon)[:, :, 0]
    out = F.conv_transpose2d(out, sd["patchrecovery2d.conv.weight"], sd["patchrecovery2d.conv.bias"], stride=p)
    hh, ww = out.shape[2], out.shape[3]
    hp, wp = hh - nlat, ww - nlon
    return out[:, :, hp // 2: hh - (hp - hp // 2), wp // 2: ww - (wp - wp // 2)]


def pangu_rollout(sd, cfg, constants, prescribed, prognostic):
    return rollout(lambda xt: pangu_one_step(sd, cfg, xt), cfg["context_size"], constants, prescribed, prognostic)
