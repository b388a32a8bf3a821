"""Backward of the window attention (csrc/window_attn_bwd.hip, dlwp_window_attn_bwd_f32) against autograd of the torch
restatement of the operator (training.window_attention_torch, itself checked against the forward kernels and -- through the
model-level gradient fixtures of tests/test_training_gpu.py -- against gradients the REAL reference classes produced).
Reference backward: scripts/train.py:271 through swin_transformer.py:122-154 / panguweather.py:176-211."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _swin_spec(h, w, wh, ww, heads, d, shifted):
    from dlwp_benchmark_amd import ops

    sh, sw = (wh // 2, ww // 2) if shifted else (0, 0)
    return ops.WindowSpec(grid=(1, h, w), padded=(1, h, w), pad_lead=(0, 0, 0), window=(1, wh, ww), shift_fwd=(0, sh, sw),
                          shift_back=(0, sh, sw), use_mask=shifted, mask_b1=(ops.BIG, h - wh, w - ww),
                          mask_b2=(ops.BIG, h - wh // 2, w - ww // 2), bias_mode=0, heads=heads, head_dim=d,
                          scale=d ** -0.5), ((2 * wh - 1) * (2 * ww - 1), heads)


def _pangu_spec(lat, lon, heads, d, roll):
    """panguweather.py:285-316 on a one-level grid: window (2, 6, 12), zero padding to a window multiple (the padded level and
    the padded latitude rows carry the qkv bias), asymmetric roll [sic] (:291 vs :310), earth-specific bias table."""
    from dlwp_benchmark_amd import ops

    win = (2, 6, 12)
    pad = lambda n, w: (w - n % w) % w
    pp, pa, po = pad(1, 2), pad(lat, 6), pad(lon, 12)
    padded = (1 + pp, lat + pa, lon + po)
    lead = (pp // 2, pa // 2, po // 2)
    spl, slat, slon = 1, 3, 6
    fwd, back = ((spl, slat, slat), (spl, slat, slon)) if roll else ((0, 0, 0), (0, 0, 0))
    ppl, plat, plon = padded
    types = (ppl // 2) * (plat // 6)
    spec = ops.WindowSpec(grid=(1, lat, lon), padded=padded, pad_lead=lead, window=win, shift_fwd=fwd, shift_back=back,
                          use_mask=roll, mask_b1=(ppl - 2, plat - 6, plon + slon - 12) if roll else (ops.BIG,) * 3,
                          mask_b2=(ppl - spl, plat - slat, plon) if roll else (ops.BIG,) * 3, bias_mode=1, heads=heads,
                          head_dim=d, scale=d ** -0.5)
    return spec, (2 * 2 * 6 * 6 * 23, types, heads)


def _check(spec, table_shape, batch, seed, tol=2e-5):
    from dlwp_benchmark_amd import ops, training as T

    g = torch.Generator().manual_seed(seed)
    l = spec.grid[0] * spec.grid[1] * spec.grid[2]
    c = spec.heads * spec.head_dim
    qkv = torch.randn(batch, l, 3 * c, generator=g).to(DEV)
    bias = (0.3 * torch.randn(3 * c, generator=g)).to(DEV)
    table = (0.5 * torch.randn(*table_shape, generator=g)).to(DEV)
    gout = torch.randn(batch, l, c, generator=g).to(DEV)
    q_, b_, t_ = (t.clone().requires_grad_(True) for t in (qkv, bias, table))
    out = T.window_attention_torch(q_, b_, t_, spec)
    want = torch.autograd.grad(out, [q_, b_, t_], gout, allow_unused=True)
    gq, gb, gt = ops.window_attention_backward(qkv, bias, table, spec, gout)
    torch.cuda.synchronize()
    assert rel_l2(gq, want[0]) <= tol, ("dqkv", rel_l2(gq, want[0]))
    assert rel_l2(gt, want[2]) <= tol, ("dtable", rel_l2(gt, want[2]))
    padded = tuple(spec.padded) != tuple(spec.grid)
    if padded:
        # (rolled Pangu blocks mask the padded level against the real one with -100: its tokens then receive e^-100 of the
        # gradient -- exactly 0 here, cancellation noise of ~1e-6 in the torch form -- so the scale is that of dqkv)
        assert gb is not None
        err = float((gb.double() - want[1].double()).norm())
        assert err <= tol * max(float(want[1].double().norm()), float(want[0].double().norm())), ("dbias", err)
    else:
        assert gb is None and (want[1] is None or float(want[1].abs().max()) == 0.0)
    return gq, gt


@pytest.mark.parametrize("shifted", [False, True])
@pytest.mark.parametrize("h,w,wh,ww,heads,d", [(16, 32, 16, 32, 2, 8), (32, 64, 32, 64, 4, 24), (16, 32, 8, 16, 2, 16),
                                               (12, 20, 12, 20, 2, 48), (8, 16, 8, 16, 1, 64)])
def test_swin_geometry_gradients(h, w, wh, ww, heads, d, shifted):
    spec, tshape = _swin_spec(h, w, wh, ww, heads, d, shifted)
    _check(spec, tshape, batch=2, seed=h + d)


@pytest.mark.parametrize("roll", [False, True])
@pytest.mark.parametrize("lat,lon,heads,d", [(12, 24, 2, 32), (16, 40, 3, 16), (8, 16, 2, 32)])
def test_earth_window_gradients_with_padding_and_asymmetric_roll(lat, lon, heads, d, roll):
    spec, tshape = _pangu_spec(lat, lon, heads, d, roll)
    _check(spec, tshape, batch=2, seed=lat + lon)


def test_backward_never_materialises_the_scores():
    """C3 stage 0 geometry (the reference window is the whole 32 x 64 map: N = 2048), batch 4: [B, heads, N, N] fp32 would be
    256 MiB per copy (the torch recomputation holds several); the HIP backward allocates qkv-sized tensors only."""
    from dlwp_benchmark_amd import ops

    spec, tshape = _swin_spec(32, 64, 32, 64, 4, 24, True)
    b, l, c = 4, 2048, 96
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(b, l, 3 * c, generator=g).to(DEV).requires_grad_(True)
    bias = torch.zeros(3 * c, device=DEV)
    table = (0.5 * torch.randn(*tshape, generator=g)).to(DEV).requires_grad_(True)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = ops.window_attention(qkv, bias, table, spec)
    out.square().sum().backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    scores = b * 4 * l * l * 4
    assert qkv.grad is not None and table.grad is not None and torch.isfinite(qkv.grad).all()
    assert peak < scores // 8, f"peak {peak / 2**20:.1f} MiB vs one score tensor {scores / 2**20:.1f} MiB"


def test_linear_training_function_matches_autograd():
    """training._LinearFn: forward, dX and dW through dlwp_linear_f32 (fp32-accurate), bias gradient as a column sum."""
    from dlwp_benchmark_amd import ops

    torch.manual_seed(5)
    m = torch.nn.Linear(96, 288).to(DEV)
    x = torch.randn(2, 512, 96, device=DEV, requires_grad=True)
    r = torch.randn(2, 512, 288, device=DEV)
    y = ops.linear(x, m, act=1)
    (y * r).sum().backward()
    got = (x.grad.clone(), m.weight.grad.clone(), m.bias.grad.clone())
    x.grad = None
    m.zero_grad()
    y2 = torch.nn.functional.gelu(torch.nn.functional.linear(x.double(), m.weight.double(), m.bias.double()))
    gx, gw, gb = torch.autograd.grad((y2 * r.double()).sum(), [x, m.weight, m.bias])
    assert rel_l2(y, y2) <= 1e-6
    assert rel_l2(got[0], gx) <= 2e-6 and rel_l2(got[1], gw) <= 2e-6 and rel_l2(got[2], gb) <= 2e-6


@pytest.mark.parametrize("h,w,c,nb,frac", [(32, 64, 16, 4, 1.0), (128, 256, 64, 4, 1.0), (32, 64, 32, 2, 0.5)])
def test_afno_filter_backward_matches_autograd(h, w, c, nb, frac):
    """ops.afno2d_filter_backward (hand-written kept-column transforms + dlwp_afno2d_mix_bwd_f32 + four einsums) against autograd
    of training.afno_filter_torch (fourcastnet.py:85-124 in torch operators)."""
    from dlwp_benchmark_amd import ops, training as T

    g = torch.Generator().manual_seed(h + c)
    bs = c // nb
    x = torch.randn(2, c, h, w, generator=g).to(DEV)
    w1, w2 = [(0.3 * torch.randn(2, nb, bs, bs, generator=g)).to(DEV) for _ in range(2)]
    b1, b2 = [(0.3 * torch.randn(2, nb, bs, generator=g)).to(DEV) for _ in range(2)]
    gy = torch.randn(2, c, h, w, generator=g).to(DEV)
    ins = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    out = T.afno_filter_torch(*ins, nb, 0.01, frac)
    want = torch.autograd.grad(out, ins, gy)
    got = ops.afno2d_filter_backward(x, gy, w1, b1, w2, b2, nb, 0.01, frac)
    assert got is not None, "grid should be covered by the hand-written transforms"
    for name, a, b in zip(("dx", "dw1", "db1", "dw2", "db2"), got, want):
        assert rel_l2(a, b) <= 2e-5, (name, rel_l2(a, b))
