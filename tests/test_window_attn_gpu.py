"""Op-level tests of the second-generation 2-D window attention (csrc/window_attn2.hip) against the first-generation
kernel's fp32-MFMA form (an independent implementation: running maximum, per-score bias gather and mask, no tile
skipping) -- itself pinned by the model goldens of the real reference classes (tests/test_backbones_gpu.py)."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


def _spec(h, w, heads, d, shifted, shift=None):
    from dlwp_benchmark_amd import ops

    sh, sw = (shift if shift is not None else (h // 2, w // 2)) if shifted else (0, 0)
    return ops.WindowSpec(grid=(1, h, w), padded=(1, h, w), pad_lead=(0, 0, 0), window=(1, h, w), shift_fwd=(0, sh, sw),
                          shift_back=(0, sh, sw), use_mask=shifted, mask_b1=(ops.BIG, 0, 0), mask_b2=(ops.BIG, h - sh, w - sw),
                          bias_mode=0, heads=heads, head_dim=d, scale=d ** -0.5), (2 * h - 1) * (2 * w - 1)


def _inputs(b, h, w, heads, d, rows, seed=0, qk_gain=1.0):
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(b, h * w, 3, heads, d, generator=g)
    qkv[:, :, :2] *= qk_gain
    bias = torch.randn(3 * heads * d, generator=g) * 0.1
    table = torch.randn(rows, heads, generator=g) * 0.5
    return qkv.reshape(b, h * w, 3 * heads * d).cuda(), bias.cuda(), table.cuda()


@pytest.mark.parametrize("h,w,heads,d", [(32, 64, 4, 24), (16, 32, 4, 48), (16, 32, 2, 8), (8, 16, 2, 16), (12, 16, 2, 8)])
@pytest.mark.parametrize("shifted", [False, True])
def test_fast_path_matches_generic_kernel(h, w, heads, d, shifted):
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd import ops
    import ctypes

    spec, rows = _spec(h, w, heads, d, shifted)
    qkv, bias, table = _inputs(2, h, w, heads, d, rows, qk_gain=2.0)
    dsc = spec.to_c()
    dsc.form = 1
    covered = L.load().dlwp_window_attn_workspace_bytes(ctypes.byref(dsc), 2, 0) > 0
    if shifted and (w // 2) % 16:
        assert not covered            # longitude region boundary inside a 16-key block: generic kernel
    else:
        assert covered, "descriptor should run on the fast path"
    want = ops.window_attention(qkv, bias, table, spec, precision="fp32_mfma")       # generic kernel, fp32 MFMA
    got, fb = ops.window_attention(qkv, bias, table, spec, precision="bf16x6", count_fallbacks=True)
    torch.cuda.synchronize()
    assert fb == 0
    assert rel_l2(got, want) <= 2e-6
    got16, fb16 = ops.window_attention(qkv, bias, table, spec, precision="bf16", count_fallbacks=True)
    assert fb16 == 0
    e = rel_l2(got16, want)
    assert 1e-6 < e <= 2e-2, e


@pytest.mark.parametrize("shifted", [False, True])
def test_exponent_slack_fallback_is_exact(shifted):
    """Scores that climb by hundreds of binades along the key order leave the 2^+-100 slack around the offset taken from
    the first tile: those workgroups must notice (row-sum check), recompute with the exact maximum and still match."""
    from dlwp_benchmark_amd import ops

    h, w, heads, d = 16, 32, 2, 24
    spec, rows = _spec(h, w, heads, d, shifted)
    qkv, bias, table = _inputs(1, h, w, heads, d, rows, seed=3)
    n = h * w
    x = qkv.view(1, n, 3, heads, d)
    ramp = torch.linspace(0.0, 1.0, n, device="cuda").view(1, n, 1, 1)
    c = torch.ones(d, device="cuda") / d ** 0.5
    x[:, :, 0] = c * 6.0                       # every query the same direction
    x[:, :, 1] = c * (ramp * 250.0)            # key norms ramp up: logits from 0 to ~300 along the token order
    want = ops.window_attention(qkv, bias, table, spec, precision="fp32_mfma")
    got, fb = ops.window_attention(qkv, bias, table, spec, precision="bf16x6", count_fallbacks=True)
    torch.cuda.synchronize()
    assert fb > 0, "the adversarial logits did not trigger the exact fallback"
    assert torch.isfinite(got).all()
    # logits of magnitude ~430 (log2 units) carry an fp32 rounding error of ~3e-5 each in EITHER implementation
    assert rel_l2(got, want) <= 2e-4


# ---------------------------------------------------------------------------------------------------------------------
# third kernel (csrc/window_attn3.hip): small 3-D windows with the earth-specific bias (every Pangu block)
# ---------------------------------------------------------------------------------------------------------------------
def _earth_spec(grid, window, heads, shifted):
    """The descriptor models/pangu.py builds for a block (panguweather.py:285-316): ZeroPad3d to the window, roll by half a
    window on all three axes with the reference's asymmetric forward / backward longitude shift, shift_window_mask.py regions."""
    from dlwp_benchmark_amd import ops
    from dlwp_benchmark_amd.models.pangu import _pad3d

    p = _pad3d(grid, window)
    padded = (grid[0] + p[4] + p[5], grid[1] + p[2] + p[3], grid[2] + p[0] + p[1])
    shift = tuple(w // 2 for w in window)
    roll = shifted and all(shift)
    spl, slat, slon = shift
    ppl, plat, plon = padded
    wpl, wlat, wlon = window
    spec = ops.WindowSpec(
        grid=grid, padded=padded, pad_lead=(p[4], p[2], p[0]), window=window,
        shift_fwd=(spl, slat, slat) if roll else (0, 0, 0), shift_back=(spl, slat, slon) if roll else (0, 0, 0), use_mask=roll,
        mask_b1=(ppl - wpl, plat - wlat, plon + slon - wlon) if roll else (ops.BIG,) * 3,
        mask_b2=(ppl - spl, plat - slat, plon) if roll else (ops.BIG,) * 3,
        bias_mode=1, heads=heads, head_dim=32, scale=32 ** -0.5)
    rows = wpl * wpl * wlat * wlat * (2 * wlon - 1)
    types = (padded[0] // wpl) * (padded[1] // wlat)
    return spec, rows, types


@pytest.mark.parametrize("grid,window", [((1, 16, 32), (2, 6, 12)),      # the reference's shape class: one level, padded everywhere
                                         ((2, 12, 24), (2, 6, 12)),      # two real levels: both planes are query planes
                                         ((1, 8, 16), (1, 4, 8)),        # one plane, no padding
                                         ((2, 11, 13), (2, 5, 7)),       # 35 tokens per plane (3 blocks), odd pads
                                         ((1, 9, 30), (2, 3, 10))])      # 30 tokens per plane (2 blocks)
@pytest.mark.parametrize("shifted", [False, True])
def test_earth_window_kernel_matches_generic_kernel(grid, window, shifted):
    import ctypes

    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd import ops

    heads, b = 3, 2
    spec, rows, types = _earth_spec(grid, window, heads, shifted)
    g = torch.Generator().manual_seed(7)
    ltok = grid[0] * grid[1] * grid[2]
    qkv = torch.randn(b, ltok, 3, heads, 32, generator=g)
    qkv[:, :, :2] *= 1.5
    qkv = qkv.reshape(b, ltok, 3 * heads * 32).cuda()
    bias = (torch.randn(3 * heads * 32, generator=g) * 0.3).cuda()
    table = (torch.randn(rows, types, heads, generator=g) * 0.5).cuda()
    dsc = spec.to_c()
    dsc.form = 1
    assert L.load().dlwp_window_attn_workspace_bytes(ctypes.byref(dsc), b, 0) > 0, "descriptor should run on the earth-window kernel"
    want = ops.window_attention(qkv, bias, table, spec, precision="fp32_mfma")       # generic kernel, fp32 MFMA
    got = ops.window_attention(qkv, bias, table, spec, precision="bf16x6")
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) <= 2e-6
    got16 = ops.window_attention(qkv, bias, table, spec, precision="bf16")
    e = rel_l2(got16, want)
    assert 1e-6 < e <= 2e-2, e


def test_earth_window_kernel_large_logits():
    """Logits in the hundreds: the exact row maximum keeps the exponentials in range (no slack to leave on this path)."""
    from dlwp_benchmark_amd import ops

    spec, rows, types = _earth_spec((1, 16, 32), (2, 6, 12), 2, True)
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(1, 512, 3, 2, 32, generator=g)
    qkv[:, :, :2] *= 6.0
    qkv = qkv.reshape(1, 512, 192).cuda()
    bias = (torch.randn(192, generator=g) * 0.3).cuda()
    table = (torch.randn(rows, types, 2, generator=g) * 20.0).cuda()
    want = ops.window_attention(qkv, bias, table, spec, precision="fp32_mfma")
    got = ops.window_attention(qkv, bias, table, spec, precision="bf16x6")
    assert torch.isfinite(got).all() and rel_l2(got, want) <= 2e-4      # fp32 rounding of logits of several hundred


@pytest.mark.parametrize("kind", ["swin", "swin_shifted", "pangu", "pangu_rolled"])
def test_bf16_tensor_handover_is_bit_identical(kind):
    """dlwp_window_attn_bf16_io (bf16 qkv in, bf16 out: the hand-over of a block in the all-bf16 form) computes exactly what
    dlwp_window_attn_bf16 computes on the same bf16-representable values, rounded once more to bf16 on the way out."""
    from dlwp_benchmark_amd import ops
    from test_window_attn_bwd_gpu import _pangu_spec

    if kind.startswith("swin"):
        spec, rows = _spec(32, 64, 4, 24, kind.endswith("shifted"))
        qkv, bias, table = _inputs(2, 32, 64, 4, 24, rows, seed=5)
    else:
        spec, tshape = _pangu_spec(24, 48, 4, 32, kind.endswith("rolled"))
        g = torch.Generator().manual_seed(6)
        qkv = torch.randn(2, 24 * 48, 3 * 4 * 32, generator=g).cuda()
        bias = (0.3 * torch.randn(3 * 4 * 32, generator=g)).cuda()
        table = (0.5 * torch.randn(*tshape, generator=g)).cuda()
    assert ops.window_attention_io_supported(spec, 2)
    q16 = qkv.to(torch.bfloat16)
    b16 = bias.to(torch.bfloat16)
    want = ops.window_attention(q16.float(), b16.float(), table, spec, precision="bf16")
    got = ops.window_attention(q16, bias, table, spec, precision="bf16")
    assert got.dtype == torch.bfloat16
    assert torch.equal(got, want.to(torch.bfloat16))


def _sub_window_spec(h, w, wh, ww, heads, d, shifted):
    """Swin block with windows smaller than the map (nwin > 1), shift = half a window (swin_transformer.py:217-251)."""
    from dlwp_benchmark_amd import ops

    sh, sw = (wh // 2, ww // 2) if shifted else (0, 0)
    return ops.WindowSpec(grid=(1, h, w), padded=(1, h, w), pad_lead=(0, 0, 0), window=(1, wh, ww), shift_fwd=(0, sh, sw),
                          shift_back=(0, sh, sw), use_mask=shifted, mask_b1=(ops.BIG, h - wh, w - ww),
                          mask_b2=(ops.BIG, h - wh // 2, w - ww // 2), bias_mode=0, heads=heads, head_dim=d,
                          scale=d ** -0.5), (2 * wh - 1) * (2 * ww - 1)


@pytest.mark.parametrize("shifted", [False, True])
@pytest.mark.parametrize("h,w,wh,ww,heads,d", [(32, 64, 32, 64, 4, 24),     # C3 stage 0: one window, 64 queries per wave
                                               (16, 32, 16, 32, 8, 48),     # C3 stage 1 head_dim
                                               (16, 64, 8, 32, 2, 8),       # 4 windows, 256 keys each
                                               (32, 64, 8, 32, 2, 16),      # 8 windows
                                               (24, 96, 12, 48, 2, 24),     # window width 48: 256 % 48 != 0 (token-map carry)
                                               (16, 160, 8, 80, 3, 8)])     # window width 80
def test_direct_gather_form_is_bit_identical_to_the_image_form(h, w, wh, ww, heads, d, shifted):
    """dlwp_window_attn_bf16_io on 2-D descriptors gathers Q / K / V from the bfloat16 qkv tensor inside the attention kernel
    (round 3: no prep kernel, no operand images); dlwp_window_attn_bf16 on the same bf16-representable values still runs prep
    kernel + images.  Same arithmetic in the same order: the outputs must agree to the last bit, for every instantiated head
    dimension, for one and for several windows per map, power-of-two and other window widths, with and without the shift."""
    from dlwp_benchmark_amd import ops

    spec, rows = _sub_window_spec(h, w, wh, ww, heads, d, shifted)
    if shifted and (ww // 2) % 16:
        pytest.skip("longitude region boundary inside a 16-key block: generic kernel, no bfloat16-tensor form")
    qkv, bias, table = _inputs(3, h, w, heads, d, rows, seed=h + ww + d, qk_gain=1.5)
    assert ops.window_attention_io_supported(spec, 3)
    q16 = qkv.to(torch.bfloat16)
    want = ops.window_attention(q16.float(), bias, table, spec, precision="bf16")
    got = ops.window_attention(q16, bias, table, spec, precision="bf16")
    assert got.dtype == torch.bfloat16 and torch.isfinite(got.float()).all()
    assert torch.equal(got, want.to(torch.bfloat16))
    ref = ops.window_attention(q16.float(), bias, table, spec, precision="fp32_mfma")
    assert rel_l2(got.float(), ref) <= 5e-2     # sanity only (bf16 operands AND a bf16 output; head_dim 8 with 1.5 x logits: 3e-2)


@pytest.mark.parametrize("shifted", [False, True])
def test_direct_gather_form_exponent_slack_fallback(shifted):
    """The exact-maximum fallback of the direct form (scores that leave the 2^+-100 slack): still bit-identical to the image form."""
    from dlwp_benchmark_amd import ops

    spec, rows = _spec(16, 32, 2, 8, shifted)
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(1, 512, 3, 2, 8, generator=g)
    ramp = torch.linspace(0.0, 60.0, 512).view(1, 512, 1, 1)
    qkv[:, :, 1] = qkv[:, :, 1] * 0.1 + ramp          # keys grow along the window order: later tiles dwarf the first one
    qkv[:, :, 0] = qkv[:, :, 0].abs() * 8.0 + 8.0
    qkv = qkv.reshape(1, 512, 48).cuda()
    bias = torch.zeros(48).cuda()
    table = (torch.randn(rows, 2, generator=g) * 0.5).cuda()
    q16 = qkv.to(torch.bfloat16)
    want, fb = ops.window_attention(q16.float(), bias, table, spec, precision="bf16", count_fallbacks=True)
    assert fb > 0, "the inputs were meant to leave the exponent slack"
    got = ops.window_attention(q16, bias, table, spec, precision="bf16")
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want.to(torch.bfloat16))
