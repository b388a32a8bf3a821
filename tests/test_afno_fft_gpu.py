"""GPU parity of the hipFFT-backed AFNO filter (dlwp_rfft2_f32 -> dlwp_afno2d_mix_scaled_f32 in place ->
dlwp_irfft2_f32) with the same filter through torch.fft and the unscaled mixing kernel, i.e. the arithmetic of
reference fourcastnet.py:87-123 (`rfft2(norm="ortho")`, block MLP, softshrink, `irfft2(norm="ortho")`).
The mixing kernel itself is pinned against the real reference class by the model fixtures (test_backbones_gpu.py,
test_fullsize_gpu.py), which run through this path.  Tolerance: 2e-6 relative L2 (two fp32 FFT libraries' rounding)."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,nb,frac", [((2, 64, 128, 256), 4, 1.0), ((3, 16, 32, 64), 4, 0.5), ((1, 32, 20, 30), 8, 1.0)])
def test_filter_matches_torch_fft_path(shape, nb, frac):
    from dlwp_benchmark_amd import ops

    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(99)
    b, c, h, w = shape
    bs = c // nb
    x = torch.randn(*shape, generator=gen).to(dev)
    w1 = (0.3 * torch.randn(2, nb, bs, bs, generator=gen)).to(dev)
    b1 = (0.3 * torch.randn(2, nb, bs, generator=gen)).to(dev)
    w2 = (0.3 * torch.randn(2, nb, bs, bs, generator=gen)).to(dev)
    b2 = (0.3 * torch.randn(2, nb, bs, generator=gen)).to(dev)
    lam = 0.01
    got = ops.afno2d_filter_cf(x, w1, b1, w2, b2, nb, lam, frac)
    xf = torch.fft.rfft2(x, norm="ortho")
    want = torch.fft.irfft2(ops.afno2d_mix(xf, w1, b1, w2, b2, nb, lam, frac), s=(h, w), norm="ortho")
    assert got.shape == want.shape
    assert rel_l2(got, want) < 2e-6
    # the input must survive (only the internal spectrum buffer is scratch for the C2R transform)
    again = ops.afno2d_filter_cf(x, w1, b1, w2, b2, nb, lam, frac)
    assert torch.equal(got, again)


@pytest.mark.parametrize("b,cin,h,w,c,use_pos", [(2, 8, 16, 32, 64, True), (3, 18, 5, 7, 32, False), (1, 1, 8, 8, 16, True)])
def test_patch_embed_1x1_matches_conv(b, cin, h, w, c, use_pos):
    """dlwp_patch_embed_1x1_f32 vs Conv2d(1x1) -> flatten(2).transpose(1, 2) -> + pos_embed (fourcastnet.py:530-543)."""
    from dlwp_benchmark_amd import ops

    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(b, cin, h, w, generator=gen).to(dev)
    wt = torch.randn(c, cin, 1, 1, generator=gen).to(dev)
    bias = torch.randn(c, generator=gen).to(dev)
    pos = torch.randn(h * w, c, generator=gen).to(dev) if use_pos else None
    got = ops.patch_embed_1x1(x, wt, bias, pos)
    want = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double()).flatten(2).transpose(1, 2)
    if use_pos:
        want = want + pos.double()
    assert got.shape == want.shape and got.is_contiguous()
    assert rel_l2(got, want) < 1e-6
