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
