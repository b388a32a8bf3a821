"""GPU parity of the hipFFT-backed AFNO filter (dlwp_rfft2_f32 -> dlwp_afno2d_mix_scaled_f32 in place ->
dlwp_irfft2_f32) with the same filter through torch.fft and the unscaled mixing kernel, i.e. the arithmetic of
reference fourcastnet.py:87-123 (`rfft2(norm="ortho")`, block MLP, softshrink, `irfft2(norm="ortho")`).
The mixing kernel itself is pinned against the real reference class by the model fixtures (test_backbones_gpu.py,
test_fullsize_gpu.py), which run through this path.  Tolerance: 2e-6 relative L2 (two fp32 FFT libraries' rounding)."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,nb,frac", [((2, 64, 128, 256), 4, 1.0), ((3, 16, 32, 64), 4, 0.5), ((1, 32, 20, 30), 8, 1.0),
                                           ((2, 16, 64, 64), 4, 1.0), ((2, 16, 32, 32), 2, 0.5), ((1, 8, 64, 128), 2, 0.7)])
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
    # hand-written kept-column transforms (default where the grid is instantiated) vs the hipFFT path on the full half spectrum
    lib = __import__("dlwp_benchmark_amd.lib", fromlist=["load"]).load()
    if lib.dlwp_afno_fft_supported(h, w, ops.afno_kept_cols(h, w, frac)):
        via_rocfft = ops.afno2d_filter_cf(x, w1, b1, w2, b2, nb, lam, frac, use_rocfft=True)
        assert rel_l2(got, via_rocfft) < 2e-6
        assert not torch.equal(got, via_rocfft), "both paths bit-identical: is the hand-written transform running?"


@pytest.mark.parametrize("h,w,kc", [(128, 256, 65), (128, 256, 20), (64, 128, 33), (32, 64, 17), (32, 64, 8), (64, 64, 33),
                                    (32, 32, 17), (32, 32, 5)])
def test_kept_column_transforms_match_torch_fft(h, w, kc):
    """dlwp_afno_rfft2_kept_f32 = the first kc columns of torch.fft.rfft2 (unnormalised);
    dlwp_afno_irfft2_kept_f32 = H * W * torch.fft.irfft2 of the spectrum zero-padded to W/2+1 columns."""
    import ctypes

    from dlwp_benchmark_amd import lib as L

    lib = L.load()
    assert lib.dlwp_afno_fft_supported(h, w, kc)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(h * 1000 + w + kc)
    planes = 5
    x = torch.randn(planes, h, w, generator=gen).to(dev)
    plan = ctypes.c_void_p()
    L.check(lib.dlwp_afno_fft_plan_create(ctypes.byref(plan), h, w, kc, L.stream_ptr()), "plan")
    try:
        spec = torch.empty(planes, h, kc, 2, device=dev)
        L.check(lib.dlwp_afno_rfft2_kept_f32(plan, x.data_ptr(), spec.data_ptr(), planes, L.stream_ptr()), "rfft2")
        want = torch.view_as_real(torch.fft.rfft2(x.double())[:, :, :kc])
        assert rel_l2(spec, want) < 5e-7
        # inverse on an arbitrary complex spectrum (not Hermitian-consistent: the DC column's imaginary part must be ignored)
        g = torch.randn(planes, h, kc, 2, generator=gen).to(dev)
        y = torch.empty(planes, h, w, device=dev)
        L.check(lib.dlwp_afno_irfft2_kept_f32(plan, g.data_ptr(), y.data_ptr(), planes, L.stream_ptr()), "irfft2")
        full = torch.zeros(planes, h, w // 2 + 1, dtype=torch.complex128, device=dev)
        full[:, :, :kc] = torch.view_as_complex(g.double())
        want_y = torch.fft.irfft2(full, s=(h, w)) * (h * w)
        assert rel_l2(y, want_y) < 5e-7
    finally:
        lib.dlwp_afno_fft_plan_destroy(plan)


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
