"""Op-level parity of the U-Net / ModernUNet kernels of csrc/conv2.hip and the fused forms of the 3x3 convolution
against the torch operators the reference modules call (models/unet/unet.py), evaluated in float64."""
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("n,c,h,w,groups,act", [(3, 16, 8, 8, 8, "gelu"), (2, 64, 2, 2, 1, "gelu"), (2, 24, 5, 7, 3, "none"),
                                                (1, 128, 32, 64, 8, "gelu")])
def test_groupnorm_act(n, c, h, w, groups, act):
    from dlwp_benchmark_amd import ops

    x = _rand(n, c, h, w, seed=1) * 3.0 + 0.7
    gamma, beta = _rand(c, seed=2), _rand(c, seed=3)
    got = ops.groupnorm_act(x, gamma, beta, groups, 1e-5, ops.ACTS[act])
    want = F.group_norm(x.double(), groups, gamma.double(), beta.double(), 1e-5)
    if act == "gelu":
        want = F.gelu(want)
    assert rel_l2(got, want) <= 1e-6


@pytest.mark.parametrize("n,cin,cout,h,w,k,s,p", [(2, 8, 8, 16, 32, 3, 2, 1), (2, 5, 7, 9, 11, 3, 2, 1), (3, 16, 4, 8, 8, 1, 1, 0),
                                                  (1, 64, 64, 4, 8, 3, 2, 1), (2, 12, 3, 64, 64, 1, 1, 0)])
def test_conv2d(n, cin, cout, h, w, k, s, p):
    from dlwp_benchmark_amd import ops

    x, wt, b = _rand(n, cin, h, w, seed=4), _rand(cout, cin, k, k, seed=5, scale=0.3), _rand(cout, seed=6)
    got = ops.conv2d(x, wt, b, stride=s, padding=p)
    want = F.conv2d(x.double(), wt.double(), b.double(), stride=s, padding=p)
    assert got.shape == want.shape and rel_l2(got, want) <= 1e-6
    resid = _rand(*want.shape, seed=7)
    got2 = ops.conv2d(x, wt, None, stride=s, padding=p, pre_act=ops.ACTS["gelu"], act=ops.ACTS["gelu"], resid=resid)
    want2 = F.gelu(F.conv2d(F.gelu(x.double()), wt.double(), None, stride=s, padding=p) + resid.double())
    assert rel_l2(got2, want2) <= 1e-6


@pytest.mark.parametrize("n,cin,cout,h,w,k,s,p", [(2, 16, 8, 8, 16, 2, 2, 0), (2, 6, 6, 4, 4, 4, 2, 1), (1, 32, 16, 2, 2, 4, 2, 1),
                                                  (3, 5, 7, 3, 5, 2, 2, 0)])
def test_conv_transpose2d(n, cin, cout, h, w, k, s, p):
    from dlwp_benchmark_amd import ops

    x, wt, b = _rand(n, cin, h, w, seed=8), _rand(cin, cout, k, k, seed=9, scale=0.3), _rand(cout, seed=10)
    got = ops.conv_transpose2d(x, wt, b, stride=s, padding=p)
    want = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=s, padding=p)
    assert got.shape == want.shape and rel_l2(got, want) <= 1e-6


def test_avgpool2x2():
    from dlwp_benchmark_amd import ops

    x = _rand(3, 5, 8, 12, seed=11)
    assert rel_l2(ops.avgpool2x2(x), F.avg_pool2d(x.double(), 2)) <= 1e-6


@pytest.mark.parametrize("hpx", [False, True])
def test_conv3x3_pre_activation_and_residual(hpx):
    """the fused forms the residual block uses (unet.py:884-901): act on the input, shortcut added in the epilogue"""
    from dlwp_benchmark_amd import ops

    n, c0, cout, h, w = (12, 6, 10, 8, 8) if hpx else (2, 6, 10, 16, 32)
    x, wt, b = _rand(n, c0, h, w, seed=12), _rand(cout, c0, 3, 3, seed=13, scale=0.3), _rand(cout, seed=14)
    resid = _rand(n, cout, h, w, seed=15)
    got = ops.conv3x3(x, wt, b, act=ops.ACTS["gelu"], pre_act=ops.ACTS["gelu"], resid=resid, hpx=hpx)
    plain = ops.conv3x3(torch.nn.functional.gelu(x.double()).float(), wt, b, act=0, hpx=hpx)   # same kernel, un-fused
    want = F.gelu(plain.double() + resid.double())
    assert rel_l2(got, want) <= 2e-6
    if not hpx:
        pad = F.pad(F.pad(F.gelu(x.double()), (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1))
        want2 = F.gelu(F.conv2d(pad, wt.double(), b.double()) + resid.double())
        assert rel_l2(got, want2) <= 1e-6
