"""GPU parity of the fused token MLP (dlwp_token_mlp_f32) with the ops it replaces in the AFNO block
(reference fourcastnet.py:41-57 `Mlp` and the second skip of :191-192): fc1 -> exact GELU -> fc2 -> + residual.

Checker: the same expression in float64 torch on the device.  Tolerance 2e-6 relative L2 / 1e-5 of the output scale per
element: the kernel evaluates both GEMMs as six-term exact bf16 splits with fp32 accumulation, i.e. at fp32-GEMM
accuracy, and the model-level fixtures from the real reference (test_backbones_gpu.py, test_fullsize_gpu.py) run
through it at the 1e-5 per-step bound."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


def _reference(n, resid, w1, b1, w2, b2):
    h = torch.nn.functional.gelu(torch.nn.functional.linear(n.double(), w1.double(), b1.double()))
    y = torch.nn.functional.linear(h, w2.double(), b2.double() if b2 is not None else None)
    return y + resid.double() if resid is not None else y


@pytest.mark.parametrize("tokens,hidden,use_resid,use_b2", [(4096, 256, True, True), (1000, 256, True, False),
                                                            (33, 64, False, True), (32 * 700 + 5, 192, True, True)])
def test_token_mlp_matches_fp64(tokens, hidden, use_resid, use_b2):
    from dlwp_benchmark_amd import ops

    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(1234 + tokens)
    c = 64
    n = torch.randn(tokens, c, generator=gen).to(dev)
    resid = (3.0 * torch.randn(tokens, c, generator=gen)).to(dev) if use_resid else None
    w1 = (torch.randn(hidden, c, generator=gen) / c ** 0.5).to(dev)
    b1 = (0.3 * torch.randn(hidden, generator=gen)).to(dev)
    w2 = (torch.randn(c, hidden, generator=gen) / hidden ** 0.5).to(dev)
    b2 = (0.3 * torch.randn(c, generator=gen)).to(dev) if use_b2 else None
    assert ops.token_mlp_supported(c, hidden)
    packed = ops.TokenMlpWeights()
    got = ops.token_mlp(n, resid, packed.get(w1, w2), b1, b2, hidden)
    want = _reference(n, resid, w1, b1, w2, b2)
    assert rel_l2(got.double().cpu(), want.cpu()) < 2e-6
    assert (got.double() - want).abs().max().item() < 1e-5 * want.abs().max().item()
    if resid is not None:   # in place on the residual, as the AFNO block calls it
        r2 = resid.clone()
        out = ops.token_mlp(n, r2, packed.get(w1, w2), b1, b2, hidden, out=r2)
        assert out.data_ptr() == r2.data_ptr()
        assert torch.equal(out, got)


@pytest.mark.parametrize("tokens", [4096, 777])
def test_token_mlp_fused_layernorm_matches_fp64(tokens):
    """LayerNorm(eps 1e-6) -> fc1 -> GELU -> fc2 -> + residual, in place on the residual, as the AFNO block calls it
    (fourcastnet.py:191-192)."""
    from dlwp_benchmark_amd import ops

    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(4321 + tokens)
    c, hidden, eps = 64, 256, 1e-6
    s = (2.0 * torch.randn(tokens, c, generator=gen) + 0.5).to(dev)
    gamma = (1.0 + 0.2 * torch.randn(c, generator=gen)).to(dev)
    beta = (0.2 * torch.randn(c, generator=gen)).to(dev)
    w1 = (torch.randn(hidden, c, generator=gen) / c ** 0.5).to(dev)
    b1 = (0.3 * torch.randn(hidden, generator=gen)).to(dev)
    w2 = (torch.randn(c, hidden, generator=gen) / hidden ** 0.5).to(dev)
    b2 = (0.3 * torch.randn(c, generator=gen)).to(dev)
    n64 = torch.nn.functional.layer_norm(s.double(), (c,), gamma.double(), beta.double(), eps)
    want = _reference(n64, s, w1, b1, w2, b2)
    packed = ops.TokenMlpWeights().get(w1, w2, gamma, beta, b1)
    buf = s.clone()
    got = ops.token_mlp(buf, buf, packed, None, b2, hidden, out=buf, ln_eps=eps)
    assert got.data_ptr() == buf.data_ptr()
    assert rel_l2(got, want) < 2e-6


def test_token_mlp_emits_next_layernorm_channels_first():
    """dlwp_token_mlp_emit_norm_f32: the second output must equal LayerNorm(out) moved to [B, C, H, W]
    (what dlwp_layernorm_nhwc_to_nchw_f32 computes from `out` in the next AFNO block, fourcastnet.py:182)."""
    from dlwp_benchmark_amd import ops

    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(77)
    b, h, w, c, hidden, eps = 3, 8, 20, 64, 256, 1e-6
    s = (2.0 * torch.randn(b, h, w, c, generator=gen) + 0.5).to(dev)
    gamma2 = (1.0 + 0.2 * torch.randn(c, generator=gen)).to(dev)
    beta2 = (0.2 * torch.randn(c, generator=gen)).to(dev)
    gamma1 = (1.0 + 0.3 * torch.randn(c, generator=gen)).to(dev)
    beta1 = (0.3 * torch.randn(c, generator=gen)).to(dev)
    w1 = (torch.randn(hidden, c, generator=gen) / c ** 0.5).to(dev)
    b1 = (0.3 * torch.randn(hidden, generator=gen)).to(dev)
    w2 = (torch.randn(c, hidden, generator=gen) / hidden ** 0.5).to(dev)
    b2 = (0.3 * torch.randn(c, generator=gen)).to(dev)
    packed = ops.TokenMlpWeights().get(w1, w2, gamma2, beta2, b1)
    plain = ops.token_mlp(s.clone(), s, packed, None, b2, hidden, ln_eps=eps)
    buf = s.clone()
    out, nxt = ops.token_mlp(buf, buf, packed, None, b2, hidden, out=buf, ln_eps=eps, emit_norm=(gamma1, beta1, eps))
    assert torch.equal(out, plain)
    want = torch.nn.functional.layer_norm(out.double(), (c,), gamma1.double(), beta1.double(), eps).permute(0, 3, 1, 2)
    assert nxt.shape == (b, c, h, w) and nxt.is_contiguous()
    assert rel_l2(nxt, want) < 1e-6
    ref_kernel = ops.layernorm_nhwc_to_nchw(out, gamma1, beta1, eps)
    assert rel_l2(nxt, ref_kernel) < 1e-6


@pytest.mark.parametrize("form", ["bf16x6", "f16x3"])
@pytest.mark.parametrize("emit", [False, True])
def test_afno_block_tail_matches_fp64(emit, form):
    """dlwp_afno_block_tail_f32 / _f16x3 = `+ bias`, first skip, norm2, Mlp, second skip (+ the next block's norm1) of the
    AFNO block (fourcastnet.py:127, :187, :191-192, :182) against the same expression in float64 -- the same bound for the
    three-part bf16 and the two-part f16 product form."""
    from dlwp_benchmark_amd import ops

    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(31)
    b, h, w, c, hidden, eps = 2, 8, 12, 64, 256, 1e-6
    f_cf = torch.randn(b, c, h, w, generator=gen).to(dev)
    l_cf = torch.randn(b, c, h, w, generator=gen).to(dev)
    x = (1.5 * torch.randn(b, h, w, c, generator=gen)).to(dev)
    g2 = (1.0 + 0.2 * torch.randn(c, generator=gen)).to(dev)
    be2 = (0.2 * torch.randn(c, generator=gen)).to(dev)
    g1 = (1.0 + 0.3 * torch.randn(c, generator=gen)).to(dev)
    be1 = (0.3 * torch.randn(c, generator=gen)).to(dev)
    w1 = (torch.randn(hidden, c, generator=gen) / c ** 0.5).to(dev)
    b1 = (0.3 * torch.randn(hidden, generator=gen)).to(dev)
    w2 = (torch.randn(c, hidden, generator=gen) / hidden ** 0.5).to(dev)
    b2 = (0.3 * torch.randn(c, generator=gen)).to(dev)
    s64 = (f_cf + l_cf).double().permute(0, 2, 3, 1) + x.double()
    n64 = torch.nn.functional.layer_norm(s64, (c,), g2.double(), be2.double(), eps)
    want = _reference(n64, s64, w1, b1, w2, b2)
    packed = ops.TokenMlpWeights().get(w1, w2, g2, be2, b1, merged=True, f16x3=form == "f16x3")
    xin = x.clone()
    res = ops.afno_block_tail(f_cf, l_cf, xin, packed, b2, hidden, eps, emit_norm=(g1, be1, eps) if emit else None, out=xin,
                              form=form)
    out = res[0] if emit else res
    assert out.data_ptr() == xin.data_ptr()
    assert rel_l2(out, want) < 2e-6
    if emit:
        want_n = torch.nn.functional.layer_norm(want, (c,), g1.double(), be1.double(), eps).permute(0, 3, 1, 2)
        assert rel_l2(res[1], want_n) < 2e-6


def test_token_mlp_repacks_after_weight_update():
    from dlwp_benchmark_amd import ops

    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(7)
    n = torch.randn(256, 64, generator=gen).to(dev)
    w1 = torch.nn.Parameter((torch.randn(256, 64, generator=gen) / 8).to(dev))
    w2 = torch.nn.Parameter((torch.randn(64, 256, generator=gen) / 16).to(dev))
    b1 = torch.zeros(256, device=dev)
    packed = ops.TokenMlpWeights()
    a = ops.token_mlp(n, None, packed.get(w1, w2), b1, None, 256)
    with torch.no_grad():
        w2.mul_(2.0)
    b = ops.token_mlp(n, None, packed.get(w1, w2), b1, None, 256)
    assert rel_l2(b.cpu(), (2.0 * a).cpu()) < 1e-6


def test_token_mlp_rejects_unsupported_width():
    from dlwp_benchmark_amd import lib as _lib
    from dlwp_benchmark_amd import ops

    assert not ops.token_mlp_supported(96, 384)
    assert not ops.token_mlp_supported(64, 320)   # weights would not fit LDS
    assert not ops.token_mlp_supported(64, 96)
    with pytest.raises(_lib.DlwpError):
        ops.TokenMlpWeights().get(torch.zeros(384, 96, device="cuda:0"), torch.zeros(96, 384, device="cuda:0"))


@pytest.mark.parametrize("b,h,w,c,cout,bias", [(2, 32, 64, 64, 3, False), (3, 8, 24, 64, 3, True), (1, 16, 20, 128, 5, True),
                                               (2, 8, 8, 256, 13, False), (1, 4, 6, 16, 1, False)])
def test_patch_recover_1x1_matches_linear_and_rearrange(b, h, w, c, cout, bias):
    """dlwp_patch_recover_1x1_f32 = the head Linear of a 1x1-patch backbone + "b h w c -> b c h w" (fourcastnet.py:144, :296-303)
    in one pass, against float64 (token counts that are not multiples of 64, a wave tile that straddles two samples)."""
    import torch.nn.functional as F

    from dlwp_benchmark_amd import ops

    g = torch.Generator().manual_seed(b * 100 + c + cout)
    x = torch.randn(b, h, w, c, generator=g).cuda()
    wt = (torch.randn(cout, c, generator=g) / c ** 0.5).cuda()
    bs = torch.randn(cout, generator=g).cuda() if bias else None
    assert ops.patch_recover_1x1_supported(c, cout)
    got = ops.patch_recover_1x1(x, wt, bs, h, w)
    want = F.linear(x.double(), wt.double(), bs.double() if bias else None).permute(0, 3, 1, 2)
    assert got.shape == (b, cout, h, w) and got.is_contiguous()
    err = float((got.double() - want).norm() / want.norm())
    assert err <= 3e-7, err
