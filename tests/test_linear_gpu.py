"""Op-level parity of dlwp_linear_f32 (csrc/linear.hip: fp32 Linear on the bf16 matrix pipe with bias / GELU / residual fused)
against torch.nn.functional.linear evaluated in float64 -- the operator the Swin / Pangu blocks call for qkv, proj, fc1 and
fc2 (swin_transformer.py:21-39, :107-120; panguweather.py:176-211).  Tolerance: fp32 GEMM accuracy (1e-6 relative L2; a plain
fp32 rocBLAS GEMM lands at 2e-7 .. 5e-7 on the same inputs)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _linear(k, n, bias, seed):
    torch.manual_seed(seed)
    m = torch.nn.Linear(k, n, bias=bias)
    with torch.no_grad():
        m.weight.mul_(3.0)
    return m.to(DEV)


# rows: a tile tail (M % 128 != 0), fewer rows than one tile, many tiles (all 8 XCD slots + tail group);
# widths: the C3 / C5 widths, the BN = 64 path (192, 576), N tail inside a tile (N = 100), K = 32 (one k-step)
@pytest.mark.parametrize("rows,k,n,bias,act,resid", [
    (1000, 96, 288, True, 0, False), (77, 96, 96, True, 0, True), (4096, 96, 384, True, 1, False),
    (4096, 384, 96, True, 0, True), (2051, 192, 576, True, 0, False), (513, 768, 192, True, 0, True),
    (300, 32, 100, False, 1, True), (129, 64, 4, True, 0, False), (9000, 384, 1536, True, 1, False),
    (1, 1536, 384, False, 0, True)])
def test_linear_matches_float64(rows, k, n, bias, act, resid):
    from dlwp_benchmark_amd import ops

    m = _linear(k, n, bias, seed=rows + k)
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, k, generator=g) * 2.0 + 0.3).to(DEV)
    r = torch.randn(rows, n, generator=g).to(DEV) if resid else None
    with torch.no_grad():
        got = ops.linear(x, m, act=act, resid=r)
        want = F.linear(x.double(), m.weight.double(), m.bias.double() if bias else None)
        if act:
            want = F.gelu(want)
        if resid:
            want = want + r.double()
    assert got.shape == want.shape
    assert rel_l2(got, want) <= 1e-6
    assert (got.double() - want).abs().max() <= 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("rows,k,n,bias,act,resid,wgain", [
    (1000, 96, 288, True, 0, False, 3.0), (77, 96, 96, True, 0, True, 3.0), (4096, 96, 384, True, 1, False, 3.0),
    (4096, 384, 96, True, 0, True, 1e-3), (2051, 192, 576, True, 0, False, 3.0), (513, 768, 192, True, 0, True, 3.0),
    (300, 32, 100, False, 1, True, 3.0), (9000, 384, 1536, True, 1, False, 30.0), (1, 1536, 384, False, 0, True, 3.0)])
def test_linear_f16x3_matches_float64(rows, k, n, bias, act, resid, wgain):
    """dlwp_linear_f16x3 (precision "f16x3"): the same operator from exact two-part f16 splits (three products).  Same
    fp32-GEMM bound as dlwp_linear_f32 against float64, at weight magnitudes three orders apart (the scaled weight residual
    keeps 22 bits at any of them); inputs are O(1) like the LayerNorm / GELU / attention outputs the blocks feed it."""
    from dlwp_benchmark_amd import ops

    torch.manual_seed(rows + k)
    m = torch.nn.Linear(k, n, bias=bias)
    with torch.no_grad():
        m.weight.mul_(wgain)
    m = m.to(DEV)
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, k, generator=g) * 2.0 + 0.3).to(DEV)
    r = torch.randn(rows, n, generator=g).to(DEV) if resid else None
    with torch.no_grad():
        got = ops.linear(x, m, act=act, resid=r, precision="f16x3")
        ref32 = ops.linear(x, m, act=act, resid=r)
        want = F.linear(x.double(), m.weight.double(), m.bias.double() if bias else None)
        if act:
            want = F.gelu(want)
        if resid:
            want = want + r.double()
    assert got.shape == want.shape
    assert rel_l2(got, want) <= 1e-6, (rel_l2(got, want), rel_l2(ref32, want))
    assert (got.double() - want).abs().max() <= 2e-5 * max(1.0, want.abs().max().item())


def test_linear_f16x3_weight_cache_follows_the_parameter():
    from dlwp_benchmark_amd import ops

    m = _linear(96, 96, True, seed=5)
    x = torch.randn(256, 96, device=DEV)
    with torch.no_grad():
        a = ops.linear(x, m, precision="f16x3")
        m.weight.mul_(2.0)
        b = ops.linear(x, m, precision="f16x3")
        want = F.linear(x.double(), m.weight.double(), m.bias.double())
    assert rel_l2(b, want) <= 1e-6 and rel_l2(a, want) > 1e-2


@pytest.mark.parametrize("rows,k,n,act,resid", [(1000, 96, 288, 0, False), (4096, 384, 96, 0, True), (2051, 192, 768, 1, False),
                                                 (300, 32, 100, 1, True)])
def test_linear_bf16_matches_bf16_rounded_operands(rows, k, n, act, resid):
    """dlwp_linear_bf16: operands rounded to bf16 (round to nearest even, like torch's .bfloat16()), products and sums in
    fp32 -- against the same rounding done in torch and the GEMM in float64."""
    from dlwp_benchmark_amd import ops

    m = _linear(k, n, True, seed=rows + k)
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, k, generator=g) * 2.0 + 0.3).to(DEV)
    r = torch.randn(rows, n, generator=g).to(DEV) if resid else None
    with torch.no_grad():
        got = ops.linear(x, m, act=act, resid=r, precision="bf16")
        want = F.linear(x.bfloat16().double(), m.weight.bfloat16().double(), m.bias.double())
        if act:
            want = F.gelu(want)
        if resid:
            want = want + r.double()
        full = F.linear(x.double(), m.weight.double(), m.bias.double())
    assert rel_l2(got, want) <= 1e-6
    assert 1e-4 < rel_l2(F.linear(x.bfloat16().double(), m.weight.bfloat16().double(), m.bias.double()), full) < 1e-2   # it IS bf16


def test_linear_in_place_residual_and_leading_dims():
    from dlwp_benchmark_amd import ops

    m = _linear(96, 96, True, seed=5)
    x = torch.randn(2, 150, 96, device=DEV)
    acc = torch.randn(2, 150, 96, device=DEV)
    with torch.no_grad():
        want = F.linear(x.double(), m.weight.double(), m.bias.double()) + acc.double()
        out = ops.linear(x, m, resid=acc, out=acc)
    assert out.data_ptr() == acc.data_ptr() and out.shape == (2, 150, 96)
    assert rel_l2(out, want) <= 1e-6


def test_linear_repacks_after_weight_update():
    from dlwp_benchmark_amd import ops

    m = _linear(64, 64, True, seed=6)
    x = torch.randn(200, 64, device=DEV)
    with torch.no_grad():
        first = ops.linear(x, m).clone()
        m.weight.mul_(-2.0)                   # in-place write bumps the version: the packed images must follow
        second = ops.linear(x, m)
        want = F.linear(x.double(), m.weight.double(), m.bias.double())
    assert rel_l2(second, want) <= 1e-6
    assert not torch.allclose(first, second)


def test_linear_rejects_unsupported_shapes_loudly():
    from dlwp_benchmark_amd import lib, ops

    assert not ops.linear_supported(48, 96) and not ops.linear_supported(64, 6)
    m = _linear(48, 96, True, seed=7)
    with torch.no_grad(), pytest.raises(lib.DlwpError):
        ops.linear(torch.randn(10, 48, device=DEV), m)
    m = _linear(64, 64, True, seed=8)
    with torch.no_grad(), pytest.raises(lib.DlwpError):
        ops.linear(torch.randn(10, 32, device=DEV), m)


def test_linear_is_differentiable_through_torch_ops():
    from dlwp_benchmark_amd import ops

    m = _linear(64, 128, True, seed=9)
    x = torch.randn(50, 64, device=DEV, requires_grad=True)
    y = ops.linear(x, m, act=1)
    y.square().sum().backward()
    ref = F.gelu(F.linear(x.detach().double(), m.weight.double(), m.bias.double()))
    assert rel_l2(y.detach(), ref) <= 1e-6 and x.grad is not None and m.weight.grad is not None


@pytest.mark.parametrize("tag", ["swin_e32_32x64", "swin_c3_full", "pangu_c5_full"])
def test_block_linear_forms_hold_the_reference_bound(tag):
    """The two forms of the blocks' Linears (dlwp_linear_f32 / rocBLAS) against the REAL reference's trajectory
    (tests/golden/model_*.npz): both must hold the 1e-5 per-step bound, and agree with each other to fp32 rounding."""
    import json

    import dlwp_benchmark_amd.models as M
    from dlwp_benchmark_amd.weights import fill_by_spec
    from helpers import load_golden, per_step_rel_l2
    from oracle.make_golden import MODEL_CASES, model_inputs

    family, cfg, (batch, frames), gain = MODEL_CASES[tag]
    name = {"swin": "SwinTransformer", "pangu": "PanguWeather"}[family]
    g = load_golden(f"model_{tag}")
    sd, _ = fill_by_spec(json.loads(str(g["param_spec"])), gain=gain)
    model = getattr(M, name)(**cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to(DEV).eval()
    constants, prescribed, prognostic = model_inputs(tag, cfg, batch, frames)
    dev = lambda t: t.to(DEV) if t is not None else None
    want = torch.from_numpy(g["y"])
    outs = {}
    for form in ("bf16x6", "rocblas"):
        model.set_linear_form(form)
        outs[form] = model(constants=dev(constants), prescribed=dev(prescribed), prognostic=dev(prognostic)).clone()
        torch.cuda.synchronize()
        errs = per_step_rel_l2(outs[form], want)
        assert max(errs) <= 1e-5, f"{tag} {form}: per-step rel L2 {['%.2e' % e for e in errs]}"
    assert rel_l2(outs["bf16x6"], outs["rocblas"]) <= 5e-6
    # bf16 Linear operands: the bound the bf16 attention line uses (tests/test_backbones_gpu.py)
    model.set_linear_form("bf16")
    got = model(constants=dev(constants), prescribed=dev(prescribed), prognostic=dev(prognostic))
    assert max(per_step_rel_l2(got, want)) <= 5e-3


@pytest.mark.parametrize("kind,cin,cout,k", [("convT", 192, 96, 2), ("convT", 96, 48, 1), ("convT", 64, 32, 4), ("conv", 96, 3, 1),
                                             ("conv", 64, 8, 1)])
def test_conv_as_linear_matches_torch_convolution(kind, cin, cout, k):
    """ops.ConvAsLinear (the Swin decoder's kernel = stride transposed convolutions and its 1x1 head as Linears over
    token-major data, swin_transformer.py:600-612) against torch's convolution in float64, GELU fused."""
    from dlwp_benchmark_amd import ops

    torch.manual_seed(cin + cout + k)
    conv = (torch.nn.ConvTranspose2d(cin, cout, k, k) if kind == "convT" else torch.nn.Conv2d(cin, cout, 1)).to(DEV)
    b, h, w = 2, 6, 10
    x = torch.randn(b, cin, h, w, device=DEV)
    lin = ops.ConvAsLinear(conv)
    with torch.no_grad():
        tok = x.permute(0, 2, 3, 1).reshape(b, h * w, cin).contiguous()
        got = lin(tok, h, w, act=1)
        want = F.gelu(conv.double()(x.double()))
        kk = k if kind == "convT" else 1
        want_tok = want.permute(0, 2, 3, 1).reshape(b, h * kk * w * kk, cout)
    assert got.shape == want_tok.shape
    assert rel_l2(got, want_tok) <= 1e-6
    # the derived weight follows the module's parameters
    with torch.no_grad():
        conv.float()
        conv.weight.mul_(0.5)
        got2 = lin(tok, h, w, act=0)
        want2 = conv.double()(x.double()).permute(0, 2, 3, 1).reshape(b, h * kk * w * kk, cout)
    assert rel_l2(got2, want2) <= 1e-6


@pytest.mark.parametrize("cin,c", [(8, 96), (18, 192), (5, 64), (3, 20)])
def test_patch_embed_1x1_any_multiple_of_four(cin, c):
    from dlwp_benchmark_amd import ops

    torch.manual_seed(cin * c)
    conv = torch.nn.Conv2d(cin, c, 1).to(DEV)
    x = torch.randn(3, cin, 7, 9, device=DEV)
    with torch.no_grad():
        got = ops.patch_embed_1x1(x, conv.weight, conv.bias, None)
        want = conv.double()(x.double()).flatten(2).transpose(1, 2)
    assert got.shape == want.shape and rel_l2(got, want) <= 1e-6


def test_linear_random_shape_sweep():
    """Random (rows, in, out) triples, every epilogue combination: the persistent tile walk (XCD-owned slabs, chunked n-tiles,
    clamped tails) must cover every output exactly once."""
    import random

    from dlwp_benchmark_amd import ops

    rng = random.Random(1234)
    for trial in range(24):
        rows = rng.choice([1, 7, 127, 128, 129, 640, 1023, 1025, 3000, 5000, 8191, 20000])
        k = 32 * rng.randint(1, 24)
        n = 4 * rng.randint(1, 200)
        act, has_res, has_bias = rng.randint(0, 1), rng.random() < 0.5, rng.random() < 0.7
        prec = "bf16" if trial % 4 == 3 else "fp32"
        m = _linear(k, n, has_bias, seed=trial)
        g = torch.Generator().manual_seed(trial)
        x = torch.randn(rows, k, generator=g).to(DEV)
        r = torch.randn(rows, n, generator=g).to(DEV) if has_res else None
        with torch.no_grad():
            got = ops.linear(x, m, act=act, resid=r, precision=prec)
            xe, we = (x.bfloat16().double(), m.weight.bfloat16().double()) if prec == "bf16" else (x.double(), m.weight.double())
            want = F.linear(xe, we, m.bias.double() if has_bias else None)
            if act:
                want = F.gelu(want)
            if has_res:
                want = want + r.double()
        assert torch.isfinite(got).all(), (trial, rows, k, n)
        assert rel_l2(got, want) <= 1e-6, (trial, rows, k, n, act, has_res, has_bias, prec, rel_l2(got, want))


@pytest.mark.parametrize("rows,c,hid", [(4096, 96, 384), (1000, 192, 768), (2051, 384, 1536), (77, 96, 384)])
def test_linear_bf16_hidden_handover_is_bit_identical(rows, c, hid):
    """dlwp_linear_bf16_io: in the bf16 form the hidden activation of a block's MLP goes from fc1 to fc2 AS bfloat16 (half the
    bytes both ways).  fc2 rounds its fp32 input to bf16 to nearest even -- the rounding fc1's epilogue now does -- so the
    block output must be bit-identical to the fp32 hand-over; the bf16 tensor itself equals the rounded fp32 one."""
    from dlwp_benchmark_amd import ops

    fc1, fc2 = _linear(c, hid, True, seed=rows), _linear(hid, c, True, seed=rows + 1)
    g = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, c, generator=g).to(DEV)
    r = torch.randn(rows, c, generator=g).to(DEV)
    with torch.no_grad():
        h32 = ops.linear(x, fc1, act=1, precision="bf16")
        h16 = ops.linear(x, fc1, act=1, precision="bf16", out_dtype=torch.bfloat16)
        assert h16.dtype == torch.bfloat16 and torch.equal(h16, h32.bfloat16())
        y32 = ops.linear(h32, fc2, resid=r, precision="bf16")
        y16 = ops.linear(h16, fc2, resid=r, precision="bf16")
    assert torch.equal(y16, y32)
    with pytest.raises(Exception):
        ops.linear(h16, fc2, precision="fp32")


@pytest.mark.parametrize("rows,c,hid", [(4096, 96, 384), (1000, 192, 768), (77, 384, 1536)])
def test_layernorm_bf16_output_feeds_the_bf16_linear_bit_identically(rows, c, hid):
    """dlwp_layernorm_prebias_bf16out + dlwp_linear_bf16_io (bf16 x, bf16 or fp32 out): LayerNorm rounded to bf16 by its own
    kernel, then fc1 -- the same bits as fp32 LayerNorm -> dlwp_linear_bf16 (which rounds its input itself)."""
    from dlwp_benchmark_amd import ops

    fc1 = _linear(c, hid, True, seed=rows)
    g = torch.Generator().manual_seed(rows + c)
    x = (torch.randn(rows, c, generator=g) * 1.7 + 0.2).to(DEV)
    gamma = (1.0 + 0.2 * torch.randn(c, generator=g)).to(DEV)
    beta = (0.2 * torch.randn(c, generator=g)).to(DEV)
    pre = (0.1 * torch.randn(c, generator=g)).to(DEV)
    with torch.no_grad():
        n32 = ops.layer_norm(x, gamma, beta, 1e-5, pre_bias=pre)
        n16 = ops.layer_norm(x, gamma, beta, 1e-5, pre_bias=pre, out_dtype=torch.bfloat16)
        assert n16.dtype == torch.bfloat16 and torch.equal(n16, n32.bfloat16())
        a = ops.linear(n32, fc1, act=1, precision="bf16")
        b = ops.linear(n16, fc1, act=1, precision="bf16")
        c16 = ops.linear(n16, fc1, act=1, precision="bf16", out_dtype=torch.bfloat16)
    assert torch.equal(a, b) and torch.equal(c16, a.bfloat16())
