"""Low end of the f16x3 product form: two-part f16 splits have the f16 exponent range.  The weight
residual and (round 3) the ACTIVATION residual are stored scaled by 2^11, so neither is an f16 subnormal at any magnitude
its leading part represents; what is left is the leading part's own range (|x| >= 2^-14 ~ 6.1e-5 for full precision, graceful
below).  These tests drive the FNO step kernel (default form of the headline config), the Linear kernel and the AFNO block
tail with small-magnitude, mixed-scale and near-constant inputs against the oracle / float64 at the same bounds the O(1) tests
hold: per-step rel-L2 <= 1e-5 for rollouts, 1e-6 for a single operator."""
import copy

import pytest
import torch
import torch.nn.functional as F

from helpers import fno_std_fn, per_step_rel_l2, rel_l2

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 1e-5

NS_KW = dict(n_modes=[12, 12], constant_channels=0, prescribed_channels=0, prognostic_channels=1,
             hidden_channels=32, lifting_channels=256, projection_channels=256, n_layers=4, context_size=1)


def _fno_pair(zero_bias=False, gain=0.85):
    from dlwp_benchmark_amd.models import FNO2DModule
    from dlwp_benchmark_amd.weights import fill_state_dict
    from oracle.restate.fno import FNO2DModuleRef

    ref = FNO2DModuleRef(**NS_KW).eval()
    fill_state_dict(ref, std_fn=fno_std_fn(gain), gain=gain)
    if zero_bias:
        with torch.no_grad():
            for n, p in ref.named_parameters():
                if n.endswith("bias") or ".bias" in n:
                    p.zero_()
    hip = FNO2DModule(**NS_KW)
    hip.load_state_dict(ref.state_dict())
    return ref, hip.to(DEV).eval()


@pytest.mark.parametrize("zero_bias", [False, True])
@pytest.mark.parametrize("scale", [1e-2, 1e-3, 1e-4])
def test_fno_rollout_on_small_magnitude_fields(scale, zero_bias):
    """BASELINE configs[1] with the input field scaled down (and, harsher, with every bias zeroed so that NOTHING in the
    network restores an O(1) scale: hidden activations are then ~scale * |w|): f16x3 -- the module default -- against the
    oracle at the north-star bound, and not measurably worse than the bf16x6 form (fp32 exponent range)."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    ref, hip = _fno_pair(zero_bias)
    _, hip6 = _fno_pair(zero_bias)
    hip6.set_execution_form(precision_form="bf16x6")
    assert hip.precision_form == "f16x3"
    _, _, prog = navier_stokes(8, 21)
    prog = prog * scale
    with torch.no_grad():
        want = ref(prognostic=prog[:2])
    p = prog.to(DEV)
    a = hip(prognostic=p)
    reruns = hip.range_reruns()
    b = hip6(prognostic=p)
    ea, eb = per_step_rel_l2(a[:2], want), per_step_rel_l2(b[:2], want)
    print(f"scale {scale:g} zero_bias {zero_bias}: f16x3 max {max(ea):.3e} (range_reruns {reruns}), bf16x6 max {max(eb):.3e}")
    assert torch.isfinite(a).all()
    assert max(eb) <= TOL, eb
    assert max(ea) <= TOL, ea
    assert max(ea) <= 3.0 * max(eb) + 3e-7, (max(ea), max(eb))


def _x_cases(rows, k, gen):
    base = torch.randn(rows, k, generator=gen)
    mixed = base.clone()
    mixed[:, ::3] *= 1e-4            # every third channel four orders below its neighbours
    mixed[:, 1::7] *= 1e-2
    rowmix = base * torch.logspace(-5, 1, rows).unsqueeze(1)      # token magnitudes from 1e-5 to 10 in one call
    const = 1.0 + 1e-4 * base        # a near-constant field: the information sits in the 14th bit
    return {"1e-2": base * 1e-2, "1e-3": base * 1e-3, "1e-4": base * 1e-4, "3e-5": base * 3e-5, "mixed": mixed,
            "rowmix": rowmix, "near_constant": const}


@pytest.mark.parametrize("case", ["1e-2", "1e-3", "1e-4", "3e-5", "mixed", "rowmix", "near_constant"])
@pytest.mark.parametrize("k,n,act", [(96, 384, 1), (384, 96, 0), (768, 192, 0)])
def test_linear_f16x3_small_and_mixed_scale_inputs(case, k, n, act):
    """dlwp_linear_f16x3 against float64 on inputs far below the O(1) of the other tests: the bound of the fp32-grade forms
    (1e-6 rel-L2 over the call, and per ROW for the row-mixed case: no token may be sacrificed to its louder neighbours)."""
    from dlwp_benchmark_amd import ops

    torch.manual_seed(k + n)
    m = torch.nn.Linear(k, n, bias=False).to(DEV)
    gen = torch.Generator().manual_seed(5)
    rows = 1024
    x = _x_cases(rows, k, gen)[case].to(DEV)
    with torch.no_grad():
        got = ops.linear(x, m, act=act, precision="f16x3").double()
        ref32 = ops.linear(x, m, act=act).double()
        want = F.linear(x.double(), m.weight.double())
        if act:
            want = F.gelu(want)
    e16, e32 = rel_l2(got, want), rel_l2(ref32, want)
    row16 = ((got - want).norm(dim=1) / want.norm(dim=1).clamp_min(1e-300)).max().item()
    row32 = ((ref32 - want).norm(dim=1) / want.norm(dim=1).clamp_min(1e-300)).max().item()
    print(f"{case} {k}->{n}: f16x3 {e16:.2e} (worst row {row16:.2e}), bf16x6 {e32:.2e} (worst row {row32:.2e})")
    assert e16 <= 1e-6, (e16, e32)
    assert row16 <= 2e-6 + 3 * row32, (row16, row32)


@pytest.mark.parametrize("hid_scale", [1.0, 1e-2, 1e-3])
def test_afno_block_tail_f16x3_small_hidden_activations(hid_scale):
    """dlwp_afno_block_tail_f16x3: the LayerNorm in front keeps fc1's input O(1), but fc2's input (GELU of fc1) is as small as
    fc1's weights make it -- the same float64 bound with hidden activations of 1e-2 and 1e-3."""
    from dlwp_benchmark_amd import ops

    gen = torch.Generator(device="cpu").manual_seed(41)
    b, h, w, c, hidden, eps = 2, 8, 12, 64, 256, 1e-6
    f_cf = torch.randn(b, c, h, w, generator=gen).to(DEV)
    l_cf = torch.randn(b, c, h, w, generator=gen).to(DEV)
    x = (1.5 * torch.randn(b, h, w, c, generator=gen)).to(DEV)
    g2 = (1.0 + 0.2 * torch.randn(c, generator=gen)).to(DEV)
    be2 = (0.2 * torch.randn(c, generator=gen)).to(DEV)
    w1 = (hid_scale * torch.randn(hidden, c, generator=gen) / c ** 0.5).to(DEV)
    b1 = (hid_scale * 0.3 * torch.randn(hidden, generator=gen)).to(DEV)
    w2 = (torch.randn(c, hidden, generator=gen) / hidden ** 0.5 / hid_scale).to(DEV)
    b2 = (0.3 * torch.randn(c, generator=gen)).to(DEV)
    s64 = (f_cf + l_cf).double().permute(0, 2, 3, 1) + x.double()
    n64 = F.layer_norm(s64, (c,), g2.double(), be2.double(), eps)
    mlp = F.linear(F.gelu(F.linear(n64, w1.double(), b1.double())), w2.double(), b2.double())
    want = s64 + mlp
    outs = {}
    for form in ("f16x3", "bf16x6"):
        packed = ops.TokenMlpWeights().get(w1, w2, g2, be2, b1, merged=True, f16x3=form == "f16x3")
        xin = x.clone()
        outs[form] = ops.afno_block_tail(f_cf, l_cf, xin, packed, b2, hidden, eps, out=xin, form=form).double()
    # the error that matters is the MLP's own: measure it on the increment, not on the skip-dominated sum
    e = {f: rel_l2(o - s64, mlp) for f, o in outs.items()}
    print(f"hidden scale {hid_scale:g}: MLP increment rel-L2 f16x3 {e['f16x3']:.2e}, bf16x6 {e['bf16x6']:.2e}")
    assert e["f16x3"] <= 2e-6 + 3 * e["bf16x6"], e


@pytest.mark.parametrize("tag,cls,fn", [("swin_e32_32x64", "SwinTransformer", "swin_rollout"),
                                        ("afno_e16_32x64", "FourCastNet", "afnonet_rollout")])
def test_backbone_f16x3_near_constant_channel(tag, cls, fn):
    """C3 / C4 architectures with one prognostic channel nearly constant (1 + 1e-4 * field) and one scaled to 1e-3: the f16x3
    forms of the Linear layers / block tail against the oracle's rollout at the fp32 bound."""
    import importlib
    import json

    import dlwp_benchmark_amd.models as M
    from dlwp_benchmark_amd.weights import fill_by_spec
    from helpers import load_golden
    from oracle.make_golden import MODEL_CASES, model_inputs

    family, cfg, (batch, frames), gain = MODEL_CASES[tag]
    g = load_golden(f"model_{tag}")
    sd, _ = fill_by_spec(json.loads(str(g["param_spec"])), gain=gain)
    c, p, x = model_inputs(tag, cfg, batch, frames)
    x = x.clone()
    x[:, :, 0] = 1.0 + 1e-4 * x[:, :, 0]
    if x.shape[2] > 1:
        x[:, :, 1] *= 1e-3
    model = getattr(M, cls)(**cfg, compute_precision="f16x3")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    model = model.to(DEV).eval()
    full_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    mod = importlib.import_module({"swin_rollout": "oracle.restate.swin", "afnonet_rollout": "oracle.restate.afno"}[fn])
    with torch.no_grad():
        want = getattr(mod, fn)(full_sd, cfg, c, p, x)
    dev = lambda t: t.to(DEV) if t is not None else None
    got = model(constants=dev(c), prescribed=dev(p), prognostic=dev(x))
    model.set_compute_precision("fp32")
    got32 = model(constants=dev(c), prescribed=dev(p), prognostic=dev(x))
    e16, e32 = per_step_rel_l2(got, want), per_step_rel_l2(got32, want)
    print(tag, "f16x3", ["%.2e" % e for e in e16], "bf16x6", ["%.2e" % e for e in e32])
    assert max(e16) <= TOL and max(e32) <= TOL
