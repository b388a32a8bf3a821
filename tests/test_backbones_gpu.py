"""GPU parity of the backbone mirrors against trajectories produced by the REAL reference classes
(tests/golden/model_*.npz, made by oracle/make_golden.py): same deterministic weights (filler),
same seeded inputs, multi-step rollout through the HIP path.

Tolerance: per-step relative L2 <= 1e-5 (fp32 path, BASELINE.json north_star)."""
import json

import pytest
import torch

from helpers import load_golden, per_step_rel_l2, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _product_class(family):
    import dlwp_benchmark_amd.models as M

    return {"swin": "SwinTransformer", "pangu": "PanguWeather", "afno": "FourCastNet", "unet": "UNet",
            "convlstm": "ConvLSTM"}[family], M


def _cases():
    from oracle.make_golden import MODEL_CASES

    return list(MODEL_CASES)


@pytest.mark.parametrize("tag", _cases())
def test_rollout_matches_reference_golden(tag):
    from dlwp_benchmark_amd.weights import fill_by_spec
    from oracle.make_golden import MODEL_CASES, model_inputs

    family, cfg, (batch, frames), gain = MODEL_CASES[tag]
    name, M = _product_class(family)
    if not hasattr(M, name):
        pytest.skip(f"{name} not built yet")
    g = load_golden(f"model_{tag}")
    sd, sha = fill_by_spec(json.loads(str(g["param_spec"])), gain=gain)
    assert sha == str(g["sha"])
    model = getattr(M, name)(**cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    buffers = {k for k, _ in model.named_buffers()}
    assert set(missing) <= buffers, f"parameters missing from the filler spec: {set(missing) - buffers}"
    model = model.to("cuda:0").eval()
    constants, prescribed, prognostic = model_inputs(tag, cfg, batch, frames)
    dev = lambda t: t.to("cuda:0") if t is not None else None
    got = model(constants=dev(constants), prescribed=dev(prescribed), prognostic=dev(prognostic))
    torch.cuda.synchronize()
    want = torch.from_numpy(g["y"])
    assert got.shape == want.shape
    errs = per_step_rel_l2(got, want)
    assert max(errs) <= TOL, f"{tag}: per-step rel L2 {['%.2e' % e for e in errs]}"


@pytest.mark.parametrize("tag", ["swin_e32_32x64", "pangu_e48_32x64"])
def test_bf16_attention_stays_close_to_reference(tag):
    """bf16-MFMA window attention (the precision BASELINE.json names for the Swin / Pangu configs):
    Q, K, V, P rounded to bf16, fp32 accumulation and softmax statistics.  Stated bound: per-step
    relative L2 of the rollout <= 5e-3 against the fp32 reference trajectory."""
    import dlwp_benchmark_amd.models as M
    from dlwp_benchmark_amd.weights import fill_by_spec
    from oracle.make_golden import MODEL_CASES, model_inputs

    family, cfg, (batch, frames), gain = MODEL_CASES[tag]
    name, _ = _product_class(family)
    g = load_golden(f"model_{tag}")
    sd, _ = fill_by_spec(json.loads(str(g["param_spec"])), gain=gain)
    model = getattr(M, name)(**cfg)
    model.load_state_dict(sd, strict=False)
    model = model.to("cuda:0").eval().set_attention_precision("bf16")
    constants, prescribed, prognostic = model_inputs(tag, cfg, batch, frames)
    dev = lambda t: t.to("cuda:0") if t is not None else None
    got = model(constants=dev(constants), prescribed=dev(prescribed), prognostic=dev(prognostic))
    torch.cuda.synchronize()
    errs = per_step_rel_l2(got, torch.from_numpy(g["y"]))
    print(tag, "bf16 attention per-step rel L2:", ["%.2e" % e for e in errs])
    assert max(errs) <= 5e-3, errs
    assert max(errs) > 1e-7, "bf16 path suspiciously exact: is it running?"


@pytest.mark.parametrize("rows,c", [(37, 16), (1000, 64), (513, 96), (64, 384), (10, 768), (3, 2048)])
def test_layernorm_kernel(rows, c):
    from dlwp_benchmark_amd import ops

    torch.manual_seed(rows + c)
    x = torch.randn(rows, c) * 3 + 1.5
    w, b = torch.randn(c), torch.randn(c)
    want = torch.nn.functional.layer_norm(x.double(), (c,), w.double(), b.double(), 1e-6)
    got = ops.layer_norm(x.cuda(), w.cuda(), b.cuda(), 1e-6)
    assert rel_l2(got, want) < 5e-7
