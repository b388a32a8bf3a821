"""Training row on the GPU: gradients through the HIP spectral kernels.

  * SpectralConv2d: dL/dx, dL/dweights1, dL/dweights2 against gradients produced by the REAL reference class
    (tests/golden/spectral_conv2d_grad_c32_32x64_m8x6.npz);
  * FNO2DModule: one optimisation step (rollout loss, backward, Adam) against the oracle module trained with the
    same data on the CPU -- loss, every parameter gradient and the updated prediction.
Tolerance 1e-4 relative (fp32 gradients accumulate over B*H*W terms in a different order than autograd on the CPU).
"""
import pytest
import torch

from helpers import load_golden, rel_l2

pytestmark = pytest.mark.gpu


def test_spectral_conv2d_gradients_match_reference():
    from dlwp_benchmark_amd import weights as W
    from dlwp_benchmark_amd.models import SpectralConv2d
    from oracle.make_golden import spectral_conv2d_case, tensor_sha

    tag, (ci, co, h, w, m1, m2, b) = "c32_32x64_m8x6", (32, 32, 32, 64, 8, 6, 1)
    g = load_golden(f"spectral_conv2d_grad_{tag}")
    x, w1, w2 = spectral_conv2d_case(ci, co, h, w, m1, m2, b, tag)
    r = W.normal(f"golden/spectral/{tag}/r", (b, co, h, w), 1.0)
    assert tensor_sha(x, w1, w2, r) == str(g["sha"])
    mod = SpectralConv2d(ci, co, m1, m2).to("cuda:0").train()
    with torch.no_grad():
        mod.weights1.copy_(w1)
        mod.weights2.copy_(w2)
    xg = x.to("cuda:0").requires_grad_(True)
    y = mod(xg)
    assert rel_l2(y, torch.from_numpy(load_golden(f"spectral_conv2d_{tag}")["y"])) < 1e-5
    (y * r.to("cuda:0")).sum().backward()
    assert rel_l2(xg.grad, torch.from_numpy(g["gx"])) < 1e-5
    assert rel_l2(mod.weights1.grad, torch.from_numpy(g["gw1"])) < 1e-5
    assert rel_l2(mod.weights2.grad, torch.from_numpy(g["gw2"])) < 1e-5


def test_fno_training_step_matches_oracle_autograd():
    from dlwp_benchmark_amd.models import FNO2DModule
    from dlwp_benchmark_amd.synthetic import navier_stokes
    from dlwp_benchmark_amd.weights import fill_state_dict
    from oracle.restate.fno import FNO2DModuleRef

    kw = dict(n_modes=[8, 8], constant_channels=0, prescribed_channels=0, prognostic_channels=2, hidden_channels=32,
              lifting_channels=64, projection_channels=64, n_layers=3, context_size=1)
    model = FNO2DModule(**kw)
    fill_state_dict(model, std_fn=lambda n, s: 0.85 / s[0] ** 0.5 if "convs.weight" in n else None, gain=0.85)
    ref = FNO2DModuleRef(**kw)
    ref.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    prog = navier_stokes(3, 4, 32, 64, channels=2, seed=11)[2]
    target = navier_stokes(3, 3, 32, 64, channels=2, seed=12)[2]

    def step(m, dev):
        m = m.to(dev).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        opt.zero_grad()
        out = m(prognostic=prog.to(dev))
        loss = torch.nn.functional.mse_loss(out, target.to(dev))
        loss.backward()
        grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        opt.step()
        with torch.no_grad():
            m.eval()
            after = m(prognostic=prog.to(dev)).cpu()
        return float(loss.detach()), grads, after

    loss_g, grads_g, after_g = step(model, "cuda:0")
    loss_c, grads_c, after_c = step(ref, "cpu")
    assert abs(loss_g - loss_c) <= 1e-5 * abs(loss_c)
    assert set(grads_g) == set(grads_c)
    for k in grads_c:
        gg = torch.view_as_real(grads_g[k]) if grads_g[k].is_complex() else grads_g[k]
        gc = torch.view_as_real(grads_c[k]) if grads_c[k].is_complex() else grads_c[k]
        assert rel_l2(gg, gc) < 1e-4, k
    assert rel_l2(after_g, after_c) < 1e-4


# -----------------------------------------------------------------------------------------------------------------
# row f4 for the other backbones: rollout-MSE gradients through the product in train mode vs gradients the REAL reference
# classes produced (tests/golden/grad_*.npz, oracle/make_golden.py gen_grads).  Tolerance 1e-4.
# -----------------------------------------------------------------------------------------------------------------
import json

import numpy as np

from helpers import load_golden


def _grad_cases():
    from oracle.make_golden import GRAD_CASES

    return list(GRAD_CASES)


@pytest.mark.parametrize("tag", _grad_cases())
def test_training_gradients_match_reference(tag):
    import dlwp_benchmark_amd.models as M
    from dlwp_benchmark_amd.weights import fill_state_dict
    from oracle.make_golden import GRAD_CASES, MODEL_CASES, grad_probe, model_inputs, rollout_mse

    base, frames = GRAD_CASES[tag]
    family, cfg, (batch, _), gain = MODEL_CASES[base]
    name = {"swin": "SwinTransformer", "pangu": "PanguWeather", "afno": "FourCastNet", "unet": "UNet", "convlstm": "ConvLSTM"}[family]
    g = load_golden(f"grad_{tag}")
    model = getattr(M, name)(**cfg)
    sha = fill_state_dict(model, gain=gain)
    assert sha == str(g["sha"]), "filler drifted: regenerate fixtures"
    model = model.to("cuda:0")
    assert model.train() is model
    dev = lambda t: t.to("cuda:0") if t is not None else None
    constants, prescribed, prognostic = [dev(t) for t in model_inputs(base, cfg, batch, frames)]
    y = model(constants=constants, prescribed=prescribed, prognostic=prognostic)        # train.py:263-267
    assert y.requires_grad
    loss = rollout_mse(y, prognostic, cfg["context_size"])
    loss.backward()                                                                     # train.py:271
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    names = json.loads(str(g["names"]))
    params = dict(model.named_parameters())
    worst = 0.0
    for i, pname in enumerate(names):
        assert pname in params and params[pname].grad is not None, f"no gradient for {pname}"
        gr = params[pname].grad.detach().double().cpu()
        n_ref, p_ref = float(g["norms"][i]), float(g["projs"][i])
        scale = max(n_ref, 1e-12)
        worst = max(worst, abs(float(gr.norm()) - n_ref) / scale)
        # projection on a fixed random direction: |<g, r> - ref| relative to |g| |r|
        r = grad_probe(tag, pname, gr.shape).double()
        worst = max(worst, abs(float((gr * r).sum()) - p_ref) / (scale * float(r.norm())))
    for key in g.files:
        if key.startswith("grad::"):
            want = torch.from_numpy(g[key]).double()
            got = params[key[6:]].grad.detach().double().cpu()
            worst = max(worst, float((got - want).norm() / want.norm().clamp_min(1e-30)))
    print(tag, "worst relative gradient deviation:", "%.2e" % worst)
    assert worst <= 1e-4


def test_differentiable_restatements_match_the_kernels():
    """The torch forms the backward passes differentiate (training.py) are the same operators as the HIP kernels."""
    from dlwp_benchmark_amd import ops, training as T
    from test_window_attn_gpu import _inputs, _spec

    for shifted in (False, True):
        spec, rows = _spec(16, 32, 2, 8, shifted)
        qkv, bias, table = _inputs(2, 16, 32, 2, 8, rows)
        a = ops.window_attention(qkv, bias, table, spec, precision="fp32_mfma")
        b = T.window_attention_torch(qkv, bias, table, spec)
        assert float((a - b).norm() / b.norm()) <= 2e-6
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 32, 64, generator=gen).cuda()
    w1, b1, w2, b2 = [(0.3 * torch.randn(*s, generator=gen)).cuda() for s in ((2, 4, 4, 4), (2, 4, 4), (2, 4, 4, 4), (2, 4, 4))]
    a = ops.afno2d_filter_cf(x, w1, b1, w2, b2, 4, 0.01, 1.0)
    b = T.afno_filter_torch(x, w1, b1, w2, b2, 4, 0.01, 1.0)
    assert float((a - b).norm() / b.norm()) <= 2e-6
