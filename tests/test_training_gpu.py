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
