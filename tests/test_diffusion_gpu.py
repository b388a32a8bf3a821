"""PDE-Refiner backbones (reference models/diffusion_models/modern_unet/modern_unet.py): the product
mirror on the GPU against trajectories the REAL reference classes produced (tests/golden/model_diff*.npz, oracle/make_golden.py
`gen_diffusion`) with the same filler weights, the same host-generated start noise (torch.manual_seed) and the same scheduler
object (oracle/restate/ddpm.py: a restatement of diffusers' DDPMScheduler, which is absent here -- the scheduler is parity
unpinned, the networks are pinned)."""
import json

import pytest
import torch

from helpers import load_golden, per_step_rel_l2

def _cases():
    from oracle.make_golden import DIFFUSION_CASES

    return list(DIFFUSION_CASES)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", _cases())
def test_diffusion_rollout_matches_reference_golden(tag):
    import dlwp_benchmark_amd.models as M
    from dlwp_benchmark_amd.weights import fill_by_spec
    from oracle.make_golden import DIFFUSION_CASES, DIFFUSION_SEED, diffusion_inputs
    from oracle.restate.ddpm import DDPMSchedulerRestated

    cls, cfg, (batch, frames), hw, betas, nsteps = DIFFUSION_CASES[tag]
    g = load_golden(f"model_{tag}")
    sd, sha = fill_by_spec(json.loads(str(g["param_spec"])), gain=0.7)
    assert sha == str(g["sha"])
    model = getattr(M, cls)(**cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing
    model = model.to("cuda:0").eval()
    constants, prescribed, prognostic = diffusion_inputs(tag, cls, cfg, batch, frames, hw)
    dev = lambda t: t.to("cuda:0") if t is not None else None
    sched = DDPMSchedulerRestated(betas, seed=7)
    sched.set_timesteps(nsteps)
    torch.manual_seed(DIFFUSION_SEED)
    got = model(constants=dev(constants), prescribed=dev(prescribed), prognostic=dev(prognostic), noise_scheduler=sched)
    torch.cuda.synchronize()
    want = torch.from_numpy(g["y"])
    assert got.shape == want.shape
    errs = per_step_rel_l2(got, want)
    assert max(errs) <= 1e-5, f"{tag}: per-step rel L2 {['%.2e' % e for e in errs]}"


def test_state_dict_layout_matches_reference():
    import dlwp_benchmark_amd.models as M
    from oracle.make_golden import DIFFUSION_CASES

    for tag, (cls, cfg, *_rest) in DIFFUSION_CASES.items():
        g = load_golden(f"model_{tag}")
        want = {k: (tuple(s), d) for k, s, d in json.loads(str(g["state_spec"]))}
        got = {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in getattr(M, cls)(**cfg).state_dict().items()}
        assert got == want, tag


def test_restated_scheduler_basics():
    """leading spacing, no variance noise at t = 0, v-prediction algebra: model_output = 0 and unit alpha keep the sample"""
    from oracle.restate.ddpm import DDPMSchedulerRestated

    s = DDPMSchedulerRestated([0.5, 0.3, 0.1, 0.05, 0.02, 0.01], seed=1)
    s.set_timesteps(5)
    assert s.timesteps.tolist() == [4, 3, 2, 1, 0]
    s.set_timesteps(3)
    assert s.timesteps.tolist() == [4, 2, 0]
    x = torch.randn(2, 1, 3, 4, 5)
    a = s.step(torch.zeros_like(x), 0, x).prev_sample
    b = s.step(torch.zeros_like(x), 0, x).prev_sample
    assert torch.equal(a, b)                                   # deterministic at t = 0
    s.reseed()
    c = s.step(torch.zeros_like(x), 2, x).prev_sample
    s.reseed()
    d = s.step(torch.zeros_like(x), 2, x).prev_sample
    assert torch.equal(c, d)                                   # the same seed, the same noise
