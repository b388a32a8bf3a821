"""GPU parity tests (run with -m gpu on the MI355X box): HIP path (through the C ABI) vs the
oracle (CPU restatement, oracle/restate/fno.py) and vs the reference-generated golden fixtures.

Tolerance (BASELINE.json north_star): per-step relative L2 <= 1e-5 in fp32.
"""
import copy

import numpy as np
import pytest
import torch

from helpers import fno_std_fn, load_golden, per_step_rel_l2, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _dev():
    return torch.device("cuda:0")


# ------------------------------------------------------------------ SpectralConv2d (pinned)
@pytest.mark.parametrize("tag,shape", [
    ("c32_64x64_m12", (32, 32, 64, 64, 12, 12, 1)),
    ("c32_32x64_m8x6", (32, 32, 32, 64, 8, 6, 1)),
])
def test_spectral_conv2d_matches_reference_golden(tag, shape):
    from dlwp_benchmark_amd.models.spectral import SpectralConv2d
    from oracle.make_golden import spectral_conv2d_case

    ci, co, h, w, m1, m2, b = shape
    g = load_golden(f"spectral_conv2d_{tag}")
    x, w1, w2 = spectral_conv2d_case(ci, co, h, w, m1, m2, b, tag)
    mod = SpectralConv2d(ci, co, m1, m2)
    with torch.no_grad():
        mod.weights1.copy_(w1)
        mod.weights2.copy_(w2)
    mod = mod.to(_dev())
    y = mod(x.to(_dev()))
    torch.cuda.synchronize()
    err = rel_l2(y, torch.from_numpy(g["y"]))
    assert err <= TOL, f"rel L2 {err:.3e}"


def test_spectral_conv2d_batch_and_linearity():
    """size-independent properties at the full batch: linearity in x and batch independence."""
    from dlwp_benchmark_amd.models.spectral import SpectralConv2d
    from oracle.restate.fno import spectral_conv2d_ref

    torch.manual_seed(3)
    mod = SpectralConv2d(32, 32, 12, 12).to(_dev())
    x1 = torch.randn(32, 32, 64, 64, device=_dev())
    x2 = torch.randn(32, 32, 64, 64, device=_dev())
    y1, y2, y12 = mod(x1), mod(x2), mod(2.0 * x1 - 0.5 * x2)
    assert rel_l2(y12, 2.0 * y1 - 0.5 * y2) < 5e-6
    ys = mod(x1[5:7].contiguous())
    assert torch.equal(ys, y1[5:7])
    ref = spectral_conv2d_ref(x1[:2].cpu(), mod.weights1.detach().cpu(), mod.weights2.detach().cpu())
    assert rel_l2(y1[:2], ref) <= TOL


# ------------------------------------------------------------------ FNO2DModule (restated; parity unpinned)
def _make_pair(gain=0.85, **kw):
    from dlwp_benchmark_amd.models import FNO2DModule
    from dlwp_benchmark_amd.weights import fill_state_dict
    from oracle.restate.fno import FNO2DModuleRef

    ref = FNO2DModuleRef(**kw).eval()
    fill_state_dict(ref, std_fn=fno_std_fn(gain), gain=gain)
    hip = FNO2DModule(**kw)
    hip.load_state_dict(ref.state_dict())
    return ref, hip.to(_dev()).eval()


NS_KW = dict(n_modes=[12, 12], constant_channels=0, prescribed_channels=0, prognostic_channels=1,
             hidden_channels=32, lifting_channels=256, projection_channels=256, n_layers=4, context_size=1)


def test_fno_one_step_increment():
    """the backbone increment f(x_t) itself (no residual hiding errors)."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    ref, hip = _make_pair(**NS_KW)
    _, _, prog = navier_stokes(4, 1)
    x = prog[:, 0]
    with torch.no_grad():
        want = ref.fno(x)
    got = hip.one_step(x.to(_dev()))
    torch.cuda.synchronize()
    err = rel_l2(got, want)
    assert err <= TOL, f"rel L2 of increment {err:.3e}"


def test_fno_rollout_20_steps_ns64():
    """BASELINE config C2 at a batch the oracle finishes in seconds."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    ref, hip = _make_pair(**NS_KW)
    _, _, prog = navier_stokes(2, 21)
    with torch.no_grad():
        want = ref(prognostic=prog)
    got = hip(prognostic=prog.to(_dev()))
    torch.cuda.synchronize()
    assert got.shape == want.shape == (2, 20, 1, 64, 64)
    errs = per_step_rel_l2(got, want)
    assert max(errs) <= TOL, f"per-step rel L2 {['%.2e' % e for e in errs]}"


def test_fno_rollout_context_constants_prescribed():
    """context_size > 1 with constants and prescribed: exercises the channel-segment table
    that replaces _prepare_inputs / the window blending of fno.py:79-100."""
    from dlwp_benchmark_amd.synthetic import weatherbench

    kw = dict(NS_KW, constant_channels=4, prescribed_channels=1, prognostic_channels=2, context_size=2)
    ref, hip = _make_pair(gain=0.7, **kw)
    cons, presc, prog = weatherbench(2, 7, 32, 64, prognostic_channels=2)
    with torch.no_grad():
        want = ref(constants=cons, prescribed=presc, prognostic=prog)
    d = _dev()
    got = hip(constants=cons.to(d), prescribed=presc.to(d), prognostic=prog.to(d))
    torch.cuda.synchronize()
    assert got.shape == want.shape == (2, 5, 2, 32, 64)
    errs = per_step_rel_l2(got, want)
    assert max(errs) <= TOL, f"per-step rel L2 {['%.2e' % e for e in errs]}"


def test_fno_full_batch_properties():
    """B=32 (the benchmark batch): batch independence + determinism, checked without the oracle."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    _, hip = _make_pair(**NS_KW)
    _, _, prog = navier_stokes(32, 6)
    p = prog.to(_dev())
    a = hip(prognostic=p)
    b = hip(prognostic=p)
    assert torch.equal(a, b)
    sub = hip(prognostic=p[7:9].contiguous())
    assert torch.equal(sub, a[7:9])
    assert torch.isfinite(a).all()


def test_fno_argument_errors():
    from dlwp_benchmark_amd import lib as L

    _, hip = _make_pair(**NS_KW)
    with pytest.raises(L.DlwpError):
        hip(prognostic=torch.zeros(1, 1, 1, 64, 64, device=_dev()))      # T <= context
    with pytest.raises(L.DlwpError):
        hip(prognostic=torch.zeros(1, 3, 2, 64, 64, device=_dev()))      # wrong channel count
    with pytest.raises(L.DlwpError):
        hip(prognostic=torch.zeros(1, 3, 1, 64, 48, device=_dev()))      # unsupported width


def test_bf16x6_matches_fp32_mfma_kernels():
    """The default MLP kernels compute fp32 GEMMs as six bf16 MFMAs (bf16x6 split, fp32 accumulate);
    the plain fp32-MFMA kernels are the cross-check: both must agree with each other far below the
    parity tolerance, and both with the oracle."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    ref, hip = _make_pair(**NS_KW)
    _, _, prog = navier_stokes(4, 6)
    p = prog.to(_dev())
    a = hip(prognostic=p)
    hip.set_execution_form(precision_form="fp32_mfma")     # a field of the plan descriptor, not a process-wide switch
    b = hip(prognostic=p)
    hip.set_execution_form(precision_form="bf16x6")
    assert not torch.equal(a, b), "both forms bit-identical: is precision_form wired to the plan?"
    torch.cuda.synchronize()
    with torch.no_grad():
        want = ref(prognostic=prog)
    assert max(per_step_rel_l2(a, b)) < 2e-6
    assert max(per_step_rel_l2(a, want)) <= TOL
    assert max(per_step_rel_l2(b, want)) <= TOL


@pytest.mark.parametrize("h,w,batch,rows", [(128, 64, 3, None), (32, 64, 5, None), (64, 64, 3, "4"), (64, 128, 2, None)])
def test_fno_trunk_variants_match_oracle(h, w, batch, rows, monkeypatch):
    """The fused trunk at its other group sizes (H = 128 -> 16 workgroups per sample, H = 32 -> 4), with four rows per
    workgroup, and the unfused kernels a 128-wide grid falls back to -- each against the oracle."""
    import subprocess, sys, os, json

    # DLWP_TRUNK_ROWS is read once per process: run the forced-rows case in a child process
    if rows is not None:
        code = ("import sys; sys.path.insert(0, 'tests'); import torch, test_fno_gpu as T;"
                f"T._trunk_case({h}, {w}, {batch})")
        env = dict(os.environ, DLWP_TRUNK_ROWS=rows)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return
    _trunk_case(h, w, batch)


def _trunk_case(h, w, batch):
    from dlwp_benchmark_amd.synthetic import navier_stokes

    ref, hip = _make_pair(**NS_KW)
    _, _, prog = navier_stokes(batch, 4, h, w, seed=5)
    with torch.no_grad():
        want = ref(prognostic=prog)
    got = hip(prognostic=prog.to(_dev()))
    torch.cuda.synchronize()
    errs = per_step_rel_l2(got, want)
    assert max(errs) <= TOL, f"{h}x{w}: per-step rel L2 {['%.2e' % e for e in errs]}"


def test_fno_large_batches_cut_into_resident_chunks():
    """More samples than fit the GPU at one workgroup per CU: the trunk is launched per resident chunk (32 + 8, 3 x 32 + 4)."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    _, hip = _make_pair(**NS_KW)
    prog = navier_stokes(100, 3, 64, 64, seed=9)[2].to(_dev())
    full = hip(prognostic=prog)
    assert bool(torch.isfinite(full).all())
    for lo, hi in ((0, 40), (37, 41), (96, 100)):
        part = hip(prognostic=prog[lo:hi].contiguous())
        assert max(per_step_rel_l2(part, full[lo:hi])) <= 1e-6


def test_fno_persistent_rollout_is_bit_reproducible():
    """The fused kernel's spectrum hand-off is timing dependent (spins on flag-in-data buffers); its arithmetic must
    not be: 60 repeated rollouts of a full and of a ragged batch are bit-identical (tools/stress_rollout.py runs
    the long version)."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    _, hip = _make_pair(**NS_KW)
    for batch in (32, 5):
        prog = navier_stokes(batch, 21, 64, 64, seed=batch)[2].to(_dev())
        ref = hip(prognostic=prog).clone()
        assert bool(torch.isfinite(ref).all())
        for _ in range(60):
            assert torch.equal(hip(prognostic=prog), ref)


def test_fused_timeout_is_loud_and_falls_back():
    """A fused launch whose hand-off spin runs out must never return silently poisoned output: the kernel
    sets a fail word beside the NaN poison, the host reads it after the launch and either raises DLWP_ERR_TIMEOUT
    (on_timeout="raise") or re-runs the range on the unfused kernels, which have no hand-offs (default).  The timeout
    is forced through the descriptor's debug_spin_limit (one poll): run ONCE."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.synthetic import navier_stokes

    ref, hip = _make_pair(**NS_KW)
    _, _, prog = navier_stokes(32, 4)        # 32 samples x 8 workgroups = the whole chip, three steps
    p = prog.to(_dev())
    with torch.no_grad():
        want = ref(prognostic=prog)
    good = hip(prognostic=p)
    assert hip.fused_timeouts() == 0
    assert max(per_step_rel_l2(good, want)) <= TOL
    hip._debug_spin_limit = 1
    hip.set_execution_form(on_timeout="raise")
    with pytest.raises(L.DlwpError, match="status -5"):
        hip(prognostic=p)
    assert hip.fused_timeouts() == 1
    hip.set_execution_form(on_timeout="rerun")   # new plan (the form is part of the plan key)
    got = hip(prognostic=p)
    torch.cuda.synchronize()
    assert hip.fused_timeouts() == 1, "the forced timeout did not fire"
    assert torch.isfinite(got).all()
    assert max(per_step_rel_l2(got, want)) <= TOL
    # deferred mode: asynchronous calls, ONE verification -- which must raise while the knob forces timeouts
    hip.set_execution_form(check="deferred")
    bad = hip(prognostic=p)
    with pytest.raises(L.DlwpError, match="status -5"):
        hip.verify()
    assert not torch.isfinite(bad).all(), "a timed-out asynchronous launch must leave NaN in its output"
    hip._debug_spin_limit = 0
    fine = hip(prognostic=p)
    hip.verify()
    assert torch.equal(fine, good)
    hip.set_execution_form(check="per_call")
    again = hip(prognostic=p)
    assert hip.fused_timeouts() == 0 and torch.equal(again, good)


def test_f16x3_form_matches_bf16x6_and_oracle():
    """precision_form "f16x3" (default of the module; dlwp_fno2d_desc.precision_form = 2): the fused step kernel forms its
    big fp32 products from two-part f16 splits.  Same bound against the oracle as the bf16x6 form (per-step rel-L2 <= 1e-5
    over a 20-step rollout), and the two forms agree far below that bound."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    ref, hip = _make_pair(**NS_KW)
    assert hip.precision_form == "f16x3"
    _, _, prog = navier_stokes(32, 21)
    p = prog.to(_dev())
    with torch.no_grad():
        want = ref(prognostic=prog[:4])
    a = hip(prognostic=p)
    assert hip.range_reruns() == 0
    hip.set_execution_form(precision_form="bf16x6")
    b = hip(prognostic=p)
    ea, eb = per_step_rel_l2(a[:4], want), per_step_rel_l2(b[:4], want)
    assert max(ea) <= TOL and max(eb) <= TOL, (ea, eb)
    assert max(ea) <= 2.0 * max(eb) + 2e-7, (max(ea), max(eb))     # fp32-grade: not measurably worse than bf16x6
    assert max(per_step_rel_l2(a, b)) <= 2e-6


def test_f16x3_range_guard_repeats_on_bf16x6():
    """Activations beyond the f16 range (|x| >= 65520) become inf in the f16x3 operands.  The fused kernel flags a non-finite
    OUTPUT (bit 1 of its fail word / the second sticky counter); a checked call repeats the range on the bf16x6 kernels
    (fp32 exponent range) and returns their result, a deferred-check evaluation raises DLWP_ERR_RANGE (status -6)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.synthetic import navier_stokes

    ref, hip = _make_pair(**NS_KW)
    _, _, prog = navier_stokes(8, 3)
    big = (prog * 3.0e6).to(_dev())          # lifting activations ~1e6: finite in fp32 / bf16 parts, inf in f16
    exact = copy.deepcopy(hip).set_execution_form(precision_form="bf16x6", launch_form=3)
    want = exact(prognostic=big)
    assert torch.isfinite(want).all()
    got = hip(prognostic=big)
    assert hip.range_reruns() == 1, "the range guard did not fire"
    assert torch.isfinite(got).all()
    assert max(per_step_rel_l2(got, want)) <= 1e-6
    small = hip(prognostic=prog.to(_dev()))
    assert hip.range_reruns() == 1 and torch.isfinite(small).all()
    hip.set_execution_form(check="deferred")
    hip(prognostic=big)
    with pytest.raises(L.DlwpError, match="status -6"):
        hip.verify()
    hip(prognostic=prog.to(_dev()))
    hip.verify()


@pytest.mark.parametrize("nprog", [2, 3, 4])
def test_fno_register_feedback_with_several_prognostic_channels(nprog):
    """Persistent rollout with the step's input and residual kept in registers (TrunkParams.feed_regs: no constants, no
    prescribed channels, context 1, <= 4 prognostic channels): the lanes of channel group g carry channel g, the other lane
    groups zeros.  2, 3 and 4 channels (3 -> the projection's FMA layer is built for 4 outputs) over 12 steps against the
    oracle, and against the same rollout issued step by step (launch_form 1: every step reloads its input from memory)."""
    from dlwp_benchmark_amd.synthetic import navier_stokes

    kw = dict(NS_KW, prognostic_channels=nprog)
    ref, hip = _make_pair(**kw)
    _, _, prog = navier_stokes(8, 13, 64, 64, channels=nprog)
    p = prog.to(_dev())
    stepwise = copy.deepcopy(hip).set_execution_form(launch_form=1)   # (before the first call: a plan cannot be copied)
    with torch.no_grad():
        want = ref(prognostic=prog[:2])
    got = hip(prognostic=p)
    assert max(per_step_rel_l2(got[:2], want)) <= TOL
    assert torch.equal(stepwise(prognostic=p), got)


def test_tfno_matches_fno_with_reconstructed_weights():
    """TFNO2DModule (fno.py:109-146) = the FNO path on the dense weight rebuilt from its Tucker factors: same
    trajectory as an FNO2DModule loaded with the reconstructed tensors, and as the oracle on them."""
    from dlwp_benchmark_amd.models import FNO2DModule, TFNO2DModule
    from dlwp_benchmark_amd.synthetic import navier_stokes
    from dlwp_benchmark_amd.weights import fill_state_dict
    from oracle.restate.fno import FNO2DModuleRef

    kw = dict(NS_KW)
    t = TFNO2DModule(rank=0.8, **kw)
    fill_state_dict(t, std_fn=lambda n, s: 0.3 if ("core" in n or "factor" in n) else None, gain=0.85)
    with torch.no_grad():   # scale every core so that the dense weight has the magnitude the FNO tests use
        for w in t.fno.fno_blocks.convs.weight:
            w.core.mul_(0.85 / 32 ** 0.5 / float(w.dense().abs().pow(2).mean().sqrt()))
    dense = FNO2DModule(**kw)
    sd = {k: v for k, v in t.state_dict().items() if ".core" not in k and ".factors." not in k}
    for l, w in enumerate(t.fno.fno_blocks.convs.weight):
        sd[f"fno.fno_blocks.convs.weight.{l}.tensor"] = w.dense().detach()
    dense.load_state_dict(sd)
    ref = FNO2DModuleRef(**kw).eval()
    ref.load_state_dict(sd)
    _, _, prog = navier_stokes(2, 4)
    with torch.no_grad():
        want = ref(prognostic=prog)
    a = t.to(_dev()).eval()(prognostic=prog.to(_dev()))
    b = dense.to(_dev()).eval()(prognostic=prog.to(_dev()))
    # the module rebuilds the dense weight on the device, the cross-check used the host: equal up to that rounding
    assert max(per_step_rel_l2(a, b)) <= 1e-6
    assert max(per_step_rel_l2(a, want)) <= TOL
    assert max(per_step_rel_l2(b, want)) <= TOL
