"""CPU emulation of the "f16x3" product form (dlwp_benchmark_amd/csrc/common.hpp), as of round 3:
    x = xh + xm,  xh = f16(x),       xm' = f16((x - xh) * 2^11)
    w = wh + wm,  wh = f16(w 2^s),   wm' = f16((w 2^s - wh) * 2^11),   s: max |w| 2^s in [8, 16)
    2^(11+s) w x ~= wm' * xh + wh * xm' + (wh * 2^11) * xh      (fp32 accumulation on the matrix instructions)
numpy's float16 rounds to nearest even and keeps subnormals, like v_cvt_pk_f16_f32 / v_fma_mixlo_f16 and the MFMA operands.
The test pins the accuracy claims of DESIGN.md section 4.5: fp32-GEMM grade at ANY weight magnitude and at any activation
magnitude down to 2^-14 (BOTH residuals are stored scaled), graceful below; and it keeps the round-2 form (activation residual
unscaled, today only behind a LayerNorm: split_f16_pair_unit) with the floor that form has on small inputs."""
import numpy as np
import pytest


def _f16(x):
    return x.astype(np.float16).astype(np.float32)


def _bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def _weight_shift(w):
    """common.hpp f16x3_weight_shift: s with max |w| 2^s in [8, 16)"""
    m = float(np.abs(w).max())
    return 0 if not (m > 0 and np.isfinite(m)) else 4 - int(np.frexp(m)[1])


def _f16x3(w, x):
    """the round-3 form: both residuals scaled by 2^11, leading product through whB = wh * 2^11, one multiply at the end"""
    s = _weight_shift(w)
    ws = (w * np.float32(2.0 ** s)).astype(np.float32)
    wh = _f16(ws)
    wm = _f16(((ws - wh) * np.float32(2048.0)).astype(np.float32))
    whb = _f16(wh * np.float32(2048.0))
    assert np.isfinite(whb).all() and np.array_equal(whb, wh * np.float32(2048.0))      # exact, no overflow
    xh = _f16(x)
    xm = _f16(((x - xh).astype(np.float32) * np.float32(2048.0)).astype(np.float32))
    acc = wm.astype(np.float64) @ xh + wh.astype(np.float64) @ xm + whb.astype(np.float64) @ xh
    return acc * 2.0 ** -(11 + s)


def _f16x3_unit(w, x):
    """the round-2 form (split_f16_pair_unit): weight residual scaled, activation residual NOT"""
    wh = _f16(w)
    wm = _f16(((w - wh) * np.float32(2048.0)).astype(np.float32))
    xh = _f16(x)
    xm = _f16((x - xh).astype(np.float32))
    xs = _f16(xh * np.float32(2.0 ** -11))
    return wm.astype(np.float64) @ xs + wh.astype(np.float64) @ xm + wh.astype(np.float64) @ xh


def _forms(w, x):
    ref = w.astype(np.float64) @ x.astype(np.float64)
    den = np.linalg.norm(ref)
    chain = np.zeros(ref.shape, np.float32)          # a plain fp32 FMA chain, the accuracy the form has to match
    for k in range(w.shape[1]):
        chain = (chain + w[:, k:k + 1] * x[k:k + 1, :]).astype(np.float32)
    f16x3 = _f16x3(w, x)
    wh3 = _bf16(w); r = (w - wh3).astype(np.float32); wm3 = _bf16(r); wl3 = _bf16((r - wm3).astype(np.float32))
    xh3 = _bf16(x); r = (x - xh3).astype(np.float32); xm3 = _bf16(r); xl3 = _bf16((r - xm3).astype(np.float32))
    bf16x6 = sum(a.astype(np.float64) @ b.astype(np.float64)
                 for a, b in [(wl3, xh3), (wh3, xl3), (wm3, xm3), (wm3, xh3), (wh3, xm3), (wh3, xh3)])
    err = lambda a: float(np.linalg.norm(a - ref) / den)
    return err(chain), err(f16x3), err(bf16x6)


@pytest.mark.parametrize("wscale", [10.0, 1.0 / 16, 1e-2, 1e-3])
def test_f16x3_lifting_shape_is_fp32_grade_at_any_weight_magnitude(wscale):
    rng = np.random.default_rng(0)
    w = (rng.uniform(-1, 1, (32, 256)) * wscale).astype(np.float32)        # lifting layer 2: K = 256
    x = _gelu(rng.normal(0, 1, (256, 2048))).astype(np.float32)            # its operand: GELU outputs
    chain, f16x3, bf16x6 = _forms(w, x)
    assert f16x3 <= 1.5e-7 and f16x3 <= chain, (chain, f16x3, bf16x6)
    assert bf16x6 <= 2e-8


@pytest.mark.parametrize("xscale", [1.0, 0.05, 1e-2, 1e-3, 1e-4, 3e-5])
@pytest.mark.parametrize("wscale", [0.18, 1e-3, 30.0])
def test_f16x3_k32_shapes_hold_the_bound_at_small_activations(xscale, wscale):
    """K = 32 (skip convolution, projection layer 1).  With the activation residual stored scaled the error no longer grows as the
    activations shrink (round 2: 2e-6 at 0.01, 2e-5 at 1e-3 -- the unit-scale form below still shows it)."""
    rng = np.random.default_rng(1)
    w = (rng.uniform(-1, 1, (256, 32)) * wscale).astype(np.float32)
    x = rng.normal(0, xscale, (32, 2048)).astype(np.float32)
    chain, f16x3, _ = _forms(w, x)
    # full precision while the leading part is a normal f16 number (|x| >= 2^-14 = 6.1e-5); below, see the next test
    assert f16x3 <= (1.5e-7 if xscale >= 1e-4 else 4e-7) and (f16x3 <= 1.5 * chain or xscale < 1e-4), (chain, f16x3)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    unit = float(np.linalg.norm(_f16x3_unit(w, x) - ref) / np.linalg.norm(ref))
    if xscale >= 1.0:
        assert unit <= 1.5e-7             # what the token MLP's layer 1 relies on behind its LayerNorm
    if xscale <= 1e-3 and wscale < 1.0:
        assert unit > 5 * f16x3           # the floor this round removed


def test_f16x3_degrades_gracefully_below_the_f16_normal_range():
    """|x| < 2^-14: xh is an f16 subnormal, the scaled residual picks up what it drops down to an absolute 2^-36"""
    rng = np.random.default_rng(2)
    w = (rng.uniform(-1, 1, (64, 64)) * 0.1).astype(np.float32)
    for xscale, bound in ((1e-5, 1.2e-6), (1e-6, 1.2e-5), (1e-7, 1.2e-4)):    # ~ 9e-12 / rms(x): an absolute 2^-36
        x = rng.normal(0, xscale, (64, 512)).astype(np.float32)
        _, f16x3, _ = _forms(w, x)
        assert f16x3 <= bound, (xscale, f16x3)


def test_f16_overflow_is_what_the_range_guard_catches():
    with np.errstate(over="ignore"):
        assert np.isinf(np.float32(65520.0).astype(np.float16)) and np.isfinite(np.float32(65519.0).astype(np.float16))
