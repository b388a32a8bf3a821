"""CPU emulation of the "f16x3" product form of the fused FNO step kernel (dlwp_benchmark_amd/csrc/common.hpp):
    x = xh + xm,  xh = f16(x), xm = f16(x - xh);   w = wh + wm,  wh = f16(w), wm' = f16((w - wh) * 2^11)
    w x ~= wm' * (xh * 2^-11) + wh * xm + wh * xh         (fp32 accumulation on the matrix instructions)
numpy's float16 rounds to nearest even and keeps subnormals, like v_cvt_pk_f16_f32 and the MFMA operands.  The test
pins the accuracy claims of DESIGN.md section 4.5: fp32-GEMM grade for O(1) activations at ANY weight magnitude (that
is what the scaled weight residual buys), and the documented floor for very small activations."""
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


def _forms(w, x):
    ref = w.astype(np.float64) @ x.astype(np.float64)
    den = np.linalg.norm(ref)
    chain = np.zeros(ref.shape, np.float32)          # a plain fp32 FMA chain, the accuracy the form has to match
    for k in range(w.shape[1]):
        chain = (chain + w[:, k:k + 1] * x[k:k + 1, :]).astype(np.float32)
    wh = _f16(w)
    wm = _f16(((w - wh) * np.float32(2048.0)).astype(np.float32))
    xh = _f16(x)
    xm = _f16((x - xh).astype(np.float32))
    xs = _f16(xh * np.float32(2.0 ** -11))
    f16x3 = wm.astype(np.float64) @ xs + wh.astype(np.float64) @ xm + wh.astype(np.float64) @ xh
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


@pytest.mark.parametrize("xscale,bound", [(1.0, 1.5e-7), (0.05, 1e-6), (0.01, 4e-6)])
def test_f16x3_k32_shapes_and_the_small_activation_floor(xscale, bound):
    """K = 32 (skip convolution, projection layer 1).  The residual of an activation below 0.125 is an f16 subnormal
    (absolute spacing 2^-24): the relative error grows as the activations shrink -- the documented limit of the form."""
    rng = np.random.default_rng(1)
    w = (rng.uniform(-1, 1, (256, 32)) * 0.18).astype(np.float32)
    x = rng.normal(0, xscale, (32, 2048)).astype(np.float32)
    _, f16x3, _ = _forms(w, x)
    assert f16x3 <= bound, f16x3


def test_f16_overflow_is_what_the_range_guard_catches():
    with np.errstate(over="ignore"):
        assert np.isinf(np.float32(65520.0).astype(np.float16)) and np.isfinite(np.float32(65519.0).astype(np.float16))
