"""The libm-free GELU of the HIP kernels (dlwp_benchmark_amd/csrc/common.hpp): emulate its fp32
arithmetic in numpy with the coefficients parsed from the header and bound the error against
torch's exact-erf GELU."""
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coeffs():
    src = open(os.path.join(ROOT, "dlwp_benchmark_amd", "csrc", "common.hpp")).read()
    q8 = float(re.search(r"#define DLWP_GELU_QTOP ([-0-9.e+]+)f", src).group(1))
    umax = float(re.search(r"#define DLWP_GELU_UMAX ([-0-9.e+]+)f", src).group(1))
    block = re.search(r"#define DLWP_GELU_COEFFS\(X\)(.*?)\n\n", src, flags=re.S).group(1)
    rest = [float(v) for v in re.findall(r"X\(([-0-9.e+]+)f\)", block)]
    assert len(rest) == 5      # degree 5: leading coefficient + five
    return q8, rest, umax


def test_gelu_polynomial_accuracy():
    q8, rest, umax = _coeffs()
    x = np.concatenate([np.linspace(-12, 12, 600001), np.random.default_rng(0).normal(size=200000) * 2]).astype(np.float32)
    u = np.minimum(np.abs(x), np.float32(umax))
    p = np.full_like(u, np.float32(q8))
    for c in rest:
        p = (p * u + np.float32(c)).astype(np.float32)
    a = (p * u - np.float32(1.0)).astype(np.float32)
    e = np.exp2(a).astype(np.float32)
    got = (np.maximum(x, 0) - np.abs(x) * e).astype(np.float32)
    want = torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()
    err = np.abs(got.astype(np.float64) - want)
    # fp32 evaluation (no FMA in this emulation: two roundings per step): the rounding of the exponent u Q(u) sets the
    # error, not the degree -- the degree-8 polynomial of round 1 measured 4.8e-7 here, this one 5.3e-7
    assert err.max() < 6e-7, err.max()
    # relative to the output scale the rollout sees
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-7
    # approximation error alone (float64 evaluation of the same coefficients)
    x64 = x.astype(np.float64)
    u64 = np.minimum(np.abs(x64), umax)
    p64 = np.full_like(u64, q8)
    for c in rest:
        p64 = p64 * u64 + c
    got64 = np.maximum(x64, 0) - np.abs(x64) * np.exp2(p64 * u64 - 1.0)
    assert np.abs(got64 - want).max() < 1.5e-7
