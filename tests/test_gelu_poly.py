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
    q8 = float(re.search(r"#define DLWP_GELU_Q8 ([-0-9.e+]+)f", src).group(1))
    umax = float(re.search(r"#define DLWP_GELU_UMAX ([-0-9.e+]+)f", src).group(1))
    block = re.search(r"#define DLWP_GELU_COEFFS\(X\)(.*?)\n\n", src, flags=re.S).group(1)
    rest = [float(v) for v in re.findall(r"X\(([-0-9.e+]+)f\)", block)]
    assert len(rest) == 8
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
    assert err.max() < 4e-7, err.max()
    # relative to the output scale the rollout sees
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-7
