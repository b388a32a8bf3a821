"""Training row, host-side math: the weight-gradient formula of dlwp_benchmark_amd/training.py
and the adjoint identity the HIP backward-data pass relies on, against gradients the REAL reference
SpectralConv2d produced (tests/golden/spectral_conv2d_grad_*.npz) and against autograd in double precision."""
import numpy as np
import pytest
import torch

from dlwp_benchmark_amd import weights as W
from dlwp_benchmark_amd.training import pde_arena_rows, spectral_weight_grad
from helpers import load_golden, rel_l2
from oracle.make_golden import spectral_conv2d_case, tensor_sha
from oracle.restate.fno import spectral_conv2d_ref

CASES = {"c32_32x64_m8x6": (32, 32, 32, 64, 8, 6, 1), "c4_16x16_m4": (4, 4, 16, 16, 4, 4, 2)}


@pytest.mark.parametrize("tag", list(CASES))
def test_weight_gradient_and_adjoint_match_reference_gradients(tag):
    ci, co, h, w, m1, m2, b = CASES[tag]
    g = load_golden(f"spectral_conv2d_grad_{tag}")
    x, w1, w2 = spectral_conv2d_case(ci, co, h, w, m1, m2, b, tag)
    r = W.normal(f"golden/spectral/{tag}/r", (b, co, h, w), 1.0)
    assert tensor_sha(x, w1, w2, r) == str(g["sha"]), "filler drifted: regenerate fixtures"
    rows, _ = pde_arena_rows(h, m1)
    gw = spectral_weight_grad(x, r, rows, rows, m2, 1.0, 1.0 / (h * w))
    assert rel_l2(gw[:, :, :m1], torch.from_numpy(g["gw1"])) < 2e-6
    assert rel_l2(gw[:, :, m1:], torch.from_numpy(g["gw2"])) < 2e-6
    # backward-data = the same operator with conjugate-transposed weights
    adj = lambda t: torch.view_as_real(torch.view_as_complex(t.contiguous()).conj().transpose(0, 1).contiguous())
    gx = spectral_conv2d_ref(r, adj(w1), adj(w2))
    assert rel_l2(gx, torch.from_numpy(g["gx"])) < 2e-6


def test_adjoint_identity_with_general_rows_in_double():
    """FNO geometry: distinct rows_in / rows_out, forward-normalised transforms, Nyquist column kept."""
    torch.manual_seed(3)
    b, c, h, w, n_cols = 2, 3, 12, 16, 9
    rows_in, rows_out = [0, 1, 2, 10, 11], [1, 2, 3, 11, 0]
    fwd, inv = 1.0 / (h * w), 1.0

    def op(x, wt, ri, ro):
        xf = torch.fft.rfft2(x) * fwd
        out = torch.zeros(x.shape[0], wt.shape[1], h, w // 2 + 1, dtype=torch.complex128)
        out[:, :, ro, :n_cols] = torch.einsum("bixy,ioxy->boxy", xf[:, :, ri, :n_cols], wt)
        return torch.fft.irfft2(out, s=(h, w)) * (h * w) * inv

    x = torch.randn(b, c, h, w, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(c, c, len(rows_in), n_cols, dtype=torch.complex128, requires_grad=True)
    y = op(x, wt, rows_in, rows_out)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    gx = op(r, wt.detach().conj().transpose(0, 1), rows_out, rows_in)
    assert rel_l2(gx, x.grad) < 1e-12
    gw = spectral_weight_grad(x.detach(), r, rows_in, rows_out, n_cols, fwd, inv)
    assert rel_l2(gw, torch.view_as_real(wt.grad)) < 1e-12
