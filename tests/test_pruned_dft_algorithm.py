"""CPU check of the ALGORITHM the HIP spectral kernels implement (pruned DFT factored as
W-direction real matmul -> per-(sample, ky) H-direction/mixing -> W-direction back), against the
oracle's FFT formulation.  Mirrors SpectralCore::build / fno_modes_kernel / fno_layer_kernel in
dlwp_benchmark_amd/csrc/fno2d.hip with numpy float64, so a mismatch here is a maths error, not a
kernel bug."""
import numpy as np
import pytest
import torch

from oracle.restate.fno import neuralop_kept_rows, neuralop_spectral_conv, spectral_conv2d_ref


def pruned_dft_conv(x, wt, rows_in, rows_out, m2, fwd_scale, inv_scale):
    """x [B,C,H,W] float64; wt complex [Ci,Co,M1,M2]."""
    B, C, H, W = x.shape
    M1 = len(rows_in)
    KP = 2 * m2
    w = np.arange(W)
    T = np.zeros((KP, W))
    for ky in range(m2):
        a = 2 * np.pi * ((ky * w) % W) / W
        T[2 * ky] = np.cos(a)
        T[2 * ky + 1] = -np.sin(a)
    Y = np.einsum("bchw,kw->bhck", x, T)                      # [B,H,C,KP]
    Yc = Y[..., 0::2] + 1j * Y[..., 1::2]                      # [B,H,C,M2]
    h = np.arange(H)
    EF = np.exp(-2j * np.pi * ((np.array(rows_in)[:, None] * h[None]) % H) / H)   # [M1,H]
    EI = np.exp(+2j * np.pi * ((np.array(rows_out)[:, None] * h[None]) % H) / H)
    X = fwd_scale * np.einsum("rh,bhck->bcrk", EF, Yc)         # [B,C,M1,M2]
    O = np.einsum("bcrk,cork->bork", X, wt)                    # [B,Co,M1,M2]
    ck = np.array([1.0 if (ky == 0 or (W % 2 == 0 and ky == W // 2)) else 2.0 for ky in range(m2)]) * inv_scale
    Z = np.einsum("rh,bork->bhko", EI, O) * ck[None, None, :, None]   # [B,H,M2,Co]
    Zr = np.zeros((B, H, KP, wt.shape[1]))
    Zr[:, :, 0::2] = Z.real
    Zr[:, :, 1::2] = Z.imag
    return np.einsum("bhko,kw->bohw", Zr, T)


@pytest.mark.parametrize("H,W,nm", [(64, 64, (12, 12)), (32, 64, (12, 12)), (33, 64, (8, 10)), (16, 64, (16, 16))])
def test_neuralop_variant(H, W, nm):
    torch.manual_seed(0)
    B, C = 2, 4
    x = torch.randn(B, C, H, W, dtype=torch.float64)
    mh, mw = min(H, nm[0]), nm[1] // 2 + 1
    wt = torch.randn(C, C, mh, mw, dtype=torch.cdouble)
    ref = neuralop_spectral_conv(x.float(), wt.to(torch.cfloat), None, list(nm)).double().numpy()
    rows_in, rows_out = neuralop_kept_rows(H, nm[0])
    got = pruned_dft_conv(x.numpy(), wt.numpy(), rows_in, rows_out, mw, 1.0 / (H * W), 1.0)
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert err < 2e-6, err


@pytest.mark.parametrize("H,W,m1,m2", [(64, 64, 12, 12), (32, 64, 8, 6), (64, 128, 12, 16)])
def test_in_tree_variant(H, W, m1, m2):
    torch.manual_seed(1)
    B, C = 2, 3
    x = torch.randn(B, C, H, W, dtype=torch.float64)
    w1 = torch.randn(C, C, m1, m2, 2, dtype=torch.float64)
    w2 = torch.randn(C, C, m1, m2, 2, dtype=torch.float64)
    ref = spectral_conv2d_ref(x.float(), w1.float(), w2.float()).double().numpy()
    rows = list(range(m1)) + [H - m1 + r for r in range(m1)]
    wt = torch.cat([torch.view_as_complex(w1), torch.view_as_complex(w2)], dim=2).numpy()
    got = pruned_dft_conv(x.numpy(), wt, rows, rows, m2, 1.0, 1.0 / (H * W))
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert err < 2e-6, err
