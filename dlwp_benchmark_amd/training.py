"""Training-side use of the spectral kernels (first slice).

Reference: scripts/train.py:263-271 runs `loss.backward()` through the backbone; for the spectral
convolutions (models/unet/unet.py:46-69 `SpectralConv2d`, and neuralop's SpectralConv inside
`FNO2DModule`, fno.py:38-47) autograd differentiates rfft2 / einsum / irfft2.  Here

  forward        y  = S_W(x)                           dlwp_spectral_conv2d_f32 (pruned-DFT MFMA kernels)
  backward-data  dx = S_{W^H}(dy)                      the SAME kernels: rows_in/rows_out swapped, weights
                                                       conjugate-transposed on the device (exact adjoint: the
                                                       Hermitian weights of the half spectrum cancel per column)
  backward-weight dW[i,o,r,k] = sum_b conj(fwd * X[b,i,rows_in[r],k]) * inv * c_k * DY[b,o,rows_out[r],k]
                                                       two rfft2 (rocFFT) at the kept modes + one einsum

with c_k = 1 for k = 0 and the Nyquist column, 2 otherwise.  The identities are checked against autograd of
the reference operator in tests (CPU, double) and against reference gradients on the GPU.

Everything pointwise around the spectral operator (1x1 convolutions, GELU, residuals) stays in torch ops on
the GPU in training mode; window-attention backward is not built yet.
"""
import ctypes
from typing import List, Sequence

import torch

from . import lib as _lib


def spectral_weight_grad(x, grad_y, rows_in, rows_out, n_cols: int, fwd_scale: float, inv_scale: float):
    """dL/dW as the real view [Ci, Co, n_rows, n_cols, 2] (formula in the module docstring); plain torch, any device."""
    w = x.shape[-1]
    ck = torch.full((n_cols,), 2.0, device=x.device, dtype=x.dtype)
    ck[0] = 1.0
    if w % 2 == 0 and n_cols == w // 2 + 1:
        ck[-1] = 1.0
    ri = torch.as_tensor(list(rows_in), device=x.device)
    ro = torch.as_tensor(list(rows_out), device=x.device)
    xf = torch.fft.rfft2(x)[:, :, ri, :n_cols]
    gf = torch.fft.rfft2(grad_y)[:, :, ro, :n_cols]
    gw = torch.einsum("bixy,boxy->ioxy", xf.conj() * fwd_scale, gf * (ck * inv_scale))
    return torch.view_as_real(gw).contiguous()


class SpectralOperator:
    """Forward + adjoint plans of one mode-truncated spectral convolution geometry (32 -> 32 channels,
    width a multiple of 64: what the HIP kernels are specialised for)."""

    def __init__(self, channels: int, height: int, width: int, rows_in: Sequence[int], rows_out: Sequence[int],
                 n_cols: int, fwd_scale: float, inv_scale: float, device):
        self.channels, self.h, self.w = channels, height, width
        self.rows_in = [int(r) for r in rows_in]
        self.rows_out = [int(r) for r in rows_out]
        self.n_cols = int(n_cols)
        self.fwd_scale, self.inv_scale = float(fwd_scale), float(inv_scale)
        self.device = torch.device(device)
        lib = _lib.load()
        n = len(self.rows_in)
        ri = (ctypes.c_int32 * n)(*self.rows_in)
        ro = (ctypes.c_int32 * n)(*self.rows_out)
        self._fwd, self._adj = ctypes.c_void_p(), ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.dlwp_spectral_conv2d_plan_create_ex(ctypes.byref(self._fwd), channels, channels, height, width,
                                                               n, self.n_cols, ri, ro, self.fwd_scale, self.inv_scale,
                                                               _lib.stream_ptr()), "dlwp_spectral_conv2d_plan_create_ex")
            _lib.check(lib.dlwp_spectral_conv2d_plan_create_ex(ctypes.byref(self._adj), channels, channels, height, width,
                                                               n, self.n_cols, ro, ri, self.fwd_scale, self.inv_scale,
                                                               _lib.stream_ptr()), "dlwp_spectral_conv2d_plan_create_ex")
        self._ws = None

    def __del__(self):
        try:
            lib = _lib.load()
            for p in (self._fwd, self._adj):
                if p:
                    lib.dlwp_spectral_conv2d_plan_destroy(p)
        except Exception:
            pass

    def _run(self, plan, x: torch.Tensor, weight_real: torch.Tensor, adjoint: bool) -> torch.Tensor:
        _lib.require_cuda_tensor(x, "x")
        x = x.contiguous().float()
        w = weight_real.detach().contiguous().float()
        b, c, h, wd = x.shape
        n = len(self.rows_in)
        if (c, h, wd) != (self.channels, self.h, self.w) or tuple(w.shape) != (c, c, n, self.n_cols, 2):
            raise _lib.DlwpError(f"spectral operator built for {self.channels}x{self.h}x{self.w}, "
                                 f"{n}x{self.n_cols} modes; got x {tuple(x.shape)}, weight {tuple(w.shape)}")
        lib = _lib.load()
        nbytes = lib.dlwp_spectral_conv2d_workspace_bytes(plan, b)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.dlwp_spectral_conv2d_set_weights_dev(plan, w.data_ptr(), 1 if adjoint else 0, _lib.stream_ptr()),
                       "dlwp_spectral_conv2d_set_weights_dev")
            _lib.check(lib.dlwp_spectral_conv2d_f32(plan, x.data_ptr(), y.data_ptr(), b, self._ws.data_ptr(), nbytes,
                                                    _lib.stream_ptr()), "dlwp_spectral_conv2d_f32")
        return y

    def forward(self, x, weight_real):
        return self._run(self._fwd, x, weight_real, False)

    def backward_data(self, grad_y, weight_real):
        return self._run(self._adj, grad_y, weight_real, True)

    def backward_weight(self, x, grad_y):
        """[C, C, n_rows, n_cols, 2] gradient of the real view of the weights."""
        return spectral_weight_grad(x.float(), grad_y.float(), self.rows_in, self.rows_out, self.n_cols, self.fwd_scale,
                                    self.inv_scale)


class _SpectralConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight_real, op: SpectralOperator):
        ctx.op = op
        ctx.save_for_backward(x, weight_real)
        return op.forward(x, weight_real)

    @staticmethod
    def backward(ctx, grad_y):
        x, weight_real = ctx.saved_tensors
        op = ctx.op
        grad_y = grad_y.contiguous()
        gx = op.backward_data(grad_y, weight_real) if ctx.needs_input_grad[0] else None
        gw = op.backward_weight(x, grad_y) if ctx.needs_input_grad[1] else None
        return gx, gw, None


def spectral_conv(x: torch.Tensor, weight_real: torch.Tensor, op: SpectralOperator) -> torch.Tensor:
    """Differentiable mode-truncated spectral convolution; weight_real [C, C, n_rows, n_cols, 2]."""
    return _SpectralConvFn.apply(x, weight_real, op)


def pde_arena_rows(height: int, modes1: int):
    """Kept rows of reference unet.py:60-65 (`[:m1]` with weights1, `[-m1:]` with weights2)."""
    rows = list(range(modes1)) + list(range(height - modes1, height))
    return rows, rows
