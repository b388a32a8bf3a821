"""SpectralConv2d -- drop-in for reference models/unet/unet.py:19-69 (the only in-tree spectral
convolution; PDE-Arena style) on MI355X.  Same constructor, parameter names (`weights1`,
`weights2` of shape [Ci, Co, m1, m2, 2]) and forward(x) as the reference class; the forward runs
`dlwp_spectral_conv2d_f32` (pruned-DFT fp32 MFMA kernels) instead of rfft2 / einsum / irfft2.
"""
import ctypes

import torch
from torch import nn

from .. import lib as _lib


def _ops_epoch():
    from .. import ops

    return ops.pack_epoch()


class SpectralConv2d(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, modes1: int, modes2: int):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.modes1 = modes1
        self.modes2 = modes2
        self.scale = 1 / (in_channels * out_channels)  # unet.py:38
        self.weights1 = nn.Parameter(self.scale * torch.rand(in_channels, out_channels, modes1, modes2, 2))
        self.weights2 = nn.Parameter(self.scale * torch.rand(in_channels, out_channels, modes1, modes2, 2))
        self._plan = None
        self._plan_key = None
        self._ws = None

    def _destroy_plan(self):
        if self._plan is not None:
            try:
                _lib.load().dlwp_spectral_conv2d_plan_destroy(self._plan)
            except Exception:
                pass
            self._plan = None

    def __del__(self):
        try:  # at interpreter shutdown torch internals may already be torn down
            self._destroy_plan()
        except Exception:
            pass

    def _get_plan(self, h, w, device):
        key = (h, w, str(device), self.weights1._version, self.weights1.data_ptr(), self.weights2._version,
               self.weights2.data_ptr(), _ops_epoch())
        if self._plan is not None and key == self._plan_key:
            return self._plan
        self._destroy_plan()
        lib = _lib.load()
        w1 = self.weights1.detach().to("cpu", torch.float32).contiguous()
        w2 = self.weights2.detach().to("cpu", torch.float32).contiguous()
        plan = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.dlwp_spectral_conv2d_plan_create(
                ctypes.byref(plan), self.in_channels, self.out_channels, h, w, self.modes1, self.modes2,
                w1.data_ptr(), w2.data_ptr(), _lib.stream_ptr()), "dlwp_spectral_conv2d_plan_create")
        self._plan, self._plan_key = plan, key
        return plan

    def _forward_train(self, x):
        """Differentiable path: HIP forward and backward-data, rocFFT + einsum weight gradient."""
        from .. import training as T

        _, _, h, w = x.shape
        key = (h, w, str(x.device))
        if getattr(self, "_train_op_key", None) != key:
            rows, _ = T.pde_arena_rows(h, self.modes1)
            self._train_op = T.SpectralOperator(self.in_channels, h, w, rows, rows, self.modes2, 1.0, 1.0 / float(h * w),
                                                x.device)
            self._train_op_key = key
        return T.spectral_conv(x, torch.cat([self.weights1, self.weights2], dim=2), self._train_op)

    def forward(self, x, x_dim=None, y_dim=None):
        _lib.require_cuda_tensor(x, "x")
        if torch.is_grad_enabled() and (x.requires_grad or self.weights1.requires_grad and self.training):
            if self.in_channels != self.out_channels:
                raise _lib.DlwpError("training path needs in_channels == out_channels (== 32)")
            return self._forward_train(x)
        with torch.no_grad():
            return self._forward_infer(x)

    def _forward_infer(self, x):
        x = x.contiguous()
        b, c, h, w = x.shape
        if c != self.in_channels:
            raise _lib.DlwpError(f"x has {c} channels, layer expects {self.in_channels}")
        lib = _lib.load()
        plan = self._get_plan(h, w, x.device)
        nbytes = lib.dlwp_spectral_conv2d_workspace_bytes(plan, b)
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != x.device:
            self._ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=x.device)
        y = torch.empty(b, self.out_channels, h, w, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(lib.dlwp_spectral_conv2d_f32(plan, x.data_ptr(), y.data_ptr(), b, self._ws.data_ptr(), nbytes,
                                                    _lib.stream_ptr()), "dlwp_spectral_conv2d_f32")
        return y
