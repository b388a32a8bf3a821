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


# =====================================================================================================================
# Training through the other hot kernels (reference scripts/train.py:263-271 `loss.backward()`)
#
# The boundary requires a differentiable forward ("wrap kernels in autograd.Function with a PyTorch-op
# backward").  For window attention, the AFNO filter and the padded 3x3 convolution the FORWARD of a training step runs
# the same HIP kernels as inference; the BACKWARD recomputes the operator from the saved inputs with torch operators on
# the GPU (a plain restatement of the reference arithmetic, below) and lets autograd differentiate that.  Nothing here
# imports the oracle: these restatements are part of the product and double as an independent cross-check of the
# kernels (tests/test_training_gpu.py).  Hand-written backward kernels are the next step; the spectral convolution
# already has one (above).
# =====================================================================================================================
import functools

import torch.nn.functional as F

_ACT_FNS = {0: lambda t: t, 1: F.gelu, 2: torch.tanh, 3: F.relu, 4: F.silu}


@functools.lru_cache(maxsize=64)
def _window_tables(grid, padded, pad_lead, window, shift_fwd, use_mask, mask_b1, mask_b2, bias_mode, device_str):
    """bias index [N, N] (long) and region ids [n_pl, n_lat, n_lon, N] of one window geometry, on the device."""
    dev = torch.device(device_str)
    wpl, wlat, wlon = window
    n = wpl * wlat * wlon
    z = torch.arange(n)
    zlon, zlat, zpl = z % wlon, (z // wlon) % wlat, z // (wlon * wlat)
    q, k = slice(None), slice(None)
    if bias_mode == 0:
        idx = (zlat[:, None] - zlat[None, :] + wlat - 1) * (2 * wlon - 1) + (zlon[:, None] - zlon[None, :] + wlon - 1)
    else:   # earth-specific (utils/earth_position_index.py): query coordinate + window * key coordinate, relative longitude
        idx = ((zpl[:, None] + zpl[None, :] * wpl) * wlat * wlat + (zlat[:, None] + zlat[None, :] * wlat)) * (2 * wlon - 1) + \
              (zlon[:, None] - zlon[None, :] + wlon - 1)
    npl, nlat, nlon = padded[0] // wpl, padded[1] // wlat, padded[2] // wlon
    region = None
    if use_mask:
        P = (torch.arange(npl)[:, None] * wpl + zpl[None, :])          # [npl, N]
        A = (torch.arange(nlat)[:, None] * wlat + zlat[None, :])
        O = (torch.arange(nlon)[:, None] * wlon + zlon[None, :])
        rp = (P >= mask_b1[0]).long() + (P >= mask_b2[0]).long()
        ra = (A >= mask_b1[1]).long() + (A >= mask_b2[1]).long()
        ro = (O >= mask_b1[2]).long() + (O >= mask_b2[2]).long()
        region = ((rp[:, None, None, :] * 3 + ra[None, :, None, :]) * 3 + ro[None, None, :, :]).to(dev)
    return idx.to(dev), region


def window_attention_torch(qkv: torch.Tensor, qkv_bias, table: torch.Tensor, spec) -> torch.Tensor:
    """What dlwp_window_attn_f32 computes, with torch operators (differentiable in qkv, qkv_bias, table):
    swin_transformer.py:217-251 + :122-154 (bias_mode 0) / panguweather.py:285-316 + :176-211 (bias_mode 1)."""
    b, l, c3 = qkv.shape
    heads, d = spec.heads, spec.head_dim
    c = heads * d
    pl, lat, lon = spec.grid
    ppl, plat, plon = spec.padded
    wpl, wlat, wlon = spec.window
    x = qkv.view(b, pl, lat, lon, c3)
    f, t, lft = spec.pad_lead
    pads = (0, 0, lft, plon - lon - lft, t, plat - lat - t, f, ppl - pl - f)
    if any(pads):
        # zero-padded tokens enter the qkv Linear as zeros: their q, k, v are the bias
        x = F.pad(x - qkv_bias, pads) + qkv_bias if qkv_bias is not None else F.pad(x, pads)
    sf = tuple(int(s) for s in spec.shift_fwd)
    if any(sf):
        x = torch.roll(x, shifts=(-sf[0], -sf[1], -sf[2]), dims=(1, 2, 3))
    npl, nlat, nlon = ppl // wpl, plat // wlat, plon // wlon
    n = wpl * wlat * wlon
    x = x.view(b, npl, wpl, nlat, wlat, nlon, wlon, 3, heads, d).permute(0, 1, 3, 5, 7, 8, 2, 4, 6, 9)
    x = x.reshape(b, npl, nlat, nlon, 3, heads, n, d)
    q, k, v = x[:, :, :, :, 0] * spec.scale, x[:, :, :, :, 1], x[:, :, :, :, 2]
    idx, region = _window_tables(tuple(spec.grid), tuple(spec.padded), tuple(spec.pad_lead), tuple(spec.window), sf,
                                 bool(spec.use_mask), tuple(int(v_) for v_ in spec.mask_b1),
                                 tuple(int(v_) for v_ in spec.mask_b2), int(spec.bias_mode), str(qkv.device))
    attn = q @ k.transpose(-1, -2)                                      # [b, npl, nlat, nlon, heads, N, N]
    if spec.bias_mode == 0:
        bias = table[idx.reshape(-1)].view(n, n, heads).permute(2, 0, 1)                    # [heads, N, N]
        attn = attn + bias
    else:
        types = npl * nlat
        bias = table[idx.reshape(-1)].view(n, n, types, heads).permute(2, 3, 0, 1)          # [types, heads, N, N]
        attn = attn + bias.view(npl, nlat, 1, heads, n, n)
    if spec.use_mask:
        diff = region.unsqueeze(-1) != region.unsqueeze(-2)                                  # [npl, nlat, nlon, N, N]
        attn = attn + torch.where(diff, -100.0, 0.0).to(attn.dtype).unsqueeze(3)
    attn = torch.softmax(attn, dim=-1)
    o = attn @ v                                                       # [b, npl, nlat, nlon, heads, N, d]
    o = o.view(b, npl, nlat, nlon, heads, wpl, wlat, wlon, d).permute(0, 1, 5, 2, 6, 3, 7, 4, 8)
    o = o.reshape(b, ppl, plat, plon, c)
    sb = tuple(int(s) for s in spec.shift_back)
    if any(sb):
        o = torch.roll(o, shifts=sb, dims=(1, 2, 3))
    o = o[:, f:f + pl, t:t + lat, lft:lft + lon]
    return o.reshape(b, l, c)


class _WindowAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, qkv_bias, table, spec, precision):
        from . import ops

        ctx.spec = spec
        ctx.save_for_backward(qkv, qkv_bias, table)
        with torch.no_grad():
            return ops.window_attention(qkv.detach(), qkv_bias.detach() if qkv_bias is not None else None,
                                        table.detach(), spec, precision=precision)

    @staticmethod
    def backward(ctx, grad_out):
        """HIP backward (dlwp_window_attn_bwd_f32): scores recomputed tile by tile, no [B, heads, N, N] tensor.  Descriptors the
        kernel does not cover (a bias column that does not fit LDS) and DLWP_TRAIN_TORCH_BACKWARD=1 (the cross-check the tests
        use) take the torch recomputation below."""
        from . import ops

        qkv, qkv_bias, table = ctx.saved_tensors
        if not _TORCH_BACKWARD():
            try:
                gq, gb, gt = ops.window_attention_backward(qkv, qkv_bias, table, ctx.spec, grad_out)
                if qkv_bias is not None and gb is None:
                    gb = torch.zeros_like(qkv_bias) if ctx.needs_input_grad[1] else None
                return (gq if ctx.needs_input_grad[0] else None, gb if ctx.needs_input_grad[1] else None,
                        gt if ctx.needs_input_grad[2] else None, None, None)
            except _lib.DlwpError as e:
                if "status -2:" not in str(e):       # DLWP_ERR_UNSUPPORTED only
                    raise
        with torch.enable_grad():
            q_ = qkv.detach().requires_grad_(ctx.needs_input_grad[0])
            b_ = qkv_bias.detach().requires_grad_(ctx.needs_input_grad[1]) if qkv_bias is not None else None
            t_ = table.detach().requires_grad_(ctx.needs_input_grad[2])
            out = window_attention_torch(q_, b_, t_, ctx.spec)
            wrt = [t for t, need in ((q_, ctx.needs_input_grad[0]), (b_, ctx.needs_input_grad[1]), (t_, ctx.needs_input_grad[2]))
                   if need and t is not None]
            grads = list(torch.autograd.grad(out, wrt, grad_out.contiguous(), allow_unused=True)) if wrt else []
        res = []
        for t, need in ((q_, ctx.needs_input_grad[0]), (b_, ctx.needs_input_grad[1]), (t_, ctx.needs_input_grad[2])):
            res.append(grads.pop(0) if (need and t is not None) else None)
        return res[0], res[1], res[2], None, None


def _TORCH_BACKWARD() -> bool:
    import os

    return os.environ.get("DLWP_TRAIN_TORCH_BACKWARD", "0") == "1"


def window_attention(qkv, qkv_bias, table, spec, precision="fp32"):
    """differentiable window attention: HIP forward, recomputed torch backward"""
    return _WindowAttentionFn.apply(qkv, qkv_bias, table, spec, precision)


def afno_filter_torch(x_cf, w1, b1, w2, b2, num_blocks: int, sparsity_threshold: float, hard_thresholding_fraction: float):
    """fourcastnet.py:85-124 on a CHANNELS-FIRST field (without the `+ bias` of :127), torch operators."""
    b, c, h, w = x_cf.shape
    bs = c // num_blocks
    xf = torch.fft.rfft2(x_cf.float(), norm="ortho")                               # [b, c, h, wf]
    xf = xf.permute(0, 2, 3, 1).reshape(b, h, w // 2 + 1, num_blocks, bs)
    total = h // 2 + 1
    kept = int(total * hard_thresholding_fraction)
    rows = slice(max(total - kept, 0), min(total + kept, h))
    xr, xi = xf.real[:, rows, :kept], xf.imag[:, rows, :kept]
    ein = lambda a, m: torch.einsum("...bi,bio->...bo", a, m)
    o1r = F.relu(ein(xr, w1[0]) - ein(xi, w1[1]) + b1[0])
    o1i = F.relu(ein(xi, w1[0]) + ein(xr, w1[1]) + b1[1])
    o2r = ein(o1r, w2[0]) - ein(o1i, w2[1]) + b2[0]
    o2i = ein(o1i, w2[0]) + ein(o1r, w2[1]) + b2[1]
    z = F.softshrink(torch.stack([o2r, o2i], dim=-1), lambd=sparsity_threshold)
    full = torch.zeros(b, h, w // 2 + 1, num_blocks, bs, 2, device=x_cf.device, dtype=torch.float32)
    full = _put(full, rows, kept, z)
    yf = torch.view_as_complex(full).reshape(b, h, w // 2 + 1, c).permute(0, 3, 1, 2)
    return torch.fft.irfft2(yf, s=(h, w), norm="ortho")


def _put(full, rows, kept, z):
    full = full.clone()
    full[:, rows, :kept] = z
    return full


class _AfnoFilterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_cf, w1, b1, w2, b2, num_blocks, lam, frac):
        from . import ops

        ctx.cfg = (num_blocks, lam, frac)
        ctx.save_for_backward(x_cf, w1, b1, w2, b2)
        with torch.no_grad():
            return ops.afno2d_filter_cf(x_cf.detach(), w1.detach(), b1.detach(), w2.detach(), b2.detach(), num_blocks, lam, frac)

    @staticmethod
    def backward(ctx, grad_out):
        """HIP backward (ops.afno2d_filter_backward: the hand-written transforms around dlwp_afno2d_mix_bwd_f32); grids / block
        sizes it does not take and DLWP_TRAIN_TORCH_BACKWARD=1 differentiate the torch form below."""
        from . import ops

        saved = ctx.saved_tensors
        if not _TORCH_BACKWARD():
            res = ops.afno2d_filter_backward(saved[0], grad_out, saved[1], saved[2], saved[3], saved[4], *ctx.cfg)
            if res is not None:
                return (*[g if need else None for g, need in zip(res, ctx.needs_input_grad[:5])], None, None, None)
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(need) for t, need in zip(saved, ctx.needs_input_grad[:5])]
            out = afno_filter_torch(*ins, *ctx.cfg)
            wrt = [t for t, need in zip(ins, ctx.needs_input_grad[:5]) if need]
            grads = list(torch.autograd.grad(out, wrt, grad_out.contiguous(), allow_unused=True)) if wrt else []
        res = [grads.pop(0) if need else None for need in ctx.needs_input_grad[:5]]
        return (*res, None, None, None)


def afno_filter(x_cf, w1, b1, w2, b2, num_blocks, lam, frac):
    return _AfnoFilterFn.apply(x_cf, w1, b1, w2, b2, num_blocks, lam, frac)


def _hpx_pad_torch(x, table):
    """HEALPixPadding(1) as a differentiable gather: x [(B*12), C, H, W], table int32 [12, (H+2)(W+2), 2]."""
    n, c, h, w = x.shape
    bsz = n // 12
    flat = x.view(bsz, 12, c, h * w).permute(0, 2, 1, 3).reshape(bsz, c, 12 * h * w)
    a = table[:, :, 0].long().reshape(-1)
    b_ = table[:, :, 1].long().reshape(-1)
    va = flat[:, :, a]
    vb = flat[:, :, b_.clamp(min=0)]
    v = torch.where((b_ >= 0).view(1, 1, -1), 0.5 * va + 0.5 * vb, va)
    return v.view(bsz, c, 12, h + 2, w + 2).permute(0, 2, 1, 3, 4).reshape(n, c, h + 2, w + 2)


def conv3x3_torch(x0, x1, weight, bias, resid, pre_act: int, act: int, hpx_table=None):
    """pad(1) + Conv2d(3x3) (+ fusions) with torch operators: CylinderPad (utils/utils.py:11-26) or HEALPixPadding."""
    x = x0 if x1 is None else torch.cat([x0, x1], dim=1)
    x = _ACT_FNS[pre_act](x)
    if hpx_table is not None:
        x = _hpx_pad_torch(x, hpx_table)
    else:
        x = F.pad(F.pad(x, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1))
    y = F.conv2d(x, weight, bias)
    if resid is not None:
        y = y + resid
    return _ACT_FNS[act](y)


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, weight, bias, resid, pre_act, act, hpx):
        from . import ops

        ctx.cfg = (pre_act, act, hpx)
        ctx.save_for_backward(x0, x1, weight, bias, resid)
        with torch.no_grad():
            d = lambda t: t.detach() if t is not None else None
            return ops.conv3x3(d(x0), d(weight), d(bias), act=act, x1=d(x1), pre_act=pre_act, resid=d(resid), hpx=hpx)

    @staticmethod
    def backward(ctx, grad_out):
        from . import healpix as _hpx
        from . import ops

        pre_act, act, hpx = ctx.cfg
        saved = ctx.saved_tensors
        if not hpx and not _TORCH_BACKWARD():
            # CylinderPad (circular in longitude, zeros in latitude) + 3x3: the input gradient is the SAME operator with the weights
            # transposed and flipped -- dlwp_conv3x3_ex_f32 again (reference backward: train.py:271 through unet.py:429-555,
            # convlstm.py:82-111); pre- / post-activation derivatives are pointwise torch operators, the weight gradient is one
            # correlation of the padded input with the output gradient (MIOpen through torch, like the other weight gradients)
            x0, x1, weight, bias, resid = saved
            with torch.no_grad():
                xcat = x0 if x1 is None else torch.cat([x0, x1], dim=1)
                gz = grad_out.contiguous()
                if act != 0:
                    z = ops.conv3x3(x0, weight, bias, act=0, x1=x1, pre_act=pre_act, resid=resid)
                    with torch.enable_grad():
                        z_ = z.detach().requires_grad_(True)
                        gz, = torch.autograd.grad(_ACT_FNS[act](z_), z_, gz)
                    gz = gz.contiguous()
                res = [None] * 5
                if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
                    wt = weight.flip(2, 3).transpose(0, 1).contiguous()
                    dxa = ops.conv3x3(gz, wt, None)
                    if pre_act != 0:
                        with torch.enable_grad():
                            xc_ = xcat.detach().requires_grad_(True)
                            dxa, = torch.autograd.grad(_ACT_FNS[pre_act](xc_), xc_, dxa)
                    c0 = x0.shape[1]
                    res[0] = dxa[:, :c0].contiguous() if ctx.needs_input_grad[0] else None
                    res[1] = dxa[:, c0:].contiguous() if (x1 is not None and ctx.needs_input_grad[1]) else None
                if ctx.needs_input_grad[2]:
                    xa = _ACT_FNS[pre_act](xcat)
                    xp = F.pad(F.pad(xa, (1, 1, 0, 0), mode="circular"), (0, 0, 1, 1))
                    res[2] = torch.nn.grad.conv2d_weight(xp, weight.shape, gz)
                if bias is not None and ctx.needs_input_grad[3]:
                    res[3] = gz.sum(dim=(0, 2, 3))
                if resid is not None and ctx.needs_input_grad[4]:
                    res[4] = gz
            return (*res, None, None, None)
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(need) if t is not None else None for t, need in zip(saved, ctx.needs_input_grad[:5])]
            table = _hpx.device_table(ins[0].shape[2], ins[0].shape[3], 1, ins[0].device) if hpx else None
            out = conv3x3_torch(ins[0], ins[1], ins[2], ins[3], ins[4], pre_act, act, table)
            wrt = [t for t, need in zip(ins, ctx.needs_input_grad[:5]) if need and t is not None]
            grads = list(torch.autograd.grad(out, wrt, grad_out.contiguous(), allow_unused=True)) if wrt else []
        res = [grads.pop(0) if (need and t is not None) else None for t, need in zip(ins, ctx.needs_input_grad[:5])]
        return (*res, None, None, None)


def conv3x3(x0, weight, bias, act=0, x1=None, pre_act=0, resid=None, hpx=False):
    return _Conv3x3Fn.apply(x0, x1, weight, bias, resid, pre_act, act, hpx)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b with the HIP Linear kernel in BOTH directions (reference backward: scripts/train.py:271 through the
    nn.Linear layers of swin_transformer.py:21-39, :107-120 and panguweather.py:176-211): the fp32-accurate GEMM of
    csrc/linear.hip (bf16x6) computes the output, the input gradient dX = dY W (the same kernel on the transposed weight) and
    the weight gradient dW = dY^T X (the same kernel with dY^T as the activation and X^T as the "weight"; the reduction runs
    over the tokens).  The bias gradient is a column sum.  Shapes the kernel does not take (in / out features not multiples
    of 32 / 4 in the roles they play in the three products) use the torch operator."""

    @staticmethod
    def supported(rows: int, k: int, n: int) -> bool:
        from . import ops

        return (ops.linear_supported(k, n) and ops.linear_supported(n, k) and ops.linear_supported(rows, k) and
                not _TORCH_BACKWARD())

    @staticmethod
    def forward(ctx, x, weight, bias):
        from . import ops

        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        with torch.no_grad():
            return ops.linear_raw(x.detach(), weight.detach(), bias.detach() if bias is not None else None)

    @staticmethod
    def backward(ctx, gy):
        from . import ops

        x, weight = ctx.saved_tensors
        n, k = weight.shape
        gy2 = gy.reshape(-1, n).contiguous()
        x2 = x.reshape(-1, k)
        gx = gw = gb = None
        with torch.no_grad():
            if ctx.needs_input_grad[0]:
                gx = ops.linear_raw(gy2, weight.t().contiguous(), None).view(x.shape)          # [M, N] x [K, N]^T
            if ctx.needs_input_grad[1]:
                gw = ops.linear_raw(gy2.t().contiguous(), x2.t().contiguous(), None)             # [N, M] x [K, M]^T -> [N, K]
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = gy2.sum(dim=0)
        return gx, gw, gb


def linear_fn(x, weight, bias):
    return _LinearFn.apply(x, weight, bias)


def wants_grad(*tensors) -> bool:
    """True when autograd is recording and one of the tensors takes part: the ops then run their differentiable form."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
