"""Synthetic inputs of the shapes the reference dataset hands to model.forward.

No dataset is available offline (reference dlwpbench/README.md:10-49), so benchmarks and
parity tests use the seeded generators of SURVEY.md section 8d:

* Navier-Stokes 64x64 (BASELINE configs C1/C2): band-limited periodic Gaussian random field,
  spectrum ~ (k^2 + tau^2)^-alpha, tau=7, alpha=2.5, standardised; no constants / prescribed.
* WeatherBench-like (C3-C5): constants = {orography-like, land mask, lat2d, lon2d},
  prescribed = analytic insolation, prognostic = smooth unit-variance fields; the tuple layout
  follows WeatherBenchDataset.__getitem__ (reference data/datasets/datasets.py:330-416):
  constants [B,1,Cc,H,W], prescribed [B,T,Cp,H,W], prognostic [B,T,Cg,H,W], fp32.
"""
import math

import numpy as np
import torch


def _grf(rng: np.random.Generator, n: int, h: int, w: int, tau: float = 7.0, alpha: float = 2.5) -> np.ndarray:
    ky = np.fft.fftfreq(h, d=1.0 / h)[:, None]
    kx = np.fft.rfftfreq(w, d=1.0 / w)[None, :]
    amp = (kx * kx + ky * ky + tau * tau) ** (-alpha / 2.0)
    amp[0, 0] = 0.0
    coef = (rng.standard_normal((n, h, w // 2 + 1)) + 1j * rng.standard_normal((n, h, w // 2 + 1))) * amp
    f = np.fft.irfft2(coef, s=(h, w))
    f -= f.mean(axis=(-2, -1), keepdims=True)
    f /= f.std(axis=(-2, -1), keepdims=True) + 1e-12
    return f.astype(np.float32)


def navier_stokes(batch: int, steps: int, h: int = 64, w: int = 64, channels: int = 1, seed: int = 1234):
    """Returns (constants=None, prescribed=None, prognostic [B, T, C, H, W])."""
    rng = np.random.default_rng(seed)
    f = _grf(rng, batch * steps * channels, h, w).reshape(batch, steps, channels, h, w)
    return None, None, torch.from_numpy(f)


def weatherbench(batch: int, steps: int, h: int, w: int, prognostic_channels: int = 3,
                 constant_channels: int = 4, prescribed_channels: int = 1, seed: int = 1234):
    """Returns (constants [B,1,Cc,H,W], prescribed [B,T,Cp,H,W], prognostic [B,T,Cg,H,W])."""
    rng = np.random.default_rng(seed)
    lat = np.linspace(-90.0 + 90.0 / h, 90.0 - 90.0 / h, h, dtype=np.float64)
    lon = np.linspace(0.0, 360.0 - 360.0 / w, w, dtype=np.float64)
    lat2d, lon2d = np.meshgrid(lat, lon, indexing="ij")
    oro = _grf(rng, 1, h, w, tau=3.0, alpha=2.0)[0]
    lsm = (oro > 0.3).astype(np.float32)
    cons = [oro, lsm, (lat2d / 90.0).astype(np.float32), ((lon2d - 180.0) / 180.0).astype(np.float32)]
    while len(cons) < constant_channels:
        cons.append(_grf(rng, 1, h, w)[0])
    constants = np.stack(cons[:constant_channels], 0)[None, None].repeat(batch, 0) if constant_channels else None
    prescribed = None
    if prescribed_channels:
        t = np.arange(steps, dtype=np.float64)[:, None, None]
        omega = 2.0 * math.pi / 4.0  # one revolution per four 6-hourly steps
        sol = np.cos(np.deg2rad(lat2d))[None] * np.maximum(0.0, np.cos(np.deg2rad(lon2d)[None] - omega * t))
        sol = (sol - sol.mean()) / (sol.std() + 1e-12)
        prescribed = np.broadcast_to(sol[None, :, None].astype(np.float32),
                                     (batch, steps, prescribed_channels, h, w)).copy()
    prog = _grf(rng, batch * steps * prognostic_channels, h, w, tau=4.0, alpha=2.0)
    prog = prog.reshape(batch, steps, prognostic_channels, h, w)
    tt = torch.from_numpy
    return (tt(constants.astype(np.float32)) if constants is not None else None,
            tt(prescribed) if prescribed is not None else None, tt(prog))
