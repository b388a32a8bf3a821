"""CPU restatement of the evaluation metrics (TEST INFRASTRUCTURE, see oracle/__init__.py).

reference scripts/evaluate.py:786-821 (`compute_metrics`) with the de-normalisation of :281-296, restated
with numpy instead of xarray (absent from this image).  PARITY UNPINNED: the reference holds no fixture
for its metrics and xarray cannot be imported here; the formulas are Eq. (2) / (A1) of arXiv:2002.00469
as written at :788-821."""
import numpy as np


def lat_weighted_metrics(outputs, targets, lats_deg, std=None, mean=None, climatology=None):
    """outputs/targets [N, K, C, H, W] (normalised); returns (rmse [K, C], acc [K, C] or None)."""
    outputs = np.asarray(outputs, dtype=np.float64)
    targets = np.asarray(targets, dtype=np.float64)
    c = outputs.shape[2]
    std = np.ones(c) if std is None else np.asarray(std, dtype=np.float64)
    mean = np.zeros(c) if mean is None else np.asarray(mean, dtype=np.float64)
    bc = (None, None, slice(None), None, None)
    outputs = outputs * std[bc] + mean[bc]          # evaluate.py:281-296
    targets = targets * std[bc] + mean[bc]
    lats = np.deg2rad(np.asarray(lats_deg, dtype=np.float64))
    w = (np.cos(lats) / np.mean(np.cos(lats)))[None, None, None, :, None]   # evaluate.py:788-790
    rmse = np.sqrt((w * (outputs - targets) ** 2).mean(axis=(0, 3, 4)))       # :798-800
    acc = None
    if climatology is not None:
        clim = np.asarray(climatology, dtype=np.float64)[None] * std[bc] + mean[bc]
        do, dt = outputs - clim, targets - clim                                # :811-812
        nom = (w * do * dt).mean(axis=(0, 3, 4))
        den = np.sqrt((w * do ** 2).mean(axis=(0, 3, 4)) * (w * dt ** 2).mean(axis=(0, 3, 4)))
        acc = nom / den
    return rmse, acc
