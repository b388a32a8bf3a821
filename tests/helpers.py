"""Shared test helpers (metrics, fillers, golden access)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """|| a - b ||_2 / || b ||_2 in float64 (b = reference)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float(torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b))


def per_step_rel_l2(a: torch.Tensor, b: torch.Tensor):
    return [rel_l2(a[:, t], b[:, t]) for t in range(b.shape[1])]


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def fno_std_fn(gain):
    def f(name, shape):
        if "convs.weight" in name:
            return gain / shape[0] ** 0.5
        return None
    return f
