"""Deterministic, numpy-only weight filler.

There are no checkpoints offline (reference dlwpbench/README.md:94 points at an external
link), so bench.py, smoke() and the full-config parity tests fill a model's state dict with a
counter-based generator that gives the same numbers on every machine: value i of tensor `name`
is a function of (seed, name, i) only.  Scales follow the
usual fan-in rule so that per-step increments are a visible fraction of the state (a filler
with std 0.02 everywhere makes every backbone an identity map to 1e-8, which would make the
rollout parity checks vacuous); index buffers are untouched.
"""
import hashlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _name_seed(name: str, seed: int) -> np.uint64:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def uniform01(name: str, n: int, seed: int = 1234) -> np.ndarray:
    """n float64 values in (0,1), a pure function of (seed, name, index)."""
    base = _name_seed(name, seed)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(2) + base) & _M64
    bits = _splitmix64(idx)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(name: str, shape, std: float = 1.0, seed: int = 1234) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(name + "/u1", n, seed)
    u2 = uniform01(name + "/u2", n, seed)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy((z * std).astype(np.float32).reshape(shape))


def default_std(name: str, shape, gain: float = 1.0) -> float:
    """std = gain / sqrt(fan_in) with fan_in = numel / shape[0] for >=2-D tensors (what
    torch's default Linear/Conv initialisers scale like), 0.02 for everything else.  Gives
    per-step increments of O(0.1) of the state, so rollout parity checks are not dominated by
    the identity (residual) term."""
    if len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= int(d)
        return gain / float(np.sqrt(max(fan_in, 1)))
    return 0.02


def value_for(name: str, shape, is_complex: bool = False, seed: int = 1234, std_fn=None, gain: float = 1.0):
    """The filler's value for one named tensor (shared by fill_state_dict and fill_by_spec).

    Rules: *norm*.weight -> 1 + N(0, 0.05), *norm*.bias -> N(0, 0.05) (so affine terms are
    exercised); otherwise N(0, std) with std = std_fn(name, shape) if that returns a number,
    else `default_std`."""
    shape = tuple(int(d) for d in shape)
    lname = name.lower()
    std = std_fn(name, shape) if std_fn is not None else None
    if is_complex:
        if std is None:
            std = default_std(name, shape, gain)
        return torch.complex(normal(name + "/re", shape, std, seed), normal(name + "/im", shape, std, seed))
    is_norm = ("norm" in lname) or (".ln" in lname)
    if std is not None:
        return normal(name, shape, std, seed)
    if is_norm and name.endswith("weight"):
        return 1.0 + normal(name, shape, 0.05, seed)
    if is_norm and name.endswith("bias"):
        return normal(name, shape, 0.05, seed)
    return normal(name, shape, default_std(name, shape, gain), seed)


def fill_state_dict(model: torch.nn.Module, seed: int = 1234, std_fn=None, gain: float = 1.0) -> str:
    """Fills every floating-point / complex PARAMETER in place (see `value_for`); returns the SHA-256
    of the blob.  Integer buffers (index tables) and float buffers (masks) are left as constructed."""
    sha = hashlib.sha256()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not (p.is_floating_point() or p.is_complex()):
                continue
            v = value_for(name, tuple(p.shape), p.is_complex(), seed, std_fn, gain)
            p.copy_(v.to(p.dtype))
            sha.update(name.encode())
            sha.update((torch.view_as_real(v) if v.is_complex() else v).numpy().tobytes())
    return sha.hexdigest()


def fill_by_spec(spec, seed: int = 1234, std_fn=None, gain: float = 1.0):
    """spec: iterable of (name, shape) of float parameters -> (state dict, sha256), identical to what
    `fill_state_dict` writes into a module whose named_parameters() match the spec."""
    sha = hashlib.sha256()
    out = {}
    for name, shape in spec:
        v = value_for(name, tuple(shape), False, seed, std_fn, gain)
        out[name] = v
        sha.update(name.encode())
        sha.update(v.numpy().tobytes())
    return out, sha.hexdigest()
