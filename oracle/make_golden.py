"""Generates tests/golden/*.npz by running the REAL reference code (imported from
/root/reference with the stand-ins of oracle/ref_import.py).  TEST INFRASTRUCTURE: run in the
build container only (`python -m oracle.make_golden`); the reference never travels, only the
small output arrays written here do.

Inputs and weights are NOT stored: they are regenerated bit-identically at test time by the
counter-based fillers in dlwp_benchmark_amd/weights.py (numpy-only, same image on both machines);
each fixture stores the SHA-256 of the weight blob it was made with so drift is detected.
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dlwp_benchmark_amd import weights as W  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _save(name, **arrays):
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def tensor_sha(*tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


# ------------------------------------------------------------------------------------------
# SpectralConv2d (reference models/unet/unet.py:19-69)
# ------------------------------------------------------------------------------------------
def spectral_conv2d_case(ci, co, h, w, m1, m2, batch, tag):
    x = W.normal(f"golden/spectral/{tag}/x", (batch, ci, h, w), 1.0)
    w1 = W.normal(f"golden/spectral/{tag}/w1", (ci, co, m1, m2, 2), 1.0 / ci)
    w2 = W.normal(f"golden/spectral/{tag}/w2", (ci, co, m1, m2, 2), 1.0 / ci)
    return x, w1, w2


def gen_spectral(ref):
    SpectralConv2d = ref["unet"].SpectralConv2d
    for tag, (ci, co, h, w, m1, m2, b) in {
        "c32_64x64_m12": (32, 32, 64, 64, 12, 12, 1),
        "c32_32x64_m8x6": (32, 32, 32, 64, 8, 6, 1),
        "c4_16x16_m4": (4, 4, 16, 16, 4, 4, 2),
    }.items():
        x, w1, w2 = spectral_conv2d_case(ci, co, h, w, m1, m2, b, tag)
        mod = SpectralConv2d(ci, co, m1, m2)
        with torch.no_grad():
            mod.weights1.copy_(w1)
            mod.weights2.copy_(w2)
            y = mod(x)
        _save(f"spectral_conv2d_{tag}", y=y.numpy(), sha=np.array(tensor_sha(x, w1, w2)))


def main():
    if not ref_import.reference_available():
        raise SystemExit("reference tree not available: golden fixtures can only be regenerated in the build container")
    ref = ref_import.load_reference()
    torch.manual_seed(1234)
    gen_spectral(ref)


if __name__ == "__main__":
    main()
