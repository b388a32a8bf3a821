"""Pins the oracle (CPU restatement) against fixtures made by the REAL reference
(oracle/make_golden.py, reference imported from /root/reference).  CPU only."""
import hashlib

import numpy as np
import pytest
import torch

from helpers import load_golden, rel_l2
from oracle.make_golden import spectral_conv2d_case, tensor_sha
from oracle.restate.fno import spectral_conv2d_ref

SPECTRAL_CASES = {
    "c32_64x64_m12": (32, 32, 64, 64, 12, 12, 1),
    "c32_32x64_m8x6": (32, 32, 32, 64, 8, 6, 1),
    "c4_16x16_m4": (4, 4, 16, 16, 4, 4, 2),
}


@pytest.mark.parametrize("tag", list(SPECTRAL_CASES))
def test_spectral_conv2d_restatement_matches_reference(tag):
    ci, co, h, w, m1, m2, b = SPECTRAL_CASES[tag]
    g = load_golden(f"spectral_conv2d_{tag}")
    x, w1, w2 = spectral_conv2d_case(ci, co, h, w, m1, m2, b, tag)
    assert tensor_sha(x, w1, w2) == str(g["sha"]), "filler drifted: regenerate fixtures"
    y = spectral_conv2d_ref(x, w1, w2)
    # same torch build on both sides -> bit-identical
    assert torch.equal(y, torch.from_numpy(g["y"]))
