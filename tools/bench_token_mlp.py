#!/usr/bin/env python3
"""Event-timed dlwp_token_mlp_f32 at the FourCastNet C4 shape (32 x 128 x 256 tokens, 64 -> 256 -> 64) next to the
torch ops it replaces (linear -> gelu -> addmm_).  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlwp_benchmark_amd import ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    t, c, hid = 32 * 128 * 256, 64, 256
    torch.manual_seed(0)
    n = torch.randn(t, c, device=dev)
    s = torch.randn(t, c, device=dev)
    w1 = torch.randn(hid, c, device=dev) / 8
    b1 = torch.randn(hid, device=dev) * 0.1
    w2 = torch.randn(c, hid, device=dev) / 16
    b2 = torch.randn(c, device=dev) * 0.1
    packed = ops.TokenMlpWeights().get(w1, w2)
    out = torch.empty_like(n)
    us_fused = timed(lambda: ops.token_mlp(n, s, packed, b1, b2, hid, out=out))

    def unfused():
        h = torch.nn.functional.gelu(torch.nn.functional.linear(n, w1, b1))
        out.copy_(s)
        out.addmm_(h, w2.t())

    us_torch = timed(unfused, reps=5)
    flop = 2.0 * t * c * hid * 2
    print(json.dumps({"tokens": t, "channels": c, "hidden": hid, "fused_us": us_fused, "torch_us": us_torch,
                      "fused_TFLOPs_fp32_equiv": flop / us_fused * 1e-6,
                      "fused_frac_of_bf16_peak_at_6x": 6 * flop / us_fused * 1e-6 / 2500.0,
                      "min_bytes_GBps": 3.0 * t * c * 4 / us_fused * 1e-3}))


if __name__ == "__main__":
    main()
