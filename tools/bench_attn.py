#!/usr/bin/env python3
"""Window-attention calls of the BASELINE configs in isolation (C3 Swin stage 0 / stage 1, C5 Pangu layer 1 / 2),
shifted and unshifted, both precisions: HIP-event time per call and achieved algorithmic TFLOP/s.  For rocprofv3:
`rocprofv3 --kernel-trace --stats -d out -- python3 tools/bench_attn.py --reps 10`."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dlwp_benchmark_amd import ops  # noqa: E402


def swin_spec(h, w, heads, d, shifted):
    sh, sw = (h // 2, w // 2) if shifted else (0, 0)
    return ops.WindowSpec(grid=(1, h, w), padded=(1, h, w), pad_lead=(0, 0, 0), window=(1, h, w), shift_fwd=(0, sh, sw),
                          shift_back=(0, sh, sw), use_mask=shifted, mask_b1=(ops.BIG, 0, 0), mask_b2=(ops.BIG, h - h // 2, w - w // 2),
                          bias_mode=0, heads=heads, head_dim=d, scale=d ** -0.5), (2 * h - 1) * (2 * w - 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", nargs="*")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cases = {"swin_s0": (32, 64, 4, 24), "swin_s1": (16, 32, 4, 48)}
    for name, (h, w, heads, d) in cases.items():
        if args.only and name not in args.only:
            continue
        for shifted in (False, True):
            spec, rows = swin_spec(h, w, heads, d, shifted)
            g = torch.Generator(device="cpu").manual_seed(1)
            qkv = torch.randn(args.batch, h * w, 3 * heads * d, generator=g).to(dev)
            bias = torch.randn(3 * heads * d, generator=g).to(dev) * 0.1
            table = (torch.randn(rows, heads, generator=g) * 0.5).to(dev)
            for prec in ("fp32", "bf16"):
                for _ in range(3):
                    ops.window_attention(qkv, bias, table, spec, precision=prec)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    ops.window_attention(qkv, bias, table, spec, precision=prec)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                fl = 4.0 * args.batch * heads * (h * w) ** 2 * d
                print(json.dumps({"case": name, "shifted": shifted, "precision": prec, "ms_per_call": round(ms, 4),
                                  "algorithmic_TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
