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


def pangu_spec(lat, lon, heads, shifted, window=(2, 6, 12)):
    """A Pangu block's descriptor (models/pangu.py: one pressure level padded to the window, asymmetric longitude roll)."""
    from dlwp_benchmark_amd.models.pangu import _pad3d

    grid = (1, lat, lon)
    p = _pad3d(grid, window)
    padded = (grid[0] + p[4] + p[5], grid[1] + p[2] + p[3], grid[2] + p[0] + p[1])
    spl, slat, slon = (w // 2 for w in window)
    ppl, plat, plon = padded
    wpl, wlat, wlon = window
    spec = ops.WindowSpec(
        grid=grid, padded=padded, pad_lead=(p[4], p[2], p[0]), window=window,
        shift_fwd=(spl, slat, slat) if shifted else (0, 0, 0), shift_back=(spl, slat, slon) if shifted else (0, 0, 0),
        use_mask=shifted, mask_b1=(ppl - wpl, plat - wlat, plon + slon - wlon) if shifted else (ops.BIG,) * 3,
        mask_b2=(ppl - spl, plat - slat, plon) if shifted else (ops.BIG,) * 3,
        bias_mode=1, heads=heads, head_dim=32, scale=32 ** -0.5)
    return spec, wpl * wpl * wlat * wlat * (2 * wlon - 1), (padded[0] // wpl) * (padded[1] // wlat)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--passes", type=int, default=1, help="repeat the Pangu cases (clock / cache steady state)")
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
    for _ in range(args.passes):
        pangu_cases(args, dev)


def pangu_cases(args, dev):
    for name, (lat, lon, heads) in {"pangu_l1": (128, 256, 6), "pangu_l2": (64, 128, 12)}.items():
        if args.only and name not in args.only:
            continue
        for shifted in (False, True):
            spec, rows, types = pangu_spec(lat, lon, heads, shifted)
            g = torch.Generator(device="cpu").manual_seed(1)
            batch = 8
            qkv = torch.randn(batch, lat * lon, 3 * heads * 32, generator=g).to(dev)
            bias = torch.randn(3 * heads * 32, generator=g).to(dev) * 0.1
            table = (torch.randn(rows, types, heads, generator=g) * 0.5).to(dev)
            for prec in ("fp32", "bf16"):
                ms = timed(lambda: ops.window_attention(qkv, bias, table, spec, precision=prec), args.reps)
                nwin = (spec.padded[1] // 6) * (spec.padded[2] // 12)
                fl = 4.0 * batch * heads * nwin * 144 * 144 * 32      # the reference's arithmetic: all 144 x 144 scores per window
                print(json.dumps({"case": name, "shifted": shifted, "precision": prec, "ms_per_call": round(ms, 4),
                                  "algorithmic_TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
