"""Micro-benchmark of dlwp_linear_f32 against torch's fp32 GEMM (rocBLAS) at the Linear shapes of the Swin (C3) and Pangu
(C5) blocks.  Usage: python tools/bench_linear.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlwp_benchmark_amd import ops  # noqa: E402

SHAPES = [   # (label, rows, in, out, act, resid)
    ("C3 s0 qkv", 32 * 2048, 96, 288, 0, False), ("C3 s0 proj+res", 32 * 2048, 96, 96, 0, True),
    ("C3 s0 fc1+gelu", 32 * 2048, 96, 384, 1, False), ("C3 s0 fc2+res", 32 * 2048, 384, 96, 0, True),
    ("C3 s1 qkv", 32 * 512, 192, 576, 0, False), ("C3 s1 fc1+gelu", 32 * 512, 192, 768, 1, False),
    ("C3 s1 fc2+res", 32 * 512, 768, 192, 0, True),
    ("C5 l1 qkv", 8 * 32768, 192, 576, 0, False), ("C5 l1 proj+res", 8 * 32768, 192, 192, 0, True),
    ("C5 l1 fc1+gelu", 8 * 32768, 192, 768, 1, False), ("C5 l1 fc2+res", 8 * 32768, 768, 192, 0, True),
    ("C5 l2 qkv", 8 * 8192, 384, 1152, 0, False), ("C5 l2 proj+res", 8 * 8192, 384, 384, 0, True), ("C5 l2 fc1+gelu", 8 * 8192, 384, 1536, 1, False),
    ("C5 l2 fc2+res", 8 * 8192, 1536, 384, 0, True),
]


def timed(fn, reps):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="", help="substring of the shape label")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16", "f16x3"))
    ap.add_argument("--io", action="store_true", help="bf16 precision: the tensors cross HBM as the blocks hand them over -- qkv / fc1 "
                                                       "take a bf16 x (LayerNorm output), fc1 writes bf16, fc2 reads it")
    args = ap.parse_args()
    dev = "cuda:0"
    print(f"{'shape':18s} {'rows':>8s} {'in':>5s} {'out':>5s} | {'hip us':>8s} {'TF/s(6x)':>9s} | {'torch us':>9s} | err hip / torch")
    for label, rows, k, n, act, resid in SHAPES:
        if args.only and args.only not in label:
            continue
        torch.manual_seed(0)
        m = torch.nn.Linear(k, n).to(dev)
        x = torch.randn(rows, k, device=dev)
        r = torch.randn(rows, n, device=dev) if resid else None
        out = torch.empty(rows, n, device=dev)
        xin, oo = x, out
        if args.io and args.precision == "bf16":
            xin = x.to(torch.bfloat16)                          # LayerNorm output / attention output / GELU hidden handed over as bf16
            if "fc1" in label or "qkv" in label:
                oo = torch.empty(rows, n, device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            def hip():
                ops.linear(xin, m, act=act, resid=r, out=oo, precision=args.precision)

            def ref():
                y = torch.nn.functional.linear(x, m.weight, m.bias)
                if act:
                    y = torch.nn.functional.gelu(y)
                if resid:
                    y = y + r
                return y

            t_hip, t_ref = timed(hip, args.reps), timed(ref, args.reps)
            sub = slice(0, 4096)
            want = torch.nn.functional.linear(x[sub].double(), m.weight.double(), m.bias.double())
            if act:
                want = torch.nn.functional.gelu(want)
            if resid:
                want = want + r[sub].double()
            hip()
            e_hip = ((oo[sub].double() - want).norm() / want.norm()).item()
            e_ref = ((ref()[sub].double() - want).norm() / want.norm()).item()
        prods = {"fp32": 6, "f16x3": 3, "bf16": 1}[args.precision]
        tf = prods * 2.0 * rows * k * n / (t_hip * 1e-6) / 1e12
        nbytes = xin.numel() * xin.element_size() + oo.numel() * oo.element_size() + (r.numel() * 4 if resid else 0)
        print(f"{label:18s} {rows:8d} {k:5d} {n:5d} | {t_hip:8.1f} {tf:9.1f} | {t_ref:9.1f} | {e_hip:.1e} / {e_ref:.1e} | "
              f"{nbytes / (t_hip * 1e-6) / 1e12:.2f} TB/s of tensor bytes", flush=True)


if __name__ == "__main__":
    main()
