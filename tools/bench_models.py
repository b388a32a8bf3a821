#!/usr/bin/env python3
"""Throughput of the non-headline BASELINE configs (C3 Swin 32x64, C4 FourCastNet 128x256, C5 Pangu
128x256x13ch, C1 U-Net 64x64) through the HIP path, one GPU.  Not the contract bench (bench.py is);
prints one JSON line per config.  `--profile` adds a torch.profiler kernel table per config."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dlwp_benchmark_amd import models as M  # noqa: E402
from dlwp_benchmark_amd.synthetic import navier_stokes, weatherbench  # noqa: E402
from dlwp_benchmark_amd.weights import fill_state_dict  # noqa: E402

CONFIGS = {
    "C3_swin_32x64": (M.SwinTransformer, dict(context_size=1, img_height=32, img_width=64, patch_size=1, constant_channels=4,
                                              prescribed_channels=1, prognostic_channels=3, embed_dim=96, depths=[4, 4],
                                              num_heads=[4, 4], mlp_ratio=4, qkv_bias=True, drop_path_rate=0.2,
                                              norm_layer="nn.LayerNorm", patch_norm=True), 32, 12, (32, 64)),
    "C4_fourcastnet_128x256": (M.FourCastNet, dict(img_height=128, img_width=256, patch_size=[1, 1], constant_channels=4,
                                                   prescribed_channels=1, prognostic_channels=3, filter="AFNO2D",
                                                   embed_dim=64, depth=4, mlp_ratio=4.0, num_blocks=4,
                                                   sparsity_threshold=0.01, hard_thresholding_fraction=1.0,
                                                   context_size=1, use_pos_embed=True), 32, 20, (128, 256)),
    "C5_pangu_128x256x13": (M.PanguWeather, dict(constant_channels=4, prescribed_channels=1, prognostic_channels=13,
                                                 embed_dim=192, num_heads=[6, 12, 12, 6], window_size=[2, 6, 12],
                                                 patch_size=[1, 1], n_lat=128, n_lon=256, context_size=1), 8, 5, (128, 256)),
    "C1_unet_64x64": (M.UNet, dict(constant_channels=0, prescribed_channels=0, prognostic_channels=1,
                                   hidden_channels=[8, 16, 32, 64], n_convolutions=2, activation="th.nn.GELU()",
                                   context_size=1), 32, 1, (64, 64)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--attn-bf16", action="store_true", help="bf16 MFMA window attention (Swin / Pangu)")
    ap.add_argument("--graphs", action="store_true", help="replay every backbone step as a HIP graph")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, (cls, cfg, batch, steps, (h, w)) in CONFIGS.items():
        if args.only and name not in args.only:
            continue
        model = cls(**cfg)
        fill_state_dict(model, gain=0.7)
        model = model.to(dev).eval()
        if args.graphs:
            model.set_step_graphs(True)
        if args.attn_bf16 and hasattr(model, "set_attention_precision"):
            model.set_attention_precision("bf16")
        if cfg["constant_channels"] == 0:
            c, p, g = navier_stokes(batch, steps + 1, h, w, channels=cfg["prognostic_channels"])
        else:
            c, p, g = weatherbench(batch, steps + 1, h, w, prognostic_channels=cfg["prognostic_channels"])
        d = lambda t: t.to(dev) if t is not None else None
        c, p, g = d(c), d(p), d(g)
        out = model(constants=c, prescribed=p, prognostic=g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = model(constants=c, prescribed=p, prognostic=g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        res = {"config": name, "graphs": bool(args.graphs), "attention": "bf16" if args.attn_bf16 else "fp32", "batch": batch, "rollout_steps": steps, "ms_per_rollout": dt * 1e3,
               "ms_per_step": dt * 1e3 / steps, "cell_steps_per_s": batch * h * w * steps / dt,
               "finite": bool(torch.isfinite(out).all())}
        print(json.dumps(res), flush=True)
        if args.profile:
            from torch.profiler import ProfilerActivity, profile

            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                model(constants=c, prescribed=p, prognostic=g)
                torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70), flush=True)
        del model, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
