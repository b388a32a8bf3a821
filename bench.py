#!/usr/bin/env python3
"""Benchmark of the autoregressive-rollout hot path on MI355X (BASELINE.json metric: grid-cells x rollout-steps / s).

  python bench.py [--gpus 1 --steps K --warmup W]                 headline: C2 (BASELINE configs[1])
  python bench.py --config C3|C4|C5 [...]                         the other GPU configs, same JSON contract
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W [--config C4]

Workloads (one bench "step" = one WHOLE rollout of the rank's batch, inputs resident in HBM):
  C2  FNO2d modes=12 hidden=32 lift/proj=256 layers=4, Navier-Stokes 64x64, B=32 per GPU, 20-step rollout, fp32
  C3  SwinTransformer 32x64, 3 prognostic vars, B=32 per GPU, 12 steps (72 h), bf16
  C4  FourCastNet/AFNO 128x256, B=32 per GPU (256 initial conditions over 8 GPUs), 20 steps, fp32
  C5  Pangu-Weather 128x256x13, B=8 per GPU, 5 steps (5 days), bf16

N > 1 (reference call site scripts/evaluate.py:205-244): every rank rolls out its own contiguous shard of initial
conditions (weak scaling, no collective inside a step); per rollout the squared-error sums are reduced on the device and
accumulated per rank, ONE RCCL all-reduce per evaluation moves them (`--collect gather`: one chunked all-gather of the
trajectories per rollout instead).

THE LAST STDOUT LINE is the contract line: one JSON object of < 4 KB (`compact_line`; the driver keeps only a short tail
of stdout).  Everything verbose (per-kernel tables, the other configs' full entries) goes to a side file
(`--detail`, default profiles/bench_detail_last.json) and never to stdout.

Extra legs, rank 0 at N = 1 only:
  roofline      HIP-event timing of the dominant kernel on its launch stream (dlwp_fno2d_rollout_profiled_f32 / KernelTimer)
  cpu_baseline  the oracle (PyTorch CPU restatement of the reference forward) timed on the host cores on a bounded
                sample; also yields the per-step rel-L2 of the HIP trajectory
  other_configs (C2 default run only) C1 / C3 / C4 / C5 on the same GPU, compact summary in the line
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # same guide: fp32 matrix peak (v_mfma_f32_16x16x4_f32)
MFMA_16BIT_PEAK_TF = 2500.0  # same guide: dense bf16 / f16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
LINE_LIMIT = 4000           # bytes of the contract line (the driver's tail is 8 KB; VERDICT r02 asks for <= 4 KB)

MODEL_KW = dict(n_modes=[12, 12], constant_channels=0, prescribed_channels=0, prognostic_channels=1,
                hidden_channels=32, lifting_channels=256, projection_channels=256, n_layers=4, context_size=1)

# matrix instructions a form executes per algorithmic fp32 product
FORM_PRODUCTS = {"bf16x6": 6, "f16x3": 3, "bf16": 1, "fp32": 6, "fp32_mfma": 1}


def std_fn(name, shape):
    if "convs.weight" in name:
        return 0.85 / shape[0] ** 0.5
    return None


def build_model(device, precision_form=None):
    from dlwp_benchmark_amd.models import FNO2DModule
    from dlwp_benchmark_amd.weights import fill_state_dict

    m = FNO2DModule(**MODEL_KW)
    sha = fill_state_dict(m, std_fn=std_fn, gain=0.85)
    m.set_execution_form(check="deferred")      # asynchronous rollouts, verified by model.verify() in finish()
    form = precision_form or os.environ.get("DLWP_BENCH_PRECISION")   # A/B of the product forms (default: the module's)
    if form:
        m.set_execution_form(precision_form=form)
    return m.to(device).eval(), sha


def algorithmic_work(B, H, W, kw, n_rows, n_cols):
    """Per-LAUNCH algorithmic bytes / flops of each kernel class."""
    P = B * H * W
    ch, cl, cp = kw["hidden_channels"], kw["lifting_channels"], kw["projection_channels"]
    cin = kw["constant_channels"] + (kw["prescribed_channels"] + kw["prognostic_channels"]) * kw["context_size"]
    cout = kw["prognostic_channels"]
    return {
        "lift": {"bytes": 4 * P * (cin + ch) + 4 * (cl * cin + cl + ch * cl + ch),
                 "flops": 2 * P * (cin * cl + cl * ch)},
        "modes": {"bytes": 8 * ch * ch * n_rows * n_cols, "flops": 8 * B * ch * ch * n_rows * n_cols},
        "layer": {"bytes": 4 * P * 2 * ch + 4 * ch * ch,
                  "flops": 2 * P * ch * ch + 2 * 2 * P * ch * 2 * n_cols},
        "proj": {"bytes": 4 * P * (ch + 2 * cout) + 4 * (cp * ch + cp + cout * cp + cout),
                 "flops": 2 * P * (ch * cp + cp * cout)},
    }


def profile_kernels(model, prog, repeats):
    """Runs the rollout with per-launch HIP event brackets; returns {class: (avg_ms, launches)}."""
    from dlwp_benchmark_amd import lib as L

    lib = L.load()
    b, t, cg, h, w = prog.shape
    plan = model._get_plan(h, w, prog.device)
    out = torch.empty(b, t - model.context_size, cg, h, w, device=prog.device)
    nbytes = lib.dlwp_fno2d_workspace_bytes(plan, b)
    ws = model._workspace(nbytes, prog.device)
    ms = (ctypes.c_double * 5)()
    cnt = (ctypes.c_int32 * 5)()
    tot = [0.0] * 5
    n = [0] * 5
    for _ in range(repeats):
        L.check(lib.dlwp_fno2d_rollout_profiled_f32(plan, None, 0, None, 0, prog.data_ptr(), cg, b, t,
                                                    model.context_size, out.data_ptr(), ws.data_ptr(), nbytes,
                                                    L.stream_ptr(), ms, cnt), "profiled rollout")
        for i in range(5):
            tot[i] += ms[i]
            n[i] += cnt[i]
    names = ["lift", "modes", "layer", "proj"]
    # An event bracket around one launch reads kernel time + event-marker latency.  An EMPTY bracket
    # (two markers back to back) is measured in the same pass; subtracting HALF of it (one marker)
    # reproduces the rocprofv3 --kernel-trace averages of all four kernels within 2 %
    # (profiles/r01_e_fno2d_kernel_stats.csv: 32.7 / 9.1 / 10.9 / 25.8 us).
    empty = tot[4] / max(n[4], 1)
    return {names[i]: (max(tot[i] / max(n[i], 1) - 0.5 * empty, 0.0), n[i] // repeats) for i in range(4)}, empty


def cpu_baseline(state_dict, prog_cpu, rollout_steps, timed=3):
    """Times the oracle on the host cores (1 warm-up rollout + `timed` timed rollouts, median);
    returns (cell-steps/s, median seconds, all seconds, trajectory)."""
    from oracle.restate.fno import FNO2DModuleRef  # checker / reported baseline only

    ref = FNO2DModuleRef(**MODEL_KW).eval()
    ref.load_state_dict(state_dict)
    secs = []
    with torch.no_grad():
        traj = ref(prognostic=prog_cpu)  # warm-up: one whole rollout (also the parity trajectory)
        for _ in range(timed):
            t0 = time.perf_counter()
            ref(prognostic=prog_cpu)
            secs.append(time.perf_counter() - t0)
    med = sorted(secs)[len(secs) // 2]
    b, _, _, h, w = prog_cpu.shape
    return b * h * w * rollout_steps / med, med, secs, traj


def config_table():
    """The BASELINE configs beside the headline at their BASELINE sizes:
    tag -> (class, ctor kwargs, batch per GPU, rollout steps, (H, W), golden fixture of the same architecture + filler weights,
            filler gain, the precision BASELINE.json names for the config)."""
    return {
        "C1": ("UNet", dict(constant_channels=0, prescribed_channels=0, prognostic_channels=1,
                            hidden_channels=[8, 16, 32, 64], n_convolutions=2, activation="th.nn.GELU()",
                            context_size=1), 32, 1, (64, 64), "unet_c1_64x64", 1.0, "fp32"),
        "C3": ("SwinTransformer", dict(context_size=1, img_height=32, img_width=64, patch_size=1, constant_channels=4,
                                       prescribed_channels=1, prognostic_channels=3, embed_dim=96, depths=[4, 4],
                                       num_heads=[4, 4], mlp_ratio=4, qkv_bias=True, drop_path_rate=0.2,
                                       norm_layer="nn.LayerNorm", patch_norm=True), 32, 12, (32, 64), "swin_c3_full", 0.7, "bf16"),
        "C4": ("FourCastNet", dict(img_height=128, img_width=256, patch_size=[1, 1], constant_channels=4,
                                   prescribed_channels=1, prognostic_channels=3, filter="AFNO2D",
                                   embed_dim=64, depth=4, mlp_ratio=4.0, num_blocks=4,
                                   sparsity_threshold=0.01, hard_thresholding_fraction=1.0,
                                   context_size=1, use_pos_embed=True), 32, 20, (128, 256), "afno_c4_full", 0.7, "fp32"),
        "C5": ("PanguWeather", dict(constant_channels=4, prescribed_channels=1, prognostic_channels=13,
                                    embed_dim=192, num_heads=[6, 12, 12, 6], window_size=[2, 6, 12],
                                    patch_size=[1, 1], n_lat=128, n_lon=256, context_size=1), 8, 5, (128, 256),
               "pangu_c5_full", 0.7, "bf16"),
    }


# precision variants of the mirrors: name -> (attention precision, Linear form, block-tail MLP form, rel-L2 bound, dtype label)
VARIANTS = {
    "fp32": ("fp32", "bf16x6", "bf16x6", 1e-5, "f32 (bf16x6 split products, fp32 accumulate)"),
    "f16x3": ("fp32", "f16x3", "f16x3", 1e-5, "f32 (f16x3 split products, fp32 accumulate)"),
    "bf16attn": ("bf16", "bf16x6", "bf16x6", 5e-3, "bf16 window attention, f32 elsewhere"),
    "bf16": ("bf16", "bf16", "bf16x6", 5e-3, "bf16 (attention + Linear operands; fp32 accumulate, LayerNorm, residual)"),
}


def apply_variant(model, variant):
    attn, lin, mlp, _, _ = VARIANTS[variant]
    if hasattr(model, "set_attention_precision"):
        model.set_attention_precision(attn)
        model.set_linear_form(lin)
    if hasattr(model, "set_mlp_form"):
        model.set_mlp_form(mlp)
    return model


def variants_of(cls):
    if cls in ("SwinTransformer", "PanguWeather"):
        return ["fp32", "f16x3", "bf16attn", "bf16"]
    if cls == "FourCastNet":
        return ["fp32", "f16x3"]
    return ["fp32"]


def _golden_inputs(cfg, batch, frames):
    """the seeded inputs the committed fixtures were made with (oracle/make_golden.py:model_inputs, seed 4321) --
    restated here because bench.py's GPU legs may not import anything under oracle/."""
    from dlwp_benchmark_amd.synthetic import navier_stokes, weatherbench

    h = cfg.get("img_height", cfg.get("n_lat", 64))
    w = cfg.get("img_width", cfg.get("n_lon", 64))
    if cfg["constant_channels"] == 0 and cfg["prescribed_channels"] == 0:
        return navier_stokes(batch, frames, h, w, channels=cfg["prognostic_channels"], seed=4321)
    return weatherbench(batch, frames, h, w, prognostic_channels=cfg["prognostic_channels"],
                        constant_channels=cfg["constant_channels"], prescribed_channels=cfg["prescribed_channels"], seed=4321)


def _attn_tag(name, a):
    if name.startswith("dlwp_window_attn"):
        d = a[0]._obj
        return (tuple(d.padded), tuple(d.window), int(d.heads), int(d.head_dim), int(a[5]), int(d.use_mask))   # a[5] = batch
    if name.startswith("dlwp_linear_") and "pack" not in name:
        io = (int(a[9]), int(a[10])) if name.endswith("_io") else (0, 0)         # bf16 tensor on the x / out side
        return (int(a[5]), int(a[6]), int(a[7]), int(a[8]), a[3] is not None) + io  # rows, in, out, act, residual operand
    return None


def _attn_flops(tag):
    """4 * B * nW * nH * N^2 * d (QK^T + PV, padding tokens included as the reference computes them)."""
    padded, window, heads, hd, batch, _ = tag
    n = window[0] * window[1] * window[2]
    nw = (padded[0] // window[0]) * (padded[1] // window[1]) * (padded[2] // window[2])
    return 4.0 * batch * nw * heads * n * n * hd


def _golden_parity(model, cfg, gold, sha, device, bound):
    """per-step rel-L2 of this architecture + these weights against the committed trajectory of the REAL reference class"""
    import numpy as np

    gpath = os.path.join(ROOT, "tests", "golden", f"model_{gold}.npz")
    if not os.path.exists(gpath):
        return None
    gz = np.load(gpath, allow_pickle=False)
    if str(gz["sha"]) != sha:
        return None
    want = torch.from_numpy(gz["y"]).double()
    frames = want.shape[1] + cfg["context_size"]
    gc, gp, gg = _golden_inputs(cfg, want.shape[0], frames)
    dev = lambda t: t.to(device) if t is not None else None
    got = model(constants=dev(gc), prescribed=dev(gp), prognostic=dev(gg)).cpu().double()
    errs = [float(torch.linalg.vector_norm(got[:, t] - want[:, t]) / torch.linalg.vector_norm(want[:, t]))
            for t in range(want.shape[1])]
    return {"rel_l2_per_step_vs_golden": [float(f"{e:.3e}") for e in errs], "rel_l2_max": max(errs), "rel_l2_bound": bound,
            "parity_ok": max(errs) <= bound,
            "golden": f"tests/golden/model_{gold}.npz (real reference class, {want.shape[0]} sample, {want.shape[1]} steps)"}


def _kernel_leg(model, c, p, g, cls, cfg, batch, h, w, variant, dt_ms):
    """per-entry-point event timing of one more rollout (eager: events cannot bracket the nodes of a graph)"""
    from dlwp_benchmark_amd import lib as L

    with L.KernelTimer(tagger=_attn_tag) as kt:
        model(constants=c, prescribed=p, prognostic=g)
    summ, _ = kt.summary()
    covered = sum(v["total_ms"] for v in summ.values())
    by_name = {}
    for (name, _), v in summ.items():
        d = by_name.setdefault(name, {"calls": 0, "total_ms": 0.0})
        d["calls"] += v["calls"]
        d["total_ms"] += v["total_ms"]
    out = {"hip_entry_points_ms_per_rollout": {k: round(v["total_ms"], 4) for k, v in
                                               sorted(by_name.items(), key=lambda kv: -kv[1]["total_ms"])[:6]},
           "share_outside_libdlwp_hip": max(0.0, 1.0 - covered / dt_ms),   # torch glue
           "roofline": _other_roofline(cls, cfg, batch, h, w, summ, variant)}
    lin = _linear_roofline(summ)
    if lin is not None:
        out["roofline_linear"] = lin
    return out


def bench_other_configs(device, only=None, reps=2):
    """C1 / C3 / C4 / C5 through the HIP path on this GPU: whole-rollout wall time (inputs resident), per-step rel-L2
    of the SAME architecture + filler weights against the committed fixture of the real reference classes
    (tests/golden/model_*_full.npz: one initial condition, two steps), and a roofline for the dominant hand-written
    kernel from ALGORITHMIC flops / bytes / HIP-event time / the peak of the pipe the kernel's products run on."""
    import dlwp_benchmark_amd.models as M
    from dlwp_benchmark_amd.synthetic import navier_stokes, weatherbench
    from dlwp_benchmark_amd.weights import fill_state_dict

    res = {}
    for tag, (cls, cfg, batch, steps, (h, w), gold, gain, _) in config_table().items():
        if only and tag not in only:
            continue
        model = getattr(M, cls)(**cfg)
        sha = fill_state_dict(model, gain=gain)
        model = model.to(device).eval()
        if cfg["constant_channels"] == 0:
            c, p, g = navier_stokes(batch, steps + 1, h, w, channels=cfg["prognostic_channels"])
        else:
            c, p, g = weatherbench(batch, steps + 1, h, w, prognostic_channels=cfg["prognostic_channels"])
        dev = lambda t: t.to(device) if t is not None else None
        c, p, g = dev(c), dev(p), dev(g)
        for variant in variants_of(cls):
            apply_variant(model, variant)
            entry = {"workload": f"{cls} {h}x{w}, {cfg['prognostic_channels']} prognostic ch, {steps}-step rollout",
                     "dtype": VARIANTS[variant][4], "batch": batch, "rollout_steps": steps,
                     "weights": "deterministic filler sha256:" + sha[:16]}
            out = model(constants=c, prescribed=p, prognostic=g)      # warm-up (plans, allocator)
            torch.cuda.synchronize()
            times = []
            for _ in range(reps):
                t0 = time.perf_counter()
                out = model(constants=c, prescribed=p, prognostic=g)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            dt = sorted(times)[len(times) // 2] if len(times) % 2 else min(times)
            entry.update(ms_per_rollout=dt * 1e3, ms_per_step=dt * 1e3 / steps,
                         cell_steps_per_s=batch * h * w * steps / dt, finite=bool(torch.isfinite(out).all()))
            par = _golden_parity(model, cfg, gold, sha, device, VARIANTS[variant][3])
            if par:
                entry.update(par)
            entry.update(_kernel_leg(model, c, p, g, cls, cfg, batch, h, w, variant, dt * 1e3))
            res[tag + ("" if variant == "fp32" else "_" + variant)] = entry
        del model, out
        torch.cuda.empty_cache()
    return res


def _linear_roofline(summ):
    """the Linear kernel's costliest shape class: algorithmic flops 2 M K N (bias / GELU / residual not counted) / event time.
    `frac` prices the matrix instructions the form EXECUTES (6 per fp32 product for bf16x6, 3 for f16x3, 1 for bf16) against
    the dense peak of the 16-bit pipe they run on -- a fraction of a real bound, never above 1."""
    lin = {k: v for k, v in summ.items() if k[0] in ("dlwp_linear_f32", "dlwp_linear_bf16", "dlwp_linear_f16x3", "dlwp_linear_bf16_io")}
    if not lin:
        return None
    (name, tag), v = max(lin.items(), key=lambda kv: kv[1]["total_ms"])
    rows, k, n, act, resid, xb, ob = tag
    fl = 2.0 * rows * k * n
    by = rows * ((2.0 if xb else 4.0) * k + (2.0 if ob else 4.0) * n + (4.0 * n if resid else 0.0))
    prods = {"dlwp_linear_f32": 6, "dlwp_linear_f16x3": 3}.get(name, 1)
    ach = fl / (v["avg_ms"] * 1e-3) / 1e12
    tot = sum(x["total_ms"] for x in lin.values())
    return {"kernel": f"linear_kernel via {name} ({rows} x {k} -> {n}{', GELU' if act else ''}{', + residual' if resid else ''}"
                      f"{', bf16 input' if xb else ''}{', bf16 output' if ob else ''})",
            "bound": "mfma", "achieved": ach * prods, "peak": MFMA_16BIT_PEAK_TF, "unit": "TFLOP/s", "frac": ach * prods / MFMA_16BIT_PEAK_TF,
            "matrix_products_per_algorithmic_product": prods, "algorithmic_TFLOPs": ach,
            "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
            "hbm_view_GBps": by / (v["avg_ms"] * 1e-3) / 1e9, "avg_launch_ms": v["avg_ms"], "launches_per_rollout": v["calls"],
            "all_linear_ms_per_rollout": tot}


def _other_roofline(cls, cfg, batch, h, w, summ, variant):
    """roofline of the config's dominant hand-written kernel: algorithmic work per launch / event time."""
    if cls in ("SwinTransformer", "PanguWeather"):
        attn = {k: v for k, v in summ.items() if k[0].startswith("dlwp_window_attn")}
        if not attn:
            return None
        (name, tag), v = max(attn.items(), key=lambda kv: kv[1]["total_ms"])   # the shape class that costs most
        fl = _attn_flops(tag)
        bf16 = VARIANTS[variant][0] == "bf16"
        # the fp32-accurate form runs exact three-way bf16 splits: six bf16 matrix instructions per algorithmic product
        prods = 1 if bf16 else 6
        ach = fl / (v["avg_ms"] * 1e-3) / 1e12
        return {"kernel": f"window attention via {name} (window {tag[1]}, {tag[2]} heads x {tag[3]}, B={tag[4]}, "
                          f"{'shifted+masked' if tag[5] else 'unshifted'}; per C call = operand prep + attention kernel)",
                "bound": "mfma", "achieved": ach * prods, "peak": MFMA_16BIT_PEAK_TF, "unit": "TFLOP/s",
                "frac": ach * prods / MFMA_16BIT_PEAK_TF, "traffic": None,
                "matrix_products_per_algorithmic_product": prods, "algorithmic_TFLOPs": ach,
                "algorithmic_flops_per_launch": fl, "avg_launch_ms": v["avg_ms"], "launches_per_rollout": v["calls"],
                "pipe": "bf16 MFMA (dense peak)"}
    if cls == "FourCastNet":
        c, hid = cfg["embed_dim"], int(cfg["embed_dim"] * cfg["mlp_ratio"])
        tail = [(k, v) for k, v in summ.items() if k[0] in ("dlwp_afno_block_tail_f32", "dlwp_afno_block_tail_f16x3", "dlwp_token_mlp_f32",
                                                            "dlwp_token_mlp_emit_norm_f32")]
        if not tail:
            return None
        tot = sum(v["total_ms"] for _, v in tail)
        n = sum(v["calls"] for _, v in tail)
        avg = tot / n
        tokens = batch * h * w
        fl = 4.0 * tokens * c * hid                      # fc1 + fc2
        by = 4.0 * tokens * c * 4                        # f, l, x read + x written (one read + one write per operand plane)
        ach = by / (avg * 1e-3) / 1e9
        f16 = any(k[0] == "dlwp_afno_block_tail_f16x3" for k, _ in tail)
        prods = 3 if f16 else 6
        return {"kernel": f"token_mlp_kernel<MERGE, NEXT> via {'dlwp_afno_block_tail_f16x3' if f16 else 'dlwp_afno_block_tail_f32'} "
                          "(irfft out + skips + LN2 + fc1/GELU/fc2 + next LN1)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": by, "avg_launch_ms": avg, "launches_per_rollout": n,
                "mfma_view": {"algorithmic_flops_per_launch": fl, "matrix_products_per_algorithmic_product": prods,
                              "executed_TFLOPs": prods * fl / (avg * 1e-3) / 1e12,
                              "frac_of_16bit_dense_peak": prods * fl / (avg * 1e-3) / 1e12 / MFMA_16BIT_PEAK_TF}}
    if cls == "UNet":
        conv = [(k, v) for k, v in summ.items() if k[0] == "dlwp_conv3x3_cyl_f32"]
        if not conv:
            return None
        tot = sum(v["total_ms"] for _, v in conv)
        n = sum(v["calls"] for _, v in conv)
        return {"kernel": "conv3x3_cyl_kernel via dlwp_conv3x3_cyl_f32 (all levels)", "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": None, "traffic": None, "avg_launch_ms": tot / n, "launches_per_rollout": n,
                "note": "64x64 maps: every launch is a few microseconds -- launch/latency-bound, no meaningful byte roofline"}
    return None


def host_threads():
    """threads for the CPU-baseline leg = the cores this process may actually use: the cgroup CPU quota when one is
    set (a quota is CPU time, so that many threads land on distinct physical cores of a larger machine), otherwise
    the affinity mask divided by the hardware threads per core.  FIXED POLICY: this function, nothing
    else, decides the thread count unless --cpu-threads overrides it; the count and the policy are printed in `sample`."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    smt = 1
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = max(1, len(sib.replace("-", ",").split(",")))
    except Exception:
        pass
    phys = max(1, n // smt)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, min(phys, int(int(q) / int(per) + 0.5)))
    except Exception:
        pass
    return phys


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=5).stdout.strip() or None
    except Exception:
        return None


def load_traffic(kernel_key):
    """HBM bytes per launch of `kernel_key` from the PMC passes (tools/pmc_fno.sh writes profiles/traffic.json from the
    same tree as the kernel-stats CSV, stamped with the commit and the launch duration it saw).  Returns (bytes | None, source)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, None
    v = t.get(kernel_key)
    if isinstance(v, dict):
        return v.get("bytes"), f"profiles/traffic.json: rocprofv3 --pmc at {v.get('commit')}, launch {v.get('avg_launch_ms')} ms"
    return v, "profiles/traffic.json (unstamped)"


# ------------------------------------------------------------------------------------------------ the contract line

def _r(x, sig=5):
    """floats to `sig` significant digits (bytes of the line), everything else untouched"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rel_l2_per_step_max", "rel_l2_bound",
             "parity_ok", "value_bf16x6", "fused_timeouts", "range_reruns", "other_configs", "detail")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_src", "bytes_per_launch",
                 "flops_per_launch", "avg_launch_ms", "launches_per_step", "hbm_view", "fp32_equivalent_TFLOPs", "timing")
# dropped in this order while the line is too long
DROP_ORDER = (("roofline", "timing"), ("roofline", "traffic_src"), ("config", "weights"), ("config", "launch"),
              ("detail",), ("roofline", "fp32_equivalent_TFLOPs"), ("config", "precision_form"), ("config", "collect"))


def compact_other(detail):
    """tag -> {ms_per_step, rel_l2_max, parity_ok, roofline_frac, bound}"""
    out = {}
    for tag, e in detail.items():
        if not isinstance(e, dict) or "ms_per_step" not in e:
            out[tag] = {"error": str(e)[:80]}
            continue
        rf = e.get("roofline") or {}
        out[tag] = {"ms_per_step": _r(e["ms_per_step"], 4), "rel_l2_max": _r(e.get("rel_l2_max"), 3), "parity_ok": e.get("parity_ok"),
                    "roofline_frac": _r(rf.get("frac"), 3), "bound": rf.get("bound")}
    return out


def compact_line(result):
    """The contract line: the keys the driver and the judge read, floats rounded, verbose tables left out, and a hard
    size guard -- optional fields are dropped (in DROP_ORDER, then other_configs entries from the end) until the line is
    below LINE_LIMIT bytes.  tests/test_bench_line_cpu.py pins the size and the JSON round trip."""
    line = {k: result[k] for k in LINE_KEYS if k in result}
    if "roofline" in line and isinstance(line["roofline"], dict):
        line["roofline"] = {k: line["roofline"][k] for k in ROOFLINE_KEYS if k in line["roofline"]}
    if isinstance(line.get("other_configs"), dict) and any(isinstance(v, dict) and "workload" in v for v in line["other_configs"].values()):
        line["other_configs"] = compact_other(line["other_configs"])
    line = _r(line)
    s = json.dumps(line, separators=(",", ":"))
    for path in DROP_ORDER:
        if len(s) < LINE_LIMIT:
            break
        d = line
        for k in path[:-1]:
            d = d.get(k, {}) if isinstance(d, dict) else {}
        if isinstance(d, dict) and path[-1] in d:
            del d[path[-1]]
            s = json.dumps(line, separators=(",", ":"))
    oc = line.get("other_configs")
    if isinstance(oc, dict):
        tags = [k for k in oc if k != "truncated"]
        while len(s) >= LINE_LIMIT and tags:
            oc.pop(tags.pop())
            oc["truncated"] = True
            s = json.dumps(line, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:      # still too long: clip the free-text fields
        for k in ("workload", "parallelism"):
            if k in line.get("config", {}):
                line["config"][k] = line["config"][k][:120]
        if "sample" in line.get("cpu_baseline", {}):
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:160]
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) < LINE_LIMIT, len(s)
    return s


def write_detail(path, result):
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
        return os.path.relpath(path, ROOT)
    except Exception as e:       # a read-only tree must not cost the line
        print(f"bench: could not write {path}: {e}", file=sys.stderr)
        return None


# ------------------------------------------------------------------------------------------------ distributed plumbing

def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # DLWP_BENCH_BACKEND=gloo + DLWP_BENCH_ONE_GPU=1: rehearse the multi-rank control flow on a
    # one-GPU box (every rank on cuda:0, gloo for the collectives); the real run uses nccl (= RCCL).
    backend = os.environ.get("DLWP_BENCH_BACKEND", "nccl")
    if os.environ.get("DLWP_BENCH_ONE_GPU") == "1":
        local_rank = 0
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, device, dist, backend


def timed_region(step, finish, steps, warmup, dist, device, backend):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; max over ranks."""
    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)   # cpu: the gloo control-flow test
    for _ in range(warmup):
        step()
    finish(reset=True)            # also initialises the communicator's all-reduce path before the clock starts
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    finish(reset=False)
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def parallelism_text(world, collect):
    if world == 1:
        return "single GPU"
    return f"batch-shard x{world}, " + {"metrics": "on-device RMSE sums per rank, one all-reduce per evaluation",
                                        "gather": "chunked all-gather of trajectories", "none": "no collective"}[collect]


def make_runner(model, world, rank, args, prog, H, W, scorer=None):
    """one bench step = one rollout of this rank's shard + its metric sums; finish() = deferred verification + the ONE collective"""
    from dlwp_benchmark_amd.metrics import RolloutMetrics
    from dlwp_benchmark_amd.sharding import ShardedRollout

    runner = ShardedRollout(model, world_size=world, rank=rank, chunks=args.gather_chunks, gather=(args.collect == "gather"))
    # evaluation scores of the rollout against the frames the synthetic generator produced: reduced on the device,
    # all-reduced across ranks -- [4, K, C] doubles instead of trajectories
    scorer = scorer or RolloutMetrics(torch.zeros(H))
    B, K = prog.shape[0], prog.shape[1] - model.context_size
    target = prog[:, model.context_size:].contiguous()
    acc = {"sums": torch.zeros(4, K, prog.shape[2], dtype=torch.float64, device=prog.device) if args.collect == "metrics" else None,
           "samples": 0, "scores": None, "out": None}

    def device_step(constants, prescribed, prognostic):
        out = runner(constants=constants, prescribed=prescribed, prognostic=prognostic)
        if args.collect == "metrics":
            # this rank's squared-error sums of the rollout, ADDED to the evaluation's running sums on the device (the
            # reference accumulates over all batches before taking the root, evaluate.py:786-821): no collective per step
            scorer.sums(out if out.shape[0] == B else out[rank * B:(rank + 1) * B], target, into=acc["sums"])
        return out

    # --graph-step: rollout + metric sums as ONE recorded HIP graph per rank (no collective inside a step in "metrics" / "none" mode)
    graphed = None
    state = {"graph_error": None}
    # (one process per GPU under torchrun keeps the plain launch sequence: a communicator's watchdog thread beside a stream capture
    # is a combination this code has never run on real multi-GPU hardware, and the recorded step buys nothing for the long launches)
    if getattr(args, "graph_step", False) and prog.is_cuda and world == 1:
        from dlwp_benchmark_amd.sharding import CapturedStep

        graphed = CapturedStep(device_step, model=model)

    def step(constants=None, prescribed=None):
        nonlocal graphed
        if graphed is not None:
            try:
                out = graphed(constants, prescribed, prog)
            except Exception as e:      # a failed recording must not cost the line: say so (stderr + config.launch) and go on eagerly
                print(f"bench: graph recording failed ({type(e).__name__}: {e}); continuing with plain launches", file=sys.stderr, flush=True)
                state["graph_error"] = f"{type(e).__name__}: {e}"[:120]
                acc["graph_error"] = state["graph_error"]
                graphed = None
                torch.cuda.synchronize()
                out = device_step(constants, prescribed, prog)
        else:
            out = device_step(constants, prescribed, prog)
        if args.collect == "metrics":
            acc["samples"] += B
        acc["out"] = out
        return out

    def finish(reset=False):
        # deferred verification of every fused launch since the last call (DLWP_ERR_TIMEOUT raises here): the rollouts of an
        # evaluation are enqueued asynchronously and verified ONCE, inside the timed region
        if hasattr(model, "verify"):
            model.verify()
        # the ONE collective of the sharded evaluation: all-reduce of [4, K, C] sums + sample count (inside the timed region)
        if args.collect == "metrics":
            acc["scores"] = scorer.finalize(acc["sums"], float(acc["samples"]), H * W, world_size=world)
            if reset:
                acc["sums"].zero_()
                acc["samples"] = 0

    return step, finish, acc


# ------------------------------------------------------------------------------------------------ C2: the headline

def run_c2(args, world, rank, device, dist, backend):
    from dlwp_benchmark_amd.synthetic import navier_stokes

    B, K_roll, H, W = args.batch or 32, args.rollout_steps or 20, 64, 64
    model, sha = build_model(device)
    _, _, prog_cpu = navier_stokes(B, K_roll + 1, H, W, seed=1234 + rank)
    prog = prog_cpu.to(device)
    step, finish, acc = make_runner(model, world, rank, args, prog, H, W)
    dt = timed_region(step, finish, args.steps, args.warmup, dist, device, backend)
    out = acc["out"]
    ms_per_step = dt / args.steps * 1e3
    form = model.precision_form
    result = {
        "metric": "rollout cell-steps/s",
        "value": world * B * H * W * K_roll * args.steps / dt,
        "unit": "grid-cells*steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f16x3": "f32 (f16x3: products from two-part f16 splits, 22-bit operands, fp32 accumulate)",
                  "bf16x6": "f32 (bf16x6: products from three-part bf16 splits, 24-bit operands, fp32 accumulate)",
                  "fp32_mfma": "f32 (fp32 matrix instructions)"}[form],
        "data": "synthetic",
        "config": {
            "workload": "C2 FNO2d modes=12 hidden=32 lift/proj=256 layers=4, Navier-Stokes 64x64, 20-step rollout (BASELINE configs[1]); "
                        "1 step = 1 rollout of the batch",
            "batch_per_gpu": B, "global_batch": B * world, "grid": [H, W], "rollout_steps": K_roll,
            "parallelism": parallelism_text(world, args.collect), "collect": args.collect, "precision_form": form,
            "launch": (("plain launches (graph recording FAILED: " + acc["graph_error"] + ")") if acc.get("graph_error") else
                       "one recorded HIP graph per step (rollout + metric sums; sharding.CapturedStep)" if (getattr(args, "graph_step", False) and world == 1)
                       else "eager") + "; fused-kernel check deferred, verified once per evaluation inside the timed region",
            "weights": "filler sha256:" + sha[:12],
        },
        "fused_timeouts": int(model.fused_timeouts()), "range_reruns": int(model.range_reruns()),
    }
    if rank != 0 or world != 1:
        return result

    # ---- roofline leg: per-kernel HIP-event timing of the same rollout
    prof, event_overhead_ms = profile_kernels(model, prog, repeats=max(2, min(args.steps, 5)))
    work = algorithmic_work(B, H, W, MODEL_KW, 12, MODEL_KW["n_modes"][1] // 2 + 1)
    per_kernel = {}
    for name, (avg_ms, launches) in prof.items():
        w_ = work[name]
        per_kernel[name] = {
            "avg_ms": avg_ms, "launches_per_rollout": launches,
            "GBps": w_["bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None,
            "TFLOPs": w_["flops"] / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else None,
            "share_of_rollout": avg_ms * launches / ms_per_step,
        }
    # The kernel the north star names (spectral-conv rollout, HBM-bound by SURVEY.md 8d).  With the fused step kernel ONE
    # launch runs lifting + all n_layers spectral layers (incl. mode mixing) + projection for EVERY step of the rollout, so the
    # algorithmic bytes of a launch are SURVEY 8d's per-step figure (163.6 MiB at B = 32) x the steps one launch processes.
    lay = per_kernel["layer"]
    fused = per_kernel["modes"]["launches_per_rollout"] == 0
    whole_step = fused and per_kernel["lift"]["launches_per_rollout"] == 0
    nl = MODEL_KW["n_layers"]
    if fused:
        lay_bytes = nl * (work["layer"]["bytes"] + work["modes"]["bytes"])
        lay_flops = nl * (work["layer"]["flops"] + work["modes"]["flops"])
        kname, tkey = "fno_trunk_kernel (all spectral layers in one launch)", "fno_trunk_kernel"
        steps_per_launch = 1
        if whole_step:
            lay_bytes += work["lift"]["bytes"] + work["proj"]["bytes"]
            lay_flops += work["lift"]["flops"] + work["proj"]["flops"]
            steps_per_launch = max(1, K_roll // max(lay["launches_per_rollout"], 1))
            lay_bytes *= steps_per_launch
            lay_flops *= steps_per_launch
            kname, tkey = "fno_trunk_kernel<STEP> (lifting + spectral layers + projection, all rollout steps, one launch)", "fno_step_kernel"
    else:
        lay_bytes, lay_flops, steps_per_launch = work["layer"]["bytes"], work["layer"]["flops"], 1
        kname, tkey = "fno_layer_kernel", "fno_layer_kernel"
    gbps = lay_bytes / (lay["avg_ms"] * 1e-3) / 1e9
    traffic, tsrc = load_traffic(tkey)
    result["roofline"] = {
        "kernel": kname, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_src": tsrc,
        "bytes_per_launch": lay_bytes, "flops_per_launch": lay_flops, "avg_launch_ms": lay["avg_ms"],
        "launches_per_step": lay["launches_per_rollout"], "rollout_steps_per_launch": steps_per_launch,
        # what the same launch is in fp32-GEMM terms (algorithmic flops; the products run as 3 f16 / 6 bf16 matrix
        # instructions each, so this is NOT a fraction of any pipe's peak and none is claimed)
        "fp32_equivalent_TFLOPs": lay_flops / (lay["avg_ms"] * 1e-3) / 1e12,
        "timing": f"HIP events on the launch stream around every launch, one marker latency ({0.5 * event_overhead_ms * 1e3:.1f} us) subtracted",
    }
    result["kernels"] = per_kernel

    # ---- CPU baseline leg (oracle on the host cores) + per-step rel-L2
    if not args.no_cpu_baseline:
        nb = min(args.cpu_batch, B)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        nthreads = args.cpu_threads or host_threads()
        torch.set_num_threads(nthreads)
        cps, med, secs, traj = cpu_baseline(sd, prog_cpu[:nb].contiguous(), K_roll)
        got = out[:nb].detach().cpu().double()
        want = traj.double()
        errs = [float(torch.linalg.vector_norm(got[:, t] - want[:, t]) / torch.linalg.vector_norm(want[:, t]))
                for t in range(K_roll)]
        result["cpu_baseline"] = {
            "value": cps, "unit": "grid-cells*steps/s", "cores": nthreads, "kind": "port",
            "sample": f"oracle (PyTorch CPU restatement) on {nb} of {B} initial conditions, {K_roll} steps; 1 warm-up + "
                      f"{len(secs)} timed rollouts, median {med:.2f} s; {nthreads} threads = "
                      f"{'--cpu-threads' if args.cpu_threads else 'cgroup quota / physical cores (host_threads)'}, os.cpu_count {os.cpu_count()}",
        }
        result["rel_l2_per_step_max"] = max(errs)
        result["rel_l2_bound"] = 1e-5
        result["rel_l2_per_step"] = [float(f"{e:.3e}") for e in errs]
        result["parity_ok"] = max(errs) <= 1e-5

    # ---- the same workload on the full-24-bit form (bf16x6) beside `value`
    if form != "bf16x6" and not args.no_second_form:
        m2, _ = build_model(device, precision_form="bf16x6")
        a2 = argparse.Namespace(**vars(args))
        step2, finish2, _ = make_runner(m2, 1, 0, a2, prog, H, W)
        n2 = max(3, min(args.steps, 10))
        dt2 = timed_region(step2, finish2, n2, 2, None, device, backend)
        result["value_bf16x6"] = B * H * W * K_roll * n2 / dt2
        del m2

    # ---- the other BASELINE configs (C1, C3, C4, C5) through the HIP path, same process, same GPU
    if not args.no_other_configs:
        try:
            result["other_configs"] = bench_other_configs(device, only=args.only_configs)
        except Exception as e:   # the headline line must survive a failure here; the failure is loud: in the line and on stderr
            import traceback

            traceback.print_exc()
            result["other_configs"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return result


# ------------------------------------------------------------------------------------------------ C3 / C4 / C5

def cpu_baseline_backbone(cls, cfg, sd, c, p, g, steps):
    """oracle rollout of a Swin / FourCastNet / Pangu config on the host cores: one warm-up STEP, then `steps` timed steps"""
    from oracle.restate.afno import afnonet_rollout          # checker / reported baseline only
    from oracle.restate.pangu import pangu_rollout
    from oracle.restate.swin import swin_rollout

    fn = {"SwinTransformer": swin_rollout, "FourCastNet": afnonet_rollout, "PanguWeather": pangu_rollout}[cls]
    ctx = cfg["context_size"]
    with torch.no_grad():
        fn(sd, cfg, c, p[:, :ctx + 1] if p is not None else None, g[:, :ctx + 1])
        t0 = time.perf_counter()
        traj = fn(sd, cfg, c, p[:, :ctx + steps] if p is not None else None, g[:, :ctx + steps])
        sec = time.perf_counter() - t0
    return sec, traj


def run_backbone(tag, args, world, rank, device, dist, backend):
    import dlwp_benchmark_amd.models as M
    from dlwp_benchmark_amd.synthetic import weatherbench
    from dlwp_benchmark_amd.weights import fill_state_dict

    cls, cfg, batch, steps, (H, W), gold, gain, named = config_table()[tag]
    B = args.batch or batch
    K_roll = args.rollout_steps or steps
    variant = args.precision or named
    model = getattr(M, cls)(**cfg)
    sha = fill_state_dict(model, gain=gain)
    model = apply_variant(model.to(device).eval(), variant)
    c_cpu, p_cpu, g_cpu = weatherbench(B, K_roll + 1, H, W, prognostic_channels=cfg["prognostic_channels"], seed=1234 + rank)
    c, p, g = c_cpu.to(device), p_cpu.to(device), g_cpu.to(device)
    step0, finish, acc = make_runner(model, world, rank, args, g, H, W)
    step = lambda: step0(constants=c, prescribed=p)
    dt = timed_region(step, finish, args.steps, args.warmup, dist, device, backend)
    out = acc["out"]
    result = {
        "metric": "rollout cell-steps/s",
        "value": world * B * H * W * K_roll * args.steps / dt,
        "unit": "grid-cells*steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": VARIANTS[variant][4], "data": "synthetic",
        "config": {
            "workload": f"{tag} {cls} {H}x{W}, {cfg['prognostic_channels']} prognostic ch, {K_roll}-step rollout "
                        f"(BASELINE configs[{'C1 C2 C3 C4 C5'.split().index(tag)}]); 1 step = 1 rollout of the batch",
            "batch_per_gpu": B, "global_batch": B * world, "grid": [H, W], "rollout_steps": K_roll,
            "parallelism": parallelism_text(world, args.collect), "collect": args.collect, "precision_form": variant,
            "weights": "filler sha256:" + sha[:12],
        },
    }
    if rank != 0 or world != 1:
        return result
    par = _golden_parity(model, cfg, gold, sha, device, VARIANTS[variant][3])
    if par:
        result["rel_l2_per_step_max"], result["rel_l2_bound"], result["parity_ok"] = par["rel_l2_max"], par["rel_l2_bound"], par["parity_ok"]
        result["parity"] = par
    leg = _kernel_leg(model, c, p, g, cls, cfg, B, H, W, variant, dt / args.steps * 1e3)
    rf = leg.pop("roofline") or {}
    rf["bytes_per_launch"] = rf.pop("algorithmic_bytes_per_launch", None)
    rf["flops_per_launch"] = rf.pop("algorithmic_flops_per_launch", None)
    rf["timing"] = "HIP events on the launch stream around every C-ABI call, one marker latency subtracted"
    result["roofline"] = rf
    result["kernels"] = leg
    if not args.no_cpu_baseline:
        nb = 1
        nsteps = {"C3": 3, "C4": 3, "C5": 2}.get(tag, 2)
        nsteps = min(nsteps, K_roll)
        nthreads = args.cpu_threads or host_threads()
        torch.set_num_threads(nthreads)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        sec, traj = cpu_baseline_backbone(cls, cfg, sd, c_cpu[:nb], p_cpu[:nb], g_cpu[:nb], nsteps)
        got, want = out[:nb, :nsteps].detach().cpu().double(), traj.double()
        errs = [float(torch.linalg.vector_norm(got[:, t] - want[:, t]) / torch.linalg.vector_norm(want[:, t])) for t in range(nsteps)]
        result["cpu_baseline"] = {
            "value": nb * H * W * nsteps / sec, "unit": "grid-cells*steps/s", "cores": nthreads, "kind": "port",
            "sample": f"oracle (PyTorch CPU restatement) on {nb} of {B} initial conditions, {nsteps} of {K_roll} steps after a "
                      f"1-step warm-up, {sec:.2f} s; {nthreads} threads (host_threads policy), os.cpu_count {os.cpu_count()}",
        }
        result["rel_l2_vs_oracle_per_step"] = [float(f"{e:.3e}") for e in errs]
        if "rel_l2_per_step_max" not in result:
            result["rel_l2_per_step_max"], result["rel_l2_bound"] = max(errs), VARIANTS[variant][3]
            result["parity_ok"] = max(errs) <= VARIANTS[variant][3]
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default="C2", help="BASELINE config to run (default: the headline)")
    ap.add_argument("--precision", choices=sorted(VARIANTS), default=None, help="C3 / C4 / C5: precision variant (default: the one BASELINE names)")
    ap.add_argument("--batch", type=int, default=0, help="initial conditions per GPU (default: the config's)")
    ap.add_argument("--rollout-steps", type=int, default=0, help="rollout length (default: the config's)")
    ap.add_argument("--cpu-batch", type=int, default=4, help="samples of the bounded CPU-baseline leg (C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU-baseline leg (0 = host_threads())")
    ap.add_argument("--no-other-configs", action="store_true", help="C2: skip the C1/C3/C4/C5 legs")
    ap.add_argument("--no-second-form", action="store_true", help="C2: skip the bf16x6 run beside the default form")
    ap.add_argument("--only-configs", nargs="*", help="subset of the other-config tags (C1 C3 C4 C5)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "profiles", "bench_detail_last.json"),
                    help="where the verbose result (per-kernel tables, full other-config entries) is written")
    ap.add_argument("--graph-step", dest="graph_step", action="store_true", default=True,
                    help="record a step (rollout + metric sums) into one HIP graph and replay it (default)")
    ap.add_argument("--no-graph-step", dest="graph_step", action="store_false", help="enqueue every launch of a step from the host")
    ap.add_argument("--gather-chunks", type=int, default=4)
    ap.add_argument("--collect", choices=["metrics", "gather", "none"], default="metrics",
                    help="what leaves a rank per rollout: per-lead-time RMSE sums reduced on the device and all-reduced "
                         "(default; SURVEY.md 8e/8f-f1), the whole trajectory (one chunked RCCL all-gather), or nothing")
    args = ap.parse_args()

    world, rank, device, dist, backend = init_dist(args)
    if args.config == "C2":
        result = run_c2(args, world, rank, device, dist, backend)
    else:
        result = run_backbone(args.config, args, world, rank, device, dist, backend)
    if rank == 0:
        result["commit"] = git_head()
        if world == 1:
            d = write_detail(args.detail, result)
            if d:
                result["detail"] = d
        print(compact_line(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and result.get("parity_ok") is False:
        print("bench: PARITY BOUND EXCEEDED (parity_ok false in the line)", file=sys.stderr)


if __name__ == "__main__":
    main()
