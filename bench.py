#!/usr/bin/env python3
"""Headline benchmark: autoregressive rollout throughput of the FNO2d backbone on MI355X.

Workload (BASELINE.json configs[1]): FNO2d modes=12, hidden 32, lifting/projection 256, 4 layers
on synthetic Navier-Stokes 64x64, batch 32 (reference configs/testing/default.yaml:1), 20-step
rollout, fp32.  One bench "step" = one whole 20-step rollout of the batch; inputs are resident in
HBM before the timed region.  Metric: grid-cells x rollout-steps per second, whole job.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: every rank rolls out its own shard of initial conditions (weak scaling, no collective in
the step); the trajectories are collected with ONE RCCL all-gather per rollout (issued in time
chunks on RCCL's stream so it overlaps the remaining rollout steps).

Extra legs in the same run (rank 0, N = 1 only):
  roofline      per-kernel HIP-event timing of the same rollout (dlwp_fno2d_rollout_profiled_f32)
  cpu_baseline  the oracle (PyTorch CPU restatement of the reference forward) timed on the host
                cores on a bounded sample; also yields the per-step rel-L2 of the HIP trajectory.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # same guide: fp32 matrix peak (v_mfma_f32_16x16x4_f32)

MODEL_KW = dict(n_modes=[12, 12], constant_channels=0, prescribed_channels=0, prognostic_channels=1,
                hidden_channels=32, lifting_channels=256, projection_channels=256, n_layers=4, context_size=1)


def std_fn(name, shape):
    if "convs.weight" in name:
        return 0.85 / shape[0] ** 0.5
    return None


def build_model(device):
    from dlwp_benchmark_amd.models import FNO2DModule
    from dlwp_benchmark_amd.weights import fill_state_dict

    m = FNO2DModule(**MODEL_KW)
    sha = fill_state_dict(m, std_fn=std_fn, gain=0.85)
    m.set_execution_form(check="deferred")      # asynchronous rollouts, verified by model.verify() in finish()
    if os.environ.get("DLWP_BENCH_PRECISION"):  # A/B of the product forms (default: the module's, "f16x3")
        m.set_execution_form(precision_form=os.environ["DLWP_BENCH_PRECISION"])
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


MFMA_BF16_PEAK_TF = 2500.0  # same guide: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)


def other_config_table():
    """The BASELINE configs that are NOT the headline line (C1, C3, C4, C5) at their BASELINE sizes:
    tag -> (class name, ctor kwargs, batch per GPU, rollout steps, (H, W), golden fixture of the same architecture + filler weights)."""
    return {
        "C1_unet_64x64": ("UNet", dict(constant_channels=0, prescribed_channels=0, prognostic_channels=1,
                                       hidden_channels=[8, 16, 32, 64], n_convolutions=2, activation="th.nn.GELU()",
                                       context_size=1), 32, 1, (64, 64), "unet_c1_64x64", 1.0),
        "C3_swin_32x64": ("SwinTransformer", dict(context_size=1, img_height=32, img_width=64, patch_size=1, constant_channels=4,
                                                  prescribed_channels=1, prognostic_channels=3, embed_dim=96, depths=[4, 4],
                                                  num_heads=[4, 4], mlp_ratio=4, qkv_bias=True, drop_path_rate=0.2,
                                                  norm_layer="nn.LayerNorm", patch_norm=True), 32, 12, (32, 64), "swin_c3_full", 0.7),
        "C4_fourcastnet_128x256": ("FourCastNet", dict(img_height=128, img_width=256, patch_size=[1, 1], constant_channels=4,
                                                       prescribed_channels=1, prognostic_channels=3, filter="AFNO2D",
                                                       embed_dim=64, depth=4, mlp_ratio=4.0, num_blocks=4,
                                                       sparsity_threshold=0.01, hard_thresholding_fraction=1.0,
                                                       context_size=1, use_pos_embed=True), 32, 20, (128, 256), "afno_c4_full", 0.7),
        "C5_pangu_128x256x13": ("PanguWeather", dict(constant_channels=4, prescribed_channels=1, prognostic_channels=13,
                                                     embed_dim=192, num_heads=[6, 12, 12, 6], window_size=[2, 6, 12],
                                                     patch_size=[1, 1], n_lat=128, n_lon=256, context_size=1), 8, 5, (128, 256),
                                "pangu_c5_full", 0.7),
    }


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


def bench_other_configs(device, only=None, reps=2):
    """C1 / C3 / C4 / C5 through the HIP path on this GPU: whole-rollout wall time (inputs resident), per-step rel-L2
    of the SAME architecture + filler weights against the committed fixture of the real reference classes
    (tests/golden/model_*_full.npz: one initial condition, two steps), and a roofline for the dominant hand-written
    kernel from ALGORITHMIC flops / bytes / HIP-event time / the peak of the pipe it runs on."""
    import numpy as np

    import dlwp_benchmark_amd.models as M
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.synthetic import navier_stokes, weatherbench
    from dlwp_benchmark_amd.weights import fill_state_dict

    gdir = os.path.join(ROOT, "tests", "golden")
    res = {}
    for tag, (cls, cfg, batch, steps, (h, w), gold, gain) in other_config_table().items():
        if only and tag not in only:
            continue
        # "bf16": bf16 window attention only; "bf16all": bf16 attention AND bf16 Linear operands (the autocast(bfloat16)
        # analogue BASELINE configs[2] names); fp32 accumulation, LayerNorm and residual stream in all of them
        # "f16x3": fp32 attention, the Linears in the fp32-grade f16x3 form (two-part f16 splits, three products)
        variants = ["fp32"] + (["f16x3", "bf16", "bf16all"] if cls in ("SwinTransformer", "PanguWeather") else []) + \
            (["f16x3"] if cls == "FourCastNet" else [])
        model = getattr(M, cls)(**cfg)
        sha = fill_state_dict(model, gain=gain)
        model = model.to(device).eval()
        # DLWP_BENCH_STEP_GRAPHS=all|unet: replay one_step as a HIP graph (HipBackbone.set_step_graphs).  Measured in round 2:
        # no gain on any config (C1 0.61 vs 0.58 ms, C3 / C4 / C5 within noise) -- none of them is bound by the host's launch
        # rate -- so the default stays eager.
        gsel = os.environ.get("DLWP_BENCH_STEP_GRAPHS", "none")
        graphs_on = gsel == "all" or (gsel == "unet" and cls == "UNet")
        model.set_step_graphs(graphs_on)
        if cfg["constant_channels"] == 0:
            c, p, g = navier_stokes(batch, steps + 1, h, w, channels=cfg["prognostic_channels"])
        else:
            c, p, g = weatherbench(batch, steps + 1, h, w, prognostic_channels=cfg["prognostic_channels"])
        dev = lambda t: t.to(device) if t is not None else None
        c, p, g = dev(c), dev(p), dev(g)
        gpath = os.path.join(gdir, f"model_{gold}.npz")
        want = None
        if os.path.exists(gpath):
            gz = np.load(gpath, allow_pickle=False)
            if str(gz["sha"]) == sha:
                want = torch.from_numpy(gz["y"])
        for variant in variants:
            prec = "fp32" if variant in ("fp32", "f16x3") else "bf16"
            if hasattr(model, "set_attention_precision"):
                model.set_attention_precision(prec)
                model.set_linear_form({"bf16all": "bf16", "f16x3": "f16x3"}.get(variant, "bf16x6"))
            if hasattr(model, "set_mlp_form"):
                model.set_mlp_form("f16x3" if variant == "f16x3" else "bf16x6")
            what = {"fp32": "fp32", "bf16": "bf16 window attention (fp32 elsewhere)",
                    "f16x3": "fp32 (block-tail MLP products from exact two-part f16 splits, dlwp_afno_block_tail_f16x3)" if cls == "FourCastNet"
                             else "fp32 (Linear products from exact two-part f16 splits, dlwp_linear_f16x3; fp32-accurate attention)",
                    "bf16all": "bf16 window attention and bf16 Linear operands (fp32 accumulation, LayerNorm, residual stream; the MLP's hidden "
                               "activation crosses HBM as bf16)"}[variant]
            entry = {"workload": f"{cls} {h}x{w}, {cfg['prognostic_channels']} prognostic ch, {steps}-step rollout, {what}",
                     "batch": batch, "rollout_steps": steps, "weights": "deterministic filler sha256:" + sha[:16],
                     "launch": "one_step replayed as a HIP graph (set_step_graphs)" if graphs_on else "eager launches"}
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
            # parity of this architecture + these weights against the committed reference trajectory
            if want is not None:
                frames = want.shape[1] + cfg["context_size"]
                gc, gp, gg = _golden_inputs(cfg, want.shape[0], frames)
                got = model(constants=dev(gc), prescribed=dev(gp), prognostic=dev(gg)).cpu().double()
                wd = want.double()
                errs = [float(torch.linalg.vector_norm(got[:, t] - wd[:, t]) / torch.linalg.vector_norm(wd[:, t]))
                        for t in range(wd.shape[1])]
                entry["rel_l2_per_step_vs_golden"] = [float(f"{e:.3e}") for e in errs]
                entry["golden"] = f"tests/golden/model_{gold}.npz (real reference class, {want.shape[0]} sample, {want.shape[1]} steps)"
                entry["rel_l2_bound"] = 5e-3 if prec == "bf16" else 1e-5
                entry["parity_ok"] = max(errs) <= entry["rel_l2_bound"]
            # per-entry-point event timing of one more rollout (eager: events cannot bracket the nodes of a graph)
            model.set_step_graphs(False)
            with L.KernelTimer(tagger=_attn_tag) as kt:
                model(constants=c, prescribed=p, prognostic=g)
            model.set_step_graphs(graphs_on)
            summ, marker = kt.summary()
            covered = sum(v["total_ms"] for v in summ.values())
            by_name = {}
            for (name, _), v in summ.items():
                d = by_name.setdefault(name, {"calls": 0, "total_ms": 0.0})
                d["calls"] += v["calls"]
                d["total_ms"] += v["total_ms"]
            entry["hip_entry_points_ms_per_rollout"] = {k: round(v["total_ms"], 4) for k, v in
                                                        sorted(by_name.items(), key=lambda kv: -kv[1]["total_ms"])[:6]}
            entry["share_outside_libdlwp_hip"] = max(0.0, 1.0 - covered / (dt * 1e3))   # rocBLAS / MIOpen / torch glue
            entry["roofline"] = _other_roofline(cls, cfg, batch, h, w, summ, prec)
            lin = _linear_roofline(summ)
            if lin is not None:
                entry["roofline_linear"] = lin
            res[tag + {"fp32": "", "f16x3": "_f16x3", "bf16": "_bf16attn", "bf16all": "_bf16"}[variant]] = entry
        del model, out
        torch.cuda.empty_cache()
    return res


def _linear_roofline(summ):
    """the Linear kernel's costliest shape class: algorithmic flops 2 M K N (bias / GELU / residual not counted) / event time,
    priced against the fp32 matrix peak for dlwp_linear_f32 (fp32-accurate: six bf16 products per fp32 one are not credited)
    and against the dense bf16 peak for dlwp_linear_bf16."""
    lin = {k: v for k, v in summ.items() if k[0] in ("dlwp_linear_f32", "dlwp_linear_bf16", "dlwp_linear_f16x3", "dlwp_linear_bf16_io")}
    if not lin:
        return None
    (name, tag), v = max(lin.items(), key=lambda kv: kv[1]["total_ms"])
    rows, k, n, act, resid, xb, ob = tag
    fl = 2.0 * rows * k * n
    by = rows * ((2.0 if xb else 4.0) * k + (2.0 if ob else 4.0) * n + (4.0 * n if resid else 0.0))
    peak = MFMA_BF16_PEAK_TF if "bf16" in name else MFMA_F32_PEAK_TF
    ach = fl / (v["avg_ms"] * 1e-3) / 1e12
    tot = sum(x["total_ms"] for x in lin.values())
    return {"kernel": f"linear_kernel via {name} ({rows} x {k} -> {n}{', GELU' if act else ''}{', + residual' if resid else ''}"
                      f"{', bf16 input' if xb else ''}{', bf16 output' if ob else ''})",
            "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
            "hbm_view_GBps": by / (v["avg_ms"] * 1e-3) / 1e9, "avg_launch_ms": v["avg_ms"], "launches_per_rollout": v["calls"],
            "all_linear_ms_per_rollout": tot,
            "note": "dlwp_linear_f32 / _f16x3 run their products on the bf16 / f16 matrix pipe: against the fp32 matrix peak their fraction can exceed 1"}


def _other_roofline(cls, cfg, batch, h, w, summ, prec):
    """roofline of the config's dominant hand-written kernel: algorithmic work per launch / event time."""
    if cls in ("SwinTransformer", "PanguWeather"):
        attn = {k: v for k, v in summ.items() if k[0].startswith("dlwp_window_attn")}
        if not attn:
            return None
        (name, tag), v = max(attn.items(), key=lambda kv: kv[1]["total_ms"])   # the shape class that costs most
        fl = _attn_flops(tag)
        peak = MFMA_BF16_PEAK_TF if prec == "bf16" else MFMA_F32_PEAK_TF
        ach = fl / (v["avg_ms"] * 1e-3) / 1e12
        return {"kernel": f"window_attn_kernel via {name} (window {tag[1]}, {tag[2]} heads x {tag[3]}, B={tag[4]}, "
                          f"{'shifted+masked' if tag[5] else 'unshifted'})",
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                "algorithmic_flops_per_launch": fl, "avg_launch_ms": v["avg_ms"], "launches_per_rollout": v["calls"],
                "pipe": "bf16 MFMA (dense peak)" if prec == "bf16" else "fp32-accurate form priced against the fp32 matrix peak"}
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
        via = "dlwp_afno_block_tail_f16x3" if any(k[0] == "dlwp_afno_block_tail_f16x3" for k, _ in tail) else "dlwp_afno_block_tail_f32"
        return {"kernel": f"token_mlp_kernel<MERGE, NEXT> via {via} (irfft out + skips + LN2 + fc1/GELU/fc2 + next LN1)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": by, "avg_launch_ms": avg, "launches_per_rollout": n,
                "mfma_view": {"algorithmic_flops_per_launch": fl, "achieved_TFLOPs": fl / (avg * 1e-3) / 1e12,
                              "frac_of_f32_matrix_peak": fl / (avg * 1e-3) / 1e12 / MFMA_F32_PEAK_TF},
                "step_hbm_view": "whole step: SURVEY 8d counts 128 MiB/sample-step; see ms_per_step"}
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
    the affinity mask divided by the hardware threads per core."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="initial conditions per GPU")
    ap.add_argument("--rollout-steps", type=int, default=20)
    ap.add_argument("--cpu-batch", type=int, default=4, help="samples of the bounded CPU-baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU-baseline leg (0 = usable physical cores)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C1/C3/C4/C5 legs")
    ap.add_argument("--only-configs", nargs="*", help="subset of the other-config tags")
    ap.add_argument("--gather-chunks", type=int, default=4)
    ap.add_argument("--collect", choices=["metrics", "gather", "none"], default="metrics",
                    help="what leaves a rank per rollout: per-lead-time RMSE sums reduced on the device and all-reduced "
                         "(default; SURVEY.md 8e/8f-f1), the whole trajectory (one chunked RCCL all-gather), or nothing")
    args = ap.parse_args()

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

    from dlwp_benchmark_amd.sharding import ShardedRollout
    from dlwp_benchmark_amd.synthetic import navier_stokes

    B, K_roll, H, W = args.batch, args.rollout_steps, 64, 64
    model, sha = build_model(device)
    _, _, prog_cpu = navier_stokes(B, K_roll + 1, H, W, seed=1234 + rank)
    prog = prog_cpu.to(device)
    runner = ShardedRollout(model, world_size=world, rank=rank, chunks=args.gather_chunks,
                            gather=(args.collect == "gather"))
    # evaluation scores of the rollout against the frames the synthetic solver produced (uniform weights on the
    # periodic box): reduced on the device, all-reduced across ranks -- [4, K, C] doubles instead of trajectories
    from dlwp_benchmark_amd.metrics import RolloutMetrics

    scorer = RolloutMetrics(torch.zeros(H))
    target = prog[:, model.context_size:].contiguous()
    scores = {}
    acc = {"sums": torch.zeros(4, K_roll, prog.shape[2], dtype=torch.float64, device=device) if args.collect == "metrics" else None,
           "samples": 0}

    def step_eager():
        out = runner(prognostic=prog)
        if args.collect == "metrics":
            # this rank's squared-error sums of the rollout, ADDED to the evaluation's running sums on the device (the
            # reference accumulates over all batches before taking the root, evaluate.py:786-821): no collective per step
            acc["sums"].add_(scorer.sums(out, target))
        return out

    # One bench step (= one rollout of the batch + its metric sums) is a fixed chain of launches on fixed buffers: three
    # workspace memsets, the persistent rollout kernel, the sums kernel, the accumulation.  It is captured ONCE into a HIP
    # graph and replayed per step (launch gaps between the dependent nodes: ~1.5 us instead of ~6 us each).  Only without
    # a collective inside the step (--collect gather issues RCCL calls per chunk) and with the deferred verification
    # (a captured call cannot synchronise).  DLWP_BENCH_GRAPH=0 runs the same step eagerly.
    # Default OFF: measured 2.006 (graph) vs 2.003 ms (eager) per step -- what is left beside the kernel is GPU work, not launch
    # gaps -- and a capture beside a live RCCL communicator (N > 1) is one more thing that can go wrong for no gain.
    use_graph = args.collect != "gather" and world == 1 and os.environ.get("DLWP_BENCH_GRAPH", "0") == "1"
    graph = {"g": None, "out": None}

    def step():
        if graph["g"] is not None:
            graph["g"].replay()
            out = graph["out"]
        else:
            out = step_eager()
        if args.collect == "metrics":
            acc["samples"] += B
        return out

    def finish():
        # deferred verification of every fused launch since the last call (DLWP_ERR_TIMEOUT raises here): the rollouts of an
        # evaluation are enqueued asynchronously and verified ONCE, inside the timed region
        model.verify()
        # the ONE collective of the sharded evaluation: all-reduce of [4, K, C] sums + sample count (inside the timed region)
        if args.collect == "metrics" and acc["sums"] is not None:
            scores["last"] = scorer.finalize(acc["sums"], float(acc["samples"]), H * W, world_size=world)

    for _ in range(max(args.warmup, 1) if use_graph else args.warmup):
        out = step()              # eager warm-up: plans, allocator
    if use_graph:
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                graph["out"] = step_eager()
            graph["g"] = g
            for _ in range(2):
                step()            # replay warm-up
        except Exception as e:    # report, never hide: the line says which form ran
            graph["g"], graph["out"] = None, None
            use_graph = False
            print(f"bench: HIP graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
    finish()                      # also initialises the communicator's all-reduce path before the clock starts
    if acc["sums"] is not None:
        acc["sums"].zero_()
    acc["samples"] = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * H * W * K_roll * args.steps / dt

    result = {
        "metric": "rollout cell-steps/s",
        "value": value,
        "unit": "grid-cells*steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "FNO2d modes=12 hidden=32 lift/proj=256 layers=4, Navier-Stokes 64x64, 20-step rollout, fp32 (BASELINE configs[1])",
            "batch_per_gpu": B, "global_batch": B * world, "grid": [H, W], "rollout_steps": K_roll,
            "parallelism": (f"batch-shard x{world}, " + {"metrics": "on-device RMSE sums accumulated per rank, one all-reduce per evaluation",
                                                          "gather": "chunked all-gather of trajectories",
                                                          "none": "no collective"}[args.collect])
            if world > 1 else "single GPU",
            "collect": args.collect,
            "precision_form": {"f16x3": "f16x3: fp32 accumulation; the big channel products from exact two-part f16 splits of fp32 operands "
                                        "(22 significant bits, three f16 matrix instructions per product; DLWP_BENCH_PRECISION=bf16x6 selects "
                                        "the three-part bf16 form)",
                               "bf16x6": "bf16x6: fp32 accumulation; products from exact three-part bf16 splits (six bf16 matrix instructions)",
                               "fp32_mfma": "plain fp32 matrix instructions, unfused"}[model.precision_form],
            "fused_kernel_check": "deferred: rollouts enqueued asynchronously, verified once per evaluation inside the timed region",
            "launch": "hip graph replay of one step (memsets + rollout kernel + metric sums)" if graph["g"] is not None else "eager",
            "weights": "deterministic filler sha256:" + sha[:16],
        },
    }

    if rank == 0 and world == 1:
        # ---- roofline leg: per-kernel HIP-event timing of the same rollout
        prof, event_overhead_ms = profile_kernels(model, prog, repeats=max(2, min(args.steps, 5)))
        rows_in = 12
        work = algorithmic_work(B, H, W, MODEL_KW, rows_in, MODEL_KW["n_modes"][1] // 2 + 1)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = None
        per_kernel = {}
        for name, (avg_ms, launches) in prof.items():
            w_ = work[name]
            per_kernel[name] = {
                "avg_ms": avg_ms, "launches_per_rollout": launches,
                "GBps": w_["bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None,
                "TFLOPs": w_["flops"] / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else None,
                "share_of_rollout": avg_ms * launches / ms_per_step,
            }
        # the kernel the north star names (spectral-conv layer, HBM-bound).  With the fused trunk
        # (fno_trunk_kernel, the default at this config) ONE launch runs all n_layers spectral layers
        # including their mode mixing, so the algorithmic bytes of a launch are n_layers x the per-layer
        # figure of SURVEY.md 8(d) (4 * P * 2 * Ch + skip weights) plus the spectral weights it reads.
        lay = per_kernel["layer"]
        fused = per_kernel["modes"]["launches_per_rollout"] == 0
        kname = "fno_trunk_kernel" if fused else "fno_layer_kernel"
        whole_step = fused and per_kernel["lift"]["launches_per_rollout"] == 0
        if whole_step:
            kname = "fno_trunk_kernel<STEP> (lifting + all spectral layers + projection in one launch)"
        if fused:
            nl = MODEL_KW["n_layers"]
            lay_bytes = nl * (work["layer"]["bytes"] + work["modes"]["bytes"])
            lay_flops = nl * (work["layer"]["flops"] + work["modes"]["flops"])
            if whole_step:   # + the lifting and projection MLPs' own bytes / flops
                lay_bytes += work["lift"]["bytes"] + work["proj"]["bytes"]
                lay_flops += work["lift"]["flops"] + work["proj"]["flops"]
                # persistent form: one launch runs every step of the rollout
                steps_per_launch = max(1, K_roll // max(lay["launches_per_rollout"], 1))
                lay_bytes *= steps_per_launch
                lay_flops *= steps_per_launch
                lay["rollout_steps_per_launch"] = steps_per_launch
            lay["GBps"] = lay_bytes / (lay["avg_ms"] * 1e-3) / 1e9
            lay["TFLOPs"] = lay_flops / (lay["avg_ms"] * 1e-3) / 1e12
            lay["spectral_layers_per_launch"] = nl
        else:
            lay_bytes = work["layer"]["bytes"]
        if whole_step:
            # lifting + 4 spectral layers + projection in one launch: 7.0 GFLOP of fp32 GEMM work over 171 MB of
            # algorithmic bytes = 41 flop/B, above the ridge of the fp32 roofline (157.3 TF / 8 TB/s = 19.7): the
            # launch is MATRIX-bound by the roofline model, so that is the bound it is priced against (fp32
            # algorithmic flops vs the dense fp32 MFMA peak; the bf16x6 kernels spend 6 bf16 MFMAs per fp32 one and
            # are not credited for that).  The byte view of the same launch is kept alongside.
            result["roofline"] = {
                "kernel": kname, "bound": "mfma", "achieved": lay["TFLOPs"], "peak": MFMA_F32_PEAK_TF,
                "unit": "TFLOP/s", "frac": lay["TFLOPs"] / MFMA_F32_PEAK_TF,
                "traffic": (traffic or {}).get("fno_step_kernel"),
                "algorithmic_flops_per_launch": lay_flops, "algorithmic_bytes_per_launch": lay_bytes,
                "avg_launch_ms": lay["avg_ms"],
                "hbm_view": {"achieved_GBps": lay["GBps"], "frac_of_8TBps": lay["GBps"] / HBM_PEAK_GBS},
                "timing": "HIP events on the launch stream around every launch of the timed rollout; one event-marker "
                          f"latency (half of an empty bracket, {0.5 * event_overhead_ms * 1e3:.2f} us) subtracted",
            }
        else:
          result["roofline"] = {
            "kernel": kname, "bound": "hbm", "achieved": lay["GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": lay["GBps"] / HBM_PEAK_GBS if lay["GBps"] else None,
            "traffic": (traffic or {}).get("fno_step_kernel" if fused and per_kernel["lift"]["launches_per_rollout"] == 0 else kname),
            "algorithmic_bytes_per_launch": lay_bytes, "avg_launch_ms": lay["avg_ms"],
            "timing": "HIP events on the launch stream around every launch of the timed rollout; one event-marker "
                      f"latency (half of an empty bracket, {0.5 * event_overhead_ms * 1e3:.2f} us) subtracted",
        }
        # the two MFMA-bound MLP kernels, priced against the fp32 matrix peak
        for nm in ("lift", "proj"):
            k = per_kernel[nm]
            k["frac_mfma_f32_peak"] = k["TFLOPs"] / MFMA_F32_PEAK_TF if k.get("TFLOPs") else None
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
                "sample": f"oracle (PyTorch {torch.__version__} CPU restatement) on {nb} of the {B} initial conditions, "
                          f"{K_roll} steps; 1 warm-up + {len(secs)} timed rollouts, median {med:.2f} s "
                          f"(all: {', '.join('%.2f' % x for x in secs)}); {nthreads} threads = usable physical cores "
                          f"(os.cpu_count() {os.cpu_count()})",
            }
            result["rel_l2_per_step_max"] = max(errs)
            result["rel_l2_per_step"] = [float(f"{e:.3e}") for e in errs]
            if max(errs) > 1e-5:
                result["parity_ok"] = False

        # ---- the other BASELINE configs (C1, C3, C4, C5) through the HIP path, same process, same GPU
        if not args.no_other_configs:
            try:
                result["other_configs"] = bench_other_configs(device, only=args.only_configs)
            except Exception as e:   # the headline line must survive a failure here; the failure is reported, not hidden
                result["other_configs"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
