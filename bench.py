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


def cpu_baseline(state_dict, prog_cpu, rollout_steps):
    """Times the oracle on the host cores; returns (cell-steps/s, seconds, trajectory)."""
    from oracle.restate.fno import FNO2DModuleRef  # checker / reported baseline only

    ref = FNO2DModuleRef(**MODEL_KW).eval()
    ref.load_state_dict(state_dict)
    with torch.no_grad():
        ref(prognostic=prog_cpu[:, :2])  # warm-up: one step
        t0 = time.perf_counter()
        traj = ref(prognostic=prog_cpu)
        dt = time.perf_counter() - t0
    b, _, _, h, w = prog_cpu.shape
    return b * h * w * rollout_steps / dt, dt, traj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="initial conditions per GPU")
    ap.add_argument("--rollout-steps", type=int, default=20)
    ap.add_argument("--cpu-batch", type=int, default=8, help="samples of the bounded CPU-baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    acc = {"sums": None, "samples": 0}

    def step():
        out = runner(prognostic=prog)
        if args.collect == "metrics":
            # this rank's squared-error sums of the rollout, ADDED to the evaluation's running sums on the device (the
            # reference accumulates over all batches before taking the root, evaluate.py:786-821): no collective per step
            s = scorer.sums(out, target)
            acc["sums"] = s if acc["sums"] is None else acc["sums"].add_(s)
            acc["samples"] += B
        return out

    def finish():
        # the ONE collective of the sharded evaluation: all-reduce of [4, K, C] sums + sample count (inside the timed region)
        if args.collect == "metrics" and acc["sums"] is not None:
            scores["last"] = scorer.finalize(acc["sums"], float(acc["samples"]), H * W, world_size=world)

    for _ in range(args.warmup):
        out = step()
    finish()                      # also initialises the communicator's all-reduce path before the clock starts
    acc["sums"], acc["samples"] = None, 0
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
            cps, secs, traj = cpu_baseline(sd, prog_cpu[:nb].contiguous(), K_roll)
            got = out[:nb].detach().cpu().double()
            want = traj.double()
            errs = [float(torch.linalg.vector_norm(got[:, t] - want[:, t]) / torch.linalg.vector_norm(want[:, t]))
                    for t in range(K_roll)]
            result["cpu_baseline"] = {
                "value": cps, "unit": "grid-cells*steps/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"oracle (PyTorch {torch.__version__} CPU restatement) on {nb} of the {B} initial conditions, "
                          f"{K_roll} steps, {secs:.2f} s",
            }
            result["rel_l2_per_step_max"] = max(errs)
            result["rel_l2_per_step"] = [float(f"{e:.3e}") for e in errs]

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
