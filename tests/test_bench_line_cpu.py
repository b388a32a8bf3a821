"""The contract line of bench.py (its LAST stdout line) must stay small enough for the driver's stdout tail and parse as
one JSON object whatever the legs put into the result (a 21.9 KB line left BENCH_r02.json unparsed)."""
import json

import bench


def _canned(n_other=14, long_text=1):
    other = {}
    for i in range(n_other):
        other[f"C{i % 5 + 1}_variant{i}"] = {
            "workload": "PanguWeather 128x256, 13 prognostic ch, 5-step rollout " * long_text, "dtype": "bf16 " * 10, "batch": 8,
            "rollout_steps": 5, "weights": "deterministic filler sha256:0123456789abcdef", "ms_per_rollout": 72.123456789,
            "ms_per_step": 14.4246913578, "cell_steps_per_s": 9087654.321, "finite": True,
            "rel_l2_per_step_vs_golden": [1.234e-3] * 20, "rel_l2_max": 4.123456e-3, "rel_l2_bound": 5e-3, "parity_ok": True,
            "golden": "tests/golden/model_pangu_c5_full.npz (real reference class, 1 sample, 2 steps)",
            "hip_entry_points_ms_per_rollout": {f"dlwp_entry_{j}": 1.23456 for j in range(6)}, "share_outside_libdlwp_hip": 0.0123,
            "roofline": {"kernel": "window attention via dlwp_window_attn_bf16 " * 3, "bound": "mfma", "achieved": 512.3456, "peak": 2500.0,
                         "unit": "TFLOP/s", "frac": 0.20493824, "traffic": None, "algorithmic_flops_per_launch": 3.08e10},
            "roofline_linear": {"kernel": "linear_kernel via dlwp_linear_bf16_io (65536 x 384 -> 1536, GELU, bf16 output)", "frac": 0.21},
        }
    return {
        "metric": "rollout cell-steps/s", "value": 1812345678.9123, "unit": "grid-cells*steps/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 1.4467891234, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (f16x3: products from two-part f16 splits, 22-bit operands, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "C2 FNO2d modes=12 hidden=32 lift/proj=256 layers=4, Navier-Stokes 64x64, 20-step rollout " * long_text,
                   "batch_per_gpu": 32, "global_batch": 32, "grid": [64, 64], "rollout_steps": 20,
                   "parallelism": "batch-shard x8, on-device RMSE sums per rank, one all-reduce per evaluation" * long_text,
                   "collect": "metrics", "precision_form": "f16x3", "launch": "eager; deferred check " * long_text, "weights": "filler sha256:0123456789ab"},
        "roofline": {"kernel": "fno_trunk_kernel<STEP> (lifting + spectral layers + projection, all rollout steps, one launch)", "bound": "hbm",
                     "achieved": 2420.123456, "peak": 8000.0, "unit": "GB/s", "frac": 0.302515432, "traffic": 1524020326.4,
                     "traffic_src": "profiles/traffic.json: rocprofv3 --pmc at 66a6799, launch 1.41 ms", "bytes_per_launch": 3430000000.0,
                     "flops_per_launch": 130.6e9, "avg_launch_ms": 1.41812345, "launches_per_step": 1, "fp32_equivalent_TFLOPs": 92.123456,
                     "timing": "HIP events on the launch stream around every launch, one marker latency (2.3 us) subtracted",
                     "not_in_line": list(range(500))},
        "cpu_baseline": {"value": 1234567.891, "unit": "grid-cells*steps/s", "cores": 16, "kind": "port",
                         "sample": "oracle (PyTorch CPU restatement) on 4 of 32 initial conditions, 20 steps; 1 warm-up + 3 timed rollouts " * long_text},
        "rel_l2_per_step_max": 5.61234e-7, "rel_l2_bound": 1e-5, "parity_ok": True, "value_bf16x6": 1512345678.9, "fused_timeouts": 0,
        "range_reruns": 0, "kernels": {f"k{i}": {"avg_ms": 0.1, "blob": "x" * 500} for i in range(10)}, "rel_l2_per_step": [5e-7] * 20,
        "other_configs": other, "detail": "profiles/bench_detail_last.json", "commit": "abcdef0",
    }


def _check(line):
    assert "\n" not in line
    assert len(line.encode()) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert "kernels" not in d and "rel_l2_per_step" not in d
    return d


def test_default_line_is_small_and_parses():
    d = _check(bench.compact_line(_canned()))
    oc = d["other_configs"]
    assert len(oc) == 14
    for e in oc.values():
        assert set(e) == {"ms_per_step", "rel_l2_max", "parity_ok", "roofline_frac", "bound"}
    assert abs(d["value"] - 1812345678.9) / 1.8e9 < 1e-4          # rounded to 5 significant digits, not mangled
    assert d["fused_timeouts"] == 0 and "value_bf16x6" in d


def test_line_survives_pathological_inputs():
    """verbose free text and many more configs than exist: optional fields go first, then trailing other-config entries;
    the contract keys never."""
    d = _check(bench.compact_line(_canned(n_other=60, long_text=6)))
    assert d["other_configs"].get("truncated") is True


def test_other_config_failure_is_visible_in_the_line():
    r = _canned()
    r["other_configs"] = {"error": "DlwpError: boom"}
    d = _check(bench.compact_line(r))
    assert d["other_configs"] == {"error": "DlwpError: boom"}


def test_roofline_fractions_are_fractions():
    """every pricing helper divides by the peak of the pipe the kernel's products RUN on: never above 1 at physical rates"""
    summ = {("dlwp_linear_f32", (65536, 384, 1536, 1, False, 0, 0)): {"calls": 4, "total_ms": 4 * 0.40, "avg_ms": 0.40},
            ("dlwp_linear_f16x3", (65536, 384, 1536, 1, False, 0, 0)): {"calls": 1, "total_ms": 0.28, "avg_ms": 0.28}}
    r = bench._linear_roofline(summ)
    assert r["peak"] == bench.MFMA_16BIT_PEAK_TF and 0 < r["frac"] < 1
    assert r["matrix_products_per_algorithmic_product"] == 6
