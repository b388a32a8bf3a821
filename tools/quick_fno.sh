#!/bin/bash
# Quick FNO check on the GPU box: headline bench line (no CPU baseline, no other configs) + the FNO GPU tests.
# usage: bash tools/quick_fno.sh <outdir-under-gpurun_out> [notest]
set -e
out=gpurun_out/$1
mkdir -p $out
timeout -k 10 200 python bench.py --cpu-batch 1 --no-other-configs --no-second-form --detail $out/bench_detail.json > $out/bench.json 2> $out/bench.err
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print("value %.4g ms/rollout %.4f kernel_ms %.4f frac %.4f rel_l2 %.3g" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("rel_l2_max", d.get("rel_l2_per_step_max", float("nan")))))
PY
if [ "$2" != "notest" ]; then
timeout -k 10 500 python -m pytest tests/test_fno_gpu.py tests/test_gelu_poly.py -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
fi
