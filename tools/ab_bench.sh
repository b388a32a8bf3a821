#!/bin/bash
# A/B timing of libdlwp_hip.so variants on ONE box: tools/ab_bench.sh <outdir> <tag> [<tag> ...]   (two interleaved passes)
out=gpurun_out/$1; shift
mkdir -p $out
for pass in 1 2; do
  for tag in "$@"; do
    DLWP_HIP_LIB=$PWD/dlwp_benchmark_amd/ab/lib_$tag.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-configs > $out/bench_${tag}_$pass.json 2> $out/bench_${tag}_$pass.err || { echo "$tag failed"; tail -5 $out/bench_${tag}_$pass.err; continue; }
    python - <<PY
import json
d=json.load(open("$out/bench_${tag}_$pass.json"))
print("%-10s pass $pass: ms/rollout %.4f kernel_ms %.4f frac %.4f" % ("$tag", d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
  done
done
