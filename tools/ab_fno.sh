#!/bin/bash
# A/B of FNO step-kernel builds on the GPU box: bash tools/ab_fno.sh <outdir-under-gpurun_out> <tag>...   (tag "base" = the shipped library;
# other tags = dlwp_benchmark_amd/ab/lib_<tag>.so from tools/ab_build2.sh).  Prints kernel ms per rollout and the parity figure per tag.
out=gpurun_out/$1; shift
mkdir -p $out
for tag in "$@"; do
  lib=""
  [ "$tag" != "base" ] && lib=$PWD/dlwp_benchmark_amd/ab/lib_$tag.so
  DLWP_HIP_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-configs --no-second-form --steps 40 --warmup 10 --detail $out/d_$tag.json > $out/b_$tag.json 2> $out/b_$tag.err || { echo "$tag FAILED"; tail -5 $out/b_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/b_$tag.json"))
print("%-14s kernel_ms %.4f  ms/rollout %.4f  value %.4g  rel_l2 %.3g" % ("$tag", d["roofline"]["avg_launch_ms"], d["ms_per_step"], d["value"], d.get("rel_l2_max", d.get("rel_l2_per_step_max", float("nan")))))
PY
done
