# Timeline of one headline rollout period on the GPU box: bash tools/trace_gaps.sh <outdir-under-gpurun_out> [bench args]
#   rocprofv3 --kernel-trace (timestamps, not only --stats) of a short bench run -> per-kernel start/end of the last periods
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
W=/tmp/gaps_$$
rm -rf $W && mkdir -p $W $O
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $W/kt -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-second-form --steps 12 --warmup 4 "$@" > $O/bench.json 2> $O/bench.err < /dev/null || echo "trace run failed"
f=$(find $W/kt -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 3 rollout kernels of the timed region and everything between them
idx = [i for i, r in enumerate(rows) if "fno_trunk_kernel" in r["Kernel_Name"]]
sel = idx[-12:-8] if len(idx) > 14 else idx[-4:]
t0 = int(rows[sel[0]]["Start_Timestamp"])
prev_end = None
for r in rows[sel[0]:sel[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  +gap %6.1f  dur %8.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev_end = e
PY
