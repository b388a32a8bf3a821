# Per-kernel breakdown of one BASELINE config on the GPU box:  bash tools/prof_config.sh <outdir-under-gpurun_out> <C3|C4|C5> [precision]
#   rocprofv3 --kernel-trace --stats of `python3 bench.py --config <C> [--precision <p>]` -> <out>/<C>_<p>_kernel_stats.csv + bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
C=$2
P=${3:-default}
W=/tmp/profc_$1_$C_$P
rm -rf $W && mkdir -p $W $O
ARGS="--config $C --no-cpu-baseline"
[ "$P" != "default" ] && ARGS="$ARGS --precision $P"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -- python3 $R/bench.py $ARGS --detail $O/${C}_${P}_detail.json > $O/${C}_${P}_bench.json 2> $O/${C}_${P}.err < /dev/null || echo "kernel-trace run failed"
f=$(find $W/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${C}_${P}_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/${C}_${P}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("$C $P: total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:22]:
    print("%6.2f%%  calls %6s  avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:150]))
PY
