# Profiles of the headline command on the GPU box: bash tools/pmc_fno.sh <outdir-under-gpurun_out> [commit]
#  1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (the default command)  -> <out>/bench_kernel_stats.csv, bench.json
#  2. HBM traffic of the FNO rollout kernel: FETCH_SIZE / WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md HBM
#     section; on gfx950 FETCH_SIZE tallies wide reads at half: doubled by the reader) -> <out>/fno_pmc.txt
#  3. SQ counters of the same kernel (own pass)
#  4. <out>/traffic.json: bytes per launch of the rollout kernel from THIS run's counters, stamped with the commit and the launch
#     duration the kernel trace of the same run saw -- copy it to profiles/traffic.json (bench.py reads it for roofline.traffic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
COMMIT=${2:-unknown}
W=/tmp/prof_$1
rm -rf $W && mkdir -p $W $O
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -- python3 $R/bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err < /dev/null || echo "kernel-trace run failed"
f=$(find $W/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $W/pmc_$tag -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-second-form --steps 6 --warmup 2 > $W/pmc_$tag.log 2>&1 < /dev/null || echo "pmc $tag failed"
  echo "pmc $tag done"
done
python3 $R/tools/pmc_summary.py $W fno_trunk_kernel > $O/fno_pmc.txt
cat $O/fno_pmc.txt
python3 - <<PY
import csv, json, re
fetch = write = None
for line in open("$O/fno_pmc.txt"):
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
    if m and m.group(1) == "FETCH_SIZE": fetch = float(m.group(2))
    if m and m.group(1) == "WRITE_SIZE": write = float(m.group(2))
avg_ms = None
try:
    for row in csv.DictReader(open("$O/bench_kernel_stats.csv")):
        if "fno_trunk_kernel" in row.get("Name", ""):
            avg_ms = float(row["AverageNs"]) / 1e6
            break
except Exception as e:
    print("kernel stats:", e)
if fetch is not None and write is not None:
    t = {"fno_step_kernel": {"bytes": (2.0 * fetch + write) * 1024.0, "commit": "$COMMIT", "avg_launch_ms": avg_ms,
                             "fetch_kb": fetch, "write_kb": write,
                             "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py; FETCH_SIZE doubled (gfx950 tallies "
                                     "128-B requests at 64 B, MI355X_MICROARCH.md HBM section); per launch = one 20-step rollout of 32 samples"}}
    json.dump(t, open("$O/traffic.json", "w"), indent=1)
    print(json.dumps(t))
PY
head -12 $O/bench_kernel_stats.csv
