# Round-2 final profiles of the headline command on the GPU box: bash tools/pmc_fno.sh <outdir-under-gpurun_out>
#  1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (the default command)  -> <out>/bench_kernel_stats.csv, bench.json
#  2. HBM traffic of the FNO rollout kernel: FETCH_SIZE / WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md HBM
#     section; on gfx950 FETCH_SIZE tallies wide reads at half: doubled by the reader) -> <out>/fno_pmc.txt
#  3. SQ counters of the same kernel (own pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
W=/tmp/prof_$1
rm -rf $W && mkdir -p $W $O
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -- python3 $R/bench.py > $O/bench.json 2> $O/bench.err < /dev/null || echo "kernel-trace run failed"
f=$(find $W/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $W/pmc_$tag -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 6 --warmup 2 > $W/pmc_$tag.log 2>&1 < /dev/null || echo "pmc $tag failed"
  echo "pmc $tag done"
done
python3 $R/tools/pmc_summary.py $W fno_trunk_kernel > $O/fno_pmc.txt
cat $O/fno_pmc.txt
head -12 $O/bench_kernel_stats.csv
