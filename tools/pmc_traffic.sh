# HBM traffic of the Linear and earth-window attention kernels (FETCH_SIZE / WRITE_SIZE in separate passes, as
# MI355X_MICROARCH.md prescribes; on gfx950 FETCH_SIZE counts wide coalesced reads at half: doubled in the summary).
# On the GPU box: bash tools/pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/pmc_traffic
rm -rf $OUT && mkdir -p $OUT $R/gpurun_out/r02w
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d $OUT/lin_$c -- python3 $R/tools/bench_linear.py --only "C5 l2" --reps 2 > $OUT/lin_$c.log 2>&1 < /dev/null || echo "linear $c failed"
  timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d $OUT/att_$c -- python3 $R/tools/bench_attn.py --only pangu_l1 --reps 2 > $OUT/att_$c.log 2>&1 < /dev/null || echo "attn $c failed"
done
python3 $R/tools/pmc_summary.py $OUT linear_kernel > $R/gpurun_out/r02w/pmc_traffic.txt
python3 $R/tools/pmc_summary.py $OUT wattn3_kernel >> $R/gpurun_out/r02w/pmc_traffic.txt
cat $R/gpurun_out/r02w/pmc_traffic.txt
