cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/pmc_fft
rm -rf $OUT && mkdir -p $OUT $R/gpurun_out/r02z
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- python3 $R/bench.py --only-configs C4_fourcastnet_128x256 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/s$i.log 2>&1 < /dev/null || echo "set $i failed"
done
python3 $R/tools/pmc_summary.py $OUT afno_ > $R/gpurun_out/r02z/pmc_fft.txt
cat $R/gpurun_out/r02z/pmc_fft.txt
