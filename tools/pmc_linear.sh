# SQ counters of the Linear kernel (one shape), separate --pmc passes; usage on the GPU box: bash tools/pmc_linear.sh "<shape label>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE="${1:-C5 l2 qkv}"
OUT=/tmp/pmc_linear
rm -rf $OUT && mkdir -p $OUT $R/gpurun_out/r02m
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $OUT/s$i -- python3 $R/tools/bench_linear.py --only "$SHAPE" --reps 3 > $OUT/s$i.log 2>&1 || { echo "set $i failed"; tail -5 $OUT/s$i.log; }
done
python3 $R/tools/pmc_summary.py $OUT linear_v2 > $R/gpurun_out/r02m/pmc_linear_v2.txt
cat $R/gpurun_out/r02m/pmc_linear_v2.txt
