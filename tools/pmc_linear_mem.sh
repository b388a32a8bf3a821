# Memory-pipeline counters of the bf16 Linear kernels for one shape (separate --pmc passes):
#   bash tools/pmc_linear_mem.sh "<shape label>" <outdir-under-gpurun_out> [ring 0|1]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE="${1:-C5 l2 qkv}"
O=$R/gpurun_out/${2:-pmc_lin}
export DLWP_LINEAR_RING=${3:-1}
W=/tmp/pmc_linear_mem
rm -rf $W && mkdir -p $W $O
i=0
for set in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $W/s$i -- python3 $R/tools/bench_linear.py --precision bf16 --io --only "$SHAPE" --reps 3 > $W/s$i.log 2>&1 || { echo "set $i failed"; tail -5 $W/s$i.log; }
done
python3 $R/tools/pmc_summary.py $W linear_ > $O/pmc_linear_mem_ring${DLWP_LINEAR_RING}.txt
cat $O/pmc_linear_mem_ring${DLWP_LINEAR_RING}.txt
