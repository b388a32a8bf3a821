# SQ / memory counters of the kernels of one BASELINE config on the GPU box: bash tools/pmc_config.sh <outdir-under-gpurun_out> <C3|C4|C5> <kernel-name-filter> [precision]
#   separate rocprofv3 --pmc passes (never combined with a trace domain) over `bench.py --config <C>`, averaged per kernel by tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
C=$2
F=$3
P=${4:-default}
W=/tmp/pmcc_$1_$C
rm -rf $W && mkdir -p $W $O
ARGS="--config $C --no-cpu-baseline --steps 3 --warmup 1 --no-graph-step"
[ "$P" != "default" ] && ARGS="$ARGS --precision $P"
for c in "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $W/pmc_$tag -- python3 $R/bench.py $ARGS > $W/pmc_$tag.log 2>&1 < /dev/null || echo "pmc $tag failed"
done
python3 $R/tools/pmc_summary.py $W "$F" > $O/${C}_${P}_pmc.txt
cat $O/${C}_${P}_pmc.txt
