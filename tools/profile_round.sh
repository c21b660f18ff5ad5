# On the GPU box: kernel-trace stats + PMC passes of the bench workload, summaries under gpurun_out/$1 and copies under profiles/r03_*$SUFFIX.
# usage: bash tools/profile_round.sh NAME [CONFIG]   (CONFIG: bench.py --config, default 3)
set -e -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
CFG=${2:-3}
SUF=$([ "$CFG" = 3 ] && echo "" || echo "_config$CFG")
export BENCH_ARGS="--config $CFG"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=$([ "$CFG" = 5 ] && echo "--steps 2 --warmup 1" || echo "--steps 20 --warmup 3")
# two kernel traces: the serial frame step (--sync-frames: one frame at a time, every kernel alone on the device — the durations bench.py's
# roofline.kernel_ms must agree with) and the default, pipelined one (two frames in flight: kernels of consecutive frames overlap and stretch each other)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $BENCH_ARGS $STEPS --no-cpu-baseline --sync-frames --no-side-pass > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipelined -- python3 $R/bench.py $BENCH_ARGS $STEPS --no-cpu-baseline --no-side-pass > $OUT/trace_pipelined.log 2>&1
cp $(find $OUT/trace_pipelined -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pipelined.csv
cd $R
if [ "$CFG" = 5 ]; then
  bash tools/pmc_pass.sh gpurun_out/$1/pmc FETCH_SIZE WRITE_SIZE
else
  bash tools/pmc_pass.sh gpurun_out/$1/pmc FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS"
fi
python3 tools/pmc_traffic.py gpurun_out/$1/pmc gpurun_out/$1/hbm_traffic.json $CFG > $OUT/traffic.txt
TJ=profiles/r03_hbm_traffic$SUF.json
cp $OUT/hbm_traffic.json $TJ   # (on the box: the bench line below reports the traffic just measured on these very sources)
timeout -k 10 600 python3 bench.py $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/kernel_stats.csv profiles/r03${SUF}_kernel_stats.csv 2>/dev/null || true
head -8 $OUT/kernel_stats.csv; tail -3 $OUT/traffic.txt; cat $OUT/bench.json | cut -c1-600
