# On the GPU box: kernel-trace stats + PMC passes of the bench workload, summaries under gpurun_out/$1 (copy the ones to keep into profiles/).
set -e -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cd $R
bash tools/pmc_pass.sh gpurun_out/$1/pmc FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS"
python3 tools/pmc_traffic.py gpurun_out/$1/pmc gpurun_out/$1/hbm_traffic.json > $OUT/traffic.txt
cp $OUT/hbm_traffic.json profiles/r02_hbm_traffic.json   # (on the box: the bench line below reports the traffic just measured on these very sources)
timeout -k 10 300 python3 bench.py --steps 600 --warmup 30 > $OUT/bench.json 2> $OUT/bench.err
head -8 $OUT/kernel_stats.csv; tail -3 $OUT/traffic.txt
