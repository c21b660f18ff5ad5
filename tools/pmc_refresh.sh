# Separate rocprofv3 --pmc passes (never combined with tracing) for the bench workload; summaries land in gpurun_out/.
set -e -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_final
mkdir -p $R/gpurun_out/pmc_final
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  d=$R/gpurun_out/pmc_final/$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $d.log 2>&1 || { echo "pass $c failed"; tail -5 $d.log; exit 1; }
  echo "pass $c done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_final > $R/gpurun_out/pmc_final_summary.txt
cat $R/gpurun_out/pmc_final_summary.txt
