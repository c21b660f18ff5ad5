set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_statistics.py -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -6 $O/pytest_gpu.txt
for v in "" var/libskr_branchy.so; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$R/skele_raytracer_amd/lib/$v; fi
  timeout -k 10 200 python3 tools/ab_nodes.py 2>/dev/null | grep G= | tee -a $O/ab.txt
done
unset SKR_LIBRARY
bash tools/pmc_pass.sh gpurun_out/r3e/pmc "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
grep -A9 "leaf" $O/pmc/summary.txt | head -12
