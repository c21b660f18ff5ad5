set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
for v in "" var/libskr_B.so var/libskr_C.so var/libskr_D.so; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$R/skele_raytracer_amd/lib/$v; fi
  timeout -k 10 300 python3 tests/check_nodes.py > $O/check_$(basename "$v" .so).txt 2>&1 || echo "check_nodes FAILED for $v"
  tail -1 $O/check_$(basename "$v" .so).txt
  timeout -k 10 200 python3 tools/ab_nodes.py 2>/dev/null | grep G= | tee -a $O/ab.txt
done
export SKR_LIBRARY=$R/skele_raytracer_amd/lib/var/libskr_C.so
bash tools/pmc_pass.sh gpurun_out/r3f/pmcC "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
grep -A9 "leaf" $O/pmcC/summary.txt | head -12
