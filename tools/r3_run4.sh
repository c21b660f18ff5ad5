set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
timeout -k 10 300 tools/ubench/issue_rates --quick > $O/ubench_branch.txt 2>&1
head -30 $O/ubench_branch.txt | cut -c1-250
bash tools/pmc_pass.sh gpurun_out/r3d/pmc "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES"
grep -A18 "leaf" $O/pmc/summary.txt | head -40
