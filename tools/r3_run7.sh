set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -4 $O/pytest_gpu.txt
bash tools/profile_round.sh r3g_cfg3 3
