set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/time_shard.py > $O/time_shard.txt 2>&1 || echo "time_shard failed"
grep "G=" $O/time_shard.txt | cut -c1-300
SKR_FLAT=0 G_LIST=8 MODES=interleave timeout -k 10 300 python3 tools/time_shard.py > $O/time_shard_persistent.txt 2>&1 || echo "time_shard failed"
grep "G=" $O/time_shard_persistent.txt | cut -c1-300
timeout -k 10 200 python3 -m pytest tests -x -q -m gpu -k "cost_aware or native_frame_step or multi_rank" > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -3 $O/pytest_gpu.txt | cut -c1-300
