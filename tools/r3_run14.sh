set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
PRINT_WORK=1 timeout -k 10 300 python3 tools/time_shard.py > $O/time_shard.txt 2>&1 || echo "time_shard failed"
grep -c WORK $O/time_shard.txt
