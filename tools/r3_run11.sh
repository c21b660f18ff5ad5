set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -12 $O/pytest_gpu.txt | cut -c1-300
timeout -k 10 300 python3 tools/time_shard.py > $O/time_shard.txt 2>&1 || echo "time_shard failed"
grep "G=" $O/time_shard.txt | cut -c1-300
timeout -k 10 120 python3 tests/check_generic.py > $O/check_generic.txt 2>&1 || echo "check_generic FAILED"
tail -2 $O/check_generic.txt
