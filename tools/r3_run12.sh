set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/time_shard.py > $O/time_shard.txt 2>&1 || echo "time_shard failed"
grep "G=" $O/time_shard.txt | cut -c1-300
timeout -k 10 200 python3 -m pytest tests -x -q -m gpu -k "cost_aware or native_frame_step or multi_rank" > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -3 $O/pytest_gpu.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
export G_LIST=8 MODES=interleave
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o share8 --output-format csv -- python3 $R/tools/time_shard.py > $O/prof_log.txt 2>&1 || echo "rocprof failed"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/share8_kernel_stats.csv
cut -d, -f1-6 $O/share8_kernel_stats.csv | cut -c1-200
