# On the GPU box: only the two kernel traces of tools/profile_round.sh (serial and pipelined frame step) for one configuration.
# usage: bash tools/profile_stats_only.sh NAME [CONFIG]
set -e -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
CFG=${2:-3}
export BENCH_ARGS="--config $CFG"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=$([ "$CFG" = 5 ] && echo "--steps 2 --warmup 1" || echo "--steps 20 --warmup 3")
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $BENCH_ARGS $STEPS --no-cpu-baseline --sync-frames --no-side-pass > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipelined -- python3 $R/bench.py $BENCH_ARGS $STEPS --no-cpu-baseline --no-side-pass > $OUT/trace_pipelined.log 2>&1
cp $(find $OUT/trace_pipelined -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_pipelined.csv
head -5 $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-120; head -5 $OUT/kernel_stats_pipelined.csv | cut -d, -f1-4 | cut -c1-120
