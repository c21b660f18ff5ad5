# usage: [BENCH_ARGS="--config 4"] pmc_pass.sh OUTDIR "COUNTER COUNTER ..." ["COUNTERS of a second pass" ...] — separate rocprofv3 --pmc passes of the bench workload
set -e -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
for c in "$@"; do
  d=$OUT/$(echo $c | tr ' ' '_' | cut -c1-48)
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $d -- python3 $R/bench.py $BENCH_ARGS --steps 2 --warmup 1 --no-cpu-baseline > $d.log 2>&1 || { echo "pass $c failed"; tail -5 $d.log; exit 1; }
  echo "pass $c done"
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
