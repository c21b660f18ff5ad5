set -e -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in 2 4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg$c -- python3 $R/tools/profile_config.py $c > $R/gpurun_out/prof_cfg$c.log 2>&1
  find $R/gpurun_out/prof_cfg$c -name "*kernel_stats.csv" | head -1 | xargs head -4
done
