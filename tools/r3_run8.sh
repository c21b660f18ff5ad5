set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
for c in 2 4 5; do bash tools/profile_round.sh r3h_cfg$c $c > $O/cfg$c.log 2>&1 || { echo "config $c failed"; tail -20 $O/cfg$c.log; }; tail -2 $O/cfg$c.log | cut -c1-400; done
cd /tmp && export TMPDIR=/tmp
for pipe in nodes queue; do
  SKR_PIPELINE=$pipe timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/test_$pipe -- python3 $R/tools/profile_scene.py test.scn 640 360 gillum=4 shadow=1 reps=10 > $O/test_$pipe.log 2>&1 || echo "test.scn $pipe failed"
  grep "ms per frame" $O/test_$pipe.log; head -7 $(find $O/test_$pipe -name "*kernel_stats.csv" | head -1) | cut -c1-200
done
