set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
cd $R
V=$R/skele_raytracer_amd/lib/var
SKR_LIBRARY=$V/libskr_timeline.so timeout -k 10 120 python3 tools/leaf_timeline.py > $O/timeline.txt 2>&1 || echo "timeline failed"
grep "G=" $O/timeline.txt | cut -c1-700
SKR_LIBRARY=$V/libskr_stamps.so timeout -k 10 120 python3 tools/stamps_nodes.py > $O/stamps.txt 2>&1 || echo "stamps failed"
grep "stamp\|phases" $O/stamps.txt
for v in "" nobreak; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$V/libskr_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || echo "bench $v failed"
  python3 -c "
import json,sys
j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', 'frame ms', round(j['ms_per_step'],4), 'kernel ms', round(j['roofline']['kernel_ms'],4))"
done
