set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3r; mkdir -p $O
cd $R
V=$R/skele_raytracer_amd/lib/var
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$V/libskr_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || echo "bench $v failed"
  python3 -c "
import json,sys
j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', 'frame ms', round(j['ms_per_step'],4), 'kernel ms', round(j['roofline']['kernel_ms'],4))"
  timeout -k 10 300 python3 -m pytest tests -x -q -m gpu -k "variants_agree or whole_frame or bands" > $O/pytest_$v.txt 2>&1 || echo "pytest $v failed"
  tail -2 $O/pytest_$v.txt | cut -c1-200
done
