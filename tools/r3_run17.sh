set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O
cd $R
V=$R/skele_raytracer_amd/lib/var
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$V/libskr_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || echo "bench $v failed"
  python3 -c "
import json,sys
j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', 'frame ms', round(j['ms_per_step'],4), 'kernel ms', round(j['roofline']['kernel_ms'],4))"
done
unset SKR_LIBRARY
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "variants_agree or parity or fuzz or golden or level_pipeline" > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -3 $O/pytest_gpu.txt | cut -c1-300
for c in 2 4; do
timeout -k 10 200 python3 bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c$c.json 2> $O/bench_c$c.err || echo "bench config $c failed"
python3 -c "
import json,sys
j=json.loads(open('$O/bench_c$c.json').read().strip().splitlines()[-1])
print('config $c', 'frame ms', round(j['ms_per_step'],4), 'kernel ms', round(j['roofline']['kernel_ms'],4))"
done
