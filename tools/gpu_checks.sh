# On the GPU box: the round's closing checks — full -m gpu suite, smoke(), the bench line with the pipelined frame step at N = 1 (so that the
# "other frame step" side pass runs), the fuzz campaign.  usage: bash tools/gpu_checks.sh OUTNAME [FUZZ_CASES] [FUZZ_SEED]
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || echo "pytest FAILED"
tail -2 $O/pytest_gpu.txt | cut -c1-200
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 || echo "smoke FAILED"
tail -1 $O/smoke.txt
timeout -k 10 200 python3 bench.py --async-frames --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_async.json 2> $O/bench_async.err || echo "bench --async-frames FAILED"
python3 -c "
import json
j=json.loads(open('$O/bench_async.json').read().strip().splitlines()[-1])
print('async bench:', round(j['ms_per_step'],4), j['config']['frame_steps_ms'])"
timeout -k 10 1000 python3 tests/fuzz_parity.py ${2:-3000} ${3:-31} > $O/fuzz.txt 2>&1 || echo "fuzz FAILED or timed out"
tail -1 $O/fuzz.txt | cut -c1-400
