set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -1 gpurun_out/bench_final.json
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err; tail -1 gpurun_out/bench_torchrun.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_final -name "*kernel_stats.csv" | head -1 | xargs head -6
