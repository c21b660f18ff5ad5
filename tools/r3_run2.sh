set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout -k 10 200 tools/ubench/exact_ops > $O/exact_ops.txt 2>&1
echo exact_ops done; cat $O/exact_ops.txt
timeout -k 10 300 tools/ubench/issue_rates --quick > $O/ubench_quick.txt 2>&1
timeout -k 10 300 tools/ubench/issue_rates --quick --iters 2048 --warm-ms 0 > $O/ubench_cold_short.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/ub_pmc -- $R/tools/ubench/issue_rates --quick --plain > $O/ub_pmc.log 2>&1 || echo "pmc pass failed"
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py -x -q -m gpu -k "units or variants or smoke or golden" > $O/pytest_subset.txt 2>&1 || echo "pytest subset failed"
tail -5 $O/pytest_subset.txt
timeout -k 10 200 python3 tools/ab_nodes.py > $O/ab_specv2.txt 2>&1
cat $O/ab_specv2.txt | grep G=
cat $O/ubench_quick.txt
