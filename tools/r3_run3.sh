set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -8 $O/pytest_gpu.txt
for v in "" var/libskr_noexact.so var/libskr_noconst.so; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$R/skele_raytracer_amd/lib/$v; fi
  timeout -k 10 200 python3 tools/ab_nodes.py 2>/dev/null | grep G= | tee -a $O/ab.txt
done
unset SKR_LIBRARY
timeout -k 10 300 tools/ubench/issue_rates --quick > $O/ubench_x128.txt 2>&1
head -18 $O/ubench_x128.txt | cut -c1-260
