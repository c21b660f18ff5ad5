set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout -k 10 300 tools/ubench/issue_rates > $O/ubench.txt 2>&1
echo ubench done
timeout -k 10 200 python3 tools/ab_nodes.py > $O/ab_base.txt 2>&1
SKR_LIBRARY=$R/skele_raytracer_amd/lib/var/libskr_stamps.so timeout -k 10 120 python3 tools/stamps_nodes.py > $O/stamps.txt 2>&1
SKR_LIBRARY=$R/skele_raytracer_amd/lib/var/libskr_timeline.so timeout -k 10 120 python3 tools/leaf_timeline.py > $O/timeline.txt 2>&1
cat $O/ab_base.txt $O/stamps.txt $O/timeline.txt; tail -30 $O/ubench.txt
