set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O
cd $R
timeout -k 10 200 python3 tools/time_split.py > $O/split2.txt 2>&1 || echo "split failed"
grep "G=\|same" $O/split2.txt
LANES=3 timeout -k 10 200 python3 tools/time_split.py > $O/split3.txt 2>&1 || echo "split failed"
grep "G=\|same" $O/split3.txt
LANES=4 timeout -k 10 200 python3 tools/time_split.py > $O/split4.txt 2>&1 || echo "split failed"
grep "G=\|same" $O/split4.txt
