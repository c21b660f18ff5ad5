# A/B of builds (SKR_LIBRARY list in $@): leaf-kernel time and its WRITE_SIZE (one PMC pass each)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export SKR_LIBRARY=$R/$lib
  python3 $R/tools/ab_nodes.py 2>/dev/null | grep "G="
  rm -rf /tmp/pmcw; timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmcw.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmcw leaf | grep -A1 "leaf"
done
