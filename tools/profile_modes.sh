# On the GPU box: rocprofv3 kernel-trace stats of the capability rows that bench.py's configurations do not cover:
# dragon.scn --strict-scn --shade-triangles, spheres2.scn --legacy-reflect, test.scn 640x360 --gillum 4 (a mesh under the tree).
# usage: bash tools/profile_modes.sh OUTNAME
set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, then profile_scene.py arguments
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 $R/tools/profile_scene.py "$@" > $O/$n.log 2>&1
  cp $(find $O/$n -name "*kernel_stats.csv" | head -1) $O/${n}_kernel_stats.csv
  grep "ms per frame" $O/$n.log; head -5 $O/${n}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
}
run dragon_surfaces dragon.scn 1920 1080 shade_triangles=1 strict=1 reps=20
run legacy_reflect spheres2.scn 1920 1080 legacy_reflect=1 shadow=1 depth=3 reps=20
run test_gillum4 test.scn 640 360 gillum=4 shadow=1 reps=20
