set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O
cd $R
V=$R/skele_raytracer_amd/lib/var
for v in $VARIANTS; do
  export SKR_LIBRARY=$V/libskr_$v.so
  timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 gillum=16 reps=20 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] dragon: #"
  timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 shade_triangles=1 strict=1 reps=10 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] dragon surfaces: #"
  timeout -k 10 100 python3 tools/profile_scene.py test.scn 640 360 gillum=4 shadow=1 reps=10 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] test.scn gillum 4: #"
  timeout -k 10 100 python3 tools/profile_scene.py test.scn 1920 1080 reps=10 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] test.scn 1080p: #"
done
