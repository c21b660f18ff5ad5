set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O
cd $R
timeout -k 10 600 python3 tests/check_generic.py > $O/check_generic.txt 2>&1 || echo "check_generic FAILED"
cat $O/check_generic.txt | cut -c1-260
for v in "" var/libskr_tw0.so var/libskr_tw0ne.so; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$R/skele_raytracer_amd/lib/$v; fi
  timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 gillum=16 reps=20 2>/dev/null | grep "ms per frame" | sed "s#^#$v #"
done
unset SKR_LIBRARY
for pipe in generic nodes queue; do SKR_PIPELINE=$pipe timeout -k 10 100 python3 tools/profile_scene.py test.scn 640 360 gillum=4 shadow=1 reps=10 2>/dev/null | grep "ms per frame"; done
timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 shade_triangles=1 strict=1 reps=10 2>/dev/null | grep "ms per frame"
timeout -k 10 100 python3 tools/profile_scene.py spheres2.scn 1920 1080 legacy_reflect=1 shadow=1 depth=3 reps=10 2>/dev/null | grep "ms per frame"
