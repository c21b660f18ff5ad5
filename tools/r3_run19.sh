set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
V=$R/skele_raytracer_amd/lib/var
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$V/libskr_$v.so; fi
  timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 gillum=16 reps=20 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] dragon: #"
  timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 shade_triangles=1 strict=1 reps=10 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] dragon surfaces: #"
  timeout -k 10 100 python3 tools/profile_scene.py test.scn 640 360 gillum=4 shadow=1 reps=10 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] test.scn gillum 4: #"
  timeout -k 10 100 python3 tools/profile_scene.py bear.scn 1920 1080 gillum=4 shadow=1 reps=5 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] bear gillum 4: #"
  timeout -k 10 300 python3 tests/check_generic.py > $O/check_generic_$v.txt 2>&1 || echo "check_generic $v FAILED"
  tail -1 $O/check_generic_$v.txt
  timeout -k 10 400 python3 -m pytest tests -x -q -m gpu -k "triangle or mesh or dragon or bear or golden" > $O/pytest_$v.txt 2>&1 || echo "pytest $v failed"
  tail -2 $O/pytest_$v.txt | cut -c1-200
done
