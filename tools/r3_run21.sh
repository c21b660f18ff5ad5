set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v; mkdir -p $O
cd $R
V=$R/skele_raytracer_amd/lib/var
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export SKR_LIBRARY=$V/libskr_$v.so; fi
  timeout -k 10 100 python3 tools/profile_scene.py spheres2.scn 1920 1080 jsample=5 shadow=1 reps=10 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] config 2: #"
  timeout -k 10 100 python3 tools/profile_scene.py spheres2.scn 1920 1080 gillum=16 shadow=1 reps=20 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] config 3: #"
  timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 gillum=16 reps=20 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] dragon: #"
  timeout -k 10 100 python3 tools/profile_scene.py test.scn 1920 1080 reps=10 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] test.scn 1080p: #"
  timeout -k 10 100 python3 tools/profile_scene.py spheres2.scn 1920 1080 legacy_reflect=1 shadow=1 depth=3 reps=10 2>/dev/null | grep "ms per frame" | sed "s#^#[$v] legacy: #"
done
unset SKR_LIBRARY
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -3 $O/pytest_gpu.txt | cut -c1-300
