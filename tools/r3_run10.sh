set -e -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || echo "pytest failed"
tail -12 $O/pytest_gpu.txt | cut -c1-300
timeout -k 10 300 python3 tests/check_generic.py > $O/check_generic.txt 2>&1 || echo "check_generic FAILED"
tail -3 $O/check_generic.txt
timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 gillum=16 reps=20 2>/dev/null | grep "ms per frame"
timeout -k 10 100 python3 tools/profile_scene.py spheres2.scn 1920 1080 jsample=5 shadow=1 reps=20 2>/dev/null | grep "ms per frame"
timeout -k 10 100 python3 tools/profile_scene.py dragon.scn 1920 1080 shade_triangles=1 strict=1 reps=10 2>/dev/null | grep "ms per frame"
timeout -k 10 100 python3 tools/profile_scene.py test.scn 640 360 gillum=4 shadow=1 reps=10 2>/dev/null | grep "ms per frame"
timeout -k 10 100 python3 tools/profile_scene.py spheres1.scn 1920 1080 gillum=16 shadow=1 reps=10 2>/dev/null | grep "ms per frame"
timeout -k 10 200 python3 tools/ab_nodes.py 2>/dev/null | grep G=
