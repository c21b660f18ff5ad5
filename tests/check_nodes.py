#!/usr/bin/env python3
"""Dev aid: the node pipeline (SKR_PIPELINE=nodes) against the oracle on a spread of cases, then timed against the default path."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import skele_raytracer_amd as skr
from oracle import pyoracle as orc

def scene(n): return os.path.join(ROOT, "tests/golden/scenes", n)
_r = {}
def renderer(scn):
    if scn not in _r:
        sc = skr.parse_scene(scene(scn)); _r[scn] = (sc, skr.Renderer(sc))
    return _r[scn][1]

CASES = [
    ("spheres2.scn", 96, 54, dict(gillum=4, shadow=True, seed=3), {}),
    ("spheres2.scn", 240, 135, dict(gillum=16, shadow=True, seed=20261004), {}),
    ("spheres2.scn", 240, 135, dict(gillum=16, shadow=True, seed=20261004), {"SKR_UNIT_HALF": "1"}),
    ("spheres2.scn", 160, 90, dict(gillum=4, jsample=2, depth=2, shadow=True, seed=5), {}),
    ("spheres2.scn", 96, 54, dict(gillum=3, depth=4, shadow=True, seed=12), {}),
    ("spheres2.scn", 64, 36, dict(gillum=2, depth=6, shadow=True, seed=8), {}),
    ("spheres2.scn", 48, 27, dict(gillum=2, depth=8, shadow=True, seed=8), {}),
    ("spheres2.scn", 100, 57, dict(gillum=5, shadow=True, seed=77), {}),
    ("spheres2.scn", 64, 36, dict(gillum=1, shadow=True, seed=1), {}),
    ("spheres2.scn", 48, 27, dict(gillum=64, shadow=True, seed=6), {}),
    ("spheres2.scn", 24, 14, dict(gillum=256, depth=2, seed=2), {}),
    ("spheres2.scn", 200, 113, dict(gillum=6, jsample=2, shadow=True, seed=9), {"SKR_LEVELS_BUDGET_MB": "2"}),
    ("spheres2.scn", 131, 77, dict(gillum=5, depth=4, shadow=True, seed=4), {"SKR_LEVELS_BUDGET_MB": "24"}),
    ("test.scn", 96, 54, dict(gillum=4, shadow=True, seed=3), {}),
    ("test.scn", 64, 36, dict(gillum=3, depth=4, shadow=True, seed=3), {}),
    ("bear.scn", 160, 90, dict(gillum=255, seed=4), {}),
    ("spheres1.scn", 64, 36, dict(gillum=7, jsample=2, shadow=True, seed=21), {}),
    ("spheres2.scn", 1, 1, dict(gillum=4, shadow=True), {}),
]
bad = 0
CASES = [c for c in CASES if "SKR_UNIT_HALF" not in c[4]]
for scn, w, h, kw, env in CASES:
    for k in ("SKR_UNIT_HALF", "SKR_LEVELS_BUDGET_MB"): os.environ.pop(k, None)
    os.environ.update(env); os.environ["SKR_PIPELINE"] = "nodes"
    r = renderer(scn); r.counters(reset=True)
    rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True); torch.cuda.synchronize()
    v = r.kernel_variant(); cnt = r.counters()
    o_rgb, o_f, st = orc.render(scene(scn), w, h, rng=orc.RNG_COUNTER, math=orc.MATH_SHARED, want_float=True, **kw)
    nb = int((rgbf.cpu().numpy().view(np.uint32) != o_f.view(np.uint32)).sum())
    nu = int((rgb.cpu().numpy() != o_rgb).sum())
    ok = nb == 0 and nu == 0 and cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1]) and v.startswith("node_levels_v5")
    bad += not ok
    print("%s %-12s %dx%d %s %s [%s]: float words differing %d, bytes %d, rays %d/%d hits %d/%d" % ("ok  " if ok else "FAIL", scn, w, h, kw, env, v, nb, nu, cnt["radiance_rays"], int(st[0]), cnt["sphere_hits"], int(st[1])), flush=True)
for k in ("SKR_UNIT_HALF", "SKR_LEVELS_BUDGET_MB", "SKR_PIPELINE"): os.environ.pop(k, None)
print("failures:", bad, flush=True)

def timeit(scn, w, h, reps, env, **kw):
    for k in ("SKR_UNIT_HALF", "SKR_PIPELINE", "SKR_FLAT"): os.environ.pop(k, None)
    os.environ.update(env)
    r = renderer(scn); opt = skr.Options(w, h, **kw)
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda"); st = torch.cuda.current_stream()
    for _ in range(3): r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize(); r.kernel_timing(True); r.kernel_ms()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    km, _ = r.kernel_ms(); r.kernel_timing(False)
    print("%-12s %dx%d %-40s %-28s frame %.3f ms  dominant kernel %.3f ms [%s]" % (scn, w, h, kw, env, e0.elapsed_time(e1) / reps, km, r.kernel_variant()), flush=True)

if "--time" in sys.argv:
    H = dict(gillum=16, shadow=True, seed=20261004)
    FLAT, PERS, GEN = {"SKR_FLAT": "1"}, {"SKR_FLAT": "0"}, {"SKR_PIPELINE": "generic"}
    for env in ({}, FLAT, GEN):
        timeit("spheres2.scn", 1920, 1080, 20, env, **H)
    for env in ({}, PERS, GEN):
        timeit("spheres2.scn", 1920, 1080, 10, env, gillum=16, shadow=True, depth=2, seed=20261004)
        timeit("bear.scn", 1920, 1080, 10, env, gillum=16, shadow=True, seed=3)
        timeit("spheres2.scn", 480, 270, 2, env, gillum=4, depth=4, shadow=True, seed=5)
    timeit("spheres2.scn", 960, 540, 3, {}, gillum=64, shadow=True, seed=5)
    timeit("test.scn", 640, 360, 2, {}, gillum=4, shadow=True)
    timeit("test.scn", 640, 360, 2, {"SKR_PIPELINE": "nodes"}, gillum=4, shadow=True)
    for k in ("SKR_FLAT", "SKR_PIPELINE"): os.environ.pop(k, None)
sys.exit(1 if bad else 0)
