#!/usr/bin/env python3
"""Dev aid: the general level pipeline (SKR_PIPELINE=generic, render_generic.hip) against the oracle on a spread of cases and modes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import skele_raytracer_amd as skr
from oracle import pyoracle as orc

def scene(n): return os.path.join(ROOT, "tests/golden/scenes", n)
_r = {}
def renderer(scn, strict):
    if (scn, strict) not in _r:
        sc = skr.parse_scene(scene(scn), strict=strict); _r[(scn, strict)] = (sc, skr.Renderer(sc))
    return _r[(scn, strict)][1]

CASES = [
    ("spheres2.scn", 96, 54, dict(shadow=True), {}),
    ("spheres2.scn", 96, 54, dict(gillum=4, shadow=True, seed=3), {}),
    ("spheres2.scn", 160, 90, dict(gillum=4, jsample=2, depth=2, shadow=True, seed=5), {}),
    ("spheres2.scn", 64, 36, dict(gillum=3, depth=4, shadow=True, seed=12), {}),
    ("spheres2.scn", 48, 27, dict(gillum=2, depth=8, shadow=True, seed=8), {}),
    ("spheres2.scn", 24, 14, dict(gillum=300, depth=2, seed=2), {}),
    ("spheres2.scn", 40, 23, dict(gillum=5, depth=3, shadow=True, seed=4), {"SKR_LEVELS_BUDGET_MB": "8"}),
    ("test.scn", 96, 54, dict(gillum=4, shadow=True, seed=3), {}),
    ("test.scn", 64, 36, dict(gillum=3, depth=4, shadow=True, seed=3), {}),
    ("dragon.scn", 96, 54, dict(gillum=4), {}),
    ("test.scn", 96, 54, dict(gillum=4, shadow=True, seed=3, shade_triangles=True), {}),
    ("test.scn", 64, 36, dict(gillum=2, depth=7, shadow=True, seed=3, shade_triangles=True), {}),
    ("dragon.scn", 96, 54, dict(shade_triangles=True, strict=True), {}),
    ("spheres2.scn", 96, 54, dict(depth=3, shadow=True, legacy_reflect=True), {}),
    ("spheres2.scn", 64, 36, dict(depth=8, shadow=True, legacy_reflect=True), {}),
    ("spheres2.scn", 64, 36, dict(depth=3, shadow=True, legacy_reflect=True, strict=True), {}),
    ("spheres2.scn", 48, 27, dict(gillum=3, depth=3, shadow=True, legacy_reflect=True, seed=5), {}),
    ("test.scn", 64, 36, dict(gillum=2, depth=3, shadow=True, legacy_reflect=True, shade_triangles=True, seed=5), {}),
    ("spheres1.scn", 64, 36, dict(gillum=7, jsample=2, shadow=True, seed=21), {}),
]
bad = 0
for scn, w, h, kw, env in CASES:
    os.environ.pop("SKR_LEVELS_BUDGET_MB", None)
    os.environ.update(env); os.environ["SKR_PIPELINE"] = "generic"
    kw = dict(kw); strict = kw.pop("strict", False)
    r = renderer(scn, strict); r.reload_switches() if hasattr(r, "reload_switches") else None
    r.counters(reset=True)
    try:
        rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True); torch.cuda.synchronize()
    except skr.SkrError as e:
        print("FAIL %-12s %dx%d %s: %s" % (scn, w, h, kw, e), flush=True); bad += 1; continue
    v = r.kernel_variant(); cnt = r.counters()
    o_rgb, o_f, st = orc.render(scene(scn), w, h, rng=orc.RNG_COUNTER, math=orc.MATH_SHARED, want_float=True, strict=strict, **kw)
    g = rgbf.cpu().numpy().view(np.uint32); o = o_f.view(np.uint32)
    nan_both = np.isnan(rgbf.cpu().numpy()) & np.isnan(o_f)
    nb = int(((g != o) & ~nan_both).sum())
    nu = int((rgb.cpu().numpy() != o_rgb).sum())
    ok = nb == 0 and nu == 0 and cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1]) and v == "level_pipeline_g1"
    bad += not ok
    print("%s %-12s %dx%d %s %s [%s]: float words differing %d, bytes %d, rays %d/%d hits %d/%d" % ("ok  " if ok else "FAIL", scn, w, h, kw, env, v, nb, nu, cnt["radiance_rays"], int(st[0]), cnt["sphere_hits"], int(st[1])), flush=True)
print("failures:", bad, flush=True)
sys.exit(1 if bad else 0)
