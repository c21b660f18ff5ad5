#!/usr/bin/env python3
"""Development aid: random FULL-SIZE frames (the persistent schedule of the node pipeline, the general level pipeline on meshes and in
the two opt-in modes, the direct kernel; several bands where the tables ask for them) on the device, a few rows of each against the oracle, bit for bit.    python tests/check_large.py [cases=12] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import skele_raytracer_amd as skr  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    mode = str(rng.choice(["gi", "gi", "gi", "gi", "legacy", "surfaces", "plain"]))
    scenes = {"gi": ["spheres2.scn", "spheres2.scn", "bear.scn", "spheres1.scn", "test.scn"], "legacy": ["spheres2.scn", "bear.scn", "spheres1.scn", "test.scn"],
              "surfaces": ["dragon.scn", "test.scn", "spheres1.scn"], "plain": ["spheres2.scn", "dragon.scn", "test.scn", "bear.scn"]}[mode]
    scn = os.path.join(ROOT, "tests/golden/scenes", str(rng.choice(scenes)))
    w, h = (1920, 1080) if rng.random() < .6 else (int(rng.integers(900, 2600)), int(rng.integers(500, 1500)))
    n = int(rng.choice([2, 3, 5, 8, 16]))
    d = int(rng.choice([2, 3, 3, 4])) if n <= 5 else int(rng.choice([2, 3]))
    kw = dict(gillum=n, depth=d, shadow=bool(rng.random() < .7), seed=int(rng.integers(1, 2 ** 32)))
    if mode == "legacy":  # (arity N + 2 L: keep the tree small)
        kw = dict(legacy_reflect=True, depth=int(rng.choice([2, 3, 4])), shadow=kw["shadow"], seed=kw["seed"])
        if rng.random() < .4:
            kw.update(gillum=int(rng.choice([1, 2])), depth=min(kw["depth"], 3))
    if mode == "surfaces":
        kw = dict(shade_triangles=True, shadow=kw["shadow"], seed=kw["seed"])
        if rng.random() < .6:
            kw.update(gillum=int(rng.choice([1, 2, 4])), depth=int(rng.choice([2, 3])))
    if mode == "plain":
        kw = dict(shadow=kw["shadow"], seed=kw["seed"], jsample=int(rng.choice([0, 2, 3])))
    strict = bool(rng.random() < .3)
    r = skr.Renderer(skr.parse_scene(scn, strict=strict))
    rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True)
    torch.cuda.synchronize()
    g = rgbf.cpu().numpy()
    rows = sorted(set(int(y) for y in rng.integers(0, h, 3)))
    ok = True
    for y in rows:
        _, o_f, _ = orc.render(scn, w, h, want_float=True, strict=strict, y0=y, y1=y + 1, **kw)
        ok = ok and bool((g[y:y + 1].view(np.uint32) == o_f.view(np.uint32)).all())
    bad += 0 if ok else 1
    print("case %d %s %dx%d %s strict=%s rows %s variant %s: %s" % (c, os.path.basename(scn), w, h, kw, strict, rows, r.kernel_variant(), "ok" if ok else "MISMATCH"), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
