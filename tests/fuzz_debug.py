#!/usr/bin/env python3
"""Replay ONE case of a tests/fuzz_parity.py campaign and localise a mismatch: per progressive pass and per image row, the device's
ray / hit / shadow-ray counts and float image against the oracle's.     python tests/fuzz_debug.py CASES SEED INDEX"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import skele_raytracer_amd as skr
from oracle import pyoracle as orc
from fuzz_parity import generate

cases, seed, index = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for c, scn, w, h, kw, strict, passes, env in generate(np.random.default_rng(seed), cases, tempfile.mkdtemp()):
    if c == index: break
print("case", c, scn, w, h, kw, "strict", strict, "passes", passes, env, flush=True)
os.environ.update(env)
r = skr.Renderer(skr.parse_scene(scn, strict=strict))
for ps in range(passes):
    k = dict(kw, seed=kw["seed"] + ps)
    for y in range(h):
        r.counters(reset=True)
        rgb, rgbf = r.render(skr.Options(w, h, **k), want_float=True, tile_rows=1, first_tile=y, tile_stride=h)
        torch.cuda.synchronize()
        cnt = r.counters(reset=True)
        o_rgb, o_f, st = orc.render(scn, w, h, want_float=True, strict=strict, y0=y, y1=y + 1, **k)
        dev = (cnt["radiance_rays"], cnt["sphere_hits"], cnt["shadow_rays"])
        ora = tuple(int(v) for v in st[:3])
        same = (rgbf.cpu().numpy().view(np.uint32) == o_f.view(np.uint32)).all()
        if dev != ora or not same:
            print("pass %d row %d: device %s oracle %s image same %s" % (ps, y, dev, ora, bool(same)), flush=True)
print("done", flush=True)
