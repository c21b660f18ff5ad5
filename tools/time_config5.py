#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: spheres2 3840x2160 --gillum 64 --jsample 5 --shadow (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
r = skr.Renderer(sc)
opt = skr.Options(3840, 2160, gillum=64, jsample=5, shadow=True, seed=5)
for it in range(2):
    r.counters()
    t0 = time.perf_counter()
    rgb, _ = r.render(opt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = r.counters()
    print("config5 run %d: %.3f s, %.2f G radiance rays, %.1f Grays/s, nominal %.3g [%s]" % (
        it, dt, c["radiance_rays"] / 1e9, c["radiance_rays"] / dt / 1e9, skr.radiance_ray_count(opt), r.kernel_variant()), flush=True)
