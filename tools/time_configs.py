#!/usr/bin/env python3
"""Time a few configurations on the GPU (kernel ms via events) — development aid."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr

def run(scn, w, h, reps=5, **kw):
    sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes", scn))
    r = skr.Renderer(sc)
    opt = skr.Options(w, h, **kw)
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize(); r.counters()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    c = r.counters()
    rays = c["radiance_rays"] / reps
    print("%-12s %dx%d %-44s %8.3f ms  %7.1f Mrays  %8.0f Mrays/s  hits %.1fM shadow %.1fM  [%s]" % (
        scn, w, h, str(kw), ms, rays / 1e6, rays / ms / 1e3, c["sphere_hits"] / reps / 1e6, c["shadow_rays"] / reps / 1e6, r.kernel_variant()), flush=True)

if __name__ == "__main__":
    run("spheres2.scn", 1920, 1080, gillum=16, shadow=True, seed=20261004)
    run("spheres2.scn", 1920, 1080, gillum=16, shadow=False, seed=20261004)
    run("spheres2.scn", 1920, 1080, gillum=16, shadow=True, depth=2, seed=20261004)
    run("spheres2.scn", 1920, 1080, gillum=16, shadow=True, depth=1, seed=20261004)
    run("spheres2.scn", 1920, 1080, jsample=5, shadow=True, seed=9)
    run("spheres2.scn", 1920, 1080, shadow=True)
    run("spheres1.scn", 1920, 1080, gillum=16, shadow=True, seed=3)
    run("bear.scn", 1920, 1080, gillum=16, shadow=True, seed=3)
    run("dragon.scn", 1920, 1080, reps=2, gillum=16)
    run("test.scn", 640, 360, reps=2, gillum=4, shadow=True)
    run("spheres2.scn", 960, 540, reps=2, gillum=64, shadow=True, seed=5)
