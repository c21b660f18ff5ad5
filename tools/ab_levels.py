#!/usr/bin/env python3
"""A/B: level-queue pipeline (SKR_PIPELINE=levels) against the parent-queue pipeline (SKR_PIPELINE=queue) — same bytes? how fast? (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import skele_raytracer_amd as skr

def run(scn, w, h, G=1, reps=5, **kw):
    sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes", scn))
    r = skr.Renderer(sc)
    opt = skr.Options(w, h, **kw)
    n = r.tile_count(opt, 8, 0, G)
    outs = {}
    for mode in ("default", "levels"):
        os.environ["SKR_PIPELINE"] = "levels" if mode == "levels" else "queue"
        buf = torch.zeros((n * 8, w, 3), dtype=torch.uint8, device="cuda")
        fb = torch.zeros((n * 8, w, 3), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream()
        r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), fb.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(); r.counters()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
        c = r.counters()
        outs[mode] = (buf.cpu().numpy(), fb.cpu().numpy(), e0.elapsed_time(e1) / reps, r.kernel_variant(), {k: v // (reps) for k, v in c.items()})
    os.environ.pop("SKR_PIPELINE", None)
    a, b = outs["default"], outs["levels"]
    same = np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    print("%-12s %dx%d G=%d %-40s same=%s  %s %.3f ms | %s %.3f ms  counters equal=%s" % (scn, w, h, G, kw, same, a[3], a[2], b[3], b[2], a[4] == b[4]), flush=True)

if __name__ == "__main__":
    run("spheres2.scn", 1920, 1080, gillum=16, shadow=True, seed=20261004)
    run("spheres2.scn", 1920, 1080, gillum=16, seed=20261004)
    run("spheres2.scn", 1920, 1080, G=8, gillum=16, shadow=True, seed=20261004)
    run("spheres1.scn", 1920, 1080, gillum=16, shadow=True, seed=3)
    run("bear.scn", 1920, 1080, gillum=16, shadow=True, seed=3)
    run("test.scn", 640, 360, gillum=4, shadow=True)
    run("test.scn", 640, 360, gillum=16, shadow=True)
    run("spheres2.scn", 1920, 1080, gillum=4, shadow=True, seed=5)
    run("spheres2.scn", 1920, 1080, gillum=8, shadow=True, seed=5)
    run("spheres2.scn", 960, 540, reps=2, gillum=64, shadow=True, seed=5)
    run("spheres2.scn", 640, 360, reps=2, gillum=16, jsample=3, shadow=True, seed=5)
    run("spheres2.scn", 320, 180, gillum=255, shadow=True, seed=5)
