#!/usr/bin/env python3
"""Fixed cost of a launch: rank 0's share of the headline frame for large G — development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
r = skr.Renderer(sc)
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
r.kernel_timing(True)
for G in (8, 16, 32, 64, 135):
    rank = 0
    n = r.tile_count(opt, 8, rank, G)
    buf = torch.zeros((n * 8, 1920, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    r.render_tiles_into(opt, 8, rank, G, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize(); r.kernel_ms(); r.counters()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        r.render_tiles_into(opt, 8, rank, G, buf.data_ptr(), None, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    km, nl = r.kernel_ms()
    c = r.counters()
    print("G=%3d tiles %3d  frame %.3f ms  gi kernel %.3f ms  parents %d rays %.2fM -> %.1f Grays/s" % (G, n, e0.elapsed_time(e1) / 5, km, r.last_parent_count(), c["radiance_rays"] / 5e6, c["radiance_rays"] / 5 / (e0.elapsed_time(e1) / 5) / 1e6), flush=True)
