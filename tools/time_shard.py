#!/usr/bin/env python3
"""Time one rank's share of the headline frame (interleaved 8-row tiles, stride G) — development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
r = skr.Renderer(sc)
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
TR = int(os.environ.get("TILE_ROWS", "8"))
for G in (1, 2, 4, 8):
    worst = 0
    for rank in range(G):
        n = r.tile_count(opt, TR, rank, G)
        buf = torch.zeros((n * TR, 1920, 3), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream()
        r.render_tiles_into(opt, TR, rank, G, buf.data_ptr(), None, st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            r.render_tiles_into(opt, TR, rank, G, buf.data_ptr(), None, st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
        worst = max(worst, e0.elapsed_time(e1) / 5)
    print("tile_rows=%d G=%d  slowest rank %.3f ms  tile=%s" % (TR, G, worst, os.environ.get("SKR_TILE", "auto")), flush=True)
