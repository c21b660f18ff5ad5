#!/usr/bin/env python3
"""Per-phase cycle sums of the GI kernel (diagnostic build -DSKR_STAMPS=1 via SKR_LIBRARY) for one rank's share."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
r = skr.Renderer(sc)
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
r.kernel_timing(True)
for G in (1, 8, 32):
    n = r.tile_count(opt, 8, 0, G)
    buf = torch.zeros((n * 8, 1920, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize(); r.kernel_ms(); r.counters()
    r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize()
    km, _ = r.kernel_ms()
    print("G=%d kernel %.3f ms parents %d" % (G, km, r.last_parent_count()), flush=True)
    sys.stderr.flush()
    os.environ["SKR_PRINT_STAMPS"] = "1"
    r.counters()
    os.environ.pop("SKR_PRINT_STAMPS")
