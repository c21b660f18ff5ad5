#!/usr/bin/env python3
"""Dev aid: frame and leaf-kernel time of the headline configuration (and one rank's 1/8 share) for the library SKR_LIBRARY names."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn")))
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
st = torch.cuda.current_stream()
for G in (1, 2, 4, 8):
    n = r.tile_count(opt, 8, 0, G)
    buf = torch.zeros((n * 8, 1920, 3), dtype=torch.uint8, device="cuda")
    for _ in range(5): r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize(); r.kernel_timing(True); r.kernel_ms()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(30): r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    km, _ = r.kernel_ms(); r.kernel_timing(False)
    print("%s G=%d frame %.3f ms leaf %.3f ms [%s]" % (os.path.basename(os.environ.get("SKR_LIBRARY", "libskr.so")), G, e0.elapsed_time(e1) / 30, km, r.kernel_variant()), flush=True)
