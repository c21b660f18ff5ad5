#!/usr/bin/env python3
"""One rank's share (tile stride G) of the headline frame a few times: `rocprofv3 --kernel-trace --stats -- python3 tools/profile_share.py G`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn")))
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
st = torch.cuda.current_stream()
n = r.tile_count(opt, 8, 0, G)
buf = torch.zeros((n * 8, 1920, 3), dtype=torch.uint8, device="cuda")
for _ in range(20):
    r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
torch.cuda.synchronize()
print("G", G, r.kernel_variant())
