#!/usr/bin/env python3
"""Render one BASELINE configuration a few times (for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_config.py N`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
CONFIGS = {
    "2": ("spheres2.scn", 1920, 1080, dict(jsample=5, shadow=True, seed=9)),
    "3": ("spheres2.scn", 1920, 1080, dict(gillum=16, shadow=True, seed=20261004)),
    "4": ("dragon.scn", 1920, 1080, dict(gillum=16)),
    "d4": ("spheres2.scn", 1920, 1080, dict(gillum=4, depth=4, shadow=True, seed=3)),   # every kernel of the node pipeline, incl. activate / prefix
    "d3n4": ("spheres2.scn", 1920, 1080, dict(gillum=4, depth=3, shadow=True, seed=3)),
}
scn, w, h, kw = CONFIGS[sys.argv[1]]
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes", scn)))
opt = skr.Options(w, h, **kw)
buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
for _ in range(12):
    r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
torch.cuda.synchronize()
print(scn, w, h, kw, r.kernel_variant(), r.counters())
