#!/usr/bin/env python3
"""Per-wave timeline of the GI kernel (diagnostic build -DSKR_STAMPS=1 via SKR_LIBRARY): when waves start and finish."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
from skele_raytracer_amd import binding
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
r = skr.Renderer(sc)
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
L = binding.lib()
for G in (1, 8):
    n = r.tile_count(opt, 8, 0, G)
    buf = torch.zeros((n * 8, 1920, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    for _ in range(2):
        r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize()
    out = np.zeros(3 * 4096, np.uint64)
    L.skr_wave_times_read(C.c_void_p(out.ctypes.data))
    t = out.reshape(4096, 3)[:3072].astype(np.float64)
    t0 = t[:, 0].min()
    start, end, groups = (t[:, 0] - t0), (t[:, 1] - t0), t[:, 2]
    total = end.max()
    print("G=%d: kernel span %.0f ticks; wave start pct 50/90/99/max: %s ; wave end pct 1/10/50/90/max: %s ; busy fraction %.3f ; groups/wave min/mean/max %d/%.1f/%d" % (
        G, total, np.round(np.percentile(start, [50, 90, 99, 100]) / total, 3), np.round(np.percentile(end, [1, 10, 50, 90, 100]) / total, 3),
        float((end - start).sum() / (3072 * total)), groups.min(), groups.mean(), groups.max()), flush=True)
