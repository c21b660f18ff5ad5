#!/usr/bin/env python3
"""Per-wave timeline of the node pipeline's leaf kernel (diagnostic build -DSKR_TIMELINE=1 via SKR_LIBRARY): entry, first unit,
last unit done, exit on the device-wide 100 MHz clock, and the shader clock the waves ran at."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
from skele_raytracer_amd import binding
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn")))
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
L = binding.lib()
for G in (1, 8):
    n = r.tile_count(opt, 8, 0, G)
    buf = torch.zeros((n * 8, 1920, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    for _ in range(30):  # the clock the chip settles at under this load
        r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize()
    out = np.zeros(6 * 4096, np.uint64)
    L.skr_leaf2_times_read(C.c_void_p(out.ctypes.data))
    t = out.reshape(4096, 6).astype(np.float64)
    t0 = t[:, 0].min()
    entry, first, last, exit_, units, cyc = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0, t[:, 3] - t0, t[:, 4], t[:, 5]
    has = units > 0
    span = exit_.max()
    pct = lambda a, q: np.round(np.percentile(a, q) / 100.0, 1)  # ticks of 10 ns -> microseconds
    ghz = cyc / np.maximum(exit_ - entry, 1) / 10.0 / 1e0 * 1e-2  # cycles per 10-ns tick -> GHz
    print("G=%d [%s]: span %.1f us; entry pct 50/99/max %s us; first unit in hand pct 1/50/99 %s us; last unit done pct 1/10/50/90/max %s us; exit - last pct 50/99 %s us; units/wave min/mean/max %d/%.2f/%d; busy %.3f; shader clock GHz pct 1/50/99 %s" % (
        G, r.kernel_variant(), span / 100, pct(entry, [50, 99, 100]), pct(first[has], [1, 50, 99]), pct(last[has], [1, 10, 50, 90, 100]), pct((exit_ - np.where(has, last, entry)), [50, 99]),
        units.min(), units.mean(), units.max(), float((np.where(has, last - first, 0)).sum() / (4096 * span)), np.round(np.percentile(ghz, [1, 50, 99]), 3)), flush=True)
