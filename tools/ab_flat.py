#!/usr/bin/env python3
"""Frame time of full-HD configurations under both schedules of the node pipeline (SKR_FLAT=0 / 1) and the automatic choice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
st = torch.cuda.current_stream()
for (w, h, n, d) in ((1920, 1080, 16, 2), (1920, 1080, 8, 2), (1920, 1080, 3, 2), (1920, 1080, 5, 3), (1920, 1080, 2, 3), (1920, 1080, 2, 4), (960, 540, 16, 3), (640, 360, 16, 3), (1920, 1080, 4, 3), (1920, 1080, 3, 4)):
    out = []
    for mode in ("0", "1", None):
        os.environ.pop("SKR_FLAT", None)
        if mode is not None:
            os.environ["SKR_FLAT"] = mode
        r = skr.Renderer(sc)
        opt = skr.Options(w, h, gillum=n, depth=d, shadow=True, seed=5)
        buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        for _ in range(4): r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20): r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
        out.append("%s %.3f ms [%s]" % ({"0": "persistent", "1": "flat", None: "auto"}[mode], e0.elapsed_time(e1) / 20, r.kernel_variant().replace("node_levels_v5", "nl5")))
    print("%dx%d gillum %d depth %d: %s" % (w, h, n, d, "; ".join(out)), flush=True)
