#!/usr/bin/env python3
"""Dev aid: render one configuration a few times (for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_scene.py ...`).
usage: profile_scene.py SCENE W H [gillum=N] [depth=D] [shadow=1] [jsample=G] [reps=10]   (+ SKR_* switches from the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr
scn, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kw = dict(a.split("=") for a in sys.argv[4:])
reps = int(kw.pop("reps", 10))
strict = bool(int(kw.pop("strict", 0)))
opts = {k: (bool(int(v)) if k in ("shadow", "shade_triangles", "legacy_reflect") else int(v)) for k, v in kw.items()}
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes", scn), strict=strict))
opt = skr.Options(w, h, seed=3, **opts)
st = torch.cuda.current_stream()
buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
for _ in range(3): r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(reps): r.render_tiles_into(opt, h, 0, 1, buf.data_ptr(), None, st.cuda_stream)
e1.record(st); torch.cuda.synchronize()
print("%s %dx%d %s [%s]: %.3f ms per frame" % (scn, w, h, opts, r.kernel_variant(), e0.elapsed_time(e1) / reps), flush=True)
