#!/usr/bin/env python3
"""How long the HOST takes to enqueue one frame of the pipelined frame step (skr_comm_render_frame_async, a world of one through a real RCCL
communicator) for a 1/8-share-sized frame: the device needs ~0.21 ms per such frame, so the enqueue has to stay well below that."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr
from skele_raytracer_amd import binding
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
r = skr.Renderer(sc)
st = torch.cuda.current_stream()
for (w, h, label) in ((1920, 136, "a 1/8 share's pixels (1920x136)"), (1920, 1080, "the whole frame")):
    opt = skr.Options(w, h, gillum=16, shadow=True, seed=20261004)
    for with_rccl in (True, False):
        c = binding.Comm(r, 0, 1, binding.comm_unique_id() if with_rccl else None)
        for _ in range(20): c.render_frame_async(opt, 8, st.cuda_stream, want_previous=False)
        c.flush(st.cuda_stream); torch.cuda.synchronize()
        n = 400
        t0 = time.perf_counter()
        for _ in range(n): c.render_frame_async(opt, 8, st.cuda_stream, want_previous=False)
        t1 = time.perf_counter()
        c.flush(st.cuda_stream); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s, %s: host enqueue %.1f us per frame, device %.1f us per frame (the enqueue loop ran %s the device)" % (
            label, "RCCL communicator" if with_rccl else "no collective", (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, "ahead of" if (t1 - t0) < 0.8 * (t2 - t0) else "in step with"), flush=True)
        c.close()
