#!/usr/bin/env python3
"""Frames in flight: K headline frames through one renderer / one stream against two renderers on two streams (each with
its own tables), alternating.  Frames are independent, so the second stream's primary + trace kernels can fill the tail
of the first stream's leaf kernel.  Prints ms per frame for both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # one rank's share of the frame cut over G GPUs (tile t -> rank t mod G) instead of the whole frame
W, H = 1920, 1080
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
opt = skr.Options(W, H, gillum=16, shadow=True, seed=20261004)
dev = torch.device("cuda", 0)
for n in (1, 2, 3):
    rs = [skr.Renderer(sc) for _ in range(n)]
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    rows = rs[0].tile_count(opt, 8, 0, G) * 8 if G > 1 else H
    bufs = [torch.zeros((rows, W, 3), dtype=torch.uint8, device=dev) for _ in range(n)]
    def frame(i):
        k = i % n
        if G > 1: rs[k].render_tiles_into(opt, 8, 0, G, bufs[k].data_ptr(), None, streams[k].cuda_stream)
        else: rs[k].render_tiles_into(opt, H, 0, 1, bufs[k].data_ptr(), None, streams[k].cuda_stream)
    for i in range(3 * n):
        frame(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(K):
            frame(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    same = all(torch.equal(bufs[0], b) for b in bufs[1:])
    print("G=%d frames in flight %d: %.4f ms per frame (frames equal: %s) [%s]" % (G, n, best, same, rs[0].kernel_variant()), flush=True)
