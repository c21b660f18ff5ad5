#!/usr/bin/env python3
"""Experiment: ONE frame rendered as two interleaved halves (tiles t mod 2) by two renderers on two streams at once, against the whole frame
by one renderer — what bands in flight inside a launch would buy a frame that takes many launches (AA samples under --gillum, bands).
usage: time_halves.py CONFIG(3|5) [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H, kw = (3840, 2160, dict(gillum=64, jsample=5, shadow=True)) if cfg == 5 else (1920, 1080, dict(gillum=16, jsample=3, shadow=True))
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
opt = skr.Options(W, H, seed=20261004, **kw)
dev = torch.device("cuda", 0)
ra, rb = skr.Renderer(sc), skr.Renderer(sc)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
TR = 8
whole = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
na, nb = ra.tile_count(opt, TR, 0, 2), ra.tile_count(opt, TR, 1, 2)
ha, hb = torch.zeros((na * TR, W, 3), dtype=torch.uint8, device=dev), torch.zeros((nb * TR, W, 3), dtype=torch.uint8, device=dev)


def one():
    ra.render_tiles_into(opt, H, 0, 1, whole.data_ptr(), None, sa.cuda_stream)


def two():
    ra.render_tiles_into(opt, TR, 0, 2, ha.data_ptr(), None, sa.cuda_stream)
    rb.render_tiles_into(opt, TR, 1, 2, hb.data_ptr(), None, sb.cuda_stream)


for name, fn in (("one renderer, whole frame", one), ("two renderers, interleaved halves on two streams", two), ("one renderer, whole frame", one)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print("config %d %-50s %.4f ms per frame [%s]" % (cfg, name, (time.perf_counter() - t0) / reps * 1e3, ra.kernel_variant()), flush=True)
