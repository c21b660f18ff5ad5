#!/usr/bin/env python3
"""Experiment: a rank's 1/8 share of the headline frame rendered as TWO half-lists on two streams (two renderers, each with its own
scratch), against the one list on one stream — do the eight short dependent kernels of a share hide each other's ramps and tails?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
from skele_raytracer_amd import binding
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
NL = int(os.environ.get("LANES", "2"))
rs = [skr.Renderer(sc) for _ in range(NL)]
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
TR, T = 8, 135
streams = [torch.cuda.Stream() for _ in range(NL)]
main = torch.cuda.current_stream()


def dev(tiles):
    return torch.from_numpy(tiles.astype(np.int64)).cuda().to(torch.int32).contiguous()


for G in (8, 4):
    k_max = binding.shard_tiles_per_rank(1080, TR, G)
    for rank in (0, 3):
        tiles = np.full(k_max, 0xFFFFFFFF, np.uint32)
        mine = [t for t in range(T) if t % G == rank]
        tiles[:len(mine)] = mine
        buf = torch.zeros((k_max * TR, 1920, 3), dtype=torch.uint8, device="cuda")
        whole = dev(tiles)
        # contiguous parts of the slot list: part j = slots [j*h, (j+1)*h)
        h = (k_max + NL - 1) // NL
        parts = [dev(tiles[j * h:(j + 1) * h]) for j in range(NL)]
        sizes = [len(tiles[j * h:(j + 1) * h]) for j in range(NL)]

        def one():
            rs[0].render_tile_list_into(opt, TR, whole.data_ptr(), k_max, buf.data_ptr(), None, main.cuda_stream)

        def split():
            ev = torch.cuda.Event(); ev.record(main)
            for j in range(NL):
                streams[j].wait_event(ev)
                if sizes[j]:
                    rs[j].render_tile_list_into(opt, TR, parts[j].data_ptr(), sizes[j], buf[j * h * TR:].data_ptr(), None, streams[j].cuda_stream)
                e2 = torch.cuda.Event(); e2.record(streams[j]); main.wait_event(e2)

        for name, fn in (("one stream", one), ("%d streams" % NL, split)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(20): fn()
            e1.record(main); torch.cuda.synchronize()
            print("G=%d rank %d %-10s %.3f ms" % (G, rank, name, e0.elapsed_time(e1) / 20), flush=True)
        ref = buf.clone(); buf.zero_(); split(); torch.cuda.synchronize()
        print("   same bytes:", bool((ref == buf).all()), flush=True)
