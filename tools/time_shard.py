#!/usr/bin/env python3
"""Time every rank's share of the headline frame on ONE GPU, one rank after the other — what a frame step of G ranks would wait for
before its collective — under the cost-aware tile map (include/skr.h skr_shard_plan) and under the blind one (tile t to rank t mod G)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
from skele_raytracer_amd import binding
sc = skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn"))
r = skr.Renderer(sc)
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
TR = int(os.environ.get("TILE_ROWS", "8"))
st = torch.cuda.current_stream()
T = (1080 + TR - 1) // TR


def time_rank(tiles, k_max):
    d = torch.from_numpy(tiles.astype(np.int64)).to(torch.int64).cuda().to(torch.int32).contiguous()  # (same bits as uint32)
    buf = torch.zeros((k_max * TR, 1920, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3): r.render_tile_list_into(opt, TR, d.data_ptr(), k_max, buf.data_ptr(), None, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): r.render_tile_list_into(opt, TR, d.data_ptr(), k_max, buf.data_ptr(), None, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


whole = time_rank(np.arange(T, dtype=np.uint32), T)
print("G=1: %.3f ms" % whole, flush=True)
GS = [int(g) for g in os.environ.get("G_LIST", "2,4,8").split(",")]
MODES = os.environ.get("MODES", "lpt,interleave").split(",")
for G in GS:
    k_max = binding.shard_tiles_per_rank(1080, TR, G)
    for name, env in (("cost-aware (LPT)", "lpt"), ("t mod G", "interleave")):
        if env not in MODES: continue
        if env: os.environ["SKR_SHARD"] = env
        else: os.environ.pop("SKR_SHARD", None)
        slot = r.shard_plan(opt, TR, G)
        times = []
        for rank in range(G):
            tiles = np.full(k_max, 0xFFFFFFFF, np.uint32)
            for t in range(T):
                if slot[t] // k_max == rank: tiles[slot[t] % k_max] = t
            times.append(time_rank(tiles, k_max))
            if os.environ.get("PRINT_WORK"):
                r.work(reset=True)
                time_rank.__globals__["r"].render_tile_list_into(opt, TR, torch.from_numpy(tiles.astype(np.int64)).cuda().to(torch.int32).contiguous().data_ptr(), k_max,
                                                                 torch.zeros((k_max * TR, 1920, 3), dtype=torch.uint8, device="cuda").data_ptr(), None, st.cuda_stream)
                torch.cuda.synchronize()
                wk = r.work(reset=True)
                print("WORK G=%d %s rank %d tiles %d time %.4f %s" % (G, env, rank, int((tiles != 0xFFFFFFFF).sum()), times[-1], " ".join("%s=%d" % kv for kv in sorted(wk.items()))), flush=True)
        print("G=%d %-18s slowest rank %.3f ms, mean %.3f ms, ranks %s  -> %.2fx of G=1 before the collective" % (G, name, max(times), sum(times) / G, " ".join("%.3f" % t for t in times), whole / max(times)), flush=True)
os.environ.pop("SKR_SHARD", None)
