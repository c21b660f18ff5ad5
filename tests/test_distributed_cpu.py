"""world_size-2/3 `gloo` tests of the multi-GPU path on CPU: partition arithmetic, the one
collective and the de-interleave.  The per-rank renderer is stood in for by the oracle (tests may
use it); on the GPU box bench.py plugs the HIP renderer into the same FrameSharder."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import scene_path
from skele_raytracer_amd import distributed as D

W, H, TILE = 96, 53, 8
KW = dict(gillum=3, shadow=True, seed=4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as orc
    scene = orc.OracleScene(scene_path("spheres2.scn"))
    sh = D.FrameSharder(W, H, TILE, rank, world, torch.device("cpu"))

    def render_into(buf):
        buf.zero_()
        for k, t in enumerate(D.my_tiles(H, TILE, rank, world)):
            y0, y1 = t * TILE, min(H, (t + 1) * TILE)
            rgb, _, _ = orc.render(scene, W, H, y0=y0, y1=y1, threads=1, **KW)
            buf[k * TILE:k * TILE + (y1 - y0)] = torch.from_numpy(rgb)

    frame = sh.step(render_into)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_process_frame(tmp_path, oracle, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    want, _, _ = oracle.render(scene_path("spheres2.scn"), W, H, **KW)
    assert got.shape == (H, W, 3) and np.array_equal(got, want)


def test_partition_arithmetic():
    assert D.tiles_total(1080, 8) == 135 and D.tiles_per_rank(1080, 8, 8) == 17
    assert D.my_tiles(20, 8, 1, 2) == [1] and D.my_tiles(20, 8, 0, 2) == [0, 2]
    for h, tile, world in [(1080, 8, 8), (53, 8, 3), (7, 16, 4), (2160, 16, 8)]:
        seen = sorted(t for r in range(world) for t in D.my_tiles(h, tile, r, world))
        assert seen == list(range(D.tiles_total(h, tile)))
        k_max = D.tiles_per_rank(h, tile, world)
        g = torch.zeros((world, k_max * tile, 4, 3), dtype=torch.uint8)
        for r in range(world):
            for k, t in enumerate(D.my_tiles(h, tile, r, world)):
                g[r, k * tile:(k + 1) * tile] = t % 251
        f = D.deinterleave(g, h, tile, world)
        assert f.shape[0] == h
        assert all(int(f[y, 0, 0]) == (y // tile) % 251 for y in range(h))


def test_native_partition_matches_the_python_one():
    """libskr's own partition helpers (include/skr.h skr_shard_*: what skr_comm_render_frame / skr_multi_render_frame
    and their de-interleave kernel implement) against distributed.py on the same buffers — host logic, no GPU."""
    from skele_raytracer_amd import binding
    rng = np.random.default_rng(5)
    for h, tile, world, w in [(1080, 8, 8, 16), (53, 8, 3, 7), (7, 16, 4, 5), (2160, 16, 8, 4), (100, 5, 1, 3), (33, 8, 6, 9)]:
        k_max = binding.shard_tiles_per_rank(h, tile, world)
        assert k_max == D.tiles_per_rank(h, tile, world)
        g = rng.integers(0, 256, (world, k_max * tile, w, 3), dtype=np.uint8)
        want = D.deinterleave(torch.from_numpy(g), h, tile, world).numpy()
        got = binding.shard_deinterleave_host(g, w, h, tile, world)
        assert np.array_equal(got, want)


def test_cost_aware_tile_map_properties():
    """libskr's cost-aware map (multi_gpu.cpp shard_lpt through skr_shard_lpt; host logic, no GPU): a permutation of the slots with
    every rank within its k_max, deterministic, never worse balanced than dealing `t mod G`, within one tile of the mean where that
    is possible — and the map-driven de-interleave inverts it."""
    from skele_raytracer_amd import binding
    rng = np.random.default_rng(5)
    for T, G in ((135, 8), (135, 4), (135, 2), (17, 8), (7, 8), (64, 3), (1, 1)):
        # a frame's profile: cheap sky tiles on top, tiles some 30x dearer below, noise on both
        cost = np.where(np.arange(T) < T // 3, 15360, 400000).astype(np.uint64) + rng.integers(0, 20000, T).astype(np.uint64)
        k_max = binding.shard_tiles_per_rank(T * 8, 8, G)
        slot = binding.shard_lpt(cost, G)
        assert len(set(slot.tolist())) == T and slot.max() < G * k_max              # every tile its own slot
        per_rank = np.bincount(slot // k_max, minlength=G)
        assert per_rank.max() <= k_max
        assert np.array_equal(slot, binding.shard_lpt(cost, G))                     # deterministic
        load = np.bincount(slot // k_max, weights=cost.astype(np.float64), minlength=G)
        blind = np.bincount(np.arange(T) % G, weights=cost.astype(np.float64), minlength=G)
        assert load.max() <= blind.max() + 1e-9
        if T >= 4 * G:
            assert load.max() <= cost.sum() / G + cost.max()                         # LPT's bound (slots were not binding here)
        # the gathered buffer of that map, de-interleaved: row y of tile t sits in slot slot[t]
        W, TR = 5, 8
        H = T * TR - 3
        frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        gathered = np.zeros((G * k_max * TR, W, 3), np.uint8)
        for t in range(T):
            rows = frame[t * TR:min(H, (t + 1) * TR)]
            gathered[slot[t] * TR:slot[t] * TR + len(rows)] = rows
        assert np.array_equal(binding.shard_deinterleave_map_host(gathered, W, H, TR, slot), frame)
        # within a rank the tiles keep their image order
        for rank in range(G):
            mine = np.flatnonzero(slot // k_max == rank)
            assert np.array_equal(np.sort(slot[mine]), slot[mine])
    # equal costs: still a valid map, and exactly as balanced as the blind one
    slot = binding.shard_lpt(np.full(135, 7, np.uint64), 8)
    assert np.bincount(slot // 17, minlength=8).max() == 17
    # the frame steps' rule: the blind map while it is within 10 % of balance (a smooth profile), LPT where it is not (expensive rows
    # that repeat with the period of the deal)
    T, G = 135, 8
    blind = (np.arange(T) % G) * 17 + np.arange(T) // G
    smooth = (np.where(np.arange(T) < 45, 15360, 400000) + 50 * np.arange(T)).astype(np.uint64)
    assert np.array_equal(binding.shard_by_cost(smooth, G), blind)
    periodic = np.where(np.arange(T) % G == 3, 900000, 20000).astype(np.uint64)
    slot = binding.shard_by_cost(periodic, G)
    assert np.array_equal(slot, binding.shard_lpt(periodic, G)) and not np.array_equal(slot, blind)
    load = np.bincount(slot // 17, weights=periodic.astype(np.float64), minlength=G)
    assert load.max() < 0.3 * np.bincount(np.arange(T) % G, weights=periodic.astype(np.float64), minlength=G).max()
