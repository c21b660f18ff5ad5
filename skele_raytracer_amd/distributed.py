"""Row-tile sharding of one frame over the GPUs of a node (one process per GPU).

The framebuffer is cut into tiles of `tile_rows` rows; tile t belongs to rank t % world
(interleaved, because cost is very non-uniform vertically: sky rows trace 1 ray per pixel,
ground rows up to 1 + N + N^2).  Every rank renders its tiles into a compact, equally sized
(padded) u8 buffer with one call of skr_render_tiles (include/skr.h:
first_tile = rank, tile_stride = world); one collective — an all-gather of those buffers
over RCCL/xGMI (`torch.distributed`, backend "nccl"; "gloo" in the CPU tests) — brings them
together and the root de-interleaves.  There is no exchange inside the frame: random numbers
are keyed by the global pixel index, so the image does not depend on the partition.

The reference has no distributed path at all; this is new design.
"""
import torch
import torch.distributed as dist


def tiles_total(height, tile_rows):
    return (height + tile_rows - 1) // tile_rows


def tiles_per_rank(height, tile_rows, world):
    """Padded tile count per rank (equal buffer sizes for the collective)."""
    return (tiles_total(height, tile_rows) + world - 1) // world


def my_tiles(height, tile_rows, rank, world):
    return list(range(rank, tiles_total(height, tile_rows), world))


def deinterleave(gathered, height, tile_rows, world):
    """gathered: [world, k_max*tile_rows, W, 3] (rank-major) -> frame [height, W, 3].
    Tile t lives at rank t % world, slot t // world."""
    _, rows, w, c = gathered.shape
    k_max = rows // tile_rows
    g = gathered.view(world, k_max, tile_rows, w, c).permute(1, 0, 2, 3, 4).reshape(-1, w, c)
    return g[:height]


class FrameSharder:
    """Owns the per-rank tile buffer and the collective.  `render_into(buf)` must enqueue the
    rendering of this rank's tiles into `buf` ([k_max*tile_rows, W, 3] uint8) on the current stream."""

    def __init__(self, width, height, tile_rows, rank, world, device):
        self.width, self.height, self.tile_rows, self.rank, self.world = width, height, tile_rows, rank, world
        self.k_max = tiles_per_rank(height, tile_rows, world)
        self.mine = torch.zeros((self.k_max * tile_rows, width, 3), dtype=torch.uint8, device=device)
        # concatenated along dim 0 (the layout both the nccl and gloo backends accept)
        self.all = torch.zeros((world * self.k_max * tile_rows, width, 3), dtype=torch.uint8, device=device) if world > 1 else None
        self.frame = None

    def step(self, render_into):
        render_into(self.mine)
        if self.world > 1:
            dist.all_gather_into_tensor(self.all, self.mine)
            if self.rank == 0:
                stacked = self.all.view(self.world, self.k_max * self.tile_rows, self.width, 3)
                self.frame = deinterleave(stacked, self.height, self.tile_rows, self.world).contiguous()
        else:
            self.frame = self.mine[:self.height]
        return self.frame
