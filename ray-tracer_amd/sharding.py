"""Tile sharding of one frame across ranks (one process per GPU).

Pixels are independent in the reference (src/render.cpp:34-42), so the frame is
split by interleaved 8x8 tiles: tile k (row-major over ceil(W/8) x ceil(H/8))
belongs to rank k % world.  Scene and BVH are replicated; the only exchange is
one gather of the packed tile buffers to rank 0.  The numpy functions mirror
the device layout (rt_render_device / rt_deinterleave) and are what the CPU
multi-process tests run.
"""
from __future__ import annotations

import numpy as np

TILE = 8


def tiles(width: int, height: int):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def shard_tiles(width: int, height: int, shard: int, world: int) -> int:
    tx, ty = tiles(width, height)
    n = tx * ty
    return (n - shard + world - 1) // world if n > shard else 0


def shard_floats(width: int, height: int, shard: int, world: int) -> int:
    """Same value as rt_shard_floats()."""
    if world == 1:
        return width * height * 3
    return shard_tiles(width, height, shard, world) * 64 * 3


def pack_shard(frame: np.ndarray, shard: int, world: int) -> np.ndarray:
    """Screen-order frame (H, W, 3) -> this shard's packed tiles, padded to the
    stride of shard 0 (the largest)."""
    h, w, _ = frame.shape
    tx, ty = tiles(w, h)
    out = np.zeros(shard_floats(w, h, 0, world), dtype=np.float32).reshape(-1, 3)
    for local, tile in enumerate(range(shard, tx * ty, world)):
        x0, y0 = (tile % tx) * TILE, (tile // tx) * TILE
        for within in range(64):
            x, y = x0 + (within & 7), y0 + (within >> 3)
            if x < w and y < h:
                out[local * 64 + within] = frame[h - 1 - y, x]
    return out.reshape(-1)


def deinterleave(packed_all: np.ndarray, width: int, height: int, world: int) -> np.ndarray:
    """(world, stride) packed buffers -> Screen-order frame (H, W, 3)."""
    tx, ty = tiles(width, height)
    packed_all = np.asarray(packed_all, dtype=np.float32).reshape(world, -1, 3)
    frame = np.zeros((height, width, 3), dtype=np.float32)
    for tile in range(tx * ty):
        shard, local = tile % world, tile // world
        x0, y0 = (tile % tx) * TILE, (tile // tx) * TILE
        for within in range(64):
            x, y = x0 + (within & 7), y0 + (within >> 3)
            if x < width and y < height:
                frame[height - 1 - y, x] = packed_all[shard, local * 64 + within]
    return frame


def gather_packed(packed, rank: int, world: int, dist, gathered=None):
    """The one collective of the path: every rank's packed buffer to rank 0.
    `packed` / `gathered` are torch tensors on the backend's device."""
    if rank == 0:
        dist.gather(packed, [gathered[i] for i in range(world)], dst=0)
    else:
        dist.gather(packed, None, dst=0)
    return gathered
