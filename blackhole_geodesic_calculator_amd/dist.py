"""Multi-GPU sharding of a frame: image tiles dealt round-robin to ranks, one gather at frame end.

Every ray is an independent ODE (no exchange during integration), so the path shards by
independent units.  All samples of a pixel stay on one rank, so the per-pixel sample mean
(raytracer/RelativisticRenderEngine.py:250) is local; cyclic tiles balance the expensive rays
near the photon sphere across ranks.  The only collective is one gather of per-pixel results to
rank 0 per frame (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).
"""
from __future__ import annotations

import numpy as np


def tile_grid(width: int, height: int, tile: int):
    return (int(width) + tile - 1) // tile, (int(height) + tile - 1) // tile


def rank_tiles(width: int, height: int, tile: int, rank: int, world: int) -> np.ndarray:
    """Tile ids (row-major over the tile grid) owned by `rank`: id % world == rank."""
    tx, ty = tile_grid(width, height, tile)
    return np.arange(rank, tx * ty, world, dtype=np.int64)


def rank_pixels(width: int, height: int, tile: int, rank: int, world: int) -> np.ndarray:
    """Flat pixel indices (y*W + x) owned by `rank`, tile after tile, row-major inside a tile."""
    W, H = int(width), int(height)
    tx, _ = tile_grid(W, H, tile)
    out = []
    for t in rank_tiles(W, H, tile, rank, world):
        ty_, tx_ = divmod(int(t), tx)
        ys = np.arange(ty_ * tile, min((ty_ + 1) * tile, H))
        xs = np.arange(tx_ * tile, min((tx_ + 1) * tile, W))
        out.append((ys[:, None] * W + xs[None, :]).reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def max_pixels_per_rank(width: int, height: int, tile: int, world: int) -> int:
    return max(len(rank_pixels(width, height, tile, r, world)) for r in range(world))


def gather_frame(local, width: int, height: int, tile: int, group=None, dst: int = 0):
    """Gather per-pixel results to rank `dst` and scatter them into frame order.

    local: torch tensor [P_local, C] for this rank's pixels in rank_pixels() order.
    Returns a [H, W, C] tensor on rank dst, None elsewhere.  One collective: dist.gather of
    equal-sized (padded) slabs.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    W, H = int(width), int(height)
    C_ = local.shape[1]
    if world == 1:
        out = torch.empty((H * W, C_), dtype=local.dtype, device=local.device)
        out[torch.from_numpy(rank_pixels(W, H, tile, 0, 1)).to(local.device)] = local
        return out.reshape(H, W, C_)
    pmax = max_pixels_per_rank(W, H, tile, world)
    slab = torch.zeros((pmax, C_), dtype=local.dtype, device=local.device)
    slab[: local.shape[0]] = local
    bufs = [torch.empty_like(slab) for _ in range(world)] if rank == dst else None
    dist.gather(slab, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty((H * W, C_), dtype=local.dtype, device=local.device)
    for r in range(world):
        px = torch.from_numpy(rank_pixels(W, H, tile, r, world)).to(local.device)
        out[px] = bufs[r][: len(px)]
    return out.reshape(H, W, C_)
