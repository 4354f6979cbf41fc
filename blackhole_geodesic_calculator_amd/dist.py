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


def _tile_costs(width, height, tile, tile_cost):
    tx, ty = tile_grid(width, height, tile)
    ids = np.arange(tx * ty, dtype=np.int64)
    return np.array([tile_cost((x + 0.5) * tile, (y + 0.5) * tile) for y, x in zip(ids // tx, ids % tx)], dtype=np.float64)


def deal_sequence(n_tiles: int, world: int, root_share: float = 1.0) -> np.ndarray:
    """The ranks the tiles of a cost ranking are dealt to, in ranking order: rounds of one tile per rank -- and, with
    root_share < 1, rank 0 sits out a fraction 1 - root_share of the rounds (evenly spread, Bresenham fashion), so that
    it ends up with root_share of an equal part, the others with correspondingly more.  Rank 0 is the frame's owner: on
    top of its shard it receives the gather and assembles the frame, and a smaller shard lets all ranks finish
    together (bench.py measures the assembly time and derives the share)."""
    world = int(world)
    q = min(max(1.0 - float(root_share), 0.0), 1.0) if world > 1 else 0.0
    seq = np.empty(int(n_tiles), dtype=np.int64)
    i = r = 0
    while i < n_tiles:
        skip = int((r + 1) * q) > int(r * q)
        for k in range(1 if skip else 0, world):
            if i >= n_tiles:
                break
            seq[i] = k
            i += 1
        r += 1
    return seq


def tile_owner(width: int, height: int, tile: int, world: int, tile_cost=None) -> np.ndarray:
    """Owner rank of every tile (row-major over the tile grid).

    Without a cost: tiles are dealt cyclically along each tile row and every row starts one rank further on,
    owner = (tile_x + tile_y) % world.  (Plain id % world hands out whole tile COLUMNS whenever the tiles per row
    divide by the world size -- 32, 64, 128 tiles per row against 2, 4, 8 ranks.)
    With tile_cost(cx, cy) -> float (expected work of the tile centred there): the tiles are sorted by decreasing
    cost (stable) and dealt round-robin in THAT order, so every rank gets one of each `world` consecutive tiles
    of the cost ranking -- longest-processing-time-first across ranks.  A thin ring of expensive tiles (the
    shadow edge) then spreads to within one tile per rank, where any fixed lattice leaves +-10 % at 8 ranks.
    An attribute tile_cost.root_share in (0, 1) deals rank 0 that share of an equal part (deal_sequence)."""
    tx, ty = tile_grid(width, height, tile)
    ids = np.arange(tx * ty, dtype=np.int64)
    if tile_cost is None:
        return (ids % tx + ids // tx) % int(world)
    order = np.argsort(-_tile_costs(width, height, tile, tile_cost), kind="stable")
    own = np.empty(tx * ty, dtype=np.int64)
    # (tile_cost.root_share < 1: rank 0 -- the frame's owner -- is dealt a smaller part, see deal_sequence)
    own[order] = deal_sequence(tx * ty, world, getattr(tile_cost, "root_share", 1.0))
    return own


def rank_tiles(width: int, height: int, tile: int, rank: int, world: int, tile_cost=None) -> np.ndarray:
    """Tile ids (row-major over the tile grid) owned by `rank`: ascending, or -- with a cost -- by decreasing cost."""
    mine = np.nonzero(tile_owner(width, height, tile, world, tile_cost) == int(rank))[0].astype(np.int64)
    if tile_cost is not None and getattr(tile_cost, "visit", "cost") == "cost":
        mine = mine[np.argsort(-_tile_costs(width, height, tile, tile_cost)[mine], kind="stable")]
    return mine


def rank_pixels(width: int, height: int, tile: int, rank: int, world: int, tile_cost=None) -> np.ndarray:
    """Flat pixel indices (y*W + x) owned by `rank`, tile after tile, row-major inside a tile.

    tile_cost(cx, cy) -> float, optional: tiles are dealt to the ranks by cost ranking (tile_owner) and each rank
    visits its tiles in order of DECREASING cost (the expensive rays near the photon sphere start early and the
    kernel's tail is made of cheap far-field rays).  Every rank -- and the FrameGatherer -- must use the same
    tile_cost."""
    W, H = int(width), int(height)
    tx, _ = tile_grid(W, H, tile)
    out = []
    for t in rank_tiles(W, H, tile, rank, world, tile_cost):
        ty_, tx_ = divmod(int(t), tx)
        ys = np.arange(ty_ * tile, min((ty_ + 1) * tile, H))
        xs = np.arange(tx_ * tile, min((tx_ + 1) * tile, W))
        out.append((ys[:, None] * W + xs[None, :]).reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def measured_tile_cost(width: int, height: int, tile: int, pixels, cost_per_pixel, group=None):
    """A tile_cost made of MEASURED work: cost_per_pixel[p] is what pixel pixels[p] cost the last time it was traced
    (its rays' attempted steps summed over samples -- DeviceFrame.pixel_cost(); the reference renders the same
    pixels `samples` times per frame and the same view frame after frame, :242-250, so the last pass prices the next).
    Returns tile_cost(cx, cy) for tile_owner / rank_pixels / FrameGatherer, with the per-tile sums in `.table`
    (row-major over the tile grid).  With torch.distributed initialised over more than one rank every rank passes ITS
    pixels and the tables are summed over the ranks (one small all-reduce, outside any timed region), so that all ranks
    hold the same cost map: the tiles can then be re-dealt by measured cost, which is what balances the shards."""
    tx, ty = tile_grid(width, height, tile)
    px = np.asarray(pixels, dtype=np.int64)
    t = (px // int(width) // tile) * tx + (px % int(width)) // tile
    table = np.bincount(t, weights=np.asarray(cost_per_pixel, dtype=np.float64), minlength=tx * ty).astype(np.float64)
    try:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            tt = torch.from_numpy(table).to(dev)
            dist.all_reduce(tt, group=group)
            table = tt.cpu().numpy()
    except ImportError:
        pass

    def tile_cost(cx, cy):
        return table[int(cy // tile) * tx + int(cx // tile)]
    tile_cost.table = table
    return tile_cost


def max_pixels_per_rank(width: int, height: int, tile: int, world: int, tile_cost=None) -> int:
    return max(len(rank_pixels(width, height, tile, r, world, tile_cost)) for r in range(world))


def gather_frame(local, width: int, height: int, tile: int, group=None, dst: int = 0, tile_cost=None):
    """Gather per-pixel results to rank `dst` and scatter them into frame order.

    local: torch tensor [P_local, C] for this rank's pixels in rank_pixels(..., tile_cost) order (every rank
    must pass the same tile_cost it built its shard with).
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
        out[torch.from_numpy(rank_pixels(W, H, tile, 0, 1, tile_cost)).to(local.device)] = local
        return out.reshape(H, W, C_)
    pmax = max_pixels_per_rank(W, H, tile, world, tile_cost)
    slab = torch.zeros((pmax, C_), dtype=local.dtype, device=local.device)
    slab[: local.shape[0]] = local
    bufs = [torch.empty_like(slab) for _ in range(world)] if rank == dst else None
    dist.gather(slab, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty((H * W, C_), dtype=local.dtype, device=local.device)
    for r in range(world):
        px = torch.from_numpy(rank_pixels(W, H, tile, r, world, tile_cost)).to(local.device)
        out[px] = bufs[r][: len(px)]
    return out.reshape(H, W, C_)


class FrameGatherer:
    """The frame-end exchange of a sharded frame: ONE gather per frame of this rank's per-pixel
    slab to rank `dst`, issued asynchronously (it overlaps the next frame's trace) with two slabs in
    rotation so a slab in flight is never overwritten.  Rank dst scatters the slabs into frame order.
    Works over RCCL ("nccl") on GPUs and over gloo on CPU tensors (tests)."""

    def __init__(self, width, height, tile, channels=4, dtype=None, device="cpu", group=None, dst=0, assemble=None,
                 tile_cost=None, collective=None):
        """assemble(slabs [world*pmax, C], perm [H*W] int64, frame [H*W, C]): optional device routine for the
        root's frame assembly, frame[p] = slabs[perm[p]] (bench.py passes libbhgeo's kernel); default: torch.
        tile_cost: the SAME function the shards were built with (rank_pixels(..., tile_cost=)): row p of a rank's
        slab belongs to the p-th pixel of that list, so the gatherer must know the order.
        collective: None = issue the gather when there is more than one rank; True = issue it even in a process
        group of ONE rank (a 1-rank gather is legal): the whole N > 1 code path -- slabs, the asynchronous
        collective on the backend's stream, the root-side assembly -- then runs on a single GPU."""
        import torch
        import torch.distributed as dist

        self.dist = dist if dist.is_initialized() else None
        self.torch = torch
        self.assemble = assemble
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rank = dist.get_rank(group) if self.dist else 0
        self.W, self.H, self.tile = int(width), int(height), int(tile)
        self.collective = (self.world > 1) if collective is None else bool(collective)
        if self.collective and self.dist is None:
            raise RuntimeError("collective=True needs an initialised torch.distributed process group")
        dtype = dtype or torch.float32
        self.pixels = rank_pixels(self.W, self.H, tile, self.rank, self.world, tile_cost)   # this rank's slab rows
        self.P = len(self.pixels)
        self.pmax = max_pixels_per_rank(self.W, self.H, tile, self.world, tile_cost)
        self.slabs = [torch.zeros((self.pmax, channels), dtype=dtype, device=device) for _ in range(2)]
        self.is_dst = self.rank == dst
        self.recv = [None, None]
        self.pix_of = None
        self.frame = None
        if self.is_dst:
            pix = [rank_pixels(self.W, self.H, tile, r, self.world, tile_cost) for r in range(self.world)]
            self.pix_of = [torch.from_numpy(p).to(device) for p in pix]
            if self.collective:
                # the ranks' slabs arrive in ONE block [world * pmax, C]; frame order is a single gather through
                # `perm` (perm[pixel] = r * pmax + position of the pixel in rank r's list): one kernel per frame
                # on the root instead of one scatter per rank
                self.recv_all = [torch.empty((self.world * self.pmax, channels), dtype=dtype, device=device)
                                 for _ in range(2)]
                self.recv = [[ra[r * self.pmax:(r + 1) * self.pmax] for r in range(self.world)] for ra in self.recv_all]
                import numpy as _np
                perm = _np.empty(self.H * self.W, dtype=_np.int64)
                for r, p in enumerate(pix):
                    perm[p] = r * self.pmax + _np.arange(len(p), dtype=_np.int64)
                self.perm = torch.from_numpy(perm).to(device)
            self.frame = torch.zeros((self.H * self.W, channels), dtype=dtype, device=device)
        self.pending = [None, None]
        self.frames_done = 0

    def finish(self, b):
        """Wait for slab b's gather (if any) and, on dst, scatter it into the frame image."""
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
            if self.is_dst:
                if self.assemble is not None:   # libbhgeo's gather kernel (float RGBA rows on a GPU)
                    self.assemble(self.recv_all[b], self.perm, self.frame)
                else:
                    for r in range(self.world):
                        self.frame[self.pix_of[r]] = self.recv[b][r][: len(self.pix_of[r])]
                self.frames_done += 1

    def submit(self, i, local):
        """Frame i's per-pixel result of this rank ([P, C], rank_pixels() order)."""
        b = i & 1
        self.finish(b)
        self.last = b
        self.slabs[b][: self.P].copy_(local)
        if self.collective:
            self.pending[b] = self.dist.gather(self.slabs[b], self.recv[b], dst=self.dst, group=self.group, async_op=True)
        else:
            self.frame[self.pix_of[0]] = self.slabs[b][: self.P]
            self.frames_done += 1

    def submit_with(self, i, fill):
        """Like submit(), but the producer writes its [P, C] result itself: fill(out, scatter) is called with
        either (this rank's slab, None) when a gather follows, or -- single rank -- (the frame image [H*W, C],
        this rank's flat pixel ids), so that the result lands in frame order without an intermediate copy."""
        b = i & 1
        self.finish(b)
        self.last = b
        if self.collective:
            fill(self.slabs[b][: self.P], None)
            self.pending[b] = self.dist.gather(self.slabs[b], self.recv[b], dst=self.dst, group=self.group, async_op=True)
        else:
            fill(self.frame, self.pix_of[0])
            self.frames_done += 1

    def drain(self):
        """Finish what is in flight, oldest frame first."""
        last = getattr(self, "last", 1)
        for b in (last ^ 1, last):
            self.finish(b)

    def image(self):
        """[H, W, C] on dst (call drain() first), None elsewhere."""
        return self.frame.reshape(self.H, self.W, -1) if self.is_dst else None
