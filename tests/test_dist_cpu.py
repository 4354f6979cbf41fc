"""CPU tests of the multi-GPU sharding path: tile dealing and the frame-end gather, with
world_size 2 over gloo (the same code runs over RCCL on the GPU box)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_tiles_partition_the_frame():
    from blackhole_geodesic_calculator_amd import dist as bd
    for (W, H, T, world) in [(64, 64, 32, 2), (100, 70, 32, 3), (1024, 1024, 32, 8), (33, 17, 16, 4)]:
        allpix = np.concatenate([bd.rank_pixels(W, H, T, r, world) for r in range(world)])
        assert len(allpix) == W * H and np.array_equal(np.sort(allpix), np.arange(W * H))
        counts = [len(bd.rank_pixels(W, H, T, r, world)) for r in range(world)]
        assert max(counts) == bd.max_pixels_per_rank(W, H, T, world)
    # 1024^2 in 32-pixel tiles deals evenly to 1, 2, 4, 8 GPUs
    for world in (1, 2, 4, 8):
        assert {len(bd.rank_pixels(1024, 1024, 32, r, world)) for r in range(world)} == {1024 * 1024 // world}


def test_sharded_raygen_equals_full_frame():
    from blackhole_geodesic_calculator_amd import camera_directions
    from blackhole_geodesic_calculator_amd import dist as bd
    from blackhole_geodesic_calculator_amd.raygen import camera_directions_for_pixels
    W, H, S = 96, 64, 3
    full = camera_directions(W, H, S, 0.6, 0.6, 42.0).reshape(S, H * W, 3)
    for r in range(3):
        px = bd.rank_pixels(W, H, 32, r, 3)
        assert np.array_equal(camera_directions_for_pixels(W, H, S, px, 0.6, 0.6, 42.0), full[:, px, :])


_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BHG_ROOT"])
from blackhole_geodesic_calculator_amd import dist as bd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, T = 100, 70, 32
px = bd.rank_pixels(W, H, T, rank, world)
# a per-pixel payload every rank can compute: (pixel id, rank, y, x)
local = torch.tensor(np.stack([px, np.full_like(px, rank), px // W, px % W], 1), dtype=torch.float64)
img = bd.gather_frame(local, W, H, T)
if rank == 0:
    img = img.numpy()
    ids = np.arange(W * H).reshape(H, W)
    assert np.array_equal(img[..., 0], ids)
    assert np.array_equal(img[..., 2], ids // W) and np.array_equal(img[..., 3], ids % W)
    assert np.array_equal(img[..., 1], ((ids // W) // T + (ids % W) // T) % world)   # owner = (tile_x + tile_y) % world
    print("GATHER_OK")
else:
    assert img is None
dist.barrier()
dist.destroy_process_group()
"""


_WORKER2 = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BHG_ROOT"])
from blackhole_geodesic_calculator_amd import dist as bd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, T = 96, 64, 32
g = bd.FrameGatherer(W, H, T, channels=4, dtype=torch.float32, device="cpu")
px = bd.rank_pixels(W, H, T, rank, world)
assert g.P == len(px)
for frame in range(5):   # five frames through the two rotating slabs
    local = torch.tensor(np.stack([px + 1000 * frame, np.full_like(px, rank), px // W, px % W], 1), dtype=torch.float32)
    if frame % 2 == 0:
        g.submit(frame, local)
    else:                 # the producer fills the slab itself (what bench.py's frame end does)
        def fill(out, scatter, local=local):
            assert scatter is None and out.shape == local.shape
            out.copy_(local)
        g.submit_with(frame, fill)
g.drain()
dist.barrier()
if rank == 0:
    img = g.image().numpy()
    ids = np.arange(W * H).reshape(H, W)
    assert g.frames_done == 5
    assert np.array_equal(img[..., 0], ids + 4000)      # the last frame is what the image holds
    assert np.array_equal(img[..., 2], ids // W) and np.array_equal(img[..., 3], ids % W)
    print("GATHERER_OK")
else:
    assert g.image() is None
dist.destroy_process_group()
"""


_WORKER3 = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BHG_ROOT"])
from blackhole_geodesic_calculator_amd import dist as bd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, T = 128, 96, 32
cost = lambda cx, cy: -abs(np.hypot(cx - W / 2, cy - H / 2) - 30.0)   # a ring of expensive tiles, like the shadow edge
px = bd.rank_pixels(W, H, T, rank, world, tile_cost=cost)
assert not np.array_equal(px, bd.rank_pixels(W, H, T, rank, world))    # the cost order really differs
g = bd.FrameGatherer(W, H, T, channels=4, dtype=torch.float32, device="cpu", tile_cost=cost)
assert np.array_equal(g.pixels, px)
for frame in range(3):
    local = torch.tensor(np.stack([px + 1000 * frame, np.full_like(px, rank), px // W, px % W], 1), dtype=torch.float32)
    g.submit(frame, local)
g.drain()
img1 = bd.gather_frame(torch.tensor(np.stack([px, px], 1), dtype=torch.float64), W, H, T, tile_cost=cost)
dist.barrier()
if rank == 0:
    img = g.image().numpy()
    ids = np.arange(W * H).reshape(H, W)
    assert np.array_equal(img[..., 0], ids + 2000)
    assert np.array_equal(img[..., 2], ids // W) and np.array_equal(img[..., 3], ids % W)
    assert np.array_equal(img1.numpy()[..., 0], ids)
    print("COST_ORDER_OK")
dist.destroy_process_group()
"""


_WORKER4 = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BHG_ROOT"])
from blackhole_geodesic_calculator_amd import dist as bd
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, T = 128, 96, 16
true_cost = lambda p: 1.0 + 9.0 * np.exp(-((p % W - 70.0) ** 2 + (p // W - 40.0) ** 2) / 900.0)   # work per pixel
px = bd.rank_pixels(W, H, T, rank, world)                     # first pass: plain dealing, every rank prices ITS pixels
tc = bd.measured_tile_cost(W, H, T, px, true_cost(px))        # ... and the tables are summed over the ranks
ids = np.arange(W * H)
tx = (W + T - 1) // T
want = np.bincount((ids // W // T) * tx + (ids % W) // T, weights=true_cost(ids))
assert np.allclose(tc.table, want), (tc.table, want)          # the whole frame's cost map, on every rank
px2 = bd.rank_pixels(W, H, T, rank, world, tile_cost=tc)      # second pass: dealt and visited by measured cost
mine = true_cost(px2).sum()
tot = torch.tensor([mine], dtype=torch.float64)
dist.all_reduce(tot)
assert abs(mine / tot.item() - 1.0 / world) < 0.08, mine / tot.item()      # balanced to a tile
first, last = true_cost(px2[: T * T]).sum(), true_cost(px2[-T * T:]).sum()
assert first >= last                                           # longest first
g = bd.FrameGatherer(W, H, T, channels=1, dtype=torch.float64, device="cpu", tile_cost=tc)
g.submit(0, torch.tensor(px2[:, None], dtype=torch.float64))
g.drain()
if rank == 0:
    assert np.array_equal(g.image().numpy().reshape(-1), ids)
    print("MEASURED_COST_OK")
dist.destroy_process_group()
"""


def test_measured_tile_cost_world2_gloo(tmp_path):
    """A calibration pass prices the tiles (summed over the ranks), the next pass is dealt and ordered by it."""
    _run_world2(tmp_path, _WORKER4, "MEASURED_COST_OK")


def test_measured_tile_cost_and_visit_order_single_process():
    from blackhole_geodesic_calculator_amd import dist as bd
    W, H, T = 96, 64, 32
    px = bd.rank_pixels(W, H, T, 0, 1)
    cost = (px % 7 + 1).astype(np.float64)
    tc = bd.measured_tile_cost(W, H, T, px, cost)
    assert tc.table.shape == (6,) and np.isclose(tc.table.sum(), cost.sum())
    assert tc(16.0, 16.0) == tc.table[0] and tc(80.0, 48.0) == tc.table[5]
    order = bd.rank_tiles(W, H, T, 0, 1, tile_cost=tc)
    assert np.all(np.diff(tc.table[order]) <= 0)               # visited longest-first ...
    tc.visit = "row"                                           # ... or dealt by cost but visited row-major
    assert np.array_equal(bd.rank_tiles(W, H, T, 0, 1, tile_cost=tc), np.arange(6))
    own_cost = bd.tile_owner(W, H, T, 2, tile_cost=tc)
    tc.visit = "cost"
    assert np.array_equal(own_cost, bd.tile_owner(W, H, T, 2, tile_cost=tc))    # the dealing does not depend on it


def _run_world2(tmp_path, body, token):
    script = tmp_path / "worker.py"
    script.write_text(body)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BHG_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BHGEO_NO_TORCH_PRELOAD="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert token in out.stdout


def test_frame_gatherer_async_double_buffer_world2_gloo(tmp_path):
    """bench.py's frame-end path: asynchronous gather, two slabs in rotation, scatter on rank 0."""
    _run_world2(tmp_path, _WORKER2, "GATHERER_OK")


def test_frame_gatherer_with_tile_cost_order_world2_gloo(tmp_path):
    """Shards built longest-first (rank_pixels(tile_cost=), what bench.py does) land in frame order."""
    _run_world2(tmp_path, _WORKER3, "COST_ORDER_OK")


def test_frame_gatherer_single_process_tile_cost():
    import torch
    from blackhole_geodesic_calculator_amd import dist as bd
    W, H, T = 128, 128, 32
    cost = lambda cx, cy: -abs(np.hypot(cx - W / 2, cy - H / 2) - 40.0)
    px = bd.rank_pixels(W, H, T, 0, 1, tile_cost=cost)
    g = bd.FrameGatherer(W, H, T, channels=2, dtype=torch.float64, tile_cost=cost)
    g.submit(0, torch.tensor(np.stack([px, 2 * px], 1), dtype=torch.float64))
    g.drain()
    assert np.array_equal(g.image().numpy()[..., 0].reshape(-1), np.arange(W * H))

    def fill(out, scatter):
        out[scatter] = torch.tensor(np.stack([3 * px, px], 1), dtype=torch.float64)
    g.submit_with(1, fill)
    g.drain()
    assert np.array_equal(g.image().numpy()[..., 0].reshape(-1), 3 * np.arange(W * H))


def test_tile_dealing_is_skewed_and_balanced():
    """Without a cost, owner = (tile_x + tile_y) % world: no rank owns whole tile columns, every rank owns 1/world of
    each tile row.  With bench.py's tile cost the tiles are dealt by cost ranking and the shadow-edge pixels spread
    evenly (id % world gave +-10 % at 8 ranks, and so does any fixed lattice)."""
    from blackhole_geodesic_calculator_amd import dist as bd
    for world, (W, H) in [(2, (2048, 1024)), (4, (2048, 2048)), (8, (4096, 2048))]:
        own = bd.tile_owner(W, H, 32, world).reshape(H // 32, W // 32)
        for r in range(world):
            assert ((own == r).sum(1) == W // 32 // world).all()
            assert not (own == r).all(0).any()
        # dealt by cost ranking (bench.py's tile_cost): the shadow-edge ring spreads to within a tile per rank
        cost = lambda cx, cy: -abs(np.hypot(0.6 * (cx - W / 2) / W, 0.6 * (cy - H / 2) / H) - 2.598 / 30.0)
        own = bd.tile_owner(W, H, 32, world, cost).reshape(H // 32, W // 32)
        assert np.bincount(own.reshape(-1), minlength=world).tolist() == [own.size // world] * world
        ys, xs = np.mgrid[0:H:4, 0:W:4]
        b = np.hypot(0.6 * (xs - W / 2) / W, 0.6 * (ys - H / 2) / H) * 30.0      # impact parameter of the pixel
        for width in (0.2, 0.4, 1.0):
            edge = (np.abs(b - 2.598) < width)
            cnt = np.array([(edge & (own[ys // 32, xs // 32] == r)).sum() for r in range(world)])
            assert cnt.max() / cnt.mean() < 1.04 and cnt.min() / cnt.mean() > 0.96, (world, width, cnt)


def test_frame_gatherer_single_process():
    import torch
    from blackhole_geodesic_calculator_amd import dist as bd
    g = bd.FrameGatherer(64, 32, 16, channels=2, dtype=torch.float64)
    px = bd.rank_pixels(64, 32, 16, 0, 1)
    g.submit(0, torch.tensor(np.stack([px, 2 * px], 1), dtype=torch.float64))
    g.drain()
    img = g.image().numpy()
    assert np.array_equal(img[..., 0].reshape(-1), np.arange(64 * 32)) and g.frames_done == 1

    def fill(out, scatter):   # single rank: the producer scatters straight into the frame image
        out[scatter] = torch.tensor(np.stack([3 * px, px], 1), dtype=torch.float64)
    g.submit_with(1, fill)
    g.drain()
    assert np.array_equal(g.image().numpy()[..., 0].reshape(-1), 3 * np.arange(64 * 32)) and g.frames_done == 2


def test_gather_frame_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BHG_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BHGEO_NO_TORCH_PRELOAD="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GATHER_OK" in out.stdout
