"""The sharded frame pipeline end to end with N ranks (torch.distributed): tile dealing, sharded ray generation, trace, shade to
float RGBA into the gather slab, asynchronous gather, assembly on rank 0 -- the gathered frame must be bit-identical to the
same frame rendered by one rank.  Started by torch.distributed.run (tests/test_gpu_multirank.py; scripts/first_node_run.py).
BHG_DISTINCT=1: one GPU per rank, the process group is nccl (= RCCL) -- the form a multi-GPU node runs; otherwise the ranks
share GPU 0 and the group is gloo (RCCL refuses two ranks on one device)."""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("BHG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackhole_geodesic_calculator_amd import _ffi, dist as bd
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
from blackhole_geodesic_calculator_amd.raygen import python_random_stream
distinct = os.environ.get("BHG_DISTINCT") == "1"
dev = int(os.environ.get("LOCAL_RANK", "0")) if distinct else 0
torch.cuda.set_device(dev)
if distinct:
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, S, T = 160, 96, 3, 32
cam = np.array([1e-4, 0.0, 30.0])
sky = synthetic_sky(256, 128)
jit = python_random_stream(42.0, 2 * S * W * H)
ctx = _ffi.Context(dev)
params = _ffi.make_params(r_s=1.0, lambda_end=60.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=8.0)
sph, rgb, lamps = [[3.0, 2.0, 9.0, 1.5]], [[1.0, 0.8, 0.6]], [[10.0, 10.0, 30.0, 25.0]]

def make(pixels):
    f = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=cam, pixels=pixels, jitter=jit)
    f.set_sky(sky); f.set_disk(3.0, 8.0); f.set_objects(sph, rgb, lamps); f.generate_rays()
    return f

mine = make(bd.rank_pixels(W, H, T, rank, world))
def assemble(slabs, perm, frame):
    ctx.assemble_frame_f32_device(slabs.data_ptr(), perm.data_ptr(), frame.shape[0], frame.data_ptr(),
                                  stream=torch.cuda.current_stream().cuda_stream)
g = bd.FrameGatherer(W, H, T, channels=4, dtype=torch.float32, device="cuda", assemble=assemble)
for frame in range(3):                      # three frames through the two rotating slabs
    mine.trace(params)
    g.submit_with(frame, mine.shade_f32)
g.drain()
dist.barrier()
if rank == 0:
    full = make(None)
    full.trace(params)
    want = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
    full.shade_f32(want)
    torch.cuda.synchronize()
    got = g.image().reshape(-1, 4)
    assert g.frames_done == 3
    assert torch.equal(got, want), float((got - want).abs().max())
    kinds = {int(f): int((full.d_flags == f).sum()) for f in torch.unique(full.d_flags)}
    assert kinds.get(1, 0) > 50 and kinds.get(128, 0) > 50 and kinds.get(0x88, 0) > 50, kinds
    print("MULTIRANK_OK", dict(world=world, backend=dist.get_backend(), distinct_devices=distinct, flags=kinds))
dist.barrier()
dist.destroy_process_group()
