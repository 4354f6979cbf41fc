"""Rank 0's shard of a world-N dealing of the fixed 1024x1024 x5 frame, traced + shaded K times on this one GPU (what
bench.py's strong_predicted block times), as a program of its own so that rocprofv3 --kernel-trace --stats sees ONE shard
size: python3 scripts/dev/dev_shard_run.py N [K] [two]   ("two": consecutive frames alternate between two streams / contexts)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi, dist as bd
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
N = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
two = len(sys.argv) > 3 and "two" in sys.argv[3]
with_events = len(sys.argv) > 3 and "ev" in sys.argv[3]     # record a timing event pair around every trace, as bench.py does
W = H = 1024; S = 5
def tile_cost(cx, cy):
    return -abs(np.hypot(0.6 * (cx - W / 2) / W, 0.6 * (cy - H / 2) / H) - 2.598 / 30.0)
tile_cost.visit = "cost" if N > 1 else "row"
pix = bd.rank_pixels(W, H, 32, 0, N, tile_cost=tile_cost)
sky = synthetic_sky(2048, 1024)
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
ctxs = [_ffi.Context(0)] + ([_ffi.Context(0)] if two else [])
frs = []
for c in ctxs:
    f = DeviceFrame(c, W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=(1e-4, 0.0, 30.0), pixels=pix, directions_only=True)
    f.set_sky(sky); f.generate_rays(); frs.append(f)
if two and len(sys.argv) > 3 and "shared" in sys.argv[3]:
    frs[1].d_k0 = frs[0].d_k0          # (bench.py's twin_of: the second frame reads the SAME rays)
streams = [torch.cuda.Stream() for _ in frs]
if not two and len(sys.argv) > 3 and "null" in sys.argv[3]:
    streams = [torch.cuda.default_stream()]
if two and len(sys.argv) > 3 and "prio" in sys.argv[3]:
    streams = [torch.cuda.Stream(), torch.cuda.Stream(priority=-1)]     # another priority = another hardware queue
if two and len(sys.argv) > 3 and "many" in sys.argv[3]:
    pool = [torch.cuda.Stream() for _ in range(8)]                       # (does ANY pair of plain streams overlap?)
    streams = [pool[0], pool[int(os.environ.get("PAIR", "1"))]]
imgs = [torch.zeros((W * H, 4), dtype=torch.float32, device="cuda") for _ in frs]
evs = []
def run(k):
    for i in range(k):
        j = i % len(frs)
        with torch.cuda.stream(streams[j]):
            if with_events:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(streams[j]); frs[j].trace(p); e1.record(streams[j]); evs.append((e0, e1))
            else:
                frs[j].trace(p)
            frs[j].shade_f32(imgs[j], frs[j].d_pixels)
import time
if len(sys.argv) > 3 and "bench" in sys.argv[3]:
    import bench
    ms, call, _ = bench.time_frame(frs[0], p, K, 20, overlap=two)
    print("shard 1/%d: bench.time_frame %.4f ms per frame, trace call %.4f ms (%s)" % (N, ms, call, "two in flight" if two else "sequential"))
    sys.exit(0)
run(300); torch.cuda.synchronize()
t = time.perf_counter(); run(K); t_host = (time.perf_counter() - t) / K; torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
print("shard 1/%d: %d rays, %.4f ms per frame, host enqueue %.4f ms per frame (%s)" % (N, frs[0].n, dt * 1e3, t_host * 1e3, (("two in flight, second stream at another priority" if "prio" in sys.argv[3] else "two in flight, two plain streams") if two else "sequential") + (", event pair per trace" if with_events else "")))
