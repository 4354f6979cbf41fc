"""Cost of putting 8 ranks' slabs into frame order on the root GPU: one index_select vs one index_put per rank."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import dist as bd
W, H, T, world = 4096, 2048, 32, 8
pix = [bd.rank_pixels(W, H, T, r, world) for r in range(world)]
pmax = max(len(p) for p in pix)
recv = torch.rand((world * pmax, 4), dtype=torch.float32, device="cuda")
frame = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
perm = np.empty(H * W, np.int64)
for r, p in enumerate(pix):
    perm[p] = r * pmax + np.arange(len(p))
d_perm = torch.from_numpy(perm).cuda()
d_pix = [torch.from_numpy(p).cuda() for p in pix]
def t(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def old():
    for r in range(world):
        frame[d_pix[r]] = recv[r * pmax:(r + 1) * pmax][: len(pix[r])]
def new():
    torch.index_select(recv, 0, d_perm, out=frame)
from blackhole_geodesic_calculator_amd import _ffi
ctx = _ffi.Context(0)
def own():
    ctx.assemble_frame_f32_device(recv.data_ptr(), d_perm.data_ptr(), H * W, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
print("8 index_put: %.1f us   1 index_select: %.1f us   libbhgeo gather: %.1f us" % (t(old), t(new), t(own)))
frame.zero_(); own(); torch.cuda.synchronize(); c = frame.clone()
a = frame.clone(); old(); b = frame.clone(); new()
print("same result:", torch.equal(b, frame), torch.equal(b, c))
