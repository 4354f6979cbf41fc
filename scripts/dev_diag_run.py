import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
ctx = _ffi.Context(0)
fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
fr.generate_rays()
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
for _ in range(300): fr.trace(p)
torch.cuda.synchronize()
os.environ["BHGEO_DIAG_DUMP"] = "gpurun_out/diag.bin"
fr.trace(p); torch.cuda.synchronize()
fr.trace(p); torch.cuda.synchronize()
