"""Round 5: config 4 (2048 x 2048 x 16 = 67.1 M rays, orbiting sphere, exit sphere) -- EVERY ray against the checker."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
from oracle import oracle as oc
from test_gpu_fullsize import _trace_device
oc.build()
CAM = np.array([1e-4, 0.0, 30.0])
ctx = _ffi.Context(0)
fr = DeviceFrame(ctx, 2048, 2048, 16, fov_x=0.6, fov_y=0.6)
fr.generate_rays()
sph = [[8.0 * np.cos(0.7), 8.0 * np.sin(0.7) * np.cos(np.radians(70.0)), 8.0 * np.sin(0.7) * np.sin(np.radians(70.0)), 1.5]]
kw = dict(r_s=1.0, lambda_end=80.0, r_exit=40.0)
end, fl, st, ac, ob = _trace_device(ctx, _ffi.make_params(**kw), fr.d_k0, x0_shared=CAM, spheres=sph)
k = fr.d_k0.cpu().numpy()
t = time.time()
o = oc.trace(k, CAM, spheres=sph, **kw)
to = time.time() - t
flg, stp, acn, obj = fl.cpu().numpy(), st.cpu().numpy().astype(np.uint32), ac.cpu().numpy().astype(np.uint32), ob.cpu().numpy()
rec = dict(rays=int(len(k)), flag_diff=int((flg != o["flags"]).sum()), attempted_diff=int((stp != o["n_attempted"]).sum()),
           accepted_diff=int((acn != o["n_accepted"]).sum()), object_id_diff=int((obj != o["object_id"]).sum()), oracle_s=round(to, 1),
           census={int(f): int(n) for f, n in zip(*np.unique(flg, return_counts=True))})
d = np.abs(end.cpu().numpy() - o["end"]).max(1)
for name, m in (("escaped", (flg == 4) | (flg == 8)), ("object", flg == 0x88), ("horizon", (flg & 1) != 0)):
    rec[name] = dict(rays=int(m.sum()), median=float(np.median(d[m])), p9999=float(np.quantile(d[m], 0.9999)), worst=float(d[m].max()))
print(json.dumps(rec))
json.dump(rec, open("gpurun_out/r05_config4_every_ray.json", "w"), indent=1)
