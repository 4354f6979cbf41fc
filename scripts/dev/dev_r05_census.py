"""Round 5 development aid: the numbers behind the every-ray parity tests (configs 2, 3, 4, Kerr off-axis), as JSON."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from blackhole_geodesic_calculator_amd import _ffi, camera_directions
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, FrameBatch
from oracle import oracle as oc
from test_gpu_parity import CLASS_OF, COND, STATED, _sensitivity
from test_gpu_fullsize import _trace_device

oc.build()
CAM = np.array([1e-4, 0.0, 30.0])
ctx = _ffi.Context(0)
out = {}


def classes(name, flags, d, k_all, x_all, o, kw, kerr=False, same=None):
    rec = {}
    for cls, sel in CLASS_OF.items():
        m = sel(flags)
        if same is not None:
            m = m & same
        bound = STATED[cls][1 if kerr else 0]
        if not m.any() or bound is None:
            continue
        dm = d[m]
        over = np.nonzero(m & ~(d <= bound))[0]
        r = dict(rays=int(m.sum()), median=float(np.median(dm)), p99=float(np.quantile(dm, 0.99)), p999=float(np.quantile(dm, 0.999)),
                 p9999=float(np.quantile(dm, 0.9999)), max=float(dm.max()), bound=bound, over=int(len(over)))
        if len(over):
            xo = x_all if x_all.ndim == 1 else x_all[over]
            S = _sensitivity(oc, k_all[over], xo, o["end"][over], **kw)
            S = np.nan_to_num(S, nan=np.inf, posinf=np.inf)
            ratio = (d[over] - bound) / np.maximum(S, 1e-300)
            r["ratio_to_S_max"] = float(ratio.max())
            r["ratio_to_S_quantiles"] = [float(np.quantile(ratio, q)) for q in (0.5, 0.9, 0.99)]
            r["beyond_COND"] = int((ratio > COND * (10 if kerr else 1)).sum())
            top = over[np.argsort(-d[over])[:12]]
            Smap = dict(zip(over.tolist(), S.tolist()))
            r["worst"] = [dict(i=int(i), d=float(d[i]), S=float(Smap[int(i)]), end_inf=float(np.abs(o["end"][i]).max()),
                               k_inf=float(np.abs(o["end"][i, 3:6]).max()), steps=int(o["n_attempted"][i])) for i in top]
            rel = d[over] / np.maximum(1.0, np.abs(o["end"][over]).max(1))
            r["over_rel_to_state_max"] = float(rel.max())
        rec[cls] = r
    out[name] = {**out.get(name, {}), "classes": rec}


# ---- config 2
k0 = camera_directions(1024, 1024, 5, 0.6, 0.6, 42.0).reshape(-1, 3)
kw = dict(r_s=1.0, lambda_end=50.0)
end, flags, steps, acc = ctx.trace(k0, CAM, _ffi.make_params(**kw))
o = oc.trace(k0, CAM, **kw)
out["config2"] = dict(nf=int((flags != o["flags"]).sum()), ns=int((steps != o["n_attempted"]).sum()), na=int((acc != o["n_accepted"]).sum()))
d = np.abs(end - o["end"]).max(1)
classes("config2", flags, d, k0, CAM, o, kw)
print(json.dumps(out["config2"]), flush=True)

# ---- config 3
cams = [dict(origin=(30 * np.sin(i), 0.0, 30 * np.cos(i)), rotation_euler=(0.0, i, 0.0)) for i in np.radians([85.0, 80.0, 60.0, 30.0, 5.0])]
fb = FrameBatch(ctx, cams, 1024, 1024, 1, fov_x=0.9, fov_y=0.9)
fb.generate_rays()
kw = dict(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)
endt, fl, st, ac, _ = _trace_device(ctx, _ffi.make_params(**kw), fb.d_k0, x0=fb.d_x0)
k_all, x_all = fb.d_k0.cpu().numpy(), fb.d_x0.cpu().numpy()
o = oc.trace(k_all, x_all, **kw)
flg, stp, acn = fl.cpu().numpy(), st.cpu().numpy().astype(np.uint32), ac.cpu().numpy().astype(np.uint32)
fbad = flg != o["flags"]
sbad = stp != o["n_attempted"]
out["config3"] = dict(nf=int(fbad.sum()), ns=int(sbad.sum()), na=int((acn != o["n_accepted"]).sum()),
                      flag_pairs=[[int(a), int(b)] for a, b in zip(flg[fbad][:20], o["flags"][fbad][:20])],
                      step_pairs=[[int(a), int(b), int(f)] for a, b, f in zip(stp[sbad][:20], o["n_attempted"][sbad][:20], flg[sbad][:20])],
                      bad_frames=[int(v) for v in (np.nonzero(fbad | sbad)[0] // (1024 * 1024))[:40]])
d = np.abs(endt.cpu().numpy() - o["end"]).max(1)
classes("config3", flg, d, k_all, x_all, o, kw, same=~(fbad | sbad))
print(json.dumps(out["config3"]), flush=True)
del fb, endt, fl, st, ac, o

# ---- config 4 (every 16th + last 2000)
fr = DeviceFrame(ctx, 2048, 2048, 16, fov_x=0.6, fov_y=0.6)
fr.generate_rays()
n = fr.n
sph = [[8.0 * np.cos(0.7), 8.0 * np.sin(0.7) * np.cos(np.radians(70.0)), 8.0 * np.sin(0.7) * np.sin(np.radians(70.0)), 1.5]]
kw = dict(r_s=1.0, lambda_end=80.0, r_exit=40.0)
endt, fl, st, ac, ob = _trace_device(ctx, _ffi.make_params(**kw), fr.d_k0, x0_shared=CAM, spheres=sph)
idx = torch.cat([torch.arange(0, n - 2000, 16, device="cuda"), torch.arange(n - 2000, n, device="cuda")])
k_idx = fr.d_k0[idx].cpu().numpy()
kwo = dict(kw, spheres=sph)
o = oc.trace(k_idx, CAM, **kwo)
flg, stp, acn, obj = fl[idx].cpu().numpy(), st[idx].cpu().numpy().astype(np.uint32), ac[idx].cpu().numpy().astype(np.uint32), ob[idx].cpu().numpy()
fbad = flg != o["flags"]
sbad = stp != o["n_attempted"]
out["config4"] = dict(compared=int(len(idx)), nf=int(fbad.sum()), ns=int(sbad.sum()), na=int((acn != o["n_accepted"]).sum()), nobj=int((obj != o["object_id"]).sum()),
                      flag_pairs=[[int(a), int(b)] for a, b in zip(flg[fbad][:20], o["flags"][fbad][:20])],
                      step_pairs=[[int(a), int(b), int(f)] for a, b, f in zip(stp[sbad][:20], o["n_attempted"][sbad][:20], flg[sbad][:20])])
d = np.abs(endt[idx].cpu().numpy() - o["end"]).max(1)
classes("config4", flg, d, k_idx, CAM, o, kwo, same=~(fbad | sbad))
print(json.dumps(out["config4"]), flush=True)
del fr, endt, fl, st, ac, ob, o

# ---- Kerr off-axis
inc = np.radians(60.0)
cam = np.array([30 * np.sin(inc), 0.0, 30 * np.cos(inc)])
fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6, origin=cam, rotation_euler=(0.0, inc, 0.0))
fr.generate_rays()
kw = dict(r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45)
endt, fl, st, ac, _ = _trace_device(ctx, _ffi.make_params(**kw), fr.d_k0, x0_shared=cam)
k_all = fr.d_k0.cpu().numpy()
o = oc.trace(k_all, cam, **kw)
flg, stp = fl.cpu().numpy(), st.cpu().numpy().astype(np.uint32)
sdiff = stp.astype(np.int64) - o["n_attempted"].astype(np.int64)
bad = np.nonzero(sdiff)[0]
out["kerr_offaxis"] = dict(nf=int((flg != o["flags"]).sum()), ns=int(len(bad)), diffs=[int(v) for v in sdiff[bad]], flags=[int(v) for v in flg[bad]],
                           steps=[int(v) for v in stp[bad]])
d = np.abs(endt.cpu().numpy() - o["end"]).max(1)
classes("kerr_offaxis", flg, d, k_all, cam, o, kw, kerr=True, same=sdiff == 0)
print(json.dumps(out["kerr_offaxis"]), flush=True)
json.dump(out, open("gpurun_out/r05_census.json", "w"), indent=1)
