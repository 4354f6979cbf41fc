"""Developer measurement on the GPU box: config 5 (Kerr a/M = 0.9, the reference's on-axis camera (1e-4, 0, 30), 1024 x 1024 x 5)
GPU against the oracle on EVERY ray: how many differ in flag, in step count only, by how much in the end state per class,
and what that does to the shaded 1024 x 1024 image."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
from oracle import oracle as oc, shade_reference as sh
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = 5
ctx = _ffi.Context(0)
fr = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6)
sky = synthetic_sky(2048, 1024)
fr.set_sky(sky); fr.generate_rays()
kw = dict(r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45)
fr.trace(_ffi.make_params(**kw)); rgba = fr.shade().cpu().numpy()
end, fl, st = fr.d_end.cpu().numpy(), fr.d_flags.cpu().numpy(), fr.d_steps.cpu().numpy().astype(np.uint32)
k0 = fr.d_k0.cpu().numpy()
t = time.time(); o = oc.trace(k0, np.array([1e-4, 0.0, 30.0]), **kw); print("oracle %.1f s" % (time.time() - t))
n = len(fl)
fbad = fl != o["flags"]; sbad = (~fbad) & (st != o["n_attempted"])
print("rays", n, "flag differs", int(fbad.sum()), "steps differ only", int(sbad.sum()))
for name, m in (("horizon", (o["flags"] & 1) != 0), ("escaped", o["flags"] == 4)):
    print("  class", name, "rays", int(m.sum()), "flag diff", int((fbad & m).sum()), "step diff only", int((sbad & m).sum()),
          "max |step diff|", int(np.abs(st[m].astype(int) - o["n_attempted"][m].astype(int)).max()))
    ok = m & ~fbad & ~sbad
    d = np.abs(end[ok] - o["end"][ok]).max(1)
    print("     end-state diff among identical-count rays: median %.2e p99 %.2e max %.2e" % (np.median(d), np.quantile(d, 0.99), d.max()))
    d2 = np.abs(end[m & sbad] - o["end"][m & sbad]).max(1)
    if len(d2): print("     end-state diff among step-count-differing rays: median %.2e max %.2e" % (np.median(d2), d2.max()))
# axis distance of the rays that differ: |k_x, k_y| small = through the polar axis
kperp = np.hypot(k0[:, 0], k0[:, 1])
print("  differing rays: median sin(angle to axis) %.3e vs all rays %.3e" % (np.median(kperp[fbad | sbad]) if (fbad | sbad).any() else 0, np.median(kperp)))
img_o = sh.shade_reduce(o["end"], o["flags"], W * H, S, sky)
dimg = np.abs(rgba - img_o)[:, :3].max(1)
print("image: max pixel diff %.3e, pixels > 1e-6: %d, > 1e-3: %d of %d" % (dimg.max(), int((dimg > 1e-6).sum()), int((dimg > 1e-3).sum()), W * H))
json.dump(dict(n=n, flag_diff=int(fbad.sum()), step_diff=int(sbad.sum()), img_max=float(dimg.max()), img_gt_1e6=int((dimg > 1e-6).sum()),
               img_gt_1e3=int((dimg > 1e-3).sum())), open("gpurun_out/config5_census.json", "w"))
