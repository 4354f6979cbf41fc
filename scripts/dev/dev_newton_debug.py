import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
from conftest import frame_rays, CAM
ctx = _ffi.Context(0)
k = frame_rays(20000, seed=3, fov=0.9)
kw = dict(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)
o = oc.trace(k, CAM, **kw)
end, flags, steps, acc = ctx.trace(k, CAM, _ffi.make_params(**kw))
m = (flags == 8)
th, g, code = end[m, 3], end[m, 4], end[m, 5]
print("exit rays", m.sum(), "fast path (code>=100):", (code >= 100).sum())
f = code >= 100
print("iterations histogram", np.unique((code[f] % 100).astype(int), return_counts=True))
print("|g| at th: max %.3e; count |g|>1e-6: %d" % (np.abs(g[f]).max(), (np.abs(g[f]) > 1e-6).sum()))
bad = f & (np.abs(g) > 1e-6)
for i in np.nonzero(bad)[0][:10]:
    print("th %.6f g %.4e it %d  g0 %.4e g1 %.4e h %.4f" % (th[i], g[i], code[i] % 100, end[m][i, 0], end[m][i, 1], end[m][i, 2]))
