"""Developer check run on the GPU box: parity vs oracle on every golden + quick timings."""
import glob, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi, camera_directions
from oracle import oracle as oc

ctx = _ffi.Context(0)
print("device:", ctx.name, "CUs", ctx.num_cus)
out = {}
for f in sorted(glob.glob("tests/golden/*.npz")):
    g = np.load(f)
    if "k0" not in g:
        continue
    for form in (0, 1):
        kw = dict(r_s=float(g["r_s"]), lambda_end=float(g["lambda_end"]), max_step=float(g["max_step"]),
                  rtol=float(g["rtol"]), atol=float(g["atol"]), rhs_form=form)
        if "r_exit" in g:
            kw["r_exit"] = float(g["r_exit"])
        o = oc.trace(g["k0"], g["x0"], **kw)
        end, flags, steps, acc = ctx.trace(g["k0"], g["x0"], _ffi.make_params(**kw))
        d = np.abs(end - o["end"]).max(1)
        print(os.path.basename(f), "form", form, "n", len(d), "flags", (flags == o["flags"]).mean(),
              "steps", (steps == o["n_attempted"]).mean(), "acc", (acc == o["n_accepted"]).mean(),
              "maxdiff %.3e" % d.max(), "launch", ctx.last_launch())
# rk4
g = np.load("tests/golden/frame64_christoffel.npz")
for form in (0, 1):
    kw = dict(r_s=1.0, lambda_end=50.0, h_fixed=0.1, method=1, rhs_form=form)
    o = oc.trace(g["k0"], g["x0"], **kw)
    end, flags, steps, acc = ctx.trace(g["k0"], g["x0"], _ffi.make_params(**kw))
    d = np.abs(end - o["end"]).max(1)
    print("rk4 form", form, "flags", (flags == o["flags"]).mean(), "steps", (steps == o["n_attempted"]).mean(),
          "maxdiff %.3e" % d.max())

# timing at config 2 via host API (includes PCIe) and device API
import torch
W = H = 1024; S = 5
t = time.time(); k0 = camera_directions(W, H, S, 0.6, 0.6, 42.0).reshape(-1, 3); print("raygen %.2fs" % (time.time() - t))
cam = np.array([1e-4, 0.0, 30.0])
n = k0.shape[0]
dk = torch.from_numpy(k0).cuda()
dend = torch.empty((n, 6), dtype=torch.float64, device="cuda")
dfl = torch.empty(n, dtype=torch.uint8, device="cuda")
dst = torch.empty(n, dtype=torch.int32, device="cuda")
dac = torch.empty(n, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for name, kw in [("adaptive/christoffel", dict()), ("adaptive/reduced", dict(rhs_form=1)),
                 ("fine/christoffel", dict(max_step=0.1)), ("fine/reduced", dict(max_step=0.1, rhs_form=1)),
                 ("rk4/christoffel", dict(method=1, h_fixed=0.1)), ("rk4/reduced", dict(method=1, h_fixed=0.1, rhs_form=1))]:
    p = _ffi.make_params(r_s=1.0, lambda_end=50.0, **kw)
    ts = []
    for it in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.trace_device(p, n, dk.data_ptr(), dend.data_ptr(), x0_shared=cam, d_flags=dfl.data_ptr(),
                         d_n_steps=dst.data_ptr(), d_n_accepted=dac.data_ptr(), stream=stream)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    steps = int(dst.sum().item())
    ms = min(ts[1:])
    F = 654 if kw.get("method", 0) == 0 else 254
    print(f"{name}: {ms:.3f} ms  {n/ms/1e3:.1f} Mrays/s  steps/ray {steps/n:.1f}  {steps/ms/1e6:.2f} Gsteps/s  "
          f"{steps*F/ms/1e9:.2f} TFLOP/s-alg  hits {(dfl&1).sum().item()} launch {ctx.last_launch()}")
    out[name] = dict(ms=ms, steps=steps, n=n)
json.dump(out, open("gpurun_out/dev_check.json", "w"))
