"""Round 5 development aid: the peak probes and what the box's sysfs says about the shader clock."""
import glob
import json
import sys
import time

sys.path.insert(0, ".")
from blackhole_geodesic_calculator_amd import _ffi

ctx = _ffi.Context(0)
out = {"name": ctx.name, "cus": ctx.num_cus}
for rep in range(3):
    for kind, nm in ((0, "fma"), (1, "mix")):
        for ms in (1.0, 5.0):
            out[f"{nm}_{ms}_{rep}"] = ctx.peak_probe(kind, ms)
files = {}
for pat in ("/sys/class/drm/card*/device/pp_dpm_sclk", "/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input",
            "/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap",
            "/sys/class/drm/card*/device/unique_id", "/sys/class/kfd/kfd/topology/nodes/*/properties"):
    for f in glob.glob(pat):
        try:
            files[f] = open(f).read()[:600]
        except Exception as e:
            files[f] = repr(e)
out["sysfs"] = files
print(json.dumps(out, indent=1))
