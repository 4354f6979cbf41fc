#!/usr/bin/env python3
"""One bench.py JSON line on stdin -> one short summary line (scripts/gpu_r06_driver_cmd.sh)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:   # noqa: BLE001
    print(f"{tag:14s} NO LINE ({type(e).__name__})")
    raise SystemExit(0)
rf = d["roofline"]
cal = rf.get("calibration") or {}
sp = rf.get("kernel_ms_spread") or {}
sclk = cal.get("sclk_mhz_timed_region")
extra = ""
if isinstance(sclk, dict):
    extra = " ".join(f"{k}={sclk[k]:.0f}" for k in ("mclk_mhz", "power_w", "power_input_w", "temp_junction_c") if k in sclk)
print(f"{tag:14s} ms/step median {d['ms_per_step']:.4f} reps [{' '.join('%.4f' % x for x in d.get('ms_per_step_samples', []))}] "
      f"kernel_ms {rf['kernel_ms']:.4f} (min {sp.get('min', float('nan')):.4f} max {sp.get('max', float('nan')):.4f}) frac {rf['frac']:.4f} "
      f"fma {cal.get('fp64_fma_tflops_before_after')} util {cal.get('valu_issue_utilisation_vs_step_mix')} "
      f"sclk/rep {d.get('sclk_mhz_per_repetition')} hbm_copy {cal.get('hbm_copy_GBps')} {extra}")
