import numpy as np, sys
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)[:32768]    # (per-wave stamps; the histogram sits behind them)
d = d[d[:, 1] > 0]
cross = ((d[:, 2] >> np.uint64(24)) & np.uint64((1 << 20) - 1)).astype(np.float64)
passed = (d[:, 2] >> np.uint64(44)).astype(np.float64)
d = d.copy(); d[:, 2] &= np.uint64((1 << 24) - 1)
long_cyc = (d[:, 3] >> np.uint64(40)).astype(np.float64)
d = d.copy(); d[:, 3] &= np.uint64((1 << 40) - 1)
t0, t1, it, ln = d[:, 0].astype(np.int64), d[:, 1].astype(np.int64), d[:, 2], d[:, 3]
T0 = t0.min()
life = (t1 - t0) / 100.0  # us (100 MHz)
print("waves", len(d), "kernel span us", (t1.max() - T0) / 100.0)
print("start spread us: p50 %.1f max %.1f" % (np.median(t0 - T0) / 100, (t0 - T0).max() / 100))
print("end   time us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((t1 - T0) / 100.0, [0, 10, 50, 90, 100])))
print("lifetime us: mean %.1f" % life.mean(), "iters/wave mean %.1f min %d max %d" % (it.mean(), it.min(), it.max()))
print("lane utilisation %.4f" % (ln.sum() / (64.0 * it.sum())), "total lane-steps", ln.sum())
print("shader clock GHz: mean %.3f" % np.mean(d[:, 4].astype(np.float64) / ((t1 - t0) * 10.0) / 1e3 * 1e3 / 1e3))
print("us per iteration: %.3f" % (life.sum() / it.sum()))
general = (d[:, 6] >> np.uint64(40)).astype(np.float64)
d = d.copy(); d[:, 6] &= np.uint64((1 << 40) - 1)
refill = (d[:, 7] >> np.uint64(32)).astype(np.float64)
d = d.copy(); d[:, 7] &= np.uint64(0xFFFFFFFF)
cyc = d[:, 4].astype(np.float64)
print("share of wave cycles: event drain %.4f  batch fill (setup) %.4f" % (d[:, 5].sum() / cyc.sum(), d[:, 7].sum() / cyc.sum()))
if d[:, 6].sum():
    print("events drained %d, cycles per drained event-wave (64 events) %.0f, per iteration %.0f" % (
        d[:, 6].sum(), d[:, 5].sum() / (d[:, 6].sum() / 64.0), (cyc.sum() - d[:, 5].sum() - d[:, 7].sum()) / it.sum()))
if general.sum():
    print("long list (Brent): %d steps = %.4f of all parked, %.4f of wave cycles, %.0f cycles per 64 steps" % (
        general.sum(), general.sum() / max(d[:, 6].sum() + general.sum(), 1), long_cyc.sum() / cyc.sum(), long_cyc.sum() / (general.sum() / 64.0)))
if refill.sum():
    other = refill.sum() - d[:, 5].sum() - d[:, 7].sum()
    print("refill() calls without drain / batch fill (LDS pops, work fetch): %.4f of wave cycles = %.0f cycles per iteration" % (other / cyc.sum(), other / it.sum()))
if cross.sum():
    print("disk plane: some lane crossed it in %.4f of the iterations; some lane passed the filter (is parked) in %.4f; "
          "the filter ran for nothing in %.4f" % (cross.sum() / it.sum(), passed.sum() / it.sum(), (cross.sum() - passed.sum()) / it.sum()))
# round 6: event coherence (BHG_DIAG_HIST = u64 index 400000 of the dump): iterations by k = lanes parking ONE short event
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
if raw.size >= 400131:
    H = raw[400000:400065].astype(np.float64)
    A = raw[400065:400130].astype(np.float64)
    allp = float(raw[400130])
    k = np.arange(65)
    steps = H * k
    if steps.sum() > 0 and it.sum() > 0:
        print("event coherence: %d steps parked with ONE short event (%.4f of all %d parked) in %d iterations (%.4f of all iterations); mean k %.2f"
              % (steps.sum(), steps.sum() / max(allp, 1), allp, H.sum(), H.sum() / it.sum(), steps.sum() / H.sum()))
        for lo, hi in ((1, 1), (2, 3), (4, 7), (8, 15), (16, 31), (32, 63), (64, 64)):
            sel = slice(lo, hi + 1)
            if H[sel].sum():
                print("   k %2d..%2d: %.4f of such iterations, %.4f of such steps, lanes stepping beside them (mean) %.1f"
                      % (lo, hi, H[sel].sum() / H.sum(), steps[sel].sum() / steps.sum(), A[sel].sum() / H[sel].sum()))
        print("   share of one-short-event steps from iterations with k >= 16: %.4f; k >= 8: %.4f" % (steps[16:].sum() / steps.sum(), steps[8:].sum() / steps.sum()))
